#!/usr/bin/env python3
"""Turn the raw output of tools/profile.sh (gpurun_out/prof_stats, prof_pmc_*) into the committed artefacts:
profiles/<tag>_bench_1Mx1M_kernel_stats.csv, profiles/<tag>_bench_1Mx1M_pmc_summary.txt and the
"1000000x1000000_n1" / "grid_..." entries of profiles/hbm_traffic.json that bench.py reports as roofline.traffic.

HBM-side bytes follow MI355X_MICROARCH.md's HBM section: FETCH_SIZE and WRITE_SIZE are collected in separate passes,
are in KB, and FETCH_SIZE counts half of the wide coalesced bytes on gfx950 (hence 2 x FETCH_SIZE + WRITE_SIZE).
Usage: python tools/summarize_profile.py <tag> ["header line"]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")


def newest(pattern):
    files = glob.glob(pattern)
    return max(files, key=os.path.getmtime) if files else None


def short(name):
    n = name.split("(")[0]
    return n.replace("void ", "").replace("oa::", "")


def main():
    tag = sys.argv[1]
    header = sys.argv[2] if len(sys.argv) > 2 else ""
    stats = newest(os.path.join(OUT, "prof_stats", "*", "*_kernel_stats.csv"))
    shutil.copy(stats, os.path.join(PROF, "%s_bench_1Mx1M_kernel_stats.csv" % tag))
    agg = collections.defaultdict(list)
    for sub in ("prof_pmc_FETCH_SIZE", "prof_pmc_WRITE_SIZE", "prof_pmc_SQ"):
        f = newest(os.path.join(OUT, sub, "*", "*_counter_collection.csv"))
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith("k_nn_search"):
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    lines = ["# %s: %s" % (tag, header),
             "# rocprofv3 PMC passes of `python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-surface` (tools/profile.sh); per-dispatch means", ""]
    mean = {}
    order = {"FETCH_SIZE": 0, "WRITE_SIZE": 1}
    for (k, c), v in sorted(agg.items(), key=lambda kv: (order.get(kv[0][1], 2), kv[0][1] if kv[0][1] not in order else "", kv[0][0])):
        mean[(k, c)] = sum(v) / len(v)
        lines.append("%-40s %-22s calls=%d mean=%g" % (k, c, len(v), mean[(k, c)]))
    lines.append("")
    traffic = {}
    for k in sorted({k for k, _ in mean}):
        if (k, "FETCH_SIZE") in mean and (k, "WRITE_SIZE") in mean:
            b = (2.0 * mean[(k, "FETCH_SIZE")] + mean[(k, "WRITE_SIZE")]) * 1024.0
            traffic[k] = b
            lines.append("%s: HBM-side traffic per launch = (2*FETCH_SIZE + WRITE_SIZE) KB = %.3g GB" % (k, b / 1e9))
    brute = [k for k in traffic if k.startswith("k_nn_search_filtered")]
    per_pair = None
    if brute and (brute[0], "SQ_INSTS_VALU") in mean:
        per_pair = mean[(brute[0], "SQ_INSTS_VALU")] * 64.0 / 1e12
        lines.append("%s: SQ_INSTS_VALU*64/1e12 pairs = %.3g VALU instructions per pair (8 algorithmic flop per pair)" % (brute[0], per_pair))
    summ = os.path.join(PROF, "%s_bench_1Mx1M_pmc_summary.txt" % tag)
    open(summ, "w").write("\n".join(lines) + "\n")
    jf = os.path.join(PROF, "hbm_traffic.json")
    tr = json.load(open(jf)) if os.path.exists(jf) else {}
    rel = os.path.relpath(summ, ROOT)
    if brute:
        k = brute[0]
        tr["1000000x1000000_n1"] = {"bytes_per_launch": traffic[k], "fetch_size_kb": mean[(k, "FETCH_SIZE")],
                                    "write_size_kb": mean[(k, "WRITE_SIZE")], "source": rel}
        if per_pair is not None:
            tr["1000000x1000000_n1"]["valu_instructions_per_pair"] = per_pair
    gk = [k for k in traffic if k.startswith("k_nn_search_grid")]          # k_nn_search_grid<1> at this size
    if gk:
        tr["grid_1000000x1000000_n1"] = {"bytes_per_launch": traffic[gk[0]], "source": rel}
    json.dump(tr, open(jf, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()

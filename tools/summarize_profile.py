#!/usr/bin/env python3
"""Turn the raw output of tools/profile.sh (gpurun_out/prof_*) into the committed artefacts:
  profiles/<tag>_bench_1Mx1M_kernel_stats.csv, profiles/<tag>_bench_1Mx1M_pmc_summary.txt,
  profiles/<tag>_surface_1M_kernel_stats.csv,  profiles/<tag>_surface_1M_pmc_summary.txt
and the entries of profiles/hbm_traffic.json that bench.py reports as roofline.traffic /
roofline.valu_instructions_per_pair -- each stamped with the kernel it was measured on, the commit and a fingerprint of
the kernel sources (bench.py withholds a figure whose kernel name no longer matches, and marks one from other sources
stale).

HBM-side bytes follow MI355X_MICROARCH.md's HBM section: FETCH_SIZE and WRITE_SIZE are collected in separate passes,
are in KB, and FETCH_SIZE counts half of the wide coalesced bytes on gfx950 (hence 2 x FETCH_SIZE + WRITE_SIZE).
Usage (here, after gpurun merged gpurun_out/): python tools/summarize_profile.py <tag> ["header line"]"""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")


BRUTE = ("k_nn_search_sorted", "k_nn_search_filtered")       # the headline's kernel (OA_NN_SORT=0: its predecessor)


def brute_of(traffic):
    """the brute-force search kernel of a pass: the sorted kernel when it ran, else the filtered one"""
    for name in BRUTE:
        ks = [k for k in traffic if k.startswith(name)]
        if ks:
            return ks
    return []


def newest(pattern):
    files = glob.glob(pattern)
    return max(files, key=os.path.getmtime) if files else None


def short(name):
    n = name.split("(")[0]
    return n.replace("void ", "").replace("oa::", "")


def csrc_sha16():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "object_alignment_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def collect(subdirs, want):
    agg = collections.defaultdict(list)
    for sub in subdirs:
        f = newest(os.path.join(OUT, sub, "*", "*_counter_collection.csv"))
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if want(k):
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg


def table(agg, lines):
    mean = {}
    order = {"FETCH_SIZE": 0, "WRITE_SIZE": 1}
    for (k, c), v in sorted(agg.items(), key=lambda kv: (order.get(kv[0][1], 2), kv[0][1] if kv[0][1] not in order else "", kv[0][0])):
        mean[(k, c)] = sum(v) / len(v)
        lines.append("%-44s %-22s calls=%d mean=%g" % (k, c, len(v), mean[(k, c)]))
    lines.append("")
    traffic = {}
    for k in sorted({k for k, _ in mean}):
        if (k, "FETCH_SIZE") in mean and (k, "WRITE_SIZE") in mean:
            b = (2.0 * mean[(k, "FETCH_SIZE")] + mean[(k, "WRITE_SIZE")]) * 1024.0
            traffic[k] = b
            lines.append("%s: HBM-side traffic per launch = (2*FETCH_SIZE + WRITE_SIZE) KB = %.4g GB" % (k, b / 1e9))
    return mean, traffic


def main():
    tag = sys.argv[1]
    header = sys.argv[2] if len(sys.argv) > 2 else ""
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True).strip()
    except Exception:
        commit = None
    stamp = {"commit": commit, "csrc_sha16": csrc_sha16()}
    jf = os.path.join(PROF, "hbm_traffic.json")
    tr = json.load(open(jf)) if os.path.exists(jf) else {}

    # ---- the bench command
    stats = newest(os.path.join(OUT, "prof_stats", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats, os.path.join(PROF, "%s_bench_1Mx1M_kernel_stats.csv" % tag))
    agg = collect(("prof_pmc_FETCH_SIZE", "prof_pmc_WRITE_SIZE", "prof_pmc_SQ", "prof_pmc_SQ2"), lambda k: k.startswith("k_nn_search"))
    lines = ["# %s: %s" % (tag, header),
             "# rocprofv3 PMC passes of `python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-surface` (tools/profile.sh); per-dispatch means",
             "# commit %s, kernel sources sha16 %s" % (commit, stamp["csrc_sha16"]), ""]
    mean, traffic = table(agg, lines)
    brute = brute_of(traffic)
    per_pair = None
    if brute and (brute[0], "SQ_INSTS_VALU") in mean:
        per_pair = mean[(brute[0], "SQ_INSTS_VALU")] * 64.0 / 1e12
        lines.append("%s: SQ_INSTS_VALU*64/1e12 pairs = %.4g VALU instructions per pair (8 algorithmic flop per pair)" % (brute[0], per_pair))
    summ = os.path.join(PROF, "%s_bench_1Mx1M_pmc_summary.txt" % tag)
    open(summ, "w").write("\n".join(lines) + "\n")
    rel = os.path.relpath(summ, ROOT)
    if brute:
        k = brute[0]
        tr["1000000x1000000_n1"] = dict(stamp, kernel=k, bytes_per_launch=traffic[k], fetch_size_kb=mean[(k, "FETCH_SIZE")],
                                        write_size_kb=mean[(k, "WRITE_SIZE")], source=rel)
        if per_pair is not None:
            tr["1000000x1000000_n1"]["valu_instructions_per_pair"] = per_pair
        # the profiled command runs `--warmup 1` and a timed loop, each from a cold start: launches 0 and 1 search without seeds
        # (more pairs reach the later filter levels), the others with.  bench.py weighs the two by its own step count.
        series = agg.get((k, "SQ_INSTS_VALU"), [])
        if len(series) >= 3:
            cold = sum(series[:2]) / 2.0 * 64.0 / 1e12
            seeded = sum(series[2:]) / len(series[2:]) * 64.0 / 1e12
            tr["1000000x1000000_n1"]["valu_instructions_per_pair_cold"] = cold
            tr["1000000x1000000_n1"]["valu_instructions_per_pair_seeded"] = seeded
            lines.append("%s: launches without seeds (the first of a loop) %.4g, with seeds %.4g VALU instructions per pair" % (k, cold, seeded))
            open(summ, "w").write("\n".join(lines) + "\n")
    gk = sorted((k for k in traffic if k.startswith("k_nn_search_grid")), key=lambda k: "true" not in k)   # the loop's kernel first: k_nn_search_grid<1, true>
    if gk:
        tr["grid_1000000x1000000_n1"] = dict(stamp, kernel=gk[0], bytes_per_launch=traffic[gk[0]], source=rel)
    # ---- N shards on this GPU (python bench.py --gpus N with OA_BENCH_SAME_DEVICE=1): the figures an N > 1 line reports
    for n in (2, 4, 8):
        aggn = collect(("prof_n%d_FETCH_SIZE" % n, "prof_n%d_WRITE_SIZE" % n, "prof_n%d_SQ" % n), lambda k: k.startswith(BRUTE))
        if not aggn:
            continue
        shard = -(-1000000 // n)
        lines.append("")
        lines.append("# %d shards of %d points on this GPU (OA_BENCH_SAME_DEVICE=1 python bench.py --gpus %d ...): per-dispatch means" % (n, shard, n))
        meann, trafficn = table(aggn, lines)
        bn = brute_of(trafficn)
        if bn:
            en = dict(stamp, kernel=bn[0], bytes_per_launch=trafficn[bn[0]], fetch_size_kb=meann[(bn[0], "FETCH_SIZE")],
                      write_size_kb=meann[(bn[0], "WRITE_SIZE")], source=rel,
                      note="measured with all %d shards on ONE GPU (OA_BENCH_SAME_DEVICE=1); per launch of one %d-point shard" % (n, shard))
            if (bn[0], "SQ_INSTS_VALU") in meann:
                en["valu_instructions_per_pair"] = meann[(bn[0], "SQ_INSTS_VALU")] * 64.0 / (shard * 1e6)
                sn = aggn.get((bn[0], "SQ_INSTS_VALU"), [])       # n shards per iteration: the first 2 n launches are unseeded
                if len(sn) >= 3 * n:
                    en["valu_instructions_per_pair_cold"] = sum(sn[:2 * n]) / (2.0 * n) * 64.0 / (shard * 1e6)
                    en["valu_instructions_per_pair_seeded"] = sum(sn[2 * n:]) / len(sn[2 * n:]) * 64.0 / (shard * 1e6)
                lines.append("%s (%d-point shard): SQ_INSTS_VALU*64 / pairs = %.4g VALU instructions per pair" % (bn[0], shard, en["valu_instructions_per_pair"]))
            tr["1000000x1000000_n%d" % n] = en
    open(summ, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))

    # ---- the surface loop
    sstats = newest(os.path.join(OUT, "prof_surf_stats", "*", "*_kernel_stats.csv"))
    if sstats:
        shutil.copy(sstats, os.path.join(PROF, "%s_surface_1M_kernel_stats.csv" % tag))
    sagg = collect(("prof_surf_FETCH_SIZE", "prof_surf_WRITE_SIZE", "prof_surf_SQ"),
                   lambda k: "search_grid" in k or k.startswith("k_bvh_search") or k.startswith("k_pair_accumulate"))
    if sagg:
        sl = ["# %s: %s" % (tag, header),
              "# rocprofv3 PMC passes of `ONLY=surface:auto python tools/time_surface.py` (1M points, 980k-vertex / 1.96M-triangle mesh: a 5-",
              "# and a 30-iteration run from a cold start); per-dispatch means over all launches",
              "# commit %s, kernel sources sha16 %s" % (commit, stamp["csrc_sha16"]), ""]
        smean, straffic = table(sagg, sl)
        ssumm = os.path.join(PROF, "%s_surface_1M_pmc_summary.txt" % tag)
        open(ssumm, "w").write("\n".join(sl) + "\n")
        tk = [k for k in straffic if k.startswith("k_tri_search_grid")]
        if tk:
            tr["surface_1000000x1957200_n1"] = dict(stamp, kernel=tk[0], bytes_per_launch=straffic[tk[0]],
                                                    source=os.path.relpath(ssumm, ROOT),
                                                    note="mean over the launches of a 5- and a 30-iteration run from a cold start")
        print("\n".join(sl))
    json.dump(tr, open(jf, "w"), indent=1)


if __name__ == "__main__":
    main()

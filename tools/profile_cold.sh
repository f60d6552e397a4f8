#!/bin/bash
# GPU box: per-launch durations and SQ / cache counters of the surface search in the cold regime (tools/cold_surface.py)
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$REPO/gpurun_out"; TAG="${TAG:-r02}"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/prof_cold_trace" "$OUT/prof_cold_pmc1" "$OUT/prof_cold_pmc2"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof_cold_trace" -- python $REPO/tools/cold_surface.py 4 5 > "$OUT/prof_cold_trace.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/prof_cold_pmc1" -- python $REPO/tools/cold_surface.py 4 5 > "$OUT/prof_cold_pmc1.log" 2>&1
timeout 300 rocprofv3 --pmc ${PMC2:-SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAVES} --output-format csv -d "$OUT/prof_cold_pmc2" -- python $REPO/tools/cold_surface.py 4 5 > "$OUT/prof_cold_pmc2.log" 2>&1
tail -2 "$OUT/prof_cold_pmc2.log"
python - <<PY > "$OUT/${TAG}_cold_profile.txt"
import csv, glob, collections
f = glob.glob("$OUT/prof_cold_trace/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
it = 0
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("oa::", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "k_reduce_solve_update" in n:
        it += 1
    per[(it, n)].append(d)
print("# per-launch durations (us), the last 4 x 5 iterations of tools/cold_surface.py 4 5 (iteration index within a run)")
its = sorted({k[0] for k in per})
its = its[-20:]
for j in range(5):
    agg = collections.defaultdict(list)
    for r_ in range(4):
        i = its[r_ * 5 + j]
        for (ii, n), v in per.items():
            if ii == i and ("search" in n or "accumulate" in n or "solve" in n):
                agg[n] += v
    print("iteration %d: " % j + "  ".join("%s %.1f" % (n[:24], sum(v) / 4) for n, v in sorted(agg.items())))
for sub in ("prof_cold_pmc1", "prof_cold_pmc2"):
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % sub)
    if not fs:
        print("# %s: no counters collected" % sub); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("oa::", "")
        if "search" in n:
            agg[(n, r["Counter_Name"])].append(float(r["Counter_Value"]))
    print("# %s: per-dispatch means over all launches of the cold runs" % sub)
    for k, v in sorted(agg.items()):
        print("%-28s %-22s n=%d mean=%.4g" % (k[0][:28], k[1], len(v), sum(v) / len(v)))
PY
cat "$OUT/${TAG}_cold_profile.txt"

#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; rm -rf "$OUT/prof_stats"
cd /tmp && export TMPDIR=/tmp
OA_GRID_STATS=1 ONLY=surface:grid timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof_stats" -- python $REPO/tools/time_surface.py > "$OUT/prof_stats.log" 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/prof_stats/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ds = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_tri_search_grid" in r["Kernel_Name"]]
print("instrumented k_tri_search_grid durations us:", ["%.0f" % d for d in ds])
PY
grep "phases" "$OUT/prof_stats.log" | sed -n "1p;5p;35p"

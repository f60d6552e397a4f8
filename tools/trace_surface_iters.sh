#!/bin/bash
# GPU box: per-launch durations of the surface-mode search kernels over a 5 + 30 iteration run (grid + tree)
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; rm -rf "$OUT/prof_surf_iters"
cd /tmp && export TMPDIR=/tmp
ONLY=surface:grid timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof_surf_iters" -- python $REPO/tools/time_surface.py > "$OUT/prof_surf_iters.log" 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/prof_surf_iters/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
it = 0
line = []
for r in rows:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "k_tri_search_grid" in n: line = ["grid %7.1f" % d]
    elif "k_bvh_search<true" in n and line: line.append("tree %7.1f" % d)
    elif "k_pair_accumulate" in n and line: line.append("acc %5.1f" % d)
    elif "k_reduce_solve_update" in n and line:
        line.append("solve %4.1f" % d); print("search %2d: " % it + "  ".join(line) + " us"); it += 1; line = []
PY

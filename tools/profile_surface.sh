#!/bin/bash
# GPU box: timings + per-kernel stats of the surface-mode / tree-search paths -> gpurun_out/r01g_*
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$REPO/gpurun_out"; TAG="${TAG:-r01g}"
mkdir -p "$OUT"; cd "$REPO"
(echo "# python tools/fuzz_parity.py 200 99   (brute / grid / bvh vs the oracle; vertex + surface mode)"; timeout 900 python tools/fuzz_parity.py 200 99 2>&1 | grep -v amdgpu.ids) > "$OUT/${TAG}_fuzz_parity_200trials.txt"
(echo "# python tools/time_surface.py  (1M queries, 980k-vertex / 1.96M-triangle target)"; timeout 300 python tools/time_surface.py 2>&1 | grep -v amdgpu.ids
 echo; echo "# PARTIAL=1 python tools/time_surface.py  (target = z > 0 half: ~47% of the queries have no partner within thresh)"; PARTIAL=1 timeout 300 python tools/time_surface.py 2>&1 | grep -v amdgpu.ids) > "$OUT/${TAG}_surface_1M_timings.txt"
(echo "# python tools/time_small.py"; timeout 300 python tools/time_small.py 2>&1 | grep -v amdgpu.ids) > "$OUT/${TAG}_small_problem_timings.txt"
(echo "# python tools/time_crossover.py"; timeout 600 python tools/time_crossover.py 2>&1 | grep -v amdgpu.ids) > "$OUT/${TAG}_search_mode_crossover.txt"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/prof_surf"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_surf" -- python $REPO/tools/time_surface.py > "$OUT/prof_surf.log" 2>&1
cp "$OUT"/prof_surf/*/*_kernel_stats.csv "$OUT/${TAG}_surface_1M_kernel_stats.csv"
bash $REPO/tools/profile_surface_pmc.sh > "$OUT/${TAG}_surface_1M_pmc_summary.txt" 2>&1
tail -3 "$OUT/${TAG}_fuzz_parity_200trials.txt"

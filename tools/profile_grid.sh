#!/bin/bash
# GPU box: per-kernel times of the grid-search loop (bench.py's grid_path leg dominates the launch count)
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_grid" -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-surface > "$OUT/prof_grid.log" 2>&1
cat $OUT/prof_grid/*/*_kernel_stats.csv | cut -c1-60,200-400 | head -20

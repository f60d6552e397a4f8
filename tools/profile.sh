#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py, then separate PMC passes for the
# HBM-side counters (MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE cannot share a pass; FETCH_SIZE reads 1/2
# of wide coalesced bytes on gfx950).  Results land in gpurun_out/prof_*; summaries are copied to profiles/ by hand.
set -u
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$REPO/gpurun_out"
STEPS="${STEPS:-5}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-surface"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -- $BENCH > "$OUT/prof_stats.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/prof_pmc_$c" -- $BENCH > "$OUT/prof_pmc_$c.log" 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --output-format csv -d "$OUT/prof_pmc_SQ" -- $BENCH > "$OUT/prof_pmc_SQ.log" 2>&1
find "$OUT" -name "*.csv" | head -50

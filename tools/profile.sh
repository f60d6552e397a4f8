#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py, then separate PMC passes for the
# HBM-side counters (MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE cannot share a pass; FETCH_SIZE reads 1/2
# of wide coalesced bytes on gfx950) and the SQ counters -- of the bench command, and of the surface-mode loop
# (tools/time_surface.py, AUTO search).  Results land in gpurun_out/prof_*; tools/summarize_profile.py <tag> turns them
# into the committed artefacts under profiles/.
set -u
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$REPO/gpurun_out"
STEPS="${STEPS:-5}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-surface --no-whole-call"
rm -rf "$OUT"/prof_stats "$OUT"/prof_pmc_* "$OUT"/prof_surf_*
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -- $BENCH > "$OUT/prof_stats.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/prof_pmc_$c" -- $BENCH > "$OUT/prof_pmc_$c.log" 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --output-format csv -d "$OUT/prof_pmc_SQ" -- $BENCH > "$OUT/prof_pmc_SQ.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d "$OUT/prof_pmc_SQ2" -- $BENCH > "$OUT/prof_pmc_SQ2.log" 2>&1
# N = 2 / 4 / 8 shards, all on this GPU (OA_BENCH_SAME_DEVICE=1), so that an N > 1 bench line has PMC figures of its own shard size
rm -rf "$OUT"/prof_n[248]_*
for n in 2 4 8; do
  BN="env OA_BENCH_SAME_DEVICE=1 python $REPO/bench.py --gpus $n --steps $STEPS --warmup 1 --no-cpu-baseline --no-surface --no-grid --no-mfma --no-whole-call"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/prof_n${n}_$c" -- $BN > "$OUT/prof_n${n}_$c.log" 2>&1
  done
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/prof_n${n}_SQ" -- $BN > "$OUT/prof_n${n}_SQ.log" 2>&1
done
SURF="env ONLY=surface:auto python $REPO/tools/time_surface.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_surf_stats" -- $SURF > "$OUT/prof_surf_stats.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/prof_surf_$c" -- $SURF > "$OUT/prof_surf_$c.log" 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d "$OUT/prof_surf_SQ" -- $SURF > "$OUT/prof_surf_SQ.log" 2>&1
find "$OUT" -name "*.csv" | head -50

#!/bin/bash
# GPU box: one measurement step of round 2 -- search-kernel parity (tests + fuzzers) and timings.
# Usage: TAG=r02b bash tools/r02_step.sh [quick]
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$REPO/gpurun_out"; TAG="${TAG:-r02}"
mkdir -p "$OUT"; cd "$REPO"
K='grid or surface or radius or crowded or turns or step_mode or golden or shards or non_finite or overflow'
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "$K" 2>&1 | tail -5) > "$OUT/${TAG}_tests.txt"
(echo "# python tools/fuzz_parity.py ${FUZZ:-150} ${SEED:-2024}"; timeout 900 python tools/fuzz_parity.py ${FUZZ:-150} ${SEED:-2024} 2>&1 | grep -v amdgpu.ids | tail -4) > "$OUT/${TAG}_fuzz.txt"
if [ "$1" != "quick" ]; then
(echo "# python tools/fuzz_modes_large.py ${FUZZL:-12} ${SEED:-2024}"; timeout 900 python tools/fuzz_modes_large.py ${FUZZL:-12} ${SEED:-2024} 2>&1 | grep -v amdgpu.ids | tail -4) > "$OUT/${TAG}_fuzz_large.txt"
fi
(echo "# ONLY=surface:auto / surface:grid / vertex:grid python tools/time_surface.py"
 for o in surface:auto surface:grid vertex:grid; do ONLY=$o timeout 300 python tools/time_surface.py 2>&1 | grep -v amdgpu.ids; done) > "$OUT/${TAG}_surface_timings.txt"
timeout 400 python bench.py --steps 10 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_bench_quick.json" 2> "$OUT/${TAG}_bench_quick.err"
python - <<PY
import json
d = json.load(open("$OUT/${TAG}_bench_quick.json"))
print("brute: %.2f ms/search; grid: %.4f ms/search, %.0f it/s; surface: %.3f ms/search (30 it), cold5 %.3f ms/step" % (
    d["ms_per_nn_search"], d["grid_path"]["ms_per_nn_search"], d["grid_path"]["value"],
    d["surface_path"]["ms_per_nn_search"], d["surface_path"]["cold_5_iterations"]["ms_per_step"]))
PY
cat "$OUT/${TAG}_tests.txt" "$OUT/${TAG}_fuzz.txt" "$OUT/${TAG}_surface_timings.txt"
[ -f "$OUT/${TAG}_fuzz_large.txt" ] && cat "$OUT/${TAG}_fuzz_large.txt"

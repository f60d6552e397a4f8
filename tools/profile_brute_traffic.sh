#!/bin/bash
# GPU box: HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and time of the brute-force search kernel alone
# (tools/time_nn.py: 1 iteration + 5 seeded searches at 1M<->1M).  Usage: [OA_NN_SPLITS=n] bash tools/profile_brute_traffic.sh
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python $REPO/tools/time_nn.py 2>&1 | grep -v amdgpu
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$OUT/prof_bt_$c"
  timeout 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/prof_bt_$c" -- python $REPO/tools/time_nn.py > "$OUT/prof_bt_$c.log" 2>&1
done
python - <<PY
import csv, glob
v = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/prof_bt_%s/*/*_counter_collection.csv" % c)[0]
    x = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_nn_search_filtered" in r["Kernel_Name"] and r["Counter_Name"] == c]
    v[c] = sum(x[1:]) / max(1, len(x) - 1)          # the seeded launches
print("k_nn_search_filtered, seeded launches: FETCH_SIZE %.0f KB  WRITE_SIZE %.0f KB  -> HBM-side traffic (2*FETCH + WRITE) = %.1f MB per launch"
      % (v["FETCH_SIZE"], v["WRITE_SIZE"], (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / 1e6))
PY

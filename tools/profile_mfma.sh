#!/bin/bash
# GPU box: the OA_NN_MFMA=1 experiment under rocprofv3 -- kernel durations (tools/time_nn.py: 1 iteration + 5 seeded
# searches at 1M<->1M), then SQ counters and HBM-side traffic of k_nn_search_mfma in separate passes.
# Usage: TAG=r02s bash tools/profile_mfma.sh
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$REPO/gpurun_out"; TAG="${TAG:-r02s}"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export OA_NN_MFMA=1
rm -rf "$OUT/prof_mf_stats"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_mf_stats" -- python $REPO/tools/time_nn.py > "$OUT/prof_mf_stats.log" 2>&1
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  d="$OUT/prof_mf_$(echo $c | cut -d' ' -f1)"
  rm -rf "$d"
  timeout 300 rocprofv3 --pmc $c --output-format csv -d "$d" -- python $REPO/tools/time_nn.py > "$d.log" 2>&1
done
python - <<PY > "$OUT/${TAG}_mfma_profile.txt"
import csv, glob, collections
print("# ${TAG}: OA_NN_MFMA=1 python tools/time_nn.py (1M<->1M: one iteration + 5 seeded searches) under rocprofv3")
f = glob.glob("$OUT/prof_mf_stats/*/*_kernel_stats.csv")
if f:
    for r in csv.DictReader(open(f[0])):
        if "search" in r["Name"] or "accumulate" in r["Name"]:
            print("%-60s calls %s  avg %.1f us  total %.1f ms" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/prof_mf_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_nn_search_mfma" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# k_nn_search_mfma, per-dispatch means over the launches (the first one is unseeded)")
m = {}
for k, v in sorted(agg.items()):
    m[k] = sum(v) / len(v)
    print("%-28s n=%d mean=%.5g  seeded mean=%.5g" % (k, len(v), m[k], sum(v[1:]) / max(1, len(v) - 1)))
if "SQ_INSTS_VALU" in m:
    print("VALU instructions per pair (x64 / 1e12): %.3f" % (m["SQ_INSTS_VALU"] * 64 / 1e12))
if "SQ_INSTS_MFMA" in m:
    print("MFMA instructions per launch: %.4g  (1e12 pairs / 1024 = 9.77e8 in the hot path)" % m["SQ_INSTS_MFMA"])
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    print("HBM-side traffic per launch (2 x FETCH_SIZE + WRITE_SIZE) KB = %.4g GB" % ((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024 / 1e9))
PY
cat "$OUT/${TAG}_mfma_profile.txt"

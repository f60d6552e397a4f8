#!/bin/bash
# GPU box: L2 (TCC) counters of the brute-force search kernel (tools/time_nn.py), to explain its HBM-side traffic
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_ATOMIC_sum TCC_WRITE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_DRAM_sum"; do
  i=$((i+1)); rm -rf "$OUT/prof_l2_$i"
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/prof_l2_$i" -- python $REPO/tools/time_nn.py > "$OUT/prof_l2_$i.log" 2>&1 || tail -3 "$OUT/prof_l2_$i.log"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/prof_l2_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_nn_search_filtered" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    w = v[1:] if len(v) > 1 else v
    print("k_nn_search_filtered %-26s seeded launches n=%d mean=%.4g" % (k, len(w), sum(w) / len(w)))
PY

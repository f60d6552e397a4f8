#!/bin/bash
# GPU box: SQ counters of k_nn_search_grid in the bench's grid_path leg
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"; rm -rf "$OUT/prof_grid_pmc"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/prof_grid_pmc" -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-surface > "$OUT/prof_grid_pmc.log" 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/prof_grid_pmc/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_nn_search_grid" in r["Kernel_Name"] or "k_pair_accumulate" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].split("(")[0][-22:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    v=v[len(v)//2:]   # converged half
    print(k, "n=%d mean=%.4g" % (len(v), sum(v)/len(v)))
PY

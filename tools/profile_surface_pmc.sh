#!/bin/bash
# GPU box: SQ counters of the surface-mode searches (tools/time_surface.py)
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"; rm -rf "$OUT/prof_surf_pmc"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/prof_surf_pmc" -- python $REPO/tools/time_surface.py > "$OUT/prof_surf_pmc.log" 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/prof_surf_pmc/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "search_grid" in n or "k_bvh_search" in n:
        agg[(n.split("(")[0][-26:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    v=v[len(v)//2:]
    print(k, "n=%d mean=%.4g" % (len(v), sum(v)/len(v)))
PY

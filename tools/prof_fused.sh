#!/bin/bash
# GPU box: per-kernel times of the AUTO-mode loop at C1 / C2 / C3, fused accumulation on and off (rocprofv3 --stats)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof_fused
for fused in 1; do
  for cfg in c1 c2 c3; do
    OA_FUSED_ACC=$fused rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fused/${cfg}_f${fused} -o out -- python $R/tools/run_cfg.py $cfg 200 > /dev/null 2>&1
    f=$(find $R/gpurun_out/prof_fused/${cfg}_f${fused} -name "*kernel_stats.csv" | head -1)
    echo "== $cfg fused=$fused"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("%-60s calls %6s avg %9.2f us total %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
  done
done

#!/bin/bash
# GPU box: a wider set of counters for the surface-mode search (k_tri_search_grid) and, for comparison, the vertex grid search of
# the bench: where the wave cycles go (VALU / LDS / VMEM issue, waits), how busy the texture path (TA / TD / TCP) and the L2 are.
# NOTE: ten rocprofv3 passes; the TCP / TCC groups are slow (the whole script took > 25 min on the box): give gpurun --timeout 3000, or run
# it for one group at a time.
# One rocprofv3 --pmc pass per group (no tracing alongside).  Output: gpurun_out/prof_deep/<group>/, summarised to stdout.
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$REPO/gpurun_out/prof_deep"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SURF="env ONLY=surface:auto python $REPO/tools/time_surface.py"
GRID="python $REPO/tools/run_cfg.py c3 60"
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
G2="SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS_ATOMIC SQ_INSTS_BRANCH SQ_IFETCH SQ_LDS_IDX_ACTIVE"
G3="TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
G4="TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
G5="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr GRBM_GUI_ACTIVE"
i=0
for g in "$G1" "$G2" "$G3" "$G4" "$G5"; do
  i=$((i+1))
  rocprofv3 --pmc $g --output-format csv -d "$OUT/surf_g$i" -- $SURF > "$OUT/surf_g$i.log" 2>&1
  rocprofv3 --pmc $g --output-format csv -d "$OUT/grid_g$i" -- $GRID > "$OUT/grid_g$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for what, pat in (("surface", "k_tri_search_grid"), ("grid", "k_nn_search_grid<1, true")):
    agg = collections.OrderedDict()
    for f in sorted(glob.glob(out + "/%s_g*/**/*counter_collection.csv" % ("surf" if what == "surface" else "grid"), recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
        by = collections.defaultdict(list)
        for r in rows: by[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in by.items():
            v = v[len(v) // 2:]                                   # the later (settled) half of the launches
            agg[k] = (len(v), sum(v) / len(v))
    print("== %s (%s), mean over the later half of its launches" % (what, pat))
    for k, (n, m) in agg.items(): print("  %-34s n=%3d  %.5g" % (k, n, m))
PY

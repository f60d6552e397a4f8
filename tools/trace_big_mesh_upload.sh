cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/tr_big; rocprofv3 --kernel-trace --output-format csv -d $O/tr_big -- python $R/tools/cold_surface.py 1 2 > $O/tr_big.log 2>&1
python - <<'PY'
import csv, glob, os
f = sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/tr_big/**/*_kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("oa::", "")) for r in csv.DictReader(open(f))]
rows.sort()
i0 = [i for i, r in enumerate(rows) if "k_pack_target" in r[2]][0]
t0 = rows[i0][0]; prev = t0
for s, e, n in rows[i0:]:
    if "k_stamp_start" in n: break
    print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n[:70]))
    prev = max(prev, e)
PY
rm -rf $O/tr_big

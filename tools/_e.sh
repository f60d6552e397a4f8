R=$PWD; cd /tmp && export TMPDIR=/tmp
cat > /tmp/e.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine
tgt, tris = synth.lattice_surface_mesh(700, 1400)
src = synth.bunny_surface(1_000_000, offset=0.37)
mxa = synth.rigid4(synth.rotation_from_rotvec([0.02, -0.015, 0.025]), [0.01, -0.008, 0.012])
with IcpEngine(0) as e:
    e.set_search_mode("grid"); e.set_target_mesh(tgt, tris); e.set_source(src, stride=1); e.set_matrices(mxa, np.identity(4, dtype=np.float32))
    r = e.run(iters=8, thresh=0.05, early_exit=False)
PY
rm -rf $R/gpurun_out/prof_e
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_e -- python /tmp/e.py > /dev/null 2>&1
python - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("$R/gpurun_out/prof_e/*/*_kernel_trace.csv")[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
for r in rows:
    n=r["Kernel_Name"]
    if "k_tri_search_grid" in n or "k_bvh_search" in n:
        print(n.split("(")[0][-24:], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, "us")
PY

#!/usr/bin/env python3
"""GPU box: lanes per query of the two grid kernels (OA_GRID_LANES = 1 / 2 / 4) by shard size -- the data behind the
lane selection in launch_nn_impl / launch_tri_search.  us per search, 40 iterations from an offset pose (settled)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine

pose = synth.rigid4(synth.rotation_from_rotvec([0.02, -0.015, 0.025]), [0.01, -0.008, 0.012])
ident = np.identity(4, dtype=np.float32)
meshes = {"82k tris / 41k verts": synth.bumpy_icosphere_mesh(6), "1.96M tris / 980k verts": synth.lattice_surface_mesh(700, 1400)}
for mname, (tgt, tris) in meshes.items():
    for ns in (16000, 32000, 64000, 128000, 256000, 400000, 600000, 1000000):
        src = synth.bunny_surface(ns, offset=0.37)
        for surf in (False, True):
            out = []
            for lanes in ("1", "2", "4"):
                os.environ["OA_GRID_LANES"] = lanes
                with IcpEngine(0) as e:
                    e.set_search_mode("grid")
                    if surf:
                        e.set_target_mesh(tgt, tris)
                    else:
                        e.set_target(tgt)
                    e.set_source(src, stride=1)
                    e.set_matrices(pose, ident)
                    e.run(iters=5, thresh=0.05, early_exit=False)
                    e.set_matrices(pose, ident)
                    r = e.run(iters=40, thresh=0.05, early_exit=False)
                    out.append("%s lane%s %7.1f" % (lanes, "" if lanes == "1" else "s", 1e3 * r.nn_ms_total / 40))
            print("%-24s %-7s ns %7d: %s" % (mname, "surface" if surf else "vertex", ns, "   ".join(out)), flush=True)

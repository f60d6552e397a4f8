#!/usr/bin/env python3
"""GPU box: grid search vs tree search (one wave per query) as the shard size grows -- the data behind the
OA_SEARCH_AUTO switch-over (oa_icp.hip: bvh_whole)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine

pose = synth.rigid4(synth.rotation_from_rotvec([0.02, -0.015, 0.025]), [0.01, -0.008, 0.012])
ident = np.identity(4, dtype=np.float32)
meshes = {"82k tris": synth.bumpy_icosphere_mesh(6), "2M tris": synth.lattice_surface_mesh(700, 1400)}
for mname, (tgt, tris) in meshes.items():
    for ns in (2000, 4000, 8000, 16000, 32000, 64000, 128000, 256000):
        src = synth.bunny_surface(ns, offset=0.37)
        for surf in (True, False):
            out = []
            for mode in ("grid", "bvh"):
                with IcpEngine(0) as e:
                    e.set_search_mode(mode)
                    if surf:
                        e.set_target_mesh(tgt, tris)
                    else:
                        e.set_target(tgt)
                    e.set_source(src, stride=1)
                    e.set_matrices(pose, ident)
                    e.run(iters=5, thresh=0.05, early_exit=False)
                    e.set_matrices(pose, ident)
                    r = e.run(iters=40, thresh=0.05, early_exit=False)
                    out.append("%s nn %7.1f us" % (mode, 1e3 * r.nn_ms_total / 40))
            print("target %-8s ns %6d %s: %s" % (mname, ns, "surface" if surf else "vertex ", "   ".join(out)), flush=True)

#!/usr/bin/env python3
"""GPU box: grid vs tree for SHORT runs -- a 50-iteration run with early exit from an offset pose converges in ~5
iterations, all of them in the early regime (stale seeds, long reach) where the tree is relatively stronger than in
the 40-iteration averages of tools/time_crossover.py.  The OA_SEARCH_AUTO switch-over is set between the two."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine

pose = synth.rigid4(synth.rotation_from_rotvec([0.02, -0.015, 0.025]), [0.01, -0.008, 0.012])
ident = np.identity(4, dtype=np.float32)
meshes = {"82k tris": synth.bumpy_icosphere_mesh(6), "2M tris": synth.lattice_surface_mesh(700, 1400)}
sizes = [int(x) for x in os.environ.get("SIZES", "8000,12000,16000,24000,32000,48000,64000,96000,128000").split(",")]
for mname, (tgt, tris) in meshes.items():
    for surf in (True, False):
        for ns in sizes:
            src = synth.bunny_surface(ns, offset=0.37)
            out = []
            for mode in ("grid", "bvh", "auto"):
                with IcpEngine(0) as e:
                    e.set_search_mode(mode)
                    if surf:
                        e.set_target_mesh(tgt, tris)
                    else:
                        e.set_target(tgt)
                    e.set_source(src, stride=1)
                    best = 1e9
                    for rep in range(4):
                        e.set_source(src, stride=1)                      # forget the seeds: every run starts cold
                        e.set_matrices(pose, ident)
                        t0 = time.perf_counter()
                        r = e.run(iters=50, thresh=0.05, target_d=0.01, early_exit=True)
                        best = min(best, time.perf_counter() - t0)
                    out.append("%s %6.0f us (%d it)" % (mode, 1e6 * best, r.iters_done))
            print("target %-8s %s ns %6d: %s" % (mname, "surface" if surf else "vertex ", ns, "   ".join(out)), flush=True)

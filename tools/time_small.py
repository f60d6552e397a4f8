#!/usr/bin/env python3
"""GPU box: per-iteration wall/device time of small problems (launch-bound regime), every search mode."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine

pose = synth.rigid4(synth.rotation_from_rotvec([0.06, -0.05, 0.08]), [0.03, -0.02, 0.025])
ident = np.identity(4, dtype=np.float32)
cases = {"C1 ico 2562": synth.c1_icospheres() + (None,),
         "bumpy 2562": (synth.bumpy_icosphere(4), synth.bumpy_icosphere(4), pose, ident, None),
         "C2 bunny 100k": synth.c2_bunny_pair(100_000) + (None,),
         "random 20k": synth.c3_random_pair(20_000) + (None,)}
for sub in (2, 4, 6):
    v, f = synth.bumpy_icosphere_mesh(sub)
    cases["surface bumpy %d tris" % len(f)] = (synth.bumpy_icosphere(4), v, pose, ident, f)
iters = 200
for name, (src, tgt, mxa, mxb, tris) in cases.items():
    for mode in ("auto", "brute", "grid", "bvh"):
        with IcpEngine(0) as e:
            e.set_search_mode(mode)
            if tris is None:
                e.set_target(tgt)
            else:
                e.set_target_mesh(tgt, tris)
            e.set_source(src, stride=1)
            e.set_matrices(mxa, mxb)
            e.run(iters=20, early_exit=False)
            e.set_matrices(mxa, mxb)
            t0 = time.perf_counter()
            r = e.run(iters=iters, early_exit=False)
            wall = time.perf_counter() - t0
            print("%-26s %-5s wall %8.1f us/iter, device loop %8.1f us/iter, nn %8.1f us/iter" % (
                name, mode, 1e6 * wall / iters, 1e3 * r.loop_ms / iters, 1e3 * r.nn_ms_total / iters), flush=True)

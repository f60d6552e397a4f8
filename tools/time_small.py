#!/usr/bin/env python3
"""GPU box: per-iteration wall/device time of small problems (launch-bound regime)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine
cases = {"C1 ico 2562": synth.c1_icospheres(), "bumpy 2562": (synth.bumpy_icosphere(4), synth.bumpy_icosphere(4), synth.rigid4(synth.rotation_from_rotvec([0.06, -0.05, 0.08]), [0.03, -0.02, 0.025]), np.identity(4, dtype=np.float32)),
         "C2 bunny 100k": synth.c2_bunny_pair(100_000), "random 20k": synth.c3_random_pair(20_000)}
for name, (src, tgt, mxa, mxb) in cases.items():
    with IcpEngine(0) as e:
        e.set_target(tgt); e.set_source(src, stride=1)
        for it in (200,):
            e.set_matrices(mxa, mxb)
            e.run(iters=20, early_exit=False)
            e.set_matrices(mxa, mxb)
            t0 = time.perf_counter()
            r = e.run(iters=it, early_exit=False)
            wall = time.perf_counter() - t0
            print("%-16s iters %d: wall %.1f us/iter, device loop %.1f us/iter, nn %.1f us/iter" % (name, it, 1e6 * wall / it, 1e3 * r.loop_ms / it, 1e3 * r.nn_ms_total / it), flush=True)

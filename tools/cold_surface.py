#!/usr/bin/env python3
"""GPU box: the COLD regime of the surface search -- R repetitions of a 5-iteration alignment from the initial pose
with no correspondence seeds (what an early-exit operator call runs in), 1M points against the 1.96M-triangle mesh.
Usage: python tools/cold_surface.py [reps] [iters] [mode]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mode = sys.argv[3] if len(sys.argv) > 3 else "auto"
ns = int(os.environ.get("NS", 1_000_000))
tgt, tris = synth.lattice_surface_mesh(700, 1400)
src = synth.bunny_surface(ns, offset=0.37)
mxa = synth.rigid4(synth.rotation_from_rotvec([0.02, -0.015, 0.025]), [0.01, -0.008, 0.012])
mxb = np.identity(4, dtype=np.float32)
with IcpEngine(0) as e:
    e.set_search_mode(mode)
    e.set_target_mesh(tgt, tris)
    e.set_source(src, stride=1)
    e.set_matrices(mxa, mxb)
    e.run(iters=2, thresh=0.05, early_exit=False)
    out = []
    for _ in range(reps):
        e.set_matrices(mxa, mxb)
        e.reset_seeds()
        t0 = time.perf_counter()
        r = e.run(iters=iters, thresh=0.05, early_exit=False)
        out.append((1e3 * (time.perf_counter() - t0) / iters, r.nn_ms_total / iters))
    print("cold %d-iteration runs, %s search, %d points / %d triangles: wall %.3f ms/iter, nn %.3f ms/iter (min over %d reps: %.3f / %.3f)"
          % (iters, mode, ns, len(tris), np.mean([o[0] for o in out]), np.mean([o[1] for o in out]), reps,
             min(o[0] for o in out), min(o[1] for o in out)))

#!/usr/bin/env python3
"""GPU box: A/B of the surface search under different environment knobs, ONE process (the meshes are generated once; every
configuration gets a fresh engine, which reads the knobs at creation / upload time).
Usage: python tools/ab_surface.py "" "OA_TRI_SHARE=0" "OA_GRID_BUDGET=384,OA_GRID_BUDGET_MOVING=4" ...   (REPS=3 by default)
Prints per configuration: ms per iteration (device loop) and per search over the first 5 and over 30 iterations from a cold start."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine

configs = sys.argv[1:] or [""]
reps = int(os.environ.get("REPS", 3))
ns = int(os.environ.get("NS", 1_000_000))
tgt, tris = synth.lattice_surface_mesh(int(os.environ.get("NU", 700)), int(os.environ.get("NV", 1400)))
src = synth.bunny_surface(ns, offset=0.37)
mxa = synth.rigid4(synth.rotation_from_rotvec([0.02, -0.015, 0.025]), [0.01, -0.008, 0.012])
mxb = np.identity(4, dtype=np.float32)
ref = None
for cfg in configs:
    pairs = [kv.split("=", 1) for kv in cfg.split(",") if kv]
    for k, v in pairs:
        os.environ[k] = v
    try:
        with IcpEngine(0) as e:
            e.set_target_mesh(tgt, tris)
            e.set_source(src, stride=1)
            e.set_matrices(mxa, mxb)
            e.run(iters=2, thresh=0.05, early_exit=False)
            out = {5: [], 30: []}
            for _ in range(reps):
                for it in (5, 30):
                    e.set_matrices(mxa, mxb)
                    e.reset_seeds()
                    r = e.run(iters=it, thresh=0.05, early_exit=False)
                    out[it].append((r.loop_ms / it, r.nn_ms_total / it))
            if ref is None:
                ref = r.matrix_world.copy()
            same = bool(np.array_equal(ref, r.matrix_world))
        f = lambda it, k: min(o[k] for o in out[it])
        print("%-60s cold-5: %.3f ms/iter (search %.3f) | 30: %.3f ms/iter (search %.3f) | final matrix == first config's: %s"
              % (cfg or "(defaults)", f(5, 0), f(5, 1), f(30, 0), f(30, 1), same), flush=True)
    finally:
        for k, _ in pairs:
            os.environ.pop(k, None)

#!/usr/bin/env python3
"""GPU box: soak test of whole alignment calls -- many calls of random sizes and kinds (vertex / surface targets, one context or
two children, masks, strides) on long-lived contexts, every call repeated once and compared bitwise with itself, device-cache
size and host RSS watched (uploads recycle device blocks by size: the cache must stay under its cap and the process must not
grow).  Usage: python tools/soak.py [calls] [seed]"""
import os, resource, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
tv, tt = synth.bumpy_icosphere_mesh(5)                                   # 20k triangles
pose = synth.rigid4(synth.rotation_from_rotvec([0.02, -0.015, 0.025]), [0.01, -0.008, 0.012])
eye = np.identity(4, dtype=np.float32)
t0 = time.time()
rss0 = None
bad = 0
with IcpEngine(0) as one, IcpEngine(devices=[0, 0]) as two:
    for k in range(calls):
        eng = two if rng.random() < 0.25 else one
        surface = rng.random() < 0.35
        n = int(rng.choice([700, 2562, 9000, 40000, 70000, 130000, 300000]))
        stride = int(rng.choice([1, 1, 2, 3]))
        if surface:
            src = synth.bunny_surface(n, offset=float(rng.random()))
            mxa, mxb, thresh = pose, eye, 0.05
        else:
            src, tgt, mxa, mxb = synth.c2_bunny_pair(n, seed=int(rng.integers(1, 1000)))
            thresh = 0.5
        vlist = None
        if rng.random() < 0.3:
            vlist = np.flatnonzero(rng.random(len(src)) > 0.2).astype(np.int64)
        outs = []
        for rep in range(2):
            if surface:
                eng.set_target_mesh(tv, tt)
            else:
                eng.set_target(tgt)
            eng.set_source(src, vlist=vlist, stride=stride)
            eng.set_matrices(mxa, mxb)
            r = eng.run(iters=int(rng.integers(3, 12)) if rep == 0 else outs[0][2], thresh=thresh, target_d=0.01, use_target=True, early_exit=False)
            outs.append((r.step_M.copy(), r.matrix_world.copy(), len(r.step_M)))
        same = np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        bad += not same
        if k == 20:
            rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        if k % 50 == 0 or not same:
            print("call %4d  %s n=%6d stride %d %s %s  cache %.1f MB  maxrss %.0f MB  %s" % (
                k, "surface" if surface else "vertex ", n, stride, "mask" if vlist is not None else "    ", "2 children" if eng is two else "          ",
                one.stat("cache_bytes") / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024, "ok" if same else "MISMATCH"), flush=True)
rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print("SOAK DONE: %d calls x 2, %d not reproducible, max RSS %.0f -> %.0f MB (after call 20 -> end), %.0f s" % (calls, bad, (rss0 or rss1) / 1024, rss1 / 1024, time.time() - t0))

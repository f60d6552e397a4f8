#!/usr/bin/env python3
"""GPU box: wall time of the drop-in functions (contract 1 and 2): make_pairs + affine_matrix_from_points per call, as a
caller that keeps the reference's Python loop would use them (host arrays in, host arrays out)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.functions import make_pairs, affine_matrix_from_points, GpuBVH, AlignObject

for name, (src, tgt, mxa, mxb) in {"C1 2562": synth.c1_icospheres(), "C2 100k": synth.c2_bunny_pair(100_000), "C3 1M": synth.c3_random_pair(1_000_000)}.items():
    align, base = AlignObject(src, mxa), AlignObject(tgt, mxb)
    t0 = time.perf_counter()
    bvh = GpuBVH.FromObject(base, None)
    t_build = time.perf_counter() - t0
    vlist = np.arange(len(src))
    make_pairs(align, base, bvh, vlist, 0.5, 1, calc_stats=True)          # first call: source upload
    ts, tk = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        A, B, ds = make_pairs(align, base, bvh, vlist, 0.5, 1, calc_stats=True)
        ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        M = affine_matrix_from_points(A, B, shear=False, scale=False, usesvd=True)
        tk.append(time.perf_counter() - t0)
    print("%-8s build %.1f ms; make_pairs %.2f ms (K = %d, %.1f MB out); affine_matrix_from_points %.2f ms" % (
        name, 1e3 * t_build, 1e3 * min(ts), A.shape[1], (A.nbytes + B.nbytes) / 1e6, 1e3 * min(tk)), flush=True)

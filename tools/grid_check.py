#!/usr/bin/env python3
"""GPU box: compare the grid search (OA_NN_GRID=1) with the brute-force kernel (OA_NN_GRID=0) on BASELINE-shaped
clouds -- identical (index, d2) required -- and print kernel times."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(tag, src, tgt, mxa, mxb, iters=4):
    from object_alignment_amd.engine import IcpEngine
    out = {}
    for mode in ("0", "1"):
        os.environ["OA_NN_GRID"] = mode
        with IcpEngine(0) as e:
            t0 = time.perf_counter()
            e.set_target(tgt)
            t_set = time.perf_counter() - t0
            e.set_source(src)
            e.set_matrices(mxa, mxb)
            idx, d2, cold = e.nn_search()
            res = e.run(iters=iters, thresh=0.5, early_exit=False)
            out[mode] = (idx, d2, res.matrix_world, cold, res.nn_ms_total / iters, res.loop_ms / iters, t_set)
    same = np.array_equal(out["0"][0], out["1"][0]) and np.array_equal(out["0"][1], out["1"][1])
    same_m = np.array_equal(out["0"][2], out["1"][2])
    print("%-28s nn identical=%s final matrix identical=%s | brute: cold %.3f ms, nn %.3f ms/iter, loop %.3f ms/iter | "
          "grid: cold %.3f ms, nn %.3f ms/iter, loop %.3f ms/iter, set_target %.1f ms"
          % (tag, same, same_m, out["0"][3], out["0"][4], out["0"][5], out["1"][3], out["1"][4], out["1"][5],
             1e3 * out["1"][6]), flush=True)
    return same and same_m


def main():
    from object_alignment_amd import synth
    ok = True
    ok &= run("C2 bunny 100k", *synth.c2_bunny_pair(100_000), iters=10)
    ok &= run("C3 random 1M", *synth.c3_random_pair(1_000_000), iters=6)
    src, tgt, mxa, mxb = synth.c3_random_pair(200_000)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.2, -0.1, 0.3]), [0.3, -0.2, 0.1])     # badly misaligned start
    ok &= run("random 200k misaligned", src, tgt, mxa, mxb, iters=6)
    src, tgt, mxa, mxb = synth.c2_bunny_pair(300_000)
    ok &= run("bunny 300k offset x1000", src + np.float32(1000.0), tgt + np.float32(1000.0), mxa, mxb, iters=6)
    print("ALL IDENTICAL" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

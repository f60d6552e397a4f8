#!/usr/bin/env python3
"""GPU box: the three search strategies against each other at sizes the CPU oracle cannot reach in seconds.

The brute-force kernels are the reference here (they are pinned against the oracle by tests/ and tools/fuzz_parity.py
at small sizes); grid and tree must return bit-identical (index, d2) per query -- 3-level trees, multi-million-cell
grids, long hand-over lists, clusters that blow the candidate budget.
Usage: python tools/fuzz_modes_large.py [trials] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def cloud(rng, kind, n):
    if kind == "uniform":
        return rng.uniform(-1, 1, size=(n, 3))
    if kind == "sphere":
        v = rng.normal(size=(n, 3))
        return v / np.maximum(1e-9, np.linalg.norm(v, axis=1, keepdims=True))
    if kind == "clusters":
        c = rng.normal(size=(32, 3))
        return c[rng.integers(0, 32, size=n)] + rng.normal(size=(n, 3)) * rng.choice([1e-4, 1e-2])
    if kind == "plane":
        p = rng.uniform(-1, 1, size=(n, 3))
        p[:, 2] = 0.01 * np.sin(7 * p[:, 0])
        return p
    raise ValueError(kind)


def main():
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for t in range(trials):
        surface = bool(t % 3 == 2)
        if surface:
            nu = int(rng.choice([150, 300]))
            tgt, tris = synth.lattice_surface_mesh(nu, 2 * nu)
            extra = rng.integers(0, len(tgt), size=(2000, 3)).astype(np.int32)          # a few huge random triangles
            tris = np.concatenate([tris, extra])
            ns = int(rng.choice([5000, 20000]))
            src = (synth.bunny_surface(ns, float(rng.uniform(0, 1))) * rng.uniform(0.9, 1.3)).astype(np.float32)
            kt = ks = "mesh"
        else:
            kt, ks = rng.choice(["uniform", "sphere", "clusters", "plane"]), rng.choice(["uniform", "sphere", "clusters", "plane"])
            nt = int(rng.choice([200_000, 1_000_000, 3_000_000]))
            ns = int(rng.choice([40_000, 300_000]))
            tgt = cloud(rng, kt, nt).astype(np.float32)
            src = (cloud(rng, ks, ns) * rng.uniform(0.7, 1.4) + rng.normal(size=3) * rng.choice([0.0, 0.3])).astype(np.float32)
            tris = None
        mxb = synth.rigid4(synth.rotation_from_rotvec(rng.normal(size=3) * 0.5) @ np.diag(rng.uniform(0.6, 1.6, size=3)), rng.normal(size=3))
        mxa = (mxb.astype(np.float64) @ synth.rigid4(synth.rotation_from_rotvec(rng.normal(size=3) * 0.05), rng.normal(size=3) * 0.02).astype(np.float64)).astype(np.float32)
        out = {}
        for mode in ("brute", "grid", "bvh"):
            with IcpEngine(0) as e:
                e.set_search_mode(mode)
                if tris is None:
                    e.set_target(tgt)
                else:
                    e.set_target_mesh(tgt, tris)
                e.set_source(src)
                e.set_matrices(mxa, mxb)
                idx, d2, ms = e.nn_search()
                thresh = float(np.sqrt(np.median(d2[np.isfinite(d2)])) * 1.5)
                A, B, ds = e.make_pairs(thresh, calc_stats=True)
                e.set_matrices(mxa, mxb)
                r = e.run(iters=3, thresh=thresh, target_d=1e-300)
                out[mode] = (idx, d2, A, B, r.step_M, r.step_K, ms)
        ok = True
        for mode in ("grid", "bvh"):
            for a, b in zip(out[mode][:6], out["brute"][:6]):
                ok = ok and np.array_equal(a, b)
        bad += 0 if ok else 1
        print("trial %d %s target %s %d source %s %d: %s  (search ms brute %.1f grid %.2f bvh %.2f; K %s)  [%.0f s]" % (
            t, "surface" if surface else "vertex", kt, len(tgt), ks, ns, "ok" if ok else "MISMATCH",
            out["brute"][6], out["grid"][6], out["bvh"][6], list(out["brute"][5]), time.time() - t0), flush=True)
    print("LARGE FUZZ DONE: %d trials, %d mismatches" % (trials, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GPU box: randomized differential test of the HIP path against the CPU oracle.

Every trial draws a cloud family, sizes, scales, offsets and two (possibly non-uniformly scaled) matrix_world
matrices, then checks, for search modes brute, grid and bvh:
  * oa_nn_search == oracle brute force (index and float32 d2, bit exact) -- vertex mode and surface mode;
  * oa_make_pairs == oracle make_pairs (A, B bit exact; d_stats to 1e-9);
  * for well-conditioned clouds, three iterations of the device loop == the oracle's loop (K per iteration equal,
    per-iteration transforms to 1e-7) -- seeds, search radius and the grid -> tree hand-over across iterations;
  * for the same clouds, the loop's PATH VARIANTS: the grid search that finishes its own leftovers and accumulates
    in its epilogue (OA_GRID_PATH=fast), the four-launch form (safe) and the adaptive choice, which depends on what the host
    last heard from the device, must leave bitwise the same per-iteration matrices -- one context and three shards, with and
    without the normal-angle test, vertex and surface targets.
Usage: python tools/fuzz_parity.py [trials] [seed]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def cloud(rng, kind, n):
    if kind == "uniform":
        return rng.uniform(-1, 1, size=(n, 3))
    if kind == "gauss":
        return rng.normal(size=(n, 3))
    if kind == "sphere":
        v = rng.normal(size=(n, 3))
        return v / np.maximum(1e-9, np.linalg.norm(v, axis=1, keepdims=True))
    if kind == "lattice":
        return rng.integers(-6, 7, size=(n, 3)) * 0.25
    if kind == "clusters":
        c = rng.normal(size=(8, 3)) * 2
        return c[rng.integers(0, 8, size=n)] + rng.normal(size=(n, 3)) * 0.01
    if kind == "line":
        t = rng.uniform(-1, 1, size=(n, 1))
        return t * np.array([[1.0, 0.5, -0.25]]) + rng.normal(size=(n, 3)) * 1e-4
    if kind == "plane":
        p = rng.uniform(-1, 1, size=(n, 3))
        p[:, 2] = 0.125
        return p
    raise ValueError(kind)


def rand_matrix(rng, scaled):
    from object_alignment_amd import synth
    R = synth.rotation_from_rotvec(rng.normal(size=3) * rng.choice([0.01, 0.3, 2.0]))
    S = np.diag(rng.uniform(0.5, 2.0, size=3) * rng.choice([1.0, 1.0, 1.0, -1.0], size=3)) if scaled else np.identity(3)   # now and then a mirrored axis
    M = np.identity(4)
    M[:3, :3] = R @ S
    M[:3, 3] = rng.normal(size=3) * rng.choice([0.0, 0.05, 1.0])
    return M.astype(np.float32)


def well_posed(A, B):
    """The rotation is only defined when the covariance has rank >= 2: skip the loop comparison when
    the pairs are degenerate (e.g. a tiny source cloud whose points all find the same target vertex)."""
    Ac, Bc = A - A.mean(axis=1, keepdims=True), B - B.mean(axis=1, keepdims=True)
    sv = np.linalg.svd(Bc @ Ac.T, compute_uv=False)
    # ... and when the cloud is resolved by float32 at all (a 1e-6-sized object 1 unit from the origin is ~10 quanta
    # across: its covariance is rounding noise and the iteration is chaotic)
    resolved = min(np.abs(Ac).max(), np.abs(Bc).max()) > 1e-3 * max(np.abs(A).max(), np.abs(B).max())
    return bool(sv[0] > 0 and sv[1] > 1e-2 * sv[0] and resolved)


def main():
    from object_alignment_amd.engine import IcpEngine
    from oracle import oracle as orc
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    kinds = ["uniform", "gauss", "sphere", "lattice", "clusters", "line", "plane"]
    engines = {m: IcpEngine(0) for m in ("brute", "grid", "bvh")}
    for m, e in engines.items():
        e.set_search_mode(m)
    variants = {}                                                  # (path, shards) -> grid-mode engine with that OA_GRID_PATH
    for path in ("fast", "safe", "adaptive"):
        if path != "adaptive":
            os.environ["OA_GRID_PATH"] = path
        for shards in (1, 3):
            variants[(path, shards)] = IcpEngine(0) if shards == 1 else IcpEngine(devices=[0] * shards)
            variants[(path, shards)].set_search_mode("grid")
        os.environ.pop("OA_GRID_PATH", None)
    bad = 0
    loops = 0
    variant_runs = 0
    t0 = time.time()
    for t in range(trials):
        kt, ks = rng.choice(kinds), rng.choice(kinds)
        nt = int(rng.choice([1, 3, 17, 500, 4096, 12000, 30000]))
        ns = int(rng.choice([1, 5, 300, 2500, 9000]))
        scale = float(rng.choice([1e-6, 1e-2, 1.0, 1.0, 50.0, 1e5]))
        offset = rng.normal(size=3) * float(rng.choice([0.0, 0.0, 10.0, 1e3])) * scale
        tgt = (cloud(rng, kt, nt) * scale + offset).astype(np.float32)
        src = (cloud(rng, ks, ns) * scale * rng.uniform(0.5, 1.5) + offset).astype(np.float32)
        mxb = rand_matrix(rng, scaled=bool(rng.integers(0, 2)))
        mxa = (mxb.astype(np.float64) @ rand_matrix(rng, False).astype(np.float64)).astype(np.float32)
        surface = nt >= 3 and bool(rng.integers(0, 2))
        tris = rng.integers(0, nt, size=(int(rng.choice([1, 50, 2000, 20000])), 3)).astype(np.int32) if surface else None
        imx2 = orc.mat4_inverted(mxb)
        cof = np.array([orc.mat4_mul_vec3(imx2, orc.mat4_mul_vec3(mxa, p)) for p in src], np.float32)
        if surface:
            ridx, _, rd2 = orc.nn_tri_brute(cof, tgt, tris)
        else:
            ridx, rd2 = orc.nn_brute(cof, tgt)
        thresh = float(np.sqrt(np.median(rd2[np.isfinite(rd2)])) * rng.uniform(0.5, 3.0)) if np.isfinite(rd2).any() else 1.0
        thresh = max(thresh, 1e-30) * float(np.abs(mxb[:3, :3]).max())
        stride = int(rng.choice([0, 1, 2, 5]))
        vlist = None if rng.integers(0, 2) else np.sort(rng.choice(ns, size=max(1, ns // 2), replace=False)).astype(np.int64)
        rA, rB, rds = orc.make_pairs(src, tgt, mxa, mxb, thresh, vlist=vlist, sample=stride, calc_stats=True, tris=tris)
        for mode, e in engines.items():
            if surface:
                e.set_target_mesh(tgt, tris)
            else:
                e.set_target(tgt)
            e.set_source(src)
            e.set_matrices(mxa, mxb)
            idx, d2, _ = e.nn_search()
            ok = np.array_equal(idx, ridx) and np.array_equal(d2, rd2)
            e.set_source(src, vlist=vlist, stride=stride)
            A, B, ds = e.make_pairs(thresh, calc_stats=True)
            ok_p = A.shape == rA.shape and np.array_equal(A, rA) and np.array_equal(B, rB)
            detail = "A/B differ" if not ok_p else ""
            if ok_p and rA.shape[1]:
                # population std: accumulated around the first pass's mean on the GPU, two-pass in the oracle
                ok_p = abs(ds[0] - rds[0]) <= 1e-9 * abs(rds[0]) and abs(ds[1] - rds[1]) <= 1e-8 * abs(rds[1]) + 1e-13 * abs(rds[0])
                detail = "d_stats %r vs %r" % (ds, rds) if not ok_p else ""
            ok_l, detail_l = True, ""
            loop_ran = False
            if ok and ok_p and kt in ("uniform", "gauss", "sphere") and ks in ("uniform", "gauss", "sphere") and rA.shape[1] >= 50 and nt >= 17 and well_posed(rA, rB):
                loops += 1
                loop_ran = True
                if mode == "brute":
                    ref_loop = orc.icp_run(src, tgt, mxa, mxb, iters=3, sample=stride, thresh=thresh, target_d=1e-300,
                                           vlist=vlist, tris=tris)
                e.set_matrices(mxa, mxb)
                try:
                    res = e.run(iters=3, thresh=thresh, target_d=1e-300)
                    sK, sM = res.step_K, res.step_M
                except ValueError:
                    sK, sM = np.array([-1]), np.zeros((1, 4, 4))
                n_it = len(ref_loop["step_K"])
                ok_l = len(sK) == n_it and np.array_equal(sK, ref_loop["step_K"])
                dM = float("nan")
                mag = float(np.abs(src).max()) + 1e-300
                if ok_l and n_it:
                    # rotation part to 1e-7; translation relative to the coordinates' magnitude
                    # first iteration: same pairs in, so the transforms agree to solver precision.  Later iterations start
                    # from a float32 matrix_world that may differ in its last bit (a float64 entry of M on a rounding
                    # boundary), which an object far from the origin amplifies: looser there.
                    dM = 0.0
                    for k in range(n_it):
                        dR = float(np.abs(sM[k, :3, :3] - ref_loop["step_M"][k, :3, :3]).max())
                        dT = float(np.abs(sM[k, :3, 3] - ref_loop["step_M"][k, :3, 3]).max()) / mag
                        dk = max(dR, dT)
                        ok_l = ok_l and dk <= (1e-9 if k == 0 else 1e-4)
                        dM = max(dM, dk)
                if mode == "brute":
                    brute_M = sM
                elif ok_l and mode == "grid":
                    ok_l = np.array_equal(sM, brute_M)           # brute force and grid both end in the canonical accumulation: bit for bit
                elif ok_l:
                    # the whole-shard tree search of a small shard accumulates per wave (another summation order): to rounding
                    # (relative to the coordinates' magnitude: a translation of 0.5 between clouds at 1e5 carries 1e-11 of noise)
                    # -- for the FIRST step.  Later steps start from a float32 matrix_world, and a float64 entry on a rounding
                    # boundary may round the other way in the two runs: one float32 ulp (6e-8) in, as much out (seen once in
                    # 500 trials: 2500 points against 500, second step 1.3e-11 apart)
                    tol = 1e-12 * max(1.0, float(np.abs(brute_M).max()), mag)
                    d_first = float(np.abs(sM[0] - brute_M[0]).max()) if len(sM) else 0.0
                    d_later = float(np.abs(sM[1:] - brute_M[1:]).max()) if len(sM) > 1 else 0.0
                    ok_l = bool(d_first <= tol and d_later <= 1e6 * tol)
                    if not ok_l:
                        dM = max(d_first, d_later)
                detail_l = "" if ok_l else "loop K %s vs %s, dM %.3g" % ([int(k) for k in sK], [int(k) for k in ref_loop["step_K"]], dM)
                if not ok_l and os.environ.get("FUZZ_DEBUG"):
                    np.set_printoptions(precision=9, linewidth=200)
                    print("stride", stride, "vlist", None if vlist is None else len(vlist), "thresh", thresh, "\nmxa\n", mxa, "\nmxb\n", mxb)
                    for k in range(min(len(sM), n_it)):
                        print("iter", k, "max |dM|", np.abs(sM[k] - ref_loop["step_M"][k]).max(), "\n", sM[k], "\n", ref_loop["step_M"][k])
            if loop_ran and ok_l and mode == "grid":
                # path variants of the loop: fast == safe == adaptive, bit for bit
                nrm_src = rng.normal(size=(ns, 3)).astype(np.float32)
                nrm_tgt = rng.normal(size=(nt, 3)).astype(np.float32)
                for with_normals in (False, True):
                    got = {}
                    for key, ve in variants.items():
                        if surface:
                            ve.set_target_mesh(tgt, tris)
                        else:
                            ve.set_target(tgt)
                        ve.set_source(src, vlist=vlist, stride=stride)
                        if with_normals:
                            ve.set_normals(nrm_src, None if surface else nrm_tgt, 75.0)
                        ve.set_matrices(mxa, mxb)
                        try:
                            r = ve.run(iters=3, thresh=thresh, target_d=1e-300)
                            got[key] = (r.step_K.copy(), r.step_M.copy(), r.matrix_world.copy())
                        except ValueError:
                            got[key] = (np.array([-1]), np.zeros((1, 4, 4)), np.zeros((4, 4), np.float32))
                        variant_runs += 1
                    for shards in (1, 3):
                        a = got[("fast", shards)]
                        for other in ("safe", "adaptive"):
                            b = got[(other, shards)]
                            same = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
                            if not same:
                                ok_l = False
                                detail_l = "path variants differ: fast vs %s, %d shard(s), normals %s" % (other, shards, with_normals)
                    if not with_normals and ok_l and not np.array_equal(got[("adaptive", 1)][1], sM):
                        ok_l = False
                        detail_l = "a second grid-mode engine differs from the first"
            if not (ok and ok_p and ok_l):
                bad += 1
                nd = int(np.count_nonzero(idx != ridx))
                print("MISMATCH trial %d mode %s surface %s kinds %s/%s ns %d nt %d scale %g: nn ok %s (%d idx differ) "
                      "pairs ok %s (K %d vs %d) %s loop ok %s %s" % (t, mode, surface, ks, kt, ns, nt, scale, ok, nd, ok_p, A.shape[1],
                                                        rA.shape[1], detail, ok_l, detail_l), flush=True)
        if (t + 1) % 10 == 0:
            print("trial %d/%d  mismatches %d  (%.0f s)" % (t + 1, trials, bad, time.time() - t0), flush=True)
    for e in list(engines.values()) + list(variants.values()):
        e.close()
    print("FUZZ DONE: %d trials, %d mismatches (%d three-iteration loop comparisons, %d path-variant loops: fast / safe / adaptive x 1 / 3 "
          "shards x normals off / on)" % (trials, bad, loops, variant_runs))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

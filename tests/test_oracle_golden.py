"""Pins the CPU oracle against fixtures produced by the imported reference
(tools/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def test_kabsch_cases_numpy_and_c(orc, golden_dir):
    g = _load(golden_dir, "kabsch")
    n = int(g["n_cases"])
    assert n >= 20
    for i in range(n):
        p = "c%02d_" % i
        A, B, M, sc = g[p + "A"], g[p + "B"], g[p + "M"], bool(g[p + "scale"])
        name = str(g[p + "name"])
        scale = max(1.0, float(np.abs(M).max()))
        Mn = orc.affine_matrix_from_points(A, B, shear=False, scale=sc, usesvd=True)
        assert np.abs(Mn - M).max() <= 1e-12 * scale, name
        Mc = orc.kabsch_c(A, B, scale=sc)
        # Jacobi vs LAPACK: R = U V^T is SVD-convention invariant; rank-2 (coplanar) inputs
        # are fixed by the right-handed completion
        tol = 5e-9 if ("K3" in name or "coplanar" in name) else 1e-11
        assert np.abs(Mc - M).max() <= tol * scale, (name, np.abs(Mc - M).max())
        assert abs(np.linalg.det(Mc[:3, :3] / (np.cbrt(np.linalg.det(Mc[:3, :3])))) - 1.0) < 1e-9


def test_kabsch_error_path(orc, golden_dir):
    g = _load(golden_dir, "kabsch")
    msg = str(g["valueerror_msg"])
    assert msg == "input arrays are of wrong shape or type"
    for fn in (orc.affine_matrix_from_points, orc.kabsch_c):
        with pytest.raises(ValueError, match=msg):
            fn(np.zeros((3, 2)), np.zeros((3, 2)))


def test_make_pairs_cases(orc, golden_dir):
    g = _load(golden_dir, "make_pairs")
    for i in range(int(g["n_cases"])):
        p = "c%02d_" % i
        for kd in (None, orc.KDTree(g[p + "tgt"])):
            A, B, ds = orc.make_pairs(g[p + "src"], g[p + "tgt"], g[p + "mx_align"], g[p + "mx_base"],
                                      float(g[p + "thresh"]), vlist=g[p + "vlist"], sample=int(g[p + "sample"]),
                                      calc_stats=bool(g[p + "calc_stats"]), kd=kd)
            assert A.shape == g[p + "A"].shape, str(g[p + "name"])
            assert np.array_equal(A, g[p + "A"]) and np.array_equal(B, g[p + "B"]), str(g[p + "name"])
            if bool(g[p + "calc_stats"]):
                assert np.allclose(ds, g[p + "d_stats"], rtol=1e-12, atol=1e-15)
            else:
                assert ds is None
    assert bool(g["thresh0_returns_none"])
    with pytest.raises(ValueError):
        orc.make_pairs(g["c00_src"][:8], g["c00_tgt"][:8], g["c00_mx_align"], g["c00_mx_base"], 0.0)


LOOPS = ["icp_loop_ico_10", "icp_loop_bumpy_converge", "icp_loop_bumpy_scale", "icp_loop_include",
         "icp_loop_exclude"]


@pytest.mark.parametrize("name", LOOPS)
def test_operator_loop(orc, golden_dir, name):
    g = _load(golden_dir, name)
    iters, frac, min_start, target_d, use_target, take_m, meth = g["prefs"]
    inc = [tuple(r) for r in g["include"]] if "include" in g.files else None
    exc = [tuple(r) for r in g["exclude"]] if "exclude" in g.files else None
    vlist = orc.build_vlist(len(g["src"]), inc, exc)
    assert np.array_equal(np.array(vlist, dtype=np.int64), g["vlist"])
    factor = round(1 / frac)                                   # operators/icp_align.py:89
    r = orc.icp_run(g["src"], g["tgt"], g["mx_align"], g["mx_base"], iters=int(iters), sample=factor,
                    thresh=min_start, target_d=target_d, use_target=bool(use_target), with_scale=(meth == 1.0),
                    vlist=vlist, kd=orc.KDTree(g["tgt"]))
    assert r["status"] == 0
    assert r["iters_done"] == int(g["iters_done"])
    assert r["converged"] == bool(g["converged"])
    assert np.array_equal(r["step_K"], g["step_K"])
    # per-iteration solve: oracle Jacobi vs the reference's LAPACK
    assert np.abs(r["step_M"] - g["step_M"]).max() < 1e-10
    # float32 matrix_world after every iteration.  Allow 1 float32 ulp: a 1e-16 difference in M can
    # flip the float32 rounding of new_mat
    assert np.abs(r["matrix_world"] - g["final_world"]).max() <= 2.5e-7
    if use_target:
        assert np.allclose(r["step_stats"], g["step_stats"], rtol=1e-9, atol=1e-12)
    # float32 matrix_world recorded by the reference after every iteration
    assert np.abs(_replay_world(orc, g["mx_align"], r["step_new"]) - g["step_world"]).max() <= 2.5e-7
    if g["m_final"].shape[0]:
        # take_m_with (operators/icp_align.py:123-127): same right-multiply on every "m_" object
        from object_alignment_amd import synth
        m0 = synth.rigid4(synth.rotation_from_rotvec([0.4, 0.1, 0.2]), [1.0, 2.0, 3.0])
        m = _replay_world(orc, m0, r["step_new"])[-1]
        assert np.abs(m - g["m_final"][0]).max() <= 1e-6


def _replay_world(orc, m0, step_new):
    out, m = [], np.asarray(m0, np.float32)
    for nm in step_new:
        m = orc.mat4_mul(m, nm)
        out.append(m)
    return np.array(out)


def test_kdtree_equals_brute_including_ties(orc):
    rng = np.random.default_rng(3)
    # integer lattice => massive exact ties; duplicated points => identical distances
    tgt = rng.integers(-4, 5, size=(4000, 3)).astype(np.float32)
    q = rng.integers(-5, 6, size=(3000, 3)).astype(np.float32) + np.float32(0.5)
    ib, db = orc.nn_brute(q, tgt)
    ik, dk = orc.KDTree(tgt).query(q)
    assert np.array_equal(ib, ik) and np.array_equal(db, dk)
    tgt = rng.normal(size=(20000, 3)).astype(np.float32)
    q = rng.normal(size=(5000, 3)).astype(np.float32)
    ib, db = orc.nn_brute(q, tgt)
    ik, dk = orc.KDTree(tgt).query(q)
    assert np.array_equal(ib, ik) and np.array_equal(db, dk)


def test_float32_matrix_ops(orc):
    rng = np.random.default_rng(5)
    for _ in range(50):
        a = rng.normal(size=(4, 4)).astype(np.float32)
        a[3] = [0, 0, 0, 1]
        inv = orc.mat4_inverted(a)
        assert inv.dtype == np.float32
        assert np.abs(inv.astype(np.float64) - np.linalg.inv(a.astype(np.float64))).max() < 1e-4 * np.abs(inv).max()
        v = rng.normal(size=3).astype(np.float32)
        mv = orc.mat4_mul_vec3(a, v)
        ref = (a.astype(np.float64) @ np.append(v.astype(np.float64), 1.0))[:3]
        assert np.abs(mv - ref).max() < 1e-5
    assert orc.vec3_length(np.array([3, 4, 12], np.float32)) == 13.0
    with pytest.raises(ValueError):
        orc.mat4_inverted(np.zeros((4, 4), np.float32))

"""Parity tests proper: the HIP path (through the C-ABI) against the CPU oracle and the
reference-derived golden fixtures.  Needs a real MI355X: run with `pytest -m gpu`."""
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

FROB_TOL = 1e-5            # north_star: final transform within 1e-5 Frobenius of the reference CPU path
F32_ULP = 2.5e-7           # one float32 ulp at magnitude ~1-2 (matrix_world entries)


@pytest.fixture(scope="module")
def eng():
    from object_alignment_amd.engine import IcpEngine
    e = IcpEngine(0)
    yield e
    e.close()


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def _cofind(orc, src, mxa, mxb):
    """co_find = imx2 @ (mx1 @ co) with the oracle's float32 arithmetic (functions/general.py:287)."""
    imx2 = orc.mat4_inverted(mxb)
    return np.array([orc.mat4_mul_vec3(imx2, orc.mat4_mul_vec3(mxa, p)) for p in src], np.float32)


# ------------------------------------------------------------------ extension really is the thing under test

def test_extension_loaded_and_gpu_present():
    from object_alignment_amd import _capi
    L = _capi.load()
    assert L.oa_device_count() >= 1
    with open("/proc/self/maps") as f:
        assert "liboa_icp.so" in f.read()


# ------------------------------------------------------------------ contract 2: affine_matrix_from_points

def test_affine_matrix_from_points_golden(golden_dir):
    from object_alignment_amd.functions import affine_matrix_from_points, calc_target_matrix
    g = _load(golden_dir, "kabsch")
    for i in range(int(g["n_cases"])):
        p = "c%02d_" % i
        name = str(g[p + "name"])
        A, B, M, sc = g[p + "A"], g[p + "B"], g[p + "M"], bool(g[p + "scale"])
        got = affine_matrix_from_points(A, B, shear=False, scale=sc, usesvd=True)
        tol = (5e-9 if ("K3" in name or "coplanar" in name) else 1e-10) * max(1.0, float(np.abs(M).max()))
        assert np.abs(got - M).max() <= tol, (name, np.abs(got - M).max())
        assert np.array_equal(calc_target_matrix(A, B, scale=sc), got)
    with pytest.raises(ValueError, match=str(g["valueerror_msg"])):
        affine_matrix_from_points(np.zeros((3, 2)), np.zeros((3, 2)), shear=False, scale=False)
    for i in range(int(g["n_horn"])):                       # usesvd=False: the reference's Horn-branch results
        p = "h%02d_" % i
        got = affine_matrix_from_points(g[p + "A"], g[p + "B"], shear=False, scale=bool(g[p + "scale"]), usesvd=False)
        assert np.abs(got - g[p + "M"]).max() < 1e-11      # the quaternion branch itself (round 6; rounds 1-5 served it with the SVD: 1e-9)
        svd = affine_matrix_from_points(g[p + "A"], g[p + "B"], shear=False, scale=bool(g[p + "scale"]), usesvd=True)
        assert 0.0 < np.abs(got - svd).max() < 1e-9        # another route to the same optimum: close, not the same bits


def test_affine_matrix_from_points_full_signature_golden(golden_dir):
    """The rest of the reference's signature through the GPU path: shear=True (the default arguments, incl. the doctest
    at functions/general.py:125-130), 2-D and 4-D point sets -- against outputs of the imported reference."""
    from object_alignment_amd.functions import affine_matrix_from_points
    g = _load(golden_dir, "affine_general")
    n = int(g["n_cases"])
    assert n >= 40
    for k in range(n):
        name = str(g["c%02d_name" % k])
        v0, v1, ref = g["c%02d_v0" % k], g["c%02d_v1" % k], g["c%02d_M" % k]
        M = affine_matrix_from_points(v0, v1, shear=bool(g["c%02d_shear" % k]), scale=bool(g["c%02d_scale" % k]))
        assert M.shape == ref.shape, name
        tol = 1e-9 * max(1.0, float(np.abs(ref).max()))
        assert np.abs(M - ref).max() <= tol, (name, float(np.abs(M - ref).max()))
    # the doctest itself, with default arguments, to the digits the reference prints
    M = affine_matrix_from_points([[0, 1031, 1031, 0], [0, 0, 1600, 1600]], [[675, 826, 826, 677], [55, 52, 281, 277]])
    want = np.array([[0.14549, 0.00062, 675.50008], [0.00048, 0.14094, 53.24971], [0.0, 0.0, 1.0]])
    assert np.allclose(M, want, atol=1e-5)
    with pytest.raises(ValueError):
        affine_matrix_from_points(np.zeros((2, 1)), np.zeros((2, 1)))


def test_kabsch_properties(eng):
    from object_alignment_amd import synth
    rng = np.random.default_rng(0)
    A = rng.uniform(-1, 1, size=(3, 1000))
    R = synth.rotation_from_rotvec([0, 0, np.deg2rad(15.0)])
    t = np.array([0.1, 0.2, 0.3])
    M = eng.kabsch(A, R @ A + t[:, None])
    assert np.abs(M[:3, :3] - R).max() < 1e-12 and np.abs(M[:3, 3] - t).max() < 1e-12
    M = eng.kabsch(A, 1.7 * (R @ A + t[:, None]), scale=True)
    assert abs(np.cbrt(np.linalg.det(M[:3, :3])) - 1.7) < 1e-12
    M = eng.kabsch(A, np.diag([1.0, 1.0, -1.0]) @ A)             # reflection input -> proper rotation
    assert abs(np.linalg.det(M[:3, :3]) - 1.0) < 1e-12
    # K = 1e6, far from the origin (pivoted one-pass accumulation)
    A = rng.uniform(-1, 1, size=(3, 1_000_000)) + np.array([[100.0], [-50.0], [25.0]])
    B = R @ A + t[:, None] + rng.normal(0, 1e-3, size=A.shape)
    from oracle import oracle as orc
    assert np.abs(eng.kabsch(A, B) - orc.affine_matrix_from_points(A, B)).max() < 1e-9


# ------------------------------------------------------------------ correspondence search: bit exact

@pytest.mark.parametrize("ns,nt", [(1, 1), (3, 5), (257, 1023), (2562, 2562), (5000, 4097), (20000, 30000)])
def test_nn_search_bit_exact(eng, orc, ns, nt):
    rng = np.random.default_rng(ns * 7919 + nt)
    tgt = rng.normal(size=(nt, 3)).astype(np.float32)
    src = rng.normal(size=(ns, 3)).astype(np.float32)
    mxa = np.identity(4, dtype=np.float32)
    mxa[:3, 3] = [0.01, -0.02, 0.03]
    mxb = np.identity(4, dtype=np.float32)
    eng.set_target(tgt)
    eng.set_source(src)
    eng.set_matrices(mxa, mxb)
    idx, d2, ms = eng.nn_search()
    ridx, rd2 = orc.nn_brute(_cofind(orc, src, mxa, mxb), tgt)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(d2, rd2)


def test_nn_search_ties_lowest_index(eng, orc):
    rng = np.random.default_rng(3)
    tgt = rng.integers(-4, 5, size=(6000, 3)).astype(np.float32)          # many exact duplicates / ties
    src = rng.integers(-5, 6, size=(3000, 3)).astype(np.float32) + np.float32(0.5)
    eye = np.identity(4, dtype=np.float32)
    eng.set_target(tgt)
    eng.set_source(src)
    eng.set_matrices(eye, eye)
    idx, d2, _ = eng.nn_search()
    ridx, rd2 = orc.nn_brute(src, tgt)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)


@pytest.mark.parametrize("R,splits,filt", [(1, 0, 1), (2, 3, 1), (8, 0, 1), (4, 7, 1), (4, 0, 0), (8, 5, 0)])
def test_nn_search_geometry_variants(orc, R, splits, filt, monkeypatch):
    """Points-per-thread, target-split (atomicMin merge) and filtered/unfiltered kernels give the same answers."""
    from object_alignment_amd.engine import IcpEngine
    monkeypatch.setenv("OA_NN_R", str(R))
    monkeypatch.setenv("OA_NN_FILTER", str(filt))
    if splits:
        monkeypatch.setenv("OA_NN_SPLITS", str(splits))
    rng = np.random.default_rng(R * 10 + splits)
    tgt = rng.normal(size=(9000, 3)).astype(np.float32)
    src = rng.normal(size=(4100, 3)).astype(np.float32)
    eye = np.identity(4, dtype=np.float32)
    with IcpEngine(0, experiments=(R == 8)) as e:                   # (8 points per thread is an OA_EXPERIMENTS instantiation)
        e.set_target(tgt)
        e.set_source(src)
        e.set_matrices(eye, eye)
        idx, d2, _ = e.nn_search()
    ridx, rd2 = orc.nn_brute(src, tgt)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)


@pytest.mark.parametrize("case", ["far_offset", "lattice_jitter", "query_outside", "flat_plane", "tiny_scale",
                                  "huge_scale", "seeded_second_pass"])
def test_nn_filter_adversarial(orc, case):
    """Inputs chosen to stress the conservative filter of k_nn_search_filtered: it may only ever skip targets that
    provably lose, so the answers must stay bit-identical to the oracle's brute force."""
    from object_alignment_amd.engine import IcpEngine
    import zlib
    rng = np.random.default_rng(zlib.crc32(case.encode()))       # fixed per case (str hashes are salted per process)
    nt, ns = 20000, 6000
    eye = np.identity(4, dtype=np.float32)
    mxa = eye.copy()
    if case == "far_offset":                      # cloud extent 1 at distance 1000 from the origin
        tgt = (rng.uniform(-0.5, 0.5, size=(nt, 3)) + [1000.0, -2000.0, 500.0]).astype(np.float32)
        src = (rng.uniform(-0.5, 0.5, size=(ns, 3)) + [1000.0, -2000.0, 500.0]).astype(np.float32)
    elif case == "lattice_jitter":                # near-ties everywhere: lattice + 1e-6 jitter
        tgt = (rng.integers(-8, 9, size=(nt, 3)) * 0.125 + rng.normal(0, 1e-6, size=(nt, 3))).astype(np.float32)
        src = (rng.integers(-8, 8, size=(ns, 3)) * 0.125 + 0.0625).astype(np.float32)
    elif case == "query_outside":                 # queries far outside the target bounding box
        tgt = rng.uniform(-1, 1, size=(nt, 3)).astype(np.float32)
        src = (rng.normal(size=(ns, 3)) * 50.0).astype(np.float32)
    elif case == "flat_plane":                    # degenerate bbox (zero thickness), duplicates
        tgt = rng.uniform(-1, 1, size=(nt, 3)).astype(np.float32)
        tgt[:, 2] = 0.25
        tgt[::7] = tgt[3]
        src = rng.uniform(-1, 1, size=(ns, 3)).astype(np.float32)
    elif case == "tiny_scale":
        tgt = (rng.uniform(-1, 1, size=(nt, 3)) * 1e-12).astype(np.float32)
        src = (rng.uniform(-1, 1, size=(ns, 3)) * 1e-12).astype(np.float32)
    elif case == "huge_scale":
        tgt = (rng.uniform(-1, 1, size=(nt, 3)) * 1e12).astype(np.float32)
        src = (rng.uniform(-1, 1, size=(ns, 3)) * 1e12).astype(np.float32)
    else:
        tgt = rng.uniform(-1, 1, size=(nt, 3)).astype(np.float32)
        src = rng.uniform(-1, 1, size=(ns, 3)).astype(np.float32)
    with IcpEngine(0) as e:
        e.set_target(tgt)
        e.set_source(src)
        e.set_matrices(mxa, eye)
        if case == "seeded_second_pass":
            # a full iteration stores every point's nearest index as the seed of the next search; then move the cloud
            e.iterate(thresh=10.0)
            mxa = e.matrix_world()
            mxa[:3, 3] += np.float32(0.013)
            e.set_matrices(mxa, eye)
        idx, d2, _ = e.nn_search()
    ridx, rd2 = orc.nn_brute(_cofind(orc, src, mxa, eye), tgt)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(d2, rd2)


@pytest.mark.parametrize("case", ["volume", "surface", "far_apart", "lattice_ties", "flat", "one_cell", "outside_box"])
def test_grid_search_identical_to_brute_force(orc, case):
    """k_nn_search_grid (+ list-mode finish) must return exactly what the brute-force kernel and the oracle return."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    rng = np.random.default_rng(len(case) * 1009)
    eye = np.identity(4, dtype=np.float32)
    mxa = eye.copy()
    if case == "volume":
        tgt = rng.uniform(-1, 1, size=(60000, 3)).astype(np.float32)
        src = (tgt[rng.permutation(60000)[:20000]] + rng.normal(0, 2e-3, size=(20000, 3))).astype(np.float32)
    elif case == "surface":
        src, tgt, mxa, _ = synth.c2_bunny_pair(40000)
        src = src[:15000]
    elif case == "far_apart":                     # most queries cannot be settled within r_max rings -> list mode
        tgt = rng.uniform(-1, 1, size=(30000, 3)).astype(np.float32)
        src = (rng.uniform(-1, 1, size=(5000, 3)) + [6.0, -3.0, 2.0]).astype(np.float32)
        tgt[:100] += np.float32(7.0)              # a far cluster stretches the grid
    elif case == "lattice_ties":
        tgt = rng.integers(-10, 11, size=(50000, 3)).astype(np.float32) * np.float32(0.1)
        src = (rng.integers(-10, 10, size=(8000, 3)) * 0.1 + 0.05).astype(np.float32)
    elif case == "flat":
        tgt = rng.uniform(-1, 1, size=(30000, 3)).astype(np.float32)
        tgt[:, 1] = -0.5
        src = rng.uniform(-1, 1, size=(6000, 3)).astype(np.float32)
    elif case == "one_cell":                      # all target vertices identical
        tgt = np.tile(np.array([[0.3, -0.2, 0.9]], np.float32), (9000, 1))
        src = rng.uniform(-1, 1, size=(3000, 3)).astype(np.float32)
    else:
        tgt = rng.uniform(-1, 1, size=(30000, 3)).astype(np.float32)
        src = (rng.normal(size=(6000, 3)) * 3.0).astype(np.float32)
    ref_idx, ref_d2 = orc.nn_brute(_cofind(orc, src, mxa, eye), tgt)
    for mode in ("brute", "grid", "bvh", "auto"):
        with IcpEngine(0) as e:
            e.set_search_mode(mode)
            e.set_target(tgt)
            e.set_source(src)
            e.set_matrices(mxa, eye)
            idx, d2, _ = e.nn_search()
            assert np.array_equal(idx, ref_idx), (case, mode)
            assert np.array_equal(d2, ref_d2), (case, mode)
            e.iterate(thresh=100.0)                    # seeds + moved pose: search again
            m2 = e.matrix_world()
            idx2, d22, _ = e.nn_search()
        r2, rd2 = orc.nn_brute(_cofind(orc, src, m2, eye), tgt)
        assert np.array_equal(idx2, r2) and np.array_equal(d22, rd2), (case, mode, "seeded")


@pytest.mark.parametrize("mode", ["brute", "grid", "bvh"])
def test_non_finite_coordinates(orc, mode):
    """NaN / Inf vertices never win and never poison their neighbours: same answers as the oracle (index -1 and
    d2 = +inf for a query that has no finite distance)."""
    from object_alignment_amd.engine import IcpEngine
    rng = np.random.default_rng(8)
    tgt = rng.uniform(-1, 1, size=(20000, 3)).astype(np.float32)
    src = rng.uniform(-1, 1, size=(5000, 3)).astype(np.float32)
    eye = np.identity(4, dtype=np.float32)
    src_bad = src.copy()
    src_bad[::97, 0] = np.nan
    src_bad[5::131, 2] = np.inf
    tgt_bad = tgt.copy()
    tgt_bad[::53, 1] = np.nan
    tgt_bad[7::211] = np.inf
    for s_, t_ in ((src_bad, tgt), (src, tgt_bad), (src_bad, tgt_bad)):
        ridx, rd2 = orc.nn_brute(s_, t_)
        with IcpEngine(0) as e:
            e.set_search_mode(mode)
            e.set_target(t_)
            e.set_source(s_)
            e.set_matrices(eye, eye)
            idx, d2, _ = e.nn_search()
            assert np.array_equal(idx, ridx)
            assert np.array_equal(d2, rd2)
            A, B, ds = e.make_pairs(0.5, calc_stats=True)
        rA, rB, rds = orc.make_pairs(s_, t_, eye, eye, 0.5, calc_stats=True)
        assert np.array_equal(A, rA) and np.array_equal(B, rB)


def test_device_resident_inputs_with_vlist(orc):
    """Source / target given as device tensors (on_device = 1) together with a host vlist and a stride."""
    import torch
    from object_alignment_amd.engine import IcpEngine
    rng = np.random.default_rng(12)
    tgt = rng.uniform(-1, 1, size=(9000, 3)).astype(np.float32)
    src = rng.uniform(-1, 1, size=(7000, 3)).astype(np.float32)
    vlist = np.sort(rng.choice(7000, size=4000, replace=False)).astype(np.int64)[::-1].copy()   # descending order
    eye = np.identity(4, dtype=np.float32)
    with IcpEngine(0) as e:
        e.set_target(torch.from_numpy(tgt).cuda())
        e.set_source(torch.from_numpy(src).cuda(), vlist=vlist, stride=3)
        e.set_matrices(eye, eye)
        A, B, ds = e.make_pairs(0.2, calc_stats=True)
    rA, rB, rds = orc.make_pairs(src, tgt, eye, eye, 0.2, vlist=vlist, sample=3, calc_stats=True)
    assert np.array_equal(A, rA) and np.array_equal(B, rB) and np.allclose(ds, rds, rtol=1e-9)


def test_nn_search_full_size_self_match(eng):
    """BASELINE config 3 size (1M <-> 1M): every point finds itself (property test, no oracle needed)."""
    rng = np.random.default_rng(1234)
    n = 1_000_000
    tgt = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    perm = rng.permutation(n)
    src = tgt[perm]
    eye = np.identity(4, dtype=np.float32)
    eng.set_target(tgt)
    eng.set_source(src)
    eng.set_matrices(eye, eye)
    idx, d2, ms = eng.nn_search()
    assert np.all(d2 == 0.0)
    # duplicates in a float32 uniform cloud are possible: idx must be <= perm (lowest index wins), equal coords
    assert np.all(idx <= perm)
    assert np.array_equal(tgt[idx], src)
    assert np.count_nonzero(idx != perm) < 100


# ------------------------------------------------------------------ contract 1: make_pairs

def test_make_pairs_golden(golden_dir):
    from object_alignment_amd.functions import make_pairs, GpuBVH, AlignObject
    g = _load(golden_dir, "make_pairs")
    for i in range(int(g["n_cases"])):
        p = "c%02d_" % i
        align = AlignObject(g[p + "src"], g[p + "mx_align"])
        base = AlignObject(g[p + "tgt"], g[p + "mx_base"])
        bvh = GpuBVH.FromObject(base, None)
        A, B, ds = make_pairs(align, base, bvh, g[p + "vlist"].tolist(), float(g[p + "thresh"]),
                              int(g[p + "sample"]), calc_stats=bool(g[p + "calc_stats"]))
        name = str(g[p + "name"])
        assert A.shape == g[p + "A"].shape, name
        assert np.array_equal(A, g[p + "A"]), name
        assert np.array_equal(B, g[p + "B"]), name
        if bool(g[p + "calc_stats"]):
            assert np.allclose(ds, g[p + "d_stats"], rtol=1e-9, atol=1e-13), name
        else:
            assert ds is None
    # thresh == 0 -> None, like the reference (functions/general.py:277)
    assert make_pairs(align, base, bvh, [0, 1, 2], 0.0) is None


def test_make_pairs_duck_typed_blender_objects(golden_dir, orc):
    """Same call the reference operator makes, with objects exposing .matrix_world / .data.vertices[i].co."""
    from object_alignment_amd.functions import make_pairs, GpuBVH
    g = _load(golden_dir, "make_pairs")
    p = "c04_"
    align = orc.MeshObject(g[p + "src"], g[p + "mx_align"])
    base = orc.MeshObject(g[p + "tgt"], g[p + "mx_base"])
    A, B, ds = make_pairs(align, base, GpuBVH.FromObject(base, None), g[p + "vlist"].tolist(), float(g[p + "thresh"]),
                          int(g[p + "sample"]), calc_stats=True)
    assert np.array_equal(A, g[p + "A"]) and np.array_equal(B, g[p + "B"])


def test_make_pairs_empty_and_ragged(eng, orc):
    rng = np.random.default_rng(9)
    tgt = rng.normal(size=(37, 3)).astype(np.float32)
    src = rng.normal(size=(301, 3)).astype(np.float32)
    eye = np.identity(4, dtype=np.float32)
    eng.set_target(tgt)
    eng.set_matrices(eye, eye)
    for vlist, stride in ((None, 0), (None, 7), (list(range(300, -1, -3)), 2), ([], 0), ([5], 0)):
        eng.set_source(src, vlist=vlist, stride=stride)
        A, B, ds = eng.make_pairs(0.8, calc_stats=True)
        rA, rB, rds = orc.make_pairs(src, tgt, eye, eye, 0.8, vlist=vlist, sample=stride, calc_stats=True)
        assert np.array_equal(A, rA) and np.array_equal(B, rB)
        if rA.shape[1]:
            assert np.allclose(ds, rds, rtol=1e-9, atol=1e-13)
        else:
            assert np.isnan(ds[0])


# ------------------------------------------------------------------ the operator loop

LOOPS = ["icp_loop_ico_10", "icp_loop_bumpy_converge", "icp_loop_bumpy_scale", "icp_loop_include",
         "icp_loop_exclude"]


def _settings_from(g):
    from object_alignment_amd.operators import IcpSettings
    iters, frac, min_start, target_d, use_target, take_m, meth = g["prefs"]
    return IcpSettings(icp_iterations=int(iters), sample_fraction=float(frac), min_start=float(min_start),
                       target_d=float(target_d), use_target=bool(use_target), take_m_with=bool(take_m),
                       align_meth=str(int(meth)))


@pytest.mark.parametrize("mode", ["brute", "grid", "bvh", "auto"])
@pytest.mark.parametrize("name", LOOPS)
def test_icp_align_run_golden(golden_dir, name, mode):
    """IcpAlign.run reproduces what the reference's execute() produced, iteration by iteration."""
    from object_alignment_amd.engine import IcpEngine
    from object_alignment_amd.operators import IcpAlign
    g = _load(golden_dir, name)
    with IcpEngine(0) as e:
        e.set_search_mode(mode)
        res = IcpAlign(_settings_from(g), engine=e).run(g["src"], g["tgt"], g["mx_align"], g["mx_base"], vlist=g["vlist"])
    assert res.iters_done == int(g["iters_done"])
    assert res.converged == bool(g["converged"])
    assert np.array_equal(res.step_K, g["step_K"])
    assert np.abs(res.step_M - g["step_M"]).max() < 1e-9
    assert np.abs(res.matrix_world - g["final_world"]).max() <= F32_ULP
    assert np.linalg.norm(res.matrix_world.astype(np.float64) - g["final_world"].astype(np.float64)) <= FROB_TOL
    if bool(g["prefs"][4]):
        assert np.allclose(res.step_stats, g["step_stats"], rtol=1e-8, atol=1e-12)


def _duck_scene(g, orc):
    """Duck-typed Blender context for a loop fixture: align / base objects with vertex groups, one m_ object."""
    from object_alignment_amd import synth

    def obj(xyz, mw, nm, inc=None, exc=None):
        o = orc.MeshObject(xyz, mw, nm)
        o.type = "MESH"
        groups, gi = {}, 1
        for v in o.data.vertices:
            v.groups = []
        for gname, members in (("icp_include", inc), ("icp_exclude", exc)):
            if members is None:
                continue
            groups[gname] = types.SimpleNamespace(name=gname, index=gi)
            for vi, w in members:
                o.data.vertices[int(vi)].groups.append(types.SimpleNamespace(group=gi, weight=float(np.float32(w))))
            gi += 1
        o.vertex_groups = list(groups.values())
        return o

    inc = [tuple(r) for r in g["include"]] if "include" in g.files else None
    exc = [tuple(r) for r in g["exclude"]] if "exclude" in g.files else None
    align = obj(g["src"], g["mx_align"], "align", inc, exc)
    base = obj(g["tgt"], g["mx_base"], "base")
    m0 = synth.rigid4(synth.rotation_from_rotvec([0.4, 0.1, 0.2]), [1.0, 2.0, 3.0])
    m_obj = types.SimpleNamespace(name="m_0", matrix_world=m0.copy())
    ctx = types.SimpleNamespace(object=align, selected_objects=[base, align],
                                scene=types.SimpleNamespace(objects=[align, base, m_obj]))
    return ctx, align, m_obj


@pytest.mark.parametrize("name", ["icp_loop_include", "icp_loop_exclude", "icp_loop_bumpy_scale"])
def test_operator_execute_duck_typed(golden_dir, orc, name):
    """OBJECT_OT_icp_align.execute on duck-typed Blender objects, vertex groups and m_ objects included."""
    from object_alignment_amd.operators import OBJECT_OT_icp_align, icp_align
    g = _load(golden_dir, name)
    st = _settings_from(g)
    for k, v in st.__dict__.items():
        setattr(icp_align.get_addon_preferences(), k, v)
    ctx, align, m_obj = _duck_scene(g, orc)
    assert OBJECT_OT_icp_align.poll(ctx)
    op = OBJECT_OT_icp_align()
    assert op.execute(ctx) == {"FINISHED"}
    got = np.array([[align.matrix_world[r][c] for c in range(4)] for r in range(4)], np.float32)
    assert np.abs(got - g["final_world"]).max() <= F32_ULP
    # the summary the reference printed at the end of its loop (operators/icp_align.py:145-158), line for line
    assert [str(x) for x in g["report"]] == op.last_report[:-1], (list(g["report"]), op.last_report)
    assert op.last_report[-1].startswith("Aligned obj in ")
    if g["m_final"].shape[0]:
        assert np.abs(np.asarray(m_obj.matrix_world) - g["m_final"][0]).max() <= 1e-6
    for k, v in icp_align.IcpSettings().__dict__.items():
        setattr(icp_align.get_addon_preferences(), k, v)


@pytest.mark.parametrize("name", ["icp_loop_include", "icp_loop_bumpy_converge"])
def test_feedback_operator_execute_runs_the_whole_loop(golden_dir, orc, name):
    """OBJECT_OT_icp_align_feedback.execute -- what bpy.ops.object.align_icp_redraw('EXEC_DEFAULT') and Redo run: the reference's
    second copy of the non-modal loop (operators/icp_align_feedback.py:130-235) -- ends where the reference's loop fixture ends."""
    from object_alignment_amd.operators import OBJECT_OT_icp_align_feedback, icp_align
    g = _load(golden_dir, name)
    st = _settings_from(g)
    for k, v in st.__dict__.items():
        setattr(icp_align.get_addon_preferences(), k, v)
    ctx, align, m_obj = _duck_scene(g, orc)
    op = OBJECT_OT_icp_align_feedback()
    assert op.poll(ctx)
    assert op.execute(ctx) == {"FINISHED"}
    got = np.array([[align.matrix_world[r][c] for c in range(4)] for r in range(4)], np.float32)
    assert np.abs(got - g["final_world"]).max() <= F32_ULP
    assert [str(x) for x in g["report"]] == op.last_report[:-1]
    assert op.last_result.iters_done == int(g["iters_done"]) and op.last_result.converged == bool(g["converged"])
    for k, v in icp_align.IcpSettings().__dict__.items():
        setattr(icp_align.get_addon_preferences(), k, v)


def test_gpubvh_find_nearest_is_the_reference_contract(orc):
    """`base_bvh.find_nearest(co) -> (co, normal, index, distance)`, the one method the reference asks of its BVH
    (functions/general.py:297): surface trees against the oracle's all-triangles search, vertex trees against its brute force."""
    from object_alignment_amd import synth
    from object_alignment_amd.functions import GpuBVH, AlignObject, make_pairs
    verts, tris = synth.lattice_surface_mesh(60, 120)
    rng = np.random.default_rng(9)
    pts = (synth.bunny_surface(40, 0.3) + rng.normal(0, 0.02, size=(40, 3))).astype(np.float32)
    base = AlignObject(verts, None, tris=tris)
    bvh = GpuBVH.FromObject(base, None)
    face, co1, d2 = orc.nn_tri_brute(pts, verts, tris)
    for k, p in enumerate(pts):
        loc, normal, index, dist = bvh.find_nearest(p)
        assert index == int(face[k]) and np.array_equal(loc, co1[k])
        assert abs(dist - float(np.sqrt(np.float64(d2[k])))) <= 1e-6 * max(1.0, dist)
        a, b, c = (verts[v].astype(np.float64) for v in tris[index])
        n = np.cross(a - b, b - c)
        assert np.abs(normal.astype(np.float64) - n / np.linalg.norm(n)).max() < 1e-5
    assert bvh.find_nearest(pts[0], distance=1e-9) == (None, None, None, None)       # nothing within the cap
    assert bvh.find_nearest([np.nan, 0.0, 0.0]) == (None, None, None, None)
    # a point cloud: nearest vertex, no normal
    vb = GpuBVH.FromObject(AlignObject(verts, None), None)
    ridx, rd2 = orc.nn_brute(pts, verts)
    for k, p in enumerate(pts[:10]):
        loc, normal, index, dist = vb.find_nearest(p)
        assert index == int(ridx[k]) and normal is None and np.array_equal(loc, verts[index])
    # the tree still serves make_pairs afterwards (find_nearest took the engine's source: it is bound again)
    src = synth.bunny_surface(3000, 0.5)
    A, B, _ = make_pairs(AlignObject(src, None), base, bvh, list(range(len(src))), 0.3)
    rA, rB, _ = orc.make_pairs(src, verts, np.identity(4, np.float32), np.identity(4, np.float32), 0.3, tris=tris)
    assert np.array_equal(A, rA) and np.array_equal(B, rB)


def test_modal_operator_ticks(golden_dir, orc):
    """OBJECT_OT_icp_align_feedback: timer ticks of `redraw_frequency` iterations reach the reference's final matrix."""
    from object_alignment_amd.operators import OBJECT_OT_icp_align_feedback, icp_align
    g = _load(golden_dir, "icp_loop_bumpy_converge")
    st = _settings_from(g)
    st.redraw_frequency = 3
    for k, v in st.__dict__.items():
        setattr(icp_align.get_addon_preferences(), k, v)
    align = orc.MeshObject(g["src"], g["mx_align"], "align")
    base = orc.MeshObject(g["tgt"], g["mx_base"], "base")
    align.type = base.type = "MESH"
    ctx = types.SimpleNamespace(object=align, selected_objects=[base, align], scene=types.SimpleNamespace(objects=[]))
    op = OBJECT_OT_icp_align_feedback()
    assert OBJECT_OT_icp_align_feedback.bl_idname == "object.align_icp_redraw"
    assert op.poll(ctx)
    assert op.invoke(ctx, None) == {"RUNNING_MODAL"}
    tick = types.SimpleNamespace(type="TIMER")
    assert op.modal(ctx, types.SimpleNamespace(type="MOUSEMOVE")) == {"PASS_THROUGH"}
    n_ticks = 0
    while op.modal(ctx, tick) == {"RUNNING_MODAL"}:
        n_ticks += 1
        assert n_ticks < 50
    assert op.converged and op.total_iters == int(g["iters_done"])
    got = np.array([[align.matrix_world[r][c] for c in range(4)] for r in range(4)], np.float32)
    assert np.abs(got - g["final_world"]).max() <= F32_ULP
    for k, v in icp_align.IcpSettings().__dict__.items():
        setattr(icp_align.get_addon_preferences(), k, v)


@pytest.mark.parametrize("mode", ["brute", "grid", "bvh"])
def test_runs_are_bitwise_reproducible(mode):
    """No float atomics anywhere on the result path: repeated runs (fresh contexts, fresh grid builds whose
    in-cell order is scheduling dependent) give bit-identical matrices, sums and statistics."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    src, tgt, mxa, mxb = synth.c2_bunny_pair(60_000)
    outs = []
    for _ in range(3):
        with IcpEngine(0) as e:
            e.set_search_mode(mode)
            e.set_target(tgt)
            e.set_source(src, stride=1)
            e.set_matrices(mxa, mxb)
            r = e.run(iters=12, thresh=0.5, early_exit=False)
            outs.append((r.matrix_world.tobytes(), r.step_M.tobytes(), r.step_stats.tobytes(), r.step_K.tobytes()))
    assert outs[0] == outs[1] == outs[2]


def test_step_mode_equals_fused_loop(golden_dir):
    from object_alignment_amd.engine import IcpEngine
    g = _load(golden_dir, "icp_loop_bumpy_converge")
    with IcpEngine(0) as e:
        e.set_target(g["tgt"])
        e.set_source(g["src"], stride=1)
        e.set_matrices(g["mx_align"], g["mx_base"])
        Ms = []
        for it in range(int(g["iters_done"])):
            M, st = e.iterate(thresh=0.5, target_d=0.01, use_target=True)
            Ms.append(M)
        assert st["converged"] == bool(g["converged"])
        assert np.abs(np.array(Ms) - g["step_M"]).max() < 1e-9
        assert np.abs(e.matrix_world() - g["final_world"]).max() <= F32_ULP


def test_error_paths(eng):
    from object_alignment_amd.operators import IcpAlign, IcpSettings
    from object_alignment_amd import _capi
    rng = np.random.default_rng(1)
    tgt = rng.normal(size=(500, 3)).astype(np.float32)
    src = rng.normal(size=(400, 3)).astype(np.float32) + np.float32(100.0)     # nothing within thresh
    eye = np.identity(4, dtype=np.float32)
    with pytest.raises(ValueError, match="input arrays are of wrong shape or type"):
        IcpAlign(IcpSettings(icp_iterations=5, sample_fraction=1.0)).run(src, tgt, eye, eye)
    with pytest.raises(TypeError):
        IcpAlign(IcpSettings(min_start=0.0)).run(src, tgt, eye, eye)
    with pytest.raises(ZeroDivisionError):
        IcpAlign(IcpSettings(sample_fraction=0.0)).run(src, tgt, eye, eye)
    with pytest.raises(_capi.OaError) as ei:
        eng.set_matrices(np.zeros((4, 4), np.float32), eye)
    assert ei.value.code == _capi.OA_E_SINGULAR
    with pytest.raises(_capi.OaError) as ei:
        eng.set_source(src, vlist=[0, 400])
    assert ei.value.code == _capi.OA_E_BAD_ARG


# ------------------------------------------------------------------ BASELINE configs against the oracle

def test_c2_bunny_100k_50_iters(orc):
    """Config 2: 100k <-> 100k, 50 iterations; final transform within 1e-5 Frobenius of the CPU path."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    src, tgt, mxa, mxb = synth.c2_bunny_pair(100_000)
    with IcpEngine(0) as e:
        e.set_search_mode("brute")
        e.set_target(tgt)
        e.set_source(src, stride=1)
        e.set_matrices(mxa, mxb)
        res = e.run(iters=50, thresh=0.5, target_d=0.01, use_target=True, early_exit=False)
    ref = orc.icp_run(src, tgt, mxa, mxb, iters=50, sample=1, thresh=0.5, target_d=1e-300, use_target=True,
                      kd=orc.KDTree(tgt))
    assert res.iters_done == 50 == ref["iters_done"]
    assert np.array_equal(res.step_K, ref["step_K"])
    err = np.linalg.norm(res.matrix_world.astype(np.float64) - ref["matrix_world"].astype(np.float64))
    assert err <= FROB_TOL, err
    assert np.abs(res.step_M - ref["step_M"]).max() < 1e-9
    with IcpEngine(0) as e:                              # grid search: bitwise the same run
        e.set_search_mode("grid")
        e.set_target(tgt)
        e.set_source(src, stride=1)
        e.set_matrices(mxa, mxb)
        res_g = e.run(iters=50, thresh=0.5, target_d=0.01, use_target=True, early_exit=False)
    assert np.array_equal(res_g.matrix_world, res.matrix_world) and np.array_equal(res_g.step_M, res.step_M)


_REF_CACHE = {}


def _c3_reference(orc):
    """BASELINE config 3 / 4: the pair and the oracle's 50 iterations (KD-tree, every host core), computed once per session."""
    if "c3" not in _REF_CACHE:
        from object_alignment_amd import synth
        src, tgt, mxa, mxb = synth.c3_random_pair(1_000_000)
        ref = orc.icp_run(src, tgt, mxa, mxb, iters=50, sample=1, thresh=0.5, target_d=1e-300, use_target=True, kd=orc.KDTree(tgt))
        assert ref["iters_done"] == 50
        _REF_CACHE["c3"] = (src, tgt, mxa, mxb, ref)
    return _REF_CACHE["c3"]


def _assert_loop_equals_oracle(res, ref, iters):
    assert res.iters_done == iters
    assert np.array_equal(res.step_K, ref["step_K"][:iters])                     # pairs per iteration: exact
    assert np.abs(res.step_M - ref["step_M"][:iters]).max() < 1e-9               # every iteration's affine_matrix_from_points
    if iters == ref["iters_done"]:
        err = np.linalg.norm(res.matrix_world.astype(np.float64) - ref["matrix_world"].astype(np.float64))
        assert err <= FROB_TOL, err


@pytest.mark.parametrize("mode", ["brute", "auto"])
def test_c3_random_1m_50_iters(orc, mode):
    """BASELINE config 3 as BASELINE states it: 1M <-> 1M, 50 iterations, against the oracle's loop -- the north-star kernel
    (brute force) and what a caller gets by default (AUTO: the grid search, whose late-iteration machinery -- settled seeds,
    the safe radii built after 8 iterations, the fused fast path -- only shows after the first dozen iterations and was only
    compared with brute force before)."""
    from object_alignment_amd.engine import IcpEngine
    src, tgt, mxa, mxb, ref = _c3_reference(orc)
    with IcpEngine(0) as e:
        e.set_search_mode(mode)
        e.set_target(tgt)
        e.set_source(src, stride=1)
        e.set_matrices(mxa, mxb)
        res = e.run(iters=50, thresh=0.5, target_d=0.01, use_target=True, early_exit=False)
        if mode == "auto":
            assert e.stat("safe_radii") == 1.0                                    # ... and really ran
            assert e.stat("fast_iterations") >= 25
    _assert_loop_equals_oracle(res, ref, 50)
    _REF_CACHE.setdefault("c3_world", res.matrix_world.copy())
    assert np.array_equal(res.matrix_world, _REF_CACHE["c3_world"])              # both modes: bitwise the same matrix


def test_c5_full_size_property():
    """BASELINE config 5's sizes (10M source, 2M target) on one GPU with the grid search: every source point is an
    exact copy of a target vertex, so it must find distance 0 and an index <= its own (lowest index on duplicates)."""
    from object_alignment_amd.engine import IcpEngine
    rng = np.random.default_rng(500)
    nt, ns = 2_000_000, 10_000_000
    tgt = rng.uniform(-1, 1, size=(nt, 3)).astype(np.float32)
    own = rng.integers(0, nt, size=ns)
    src = tgt[own]
    eye = np.identity(4, dtype=np.float32)
    with IcpEngine(0) as e:
        e.set_search_mode("grid")
        e.set_target(tgt)
        e.set_source(src)
        e.set_matrices(eye, eye)
        idx, d2, ms = e.nn_search()
    assert np.all(d2 == 0.0)
    assert np.all(idx <= own)
    assert np.array_equal(tgt[idx], src)


@pytest.mark.parametrize("how", ["auto", "grid", "multi8"])
def test_c5_full_size_masked_loop(orc, how):
    """BASELINE config 5 at FULL size as a loop: 10M source points on a surface, a seeded 10 % cap excluded through the
    `icp_exclude` mask semantics (operators/icp_align.py:67-76, via vlist_from_weights), 2M target, ten iterations --
    one context in AUTO and in grid mode, and the same job dealt to 8 shards by a multi-device context (all on this
    GPU) -- against the oracle's KD-tree loop: pairs per iteration exact, transform within 1e-5 Frobenius."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    from object_alignment_amd.operators.icp_align import vlist_from_weights
    src = synth.bunny_surface(10_000_000, 0.5)
    tgt = synth.bunny_surface(2_000_000, 0.0)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.003, -0.002, 0.004]), [0.002, -0.001, 0.0015])
    eye = np.identity(4, dtype=np.float32)
    rng = np.random.default_rng(500)
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    h = src.astype(np.float64) @ axis
    cap = np.nonzero(h > np.quantile(h, 0.9))[0]                                  # the seeded 10 % cap
    vlist = np.array(vlist_from_weights(len(src), exclude=[(int(v), 1.0) for v in cap]), dtype=np.int64)
    assert len(vlist) == len(src) - len(cap)
    iters = 10
    kw = dict(iters=iters, thresh=0.5, target_d=0.01, use_target=True, early_exit=False)
    eng = IcpEngine(devices=[0] * 8) if how == "multi8" else IcpEngine(0)
    try:
        if how != "multi8":
            eng.set_search_mode(how)
        eng.set_target(tgt)
        eng.set_source(src, vlist=vlist, stride=1)
        assert eng.n_selected == len(vlist)
        eng.set_matrices(mxa, eye)
        res = eng.run(**kw)
        if how == "auto":
            assert eng.stat("safe_radii") == 1.0                                  # the late-iteration path ran (built after 8)
    finally:
        eng.close()
    if "c5" not in _REF_CACHE:                                                    # the oracle's ten iterations, once for the three variants
        _REF_CACHE["c5"] = orc.icp_run(src, tgt, mxa, eye, iters=iters, sample=1, thresh=0.5, target_d=1e-300, use_target=True,
                                       vlist=vlist, kd=orc.KDTree(tgt))
    _assert_loop_equals_oracle(res, _REF_CACHE["c5"], iters)


def test_c5_shard_through_the_brute_force_kernel(orc):
    """The north-star kernel on BASELINE config 5's shape (only timed there so far): one eighth of the 9M selected
    points -- a compact slab, as a Morton shard is -- against the 2M-vertex target with the normal-angle test on,
    set_search_mode("brute") (k_nn_search_sorted: 1954 LDS tiles, 2.2e12 pairs per search), three iterations against the
    oracle's loop over the same vertex list: pairs per iteration exact, transforms to 1e-9."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    from object_alignment_amd.operators.icp_align import vlist_from_weights
    src, sn = synth.bunny_surface_with_normals(10_000_000, 0.5)
    tgt, tn = synth.bunny_surface_with_normals(2_000_000, 0.0)
    rng = np.random.default_rng(501)
    bent = rng.random(len(src)) < 0.33
    sn = sn.copy()
    sn[bent] += rng.normal(size=(int(bent.sum()), 3)).astype(np.float32) * np.float32(0.9)
    sn /= np.maximum(np.linalg.norm(sn, axis=1, keepdims=True), np.float32(1e-12))
    sn = sn.astype(np.float32)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.003, -0.002, 0.004]), [0.002, -0.001, 0.0015])
    eye = np.identity(4, dtype=np.float32)
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    h = src.astype(np.float64) @ axis
    cap = np.nonzero(h > np.quantile(h, 0.9))[0]
    vlist = np.array(vlist_from_weights(len(src), exclude=[(int(v), 1.0) for v in cap]), dtype=np.int64)
    order = vlist[np.argsort(src[vlist, 0], kind="stable")]                      # the selection along x: shard 1 of 8 is its second eighth
    n8 = len(vlist) // 8
    shard = np.sort(order[n8:2 * n8])
    assert 1_100_000 < len(shard) < 1_150_000
    iters = 3
    with IcpEngine(0) as e:
        e.set_search_mode("brute")
        e.set_target(tgt)
        e.set_source(src, vlist=shard, stride=1)
        e.set_normals(sn, tn, 45.0)
        e.set_matrices(mxa, eye)
        assert e.stat("brute_kernel") == 3.0
        res = e.run(iters=iters, thresh=0.5, target_d=0.01, use_target=True, early_exit=False)
    ref = orc.icp_run(src, tgt, mxa, eye, iters=iters, sample=1, thresh=0.5, target_d=1e-300, use_target=True,
                      vlist=shard, kd=orc.KDTree(tgt), normals=(sn, tn), max_angle_deg=45.0)
    assert 0.5 * len(shard) < ref["step_K"][0] < 0.95 * len(shard)               # the angle test bites
    _assert_loop_equals_oracle(res, ref, iters)


def test_c5_shaped_masked_sharded(orc):
    """BASELINE config 5's shape at 1/10 scale: 1M source on a surface, 200k target, a 10 % cap of the source excluded
    (icp_exclude semantics -> vlist), source split over two contexts; against the oracle's KD-tree loop."""
    import torch
    from object_alignment_amd import synth, _capi
    from object_alignment_amd.engine import IcpEngine
    from object_alignment_amd.operators.icp_align import vlist_from_weights
    src = synth.bunny_surface(1_000_000, 0.5)
    tgt = synth.bunny_surface(200_000, 0.0)
    R = synth.rotation_from_rotvec([0.03, -0.02, 0.04])
    mxa = synth.rigid4(R, [0.02, -0.01, 0.015])
    eye = np.identity(4, dtype=np.float32)
    cap = np.nonzero(src[:, 2] > np.quantile(src[:, 2], 0.9))[0]
    vlist = np.array(vlist_from_weights(len(src), exclude=[(int(v), 1.0) for v in cap]), dtype=np.int64)
    assert len(vlist) == len(src) - len(cap)
    iters = 4
    dev = torch.device("cuda:0")
    engs = [IcpEngine(0) for _ in range(2)]
    try:
        sums = [torch.zeros(_capi.OA_NSUMS, dtype=torch.float64, device=dev) for _ in range(2)]
        for r, e in enumerate(engs):
            e.set_stream(torch.cuda.current_stream().cuda_stream)
            e.set_target(tgt)
            e.set_source(src, vlist=vlist, stride=1, shard_index=r, shard_count=2)
            e.set_matrices(mxa, eye)
            e.run_begin(iters=iters, thresh=0.5, target_d=0.01, use_target=True, early_exit=False)
        for _ in range(iters):
            for r, e in enumerate(engs):
                e.iter_partial(sums[r].data_ptr())
            total = sums[0] + sums[1]
            for e in engs:
                e.iter_finish(total.data_ptr())
        res = [e.run_end() for e in engs]
    finally:
        for e in engs:
            e.close()
    ref = orc.icp_run(src, tgt, mxa, eye, iters=iters, sample=1, thresh=0.5, target_d=1e-300, use_target=True,
                      vlist=vlist, kd=orc.KDTree(tgt))
    assert np.array_equal(res[0].matrix_world, res[1].matrix_world)
    assert np.array_equal(res[0].step_K, ref["step_K"])
    err = np.linalg.norm(res[0].matrix_world.astype(np.float64) - ref["matrix_world"].astype(np.float64))
    assert err <= FROB_TOL, err
    assert np.abs(res[0].step_M - ref["step_M"]).max() < 1e-9


# ------------------------------------------------------------------ sharded (split-phase) path on one GPU

def test_two_shards_on_one_gpu_equal_unsharded(golden_dir):
    """Two contexts holding shard 0/2 and 1/2, sums added, identical solve: what 2 ranks would do."""
    import torch
    from object_alignment_amd.engine import IcpEngine, shard_bounds
    from object_alignment_amd import _capi
    g = _load(golden_dir, "icp_loop_bumpy_converge")
    dev = torch.device("cuda:0")
    kw = dict(iters=int(g["iters_done"]), thresh=0.5, target_d=0.01, use_target=True, early_exit=True)
    engs = [IcpEngine(0) for _ in range(2)]
    try:
        sums = [torch.zeros(_capi.OA_NSUMS, dtype=torch.float64, device=dev) for _ in range(2)]
        for r, e in enumerate(engs):
            e.set_stream(torch.cuda.current_stream().cuda_stream)
            e.set_target(g["tgt"])
            e.set_source(g["src"], stride=1, shard_index=r, shard_count=2)
            b, en = shard_bounds(len(g["src"]), r, 2)
            assert e.n_selected == en - b
            e.set_matrices(g["mx_align"], g["mx_base"])
            e.run_begin(**kw)
        for it in range(kw["iters"]):
            for r, e in enumerate(engs):
                e.iter_partial(sums[r].data_ptr())
            total = sums[0] + sums[1]                       # stands in for the all-reduce
            for e in engs:
                e.iter_finish(total.data_ptr())
        res = [e.run_end() for e in engs]
    finally:
        for e in engs:
            e.close()
    assert np.array_equal(res[0].matrix_world, res[1].matrix_world)
    assert res[0].iters_done == int(g["iters_done"]) and res[0].converged == bool(g["converged"])
    assert np.array_equal(res[0].step_K, g["step_K"])
    assert np.abs(res[0].step_M - g["step_M"]).max() < 1e-9
    assert np.abs(res[0].matrix_world - g["final_world"]).max() <= F32_ULP


def test_c4_shaped_eight_shards_one_gpu(orc):
    """BASELINE config 4's shape for its 50 iterations: 1M <-> 1M with the source cut into 8 shards of 125k (here 8 contexts on
    one GPU, brute-force kernel, the all-reduce replaced by a tensor sum: the split-phase loop one process per GPU drives),
    against the oracle's loop."""
    import torch
    from object_alignment_amd import _capi
    from object_alignment_amd.engine import IcpEngine
    src, tgt, mxa, mxb, ref = _c3_reference(orc)
    world, iters = 8, 50
    dev = torch.device("cuda:0")
    engs = [IcpEngine(0) for _ in range(world)]
    try:
        sums = [torch.zeros(_capi.OA_NSUMS, dtype=torch.float64, device=dev) for _ in range(world)]
        tgt_dev = torch.from_numpy(tgt).to(dev)
        for r, e in enumerate(engs):
            e.set_search_mode("brute")
            e.set_stream(torch.cuda.current_stream().cuda_stream)
            e.set_target(tgt_dev)
            e.set_source(src, stride=1, shard_index=r, shard_count=world)
            assert e.n_selected == 125_000
            e.set_matrices(mxa, mxb)
            e.run_begin(iters=iters, thresh=0.5, target_d=0.01, use_target=True, early_exit=False)
        for _ in range(iters):
            for r, e in enumerate(engs):
                e.iter_partial(sums[r].data_ptr())
            total = torch.stack(sums).sum(0)
            for e in engs:
                e.iter_finish(total.data_ptr())
        res = [e.run_end() for e in engs]
    finally:
        for e in engs:
            e.close()
    for r in res[1:]:
        assert np.array_equal(r.matrix_world, res[0].matrix_world)
    _assert_loop_equals_oracle(res[0], ref, iters)


def test_c4_shaped_multi_device_context_50_iters(orc):
    """The same job through ONE multi-device context (oa_create_multi, eight children on this GPU, the in-library exchange
    through the mailboxes, AUTO search: Morton-range shards, grid search, safe radii, fused path) for its 50 iterations."""
    from object_alignment_amd.engine import IcpEngine
    src, tgt, mxa, mxb, ref = _c3_reference(orc)
    with IcpEngine(devices=[0] * 8) as eng:
        eng.set_target(tgt)
        eng.set_source(src, stride=1)
        eng.set_matrices(mxa, mxb)
        res = eng.run(iters=50, thresh=0.5, target_d=0.01, use_target=True, early_exit=False)
        assert eng.stat("enqueued_min") == eng.stat("enqueued_max") == 50
        assert eng.stat("safe_radii") == 1.0
    _assert_loop_equals_oracle(res, ref, 50)


def test_more_shards_than_points(orc):
    """Empty shards (shard_count > selected points) contribute zero sums and do not disturb the others."""
    import torch
    from object_alignment_amd.engine import IcpEngine
    from object_alignment_amd import _capi
    rng = np.random.default_rng(4)
    tgt = rng.uniform(-1, 1, size=(500, 3)).astype(np.float32)
    src = (tgt[:10] + np.float32(0.01)).astype(np.float32)
    eye = np.identity(4, dtype=np.float32)
    world = 16                                   # 10 points over 16 shards: 6 contexts hold nothing
    dev = torch.device("cuda:0")
    engs = [IcpEngine(0) for _ in range(world)]
    try:
        sums = [torch.zeros(_capi.OA_NSUMS, dtype=torch.float64, device=dev) for _ in range(world)]
        for r, e in enumerate(engs):
            e.set_stream(torch.cuda.current_stream().cuda_stream)
            e.set_target(tgt)
            e.set_source(src, shard_index=r, shard_count=world)
            e.set_matrices(eye, eye)
            e.run_begin(iters=3, thresh=0.5, use_target=True, early_exit=False)
        assert sum(e.n_selected for e in engs) == 10 and min(e.n_selected for e in engs) == 0
        for _ in range(3):
            for r, e in enumerate(engs):
                e.iter_partial(sums[r].data_ptr())
            total = torch.stack(sums).sum(0)
            for e in engs:
                e.iter_finish(total.data_ptr())
        res = [e.run_end() for e in engs]
    finally:
        for e in engs:
            e.close()
    ref = orc.icp_run(src, tgt, eye, eye, iters=3, sample=1, thresh=0.5, target_d=1e-300, use_target=True)
    for r in res:
        assert np.array_equal(r.matrix_world, res[0].matrix_world)
    assert np.array_equal(res[0].step_K, ref["step_K"])
    assert np.abs(res[0].step_M - ref["step_M"]).max() < 1e-9


def test_run_sharded_world1_torch_stream(golden_dir):
    """distributed.run_sharded with world_size 1 (no collective) on torch's current stream."""
    import torch
    from object_alignment_amd.engine import IcpEngine
    from object_alignment_amd.distributed import EngineShard, run_sharded, new_sums_tensor
    g = _load(golden_dir, "icp_loop_bumpy_converge")
    with IcpEngine(0) as e:
        e.set_target(torch.from_numpy(g["tgt"]).cuda())          # device-resident inputs
        e.set_source(torch.from_numpy(g["src"]).cuda(), stride=1)
        e.set_matrices(g["mx_align"], g["mx_base"])
        sums = new_sums_tensor(torch.device("cuda:0"))
        res = run_sharded(EngineShard(e, iters=30, thresh=0.5, target_d=0.01, use_target=True, early_exit=True),
                          30, sums, world_size=1)
    assert res.iters_done == int(g["iters_done"]) and res.converged
    assert np.abs(res.matrix_world - g["final_world"]).max() <= F32_ULP


# ------------------------------------------------------------------ surface mode (closest point on triangles, D2)

def _surface_cases():
    from object_alignment_amd import synth
    rng = np.random.default_rng(77)
    v1, t1 = synth.bumpy_icosphere_mesh(4)                       # 2562 vertices, 5120 triangles
    q1 = (synth.bumpy_icosphere(4)[::2] * np.float32(1.03) + np.float32(0.01)).astype(np.float32)
    v2, t2 = synth.lattice_surface_mesh(120, 240)                # 28800 vertices, 57120 triangles
    q2 = (synth.bunny_surface(12000, 0.5) + rng.normal(0, 0.01, size=(12000, 3))).astype(np.float32)
    q3 = (rng.normal(size=(3000, 3)) * 3.0).astype(np.float32)   # far outside: list-mode finish
    v4 = rng.integers(-5, 6, size=(400, 3)).astype(np.float32)   # integer lattice: exact ties, degenerate triangles
    t4 = rng.integers(0, 400, size=(3000, 3)).astype(np.int32)
    q4 = (rng.integers(-5, 5, size=(2000, 3)) + 0.5).astype(np.float32)
    return {"ico": (v1, t1, q1), "lattice": (v2, t2, q2), "far": (v2, t2, q3), "ties": (v4, t4, q4)}


@pytest.mark.parametrize("mode", ["brute", "grid", "bvh", "auto"])
@pytest.mark.parametrize("case", ["ico", "lattice", "far", "ties"])
def test_surface_search_bit_exact(orc, case, mode):
    """Nearest triangle index and float32 squared distance equal the oracle's brute force over all triangles."""
    from object_alignment_amd.engine import IcpEngine
    verts, tris, q = _surface_cases()[case]
    eye = np.identity(4, dtype=np.float32)
    face, co1, d2 = orc.nn_tri_brute(q, verts, tris)
    with IcpEngine(0) as e:
        e.set_search_mode(mode)
        e.set_target_mesh(verts, tris)
        e.set_source(q)
        e.set_matrices(eye, eye)
        idx, gd2, _ = e.nn_search()
        assert np.array_equal(gd2, d2), (case, mode)
        assert np.array_equal(idx, face), (case, mode)
        e.iterate(thresh=100.0)                                   # seeds (previous triangle) + new pose
        m2 = e.matrix_world()
        idx2, gd22, _ = e.nn_search()
    f2, _, d22 = orc.nn_tri_brute(_cofind(orc, q, m2, eye), verts, tris)
    assert np.array_equal(idx2, f2) and np.array_equal(gd22, d22), (case, mode, "seeded")


def _exact_point_triangle_distance(p, a, b, c):
    """float64 distance from points p (n, 3) to triangles (a, b, c) (m, 3) each -> (n, m); written independently of the
    search's Ericson routine: the closest point is either the projection onto the plane (when it falls inside the
    triangle) or lies on one of the three edges."""
    def seg(p, u, v):                                             # (n, 1, 3) x (1, m, 3) -> (n, m)
        d = v - u
        dd = np.einsum("ijk,ijk->ij", d, d)
        t = np.einsum("ijk,ijk->ij", p - u, d) / np.where(dd > 0, dd, 1.0)
        t = np.clip(np.where(dd > 0, t, 0.0), 0.0, 1.0)
        q = u + t[..., None] * d
        return np.linalg.norm(p - q, axis=-1)
    P = p[:, None, :]
    A, B, C = a[None], b[None], c[None]
    out = np.minimum(np.minimum(seg(P, A, B), seg(P, B, C)), seg(P, C, A))
    n = np.cross(b - a, c - a)
    nn = np.linalg.norm(n, axis=1)
    ok = nn > 0
    nu = n / np.where(ok, nn, 1.0)[:, None]
    dist = np.einsum("ijk,jk->ij", P - A, nu)                     # signed plane distance
    q = P - dist[..., None] * nu[None]                            # projection onto the plane
    def side(u, v):                                               # inside test: (v - u) x (q - u) . n >= 0
        return np.einsum("ijk,jk->ij", np.cross(np.broadcast_to(v - u, q.shape), q - u), nu) >= 0
    inside = side(A, B) & side(B, C) & side(C, A) & ok[None]
    return np.where(inside, np.minimum(np.abs(dist), out), out)


@pytest.mark.parametrize("case", ["ico", "lattice"])
def test_surface_winner_is_the_exact_nearest_triangle(case):
    """An independent check of the correspondence itself (no shared code with the search or the oracle): for every query
    the triangle the search returns is, in float64 arithmetic written differently, within the documented delta of the
    exact minimum over ALL triangles, and the float32 distance it reports is that triangle's exact distance within
    delta.  What stays unpinned against Blender after this is only the order of exact ties."""
    from object_alignment_amd.engine import IcpEngine
    v, t, q = _surface_cases()[case]
    if case == "lattice":
        q = q[::8]                                                # 1500 queries x 57k triangles in float64
    eye = np.identity(4, dtype=np.float32)
    with IcpEngine(0) as e:
        e.set_target_mesh(v, t)
        e.set_source(q)
        e.set_matrices(eye, eye)
        idx, d2, _ = e.nn_search()
    V = v.astype(np.float64)
    a, b, c = V[t[:, 0]], V[t[:, 1]], V[t[:, 2]]
    Q = q.astype(np.float64)
    scale = float(np.abs(V).max())
    worst_gap = worst_err = 0.0
    for s0 in range(0, len(Q), 64):
        D = _exact_point_triangle_distance(Q[s0:s0 + 64], a, b, c)              # (64, n_tris)
        rows = np.arange(D.shape[0])
        won = D[rows, idx[s0:s0 + 64]]
        delta = 64.0 * 2.0 ** -24 * (scale + np.abs(Q[s0:s0 + 64]).sum(axis=1))   # docs/HISTORY.md 4.5: float32 evaluation error
        gap = won - D.min(axis=1)
        assert np.all(gap <= 2.0 * delta), float((gap / delta).max())
        err = np.abs(np.sqrt(d2[s0:s0 + 64].astype(np.float64)) - won)
        assert np.all(err <= delta), float((err / delta).max())
        worst_gap = max(worst_gap, float((gap / delta).max()))
        worst_err = max(worst_err, float((err / delta).max()))
    print("surface winner check (%s): worst gap %.3g delta, worst distance error %.3g delta" % (case, worst_gap, worst_err))


@pytest.mark.parametrize("lanes", ["1", "2", "4"])
def test_surface_grid_lane_variants_and_ragged_last_wave(orc, lanes, monkeypatch):
    """k_tri_search_grid<L> for every L on a query count that leaves the last wave partly empty: phase 2 of the
    triangle grid search deals its pool of survivors to all 64 lanes of a wave, so the lanes past the last query must
    keep working (a round-2 bug: they left, and took their share of the candidates with them)."""
    from object_alignment_amd.engine import IcpEngine
    v, t, q = _surface_cases()["lattice"]
    q = np.concatenate([q, (q[:37] * np.float32(1.002)).astype(np.float32)])        # 12037 queries: ragged in every L
    eye = np.identity(4, dtype=np.float32)
    monkeypatch.setenv("OA_GRID_LANES", lanes)
    with IcpEngine(0) as e:
        e.set_search_mode("grid")
        e.set_target_mesh(v, t)
        e.set_source(q)
        e.set_matrices(eye, eye)
        idx, d2, _ = e.nn_search()
        idx2, d22, _ = e.nn_search()                                                 # seeded
    ridx, _, rd2 = orc.nn_tri_brute(q, v, t)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)
    assert np.array_equal(idx2, ridx) and np.array_equal(d22, rd2)


def test_surface_make_pairs_and_loop(orc):
    """make_pairs / the ICP loop in surface mode against the oracle's mesh-mode restatement."""
    from object_alignment_amd import synth
    from object_alignment_amd.functions import make_pairs, GpuBVH, AlignObject
    from object_alignment_amd.operators import IcpAlign, IcpSettings
    verts, tris = synth.lattice_surface_mesh(80, 160)
    src = synth.bunny_surface(20000, 0.5)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.05, -0.04, 0.06]), [0.03, -0.02, 0.02])
    mxb = synth.rigid4(synth.rotation_from_rotvec([0.2, 0.1, -0.3]) @ np.diag([1.2, 0.9, 1.1]), [0.1, 0.0, -0.1])
    align = AlignObject(src, mxb @ mxa)
    base = AlignObject(verts, mxb, tris=tris)
    bvh = GpuBVH.FromObject(base, None)
    assert bvh.tris is not None
    A, B, ds = make_pairs(align, base, bvh, list(range(len(src))), 0.3, 2, calc_stats=True)
    rA, rB, rds = orc.make_pairs(src, verts, mxb @ mxa, mxb, 0.3, sample=2, calc_stats=True, tris=tris)
    assert np.array_equal(A, rA) and np.array_equal(B, rB)
    assert np.allclose(ds, rds, rtol=1e-9, atol=1e-13)
    # the surface is at most as far as the nearest vertex
    Av, Bv, dsv = make_pairs(align, base, GpuBVH.FromObject(base, None, surface=False), list(range(len(src))), 0.3, 2,
                             calc_stats=True)
    assert ds[0] < dsv[0]
    res = IcpAlign(IcpSettings(icp_iterations=12, sample_fraction=1.0)).run(src, verts, mxb @ mxa, mxb, target_tris=tris)
    ref = orc.icp_run(src, verts, mxb @ mxa, mxb, iters=12, sample=1, tris=tris)
    assert res.iters_done == ref["iters_done"] and res.converged == ref["converged"]
    assert np.array_equal(res.step_K, ref["step_K"])
    assert np.abs(res.step_M - ref["step_M"]).max() < 1e-9
    assert np.abs(res.matrix_world - ref["matrix_world"]).max() <= F32_ULP


def test_surface_bad_triangle_index(eng):
    from object_alignment_amd import _capi
    v = np.zeros((4, 3), np.float32)
    with pytest.raises(_capi.OaError) as ei:
        eng.set_target_mesh(v, np.array([[0, 1, 4]], np.int32))
    assert ei.value.code == _capi.OA_E_BAD_ARG
    eng.set_target(np.random.default_rng(0).normal(size=(10, 3)).astype(np.float32))   # back to vertex mode


# ------------------------------------------------------------------ normal-angle rejection (extension)

@pytest.mark.parametrize("surface", [False, True])
def test_normal_angle_rejection_extension(orc, surface):
    """BASELINE config 5 names a normal-angle outlier rejection the reference does not have; the build's extension is
    pinned against the oracle's restatement of the same test (parity unpinned against the reference by construction)."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    rng = np.random.default_rng(5)
    src, src_n = synth.bunny_surface_with_normals(20000, 0.5)
    flip = rng.random(len(src)) < 0.2
    src_n[flip] = -src_n[flip]                                   # 20 % of the source normals point inwards
    src_n += rng.normal(0, 0.2, size=src_n.shape).astype(np.float32)
    mxb = synth.rigid4(synth.rotation_from_rotvec([0.2, 0.1, -0.3]) @ np.diag([1.2, 0.9, 1.1]), [0.1, 0.0, -0.1])
    mxa = (mxb.astype(np.float64) @ synth.rigid4(synth.rotation_from_rotvec([0.05, -0.04, 0.06]), [0.03, -0.02, 0.02],
                                                  np.float64)).astype(np.float32)
    if surface:
        tgt, tris = synth.lattice_surface_mesh(80, 160)
        tgt_n = None
    else:
        tgt, tgt_n = synth.bunny_surface_with_normals(30000, 0.0)
        tris = None
    kw = dict(tris=tris, normals=(src_n, tgt_n), max_angle_deg=50.0)
    rA, rB, rds = orc.make_pairs(src, tgt, mxa, mxb, 0.3, sample=1, calc_stats=True, **kw)
    rA0, _, _ = orc.make_pairs(src, tgt, mxa, mxb, 0.3, sample=1, calc_stats=True, tris=tris)
    assert 0.5 * rA0.shape[1] < rA.shape[1] < 0.9 * rA0.shape[1]          # the test really rejects (~20-25 %)
    with IcpEngine(0) as e:
        if surface:
            e.set_target_mesh(tgt, tris)
        else:
            e.set_target(tgt)
        e.set_source(src, stride=1)
        e.set_normals(src_n, tgt_n, 50.0)
        e.set_matrices(mxa, mxb)
        A, B, ds = e.make_pairs(0.3, calc_stats=True)
        assert np.array_equal(A, rA) and np.array_equal(B, rB)
        assert np.allclose(ds, rds, rtol=1e-9, atol=1e-13)
        res = e.run(iters=8, thresh=0.3, use_target=True, early_exit=False)
        ref = orc.icp_run(src, tgt, mxa, mxb, iters=8, sample=1, thresh=0.3, target_d=1e-300, **kw)
        assert np.array_equal(res.step_K, ref["step_K"])
        assert np.abs(res.step_M - ref["step_M"]).max() < 1e-9
        assert np.abs(res.matrix_world - ref["matrix_world"]).max() <= F32_ULP
        e.set_normals(None)                                       # switched off again
        A2, _, _ = e.make_pairs(0.3)
        e.set_matrices(mxa, mxb)
    rA2, _, _ = orc.make_pairs(src, tgt, res.matrix_world, mxb, 0.3, sample=1, tris=tris)
    assert A2.shape == rA2.shape


@pytest.mark.gpu
@pytest.mark.parametrize("surface", [False, True])
@pytest.mark.parametrize("mode", ["grid", "bvh", "auto"])
def test_search_radius_does_not_change_pairs(orc, mode, surface, monkeypatch):
    """The grid / tree searches stop at the radius beyond which `dist < thresh` (general.py:300) cannot hold
    (DevState::cut_a): pairs, statistics and the loop must equal the oracle's exact search, for partial overlaps,
    a non-uniformly scaled base matrix and thresholds from "almost nothing passes" to "everything passes"."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    verts, tris = synth.lattice_surface_mesh(90, 180)
    keep = verts[:, 2] > -0.2                                      # the base misses a cap: those queries are far
    remap = np.cumsum(keep) - 1
    tris = remap[tris[keep[tris].all(axis=1)]].astype(np.int32)
    verts = verts[keep]
    src = synth.bunny_surface(40000, 0.3)
    mxb = synth.rigid4(synth.rotation_from_rotvec([0.3, -0.2, 0.1]) @ np.diag([0.4, 1.0, 2.5]), [5.0, -3.0, 1.0])
    mxa = (mxb.astype(np.float64) @ synth.rigid4(synth.rotation_from_rotvec([0.03, 0.02, -0.04]), [0.02, 0.01, -0.015]).astype(np.float64)).astype(np.float32)
    kw = dict(tris=tris) if surface else {}
    with IcpEngine(0) as e:
        e.set_search_mode(mode)
        if surface:
            e.set_target_mesh(verts, tris)
        else:
            e.set_target(verts)
        e.set_source(src)
        for thresh in (1e-4, 0.01, 0.08, 0.6, 50.0):
            e.set_matrices(mxa, mxb)
            A, B, ds = e.make_pairs(thresh, calc_stats=True)
            rA, rB, rds = orc.make_pairs(src, verts, mxa, mxb, thresh, calc_stats=True, **kw)
            assert A.shape == rA.shape and np.array_equal(A, rA) and np.array_equal(B, rB), (thresh, A.shape, rA.shape)
            if rA.shape[1]:
                assert np.allclose(ds, rds, rtol=1e-8, atol=1e-13)
        e.set_matrices(mxa, mxb)
        res = e.run(iters=8, thresh=0.08, target_d=1e-12)
    ref = orc.icp_run(src, verts, mxa, mxb, iters=8, thresh=0.08, target_d=1e-12, sample=1, **kw)
    assert np.array_equal(res.step_K, ref["step_K"])
    assert 0 < res.step_K[-1] < len(src)                           # a real mix of kept and dropped pairs
    assert np.abs(res.step_M - ref["step_M"]).max() < 1e-9
    assert np.abs(res.matrix_world - ref["matrix_world"]).max() <= F32_ULP
    # with the radius switched off the answers are the same
    monkeypatch.setenv("OA_NN_CUTOFF", "0")
    with IcpEngine(0) as e:
        e.set_search_mode(mode)
        if surface:
            e.set_target_mesh(verts, tris)
        else:
            e.set_target(verts)
        e.set_source(src)
        e.set_matrices(mxa, mxb)
        res0 = e.run(iters=8, thresh=0.08, target_d=1e-12)
    assert np.array_equal(res0.matrix_world, res.matrix_world)


@pytest.mark.gpu
@pytest.mark.parametrize("surface", [False, True])
def test_crowded_cells_are_handed_to_the_tree(orc, surface, monkeypatch):
    """Grid cells holding more candidates than a thread's budget (a fan of 4000 thin triangles around one vertex; a
    cluster of 5000 coincident-ish vertices) are finished by k_bvh_search: same answers, for any budget."""
    from object_alignment_amd.engine import IcpEngine
    rng = np.random.default_rng(77)
    n_fan = 4000
    ang = np.linspace(0, 2 * np.pi, n_fan, endpoint=False)
    rim = np.stack([np.cos(ang), np.sin(ang), 0.05 * np.sin(7 * ang)], 1)
    cloud = rng.uniform(-1, 1, size=(20000, 3)) * [1.0, 1.0, 0.2] + [0, 0, 1.5]
    verts = np.concatenate([[[0.0, 0.0, 0.3]], rim, cloud]).astype(np.float32)
    fan = np.stack([np.zeros(n_fan, int), 1 + np.arange(n_fan), 1 + (np.arange(n_fan) + 1) % n_fan], 1)
    extra = rng.integers(1 + n_fan, len(verts), size=(3000, 3))
    tris = np.concatenate([fan, extra]).astype(np.int32)
    if not surface:
        verts = np.concatenate([verts, (rng.normal(size=(5000, 3)) * 1e-5 + [0.2, 0.2, 0.2]).astype(np.float32)])
    q = np.concatenate([rng.normal(size=(3000, 3)) * 0.05 + [0, 0, 0.3], rng.normal(size=(2000, 3)) * 0.01 + [0.2, 0.2, 0.2],
                        rng.uniform(-1.5, 1.5, size=(3000, 3))]).astype(np.float32)
    eye = np.identity(4, dtype=np.float32)
    if surface:
        ridx, _, rd2 = orc.nn_tri_brute(q, verts, tris)
    else:
        ridx, rd2 = orc.nn_brute(q, verts)
    for budget in ("4", "128", "1000000"):
        monkeypatch.setenv("OA_GRID_BUDGET", budget)
        with IcpEngine(0) as e:
            e.set_search_mode("grid")
            if surface:
                e.set_target_mesh(verts, tris)
            else:
                e.set_target(verts)
            e.set_source(q)
            e.set_matrices(eye, eye)
            idx, d2, _ = e.nn_search()
        assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2), budget


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["brute", "grid", "bvh"])
def test_overflowing_distances_select_nothing(orc, mode):
    """Squared distances that overflow float32 are +inf: the oracle's strict `d < best` never selects them (index -1)."""
    from object_alignment_amd.engine import IcpEngine
    rng = np.random.default_rng(3)
    tgt = rng.uniform(-1, 1, size=(9000, 3)).astype(np.float32)
    src = rng.uniform(-1, 1, size=(3000, 3)).astype(np.float32)
    src[::3] = np.where(src[::3] < 0, np.float32(-3e19), np.float32(3e19)) * (1 + np.abs(src[::3]))   # |d|^2 > FLT_MAX
    tris = rng.integers(0, len(tgt), size=(7000, 3)).astype(np.int32)
    eye = np.identity(4, dtype=np.float32)
    ridx, rd2 = orc.nn_brute(src, tgt)
    assert (ridx[::3] == -1).all() and (ridx[1::3] >= 0).all()
    fidx, _, fd2 = orc.nn_tri_brute(src, tgt, tris)
    with IcpEngine(0) as e:
        e.set_search_mode(mode)
        e.set_target(tgt)
        e.set_source(src)
        e.set_matrices(eye, eye)
        idx, d2, _ = e.nn_search()
        assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)
        e.set_target_mesh(tgt, tris)
        idx, d2, _ = e.nn_search()
        assert np.array_equal(idx, fidx) and np.array_equal(d2, fd2)


@pytest.mark.gpu
def test_landmark_prealignment_golden(golden_dir):
    """OBJECT_OT_align_pick_points.align_obj (operators/align_pick_points.py:415-465) against what the reference's own
    method produced: unequal pick counts, rigid and rot+loc+scale, take_m_with."""
    import types
    from object_alignment_amd.operators import LandmarkAlign, OBJECT_OT_align_pick_points, IcpSettings
    from object_alignment_amd.operators import icp_align as icp_mod, align_pick_points as pick_mod
    g = np.load(os.path.join(golden_dir, "landmarks.npz"))
    for i in range(int(g["n_cases"])):
        p = "c%02d_" % i
        settings = IcpSettings(take_m_with=bool(g[p + "take_m"]), align_meth=str(g[p + "meth"]))
        align = types.SimpleNamespace(name="align", type="MESH", matrix_world=g[p + "mx_align"].copy())
        base = types.SimpleNamespace(name="base", type="MESH", matrix_world=g[p + "mx_base"].copy())
        marker = types.SimpleNamespace(name="m_marker", matrix_world=g[p + "m_start"].copy())
        ctx = types.SimpleNamespace(object=align, selected_objects=[base, align],
                                    scene=types.SimpleNamespace(objects=[align, base, marker]))
        old = pick_mod.get_addon_preferences
        pick_mod.get_addon_preferences = lambda: settings
        try:
            assert OBJECT_OT_align_pick_points.poll(ctx)
            op = OBJECT_OT_align_pick_points().begin(ctx)
            for h in g[p + "hits_align"]:
                op.pick_align(h)
            for h in g[p + "hits_base"]:
                op.pick_base(h)
            M = op.align_obj(ctx)
        finally:
            pick_mod.get_addon_preferences = old
        name = str(g[p + "name"])
        assert np.abs(np.asarray(align.matrix_world) - g[p + "final_world"]).max() <= 4 * F32_ULP, name
        assert np.abs(np.asarray(marker.matrix_world) - g[p + "m_final"]).max() <= 8 * F32_ULP, name
        if not bool(g[p + "take_m"]):
            assert np.array_equal(np.asarray(marker.matrix_world), g[p + "m_start"])
        M2, new_mat = LandmarkAlign(settings).solve(list(g[p + "hits_align"]), list(g[p + "stored_base"]))
        assert np.array_equal(M, M2) and new_mat.dtype == np.float32
    with pytest.raises(ValueError, match="input arrays are of wrong shape or type"):
        LandmarkAlign(IcpSettings()).solve([np.zeros(3)] * 2, [np.zeros(3)] * 2)


@pytest.mark.gpu
def test_spatial_shards_partition_the_selection(orc, monkeypatch):
    """shard_count > 1: shards are equal ranges of the Morton-ordered selection.  Together they hold every selected
    point exactly once (pairs of all shards == pairs of the unsharded run, as sets), each lists its points in the
    caller's order, each is spatially compact, and OA_SHARD_SPATIAL=0 (contiguous ranges) gives the same pairs."""
    from object_alignment_amd.engine import IcpEngine, shard_bounds
    rng = np.random.default_rng(12)
    tgt = rng.uniform(-1, 1, size=(30000, 3)).astype(np.float32)
    src = (tgt[rng.permutation(30000)[:20001]] + rng.normal(0, 5e-3, size=(20001, 3))).astype(np.float32)
    vlist = np.sort(rng.choice(len(src), size=15000, replace=False)).astype(np.int64)
    eye = np.identity(4, dtype=np.float32)
    mxa = np.identity(4, dtype=np.float32); mxa[:3, 3] = [0.01, -0.02, 0.005]
    rA, rB, _ = orc.make_pairs(src, tgt, mxa, eye, 0.05, vlist=vlist, sample=2, calc_stats=True)
    world = 4                                     # a power of two: each Morton range is a union of octants

    def run(spatial):
        monkeypatch.setenv("OA_SHARD_SPATIAL", "1" if spatial else "0")
        parts, boxes = [], []
        for r in range(world):
            with IcpEngine(0) as e:
                e.set_target(tgt)
                e.set_source(src, vlist=vlist, stride=2, shard_index=r, shard_count=world)
                b, en = shard_bounds(7500, r, world)
                assert e.n_selected == en - b
                e.set_matrices(mxa, eye)
                A, B, _ = e.make_pairs(0.05, calc_stats=False)
            parts.append((A, B))
            boxes.append(np.prod(A.max(axis=1) - A.min(axis=1)) if A.shape[1] else 0.0)
        return parts, boxes

    for spatial in (True, False):
        parts, boxes = run(spatial)
        A = np.concatenate([p[0] for p in parts], axis=1)
        B = np.concatenate([p[1] for p in parts], axis=1)
        assert A.shape == rA.shape
        order, rorder = np.lexsort(A), np.lexsort(rA)
        assert np.array_equal(A[:, order], rA[:, rorder]) and np.array_equal(B[:, order], rB[:, rorder])
        for pA, _ in parts:                       # caller order inside a shard: a subsequence of the unsharded columns
            idx = [np.flatnonzero((rA == pA[:, [k]]).all(axis=0))[0] for k in range(0, pA.shape[1], 97)]
            assert idx == sorted(idx)
        if spatial:
            assert max(boxes) < 0.7 * np.prod(rA.max(axis=1) - rA.min(axis=1))
        else:
            assert np.array_equal(A, rA)          # contiguous ranges: the concatenation IS the unsharded order


@pytest.mark.gpu
def test_surface_full_size_property(orc):
    """Surface mode at the bench's surface_path sizes (1M queries, 980k-vertex / 1.96M-triangle mesh).  Size-independent
    properties: a query that IS a point of the mesh (a vertex, or a barycentric combination of one triangle's corners)
    finds distance ~0; for queries off the surface the distance to the triangles is never larger than the distance to
    the nearest vertex; a sample of the answers equals the oracle's brute force over all triangles, bit for bit; and
    every search mode returns the same answers."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    verts, tris = synth.lattice_surface_mesh(700, 1400)
    rng = np.random.default_rng(77)
    n = 1_000_000
    t = rng.integers(0, len(tris), size=n)
    w = rng.dirichlet([1.0, 1.0, 1.0], size=n)
    on_surface = np.einsum("nk,nkd->nd", w, verts[tris[t]].astype(np.float64)).astype(np.float32)
    on_surface[:1000] = verts[rng.integers(0, len(verts), size=1000)]          # exact vertices
    off = (on_surface.astype(np.float64) * (1.0 + rng.normal(0, 2e-3, size=(n, 1)))).astype(np.float32)
    eye = np.identity(4, dtype=np.float32)
    out = {}
    for mode in ("auto", "bvh"):
        with IcpEngine(0) as e:
            e.set_search_mode(mode)
            e.set_target_mesh(verts, tris)
            e.set_source(on_surface)
            e.set_matrices(eye, eye)
            idx_on, d2_on, _ = e.nn_search()
            e.set_source(off)
            idx_off, d2_off, _ = e.nn_search()
            if mode == "auto":
                e.set_target(verts)                                              # vertex mode on the same queries
                _, d2_vert, _ = e.nn_search()
        out[mode] = (idx_on, d2_on, idx_off, d2_off)
    idx_on, d2_on, idx_off, d2_off = out["auto"]
    assert np.all(d2_on[:1000] == 0.0)
    assert d2_on.max() <= (4e-7) ** 2 * 3 * 4                                    # float32 rounding of the combination
    assert np.all(idx_on >= 0) and np.all(idx_off >= 0)
    # the surface contains the vertices (the two float32 metrics may differ in the last bits when the closest point
    # IS a vertex: fma chain there, Blender's operand order here)
    assert np.all(d2_off <= d2_vert * np.float32(1 + 1e-5))
    assert np.count_nonzero(d2_off < d2_vert) > n // 2
    for a, b in zip(out["auto"], out["bvh"]):
        assert np.array_equal(a, b)
    pick = rng.choice(n, size=150, replace=False)
    face, _, rd2 = orc.nn_tri_brute(off[pick], verts, tris)
    assert np.array_equal(idx_off[pick], face) and np.array_equal(d2_off[pick], rd2)


@pytest.mark.gpu
def test_surface_launch_shapes_change_nothing_but_speed(monkeypatch):
    """Round 6's launch shapes of the plain surface search -- one wave per workgroup for long launches (OA_TRI_WAVE_WGS), an XCD's
    share in chunks while the pose is far off and contiguous once it has settled (OA_TRI_XCD_CHUNK) -- against workgroups of 256
    queries on contiguous shares throughout: twelve iterations from a cold start at the bench's surface_path sizes pass through
    both regimes; every iteration's statistics, the final matrix and a one-shot search's answers are the same bits."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    verts, tris = synth.lattice_surface_mesh(700, 1400)
    src = synth.bunny_surface(1_000_000, offset=0.37)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.02, -0.015, 0.025]), [0.01, -0.008, 0.012])
    eye = np.identity(4, dtype=np.float32)
    got = []
    for wave_wgs, chunk in (("0", "0"), ("1", "8"), ("1", "0")):
        monkeypatch.setenv("OA_TRI_WAVE_WGS", wave_wgs)
        monkeypatch.setenv("OA_TRI_XCD_CHUNK", chunk)
        with IcpEngine(0) as e:
            e.set_target_mesh(verts, tris)
            e.set_source(src)
            e.set_matrices(mxa, eye)
            idx, d2, _ = e.nn_search()
            r = e.run(iters=12, thresh=0.05, early_exit=False)
            got.append((idx, d2, np.array(e.matrix_world()), r.last_K, r.mean_dist))
    for other in got[1:]:
        assert np.array_equal(got[0][0], other[0]) and np.array_equal(got[0][1], other[1])
        assert np.array_equal(got[0][2], other[2]) and got[0][3] == other[3] and got[0][4] == other[4]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["auto", "grid"])
def test_step_mode_equals_fused_loop_surface_and_shards(mode):
    """oa_iterate (one iteration per call, the modal operator's step) and oa_run (the whole loop enqueued ahead, with
    the halt flag polled) walk through bitwise the same iterations -- surface mode, masked source, and the same again
    with the search strategies forced."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    verts, tris = synth.lattice_surface_mesh(120, 240)
    src = synth.bunny_surface(30000, 0.5)
    vlist = np.arange(0, len(src), 3, dtype=np.int64)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.05, -0.04, 0.06]), [0.03, -0.02, 0.02])
    mxb = synth.rigid4(synth.rotation_from_rotvec([0.2, 0.1, -0.3]), [0.1, 0.0, -0.1])
    kw = dict(thresh=0.2, target_d=1e-4, use_target=True)
    with IcpEngine(0) as e:
        e.set_search_mode(mode)
        e.set_target_mesh(verts, tris)
        e.set_source(src, vlist=vlist, stride=1)
        e.set_matrices(mxb @ mxa, mxb)
        ref = e.run(iters=40, early_exit=True, **kw)
        assert 5 < ref.iters_done <= 40
        e.set_matrices(mxb @ mxa, mxb)
        Ms, Ks = [], []
        for _ in range(ref.iters_done):
            M, st = e.iterate(**kw)
            Ms.append(M)
            Ks.append(st["K"])
        assert st["converged"] == ref.converged
        assert np.array_equal(np.array(Ms), ref.step_M)
        assert np.array_equal(np.array(Ks), ref.step_K)
        assert np.array_equal(e.matrix_world(), ref.matrix_world)


def test_search_mode_changes_between_steps():
    """The per-slot winner records (vertex mode) are written by the grid / tree searches and by k_pair_accumulate
    after a brute-force search; whatever mode ran last, the next step must see a consistent seed.  Step-by-step
    loops that switch the search mode every iteration give the same steps as one mode throughout: bitwise between the
    brute-force and the grid search (both end in the canonical accumulation, k_pair_accumulate_canon or its fused twin in
    the grid search's epilogue); K exact and M to 1e-12 once the whole-shard tree search takes a step (it keeps running
    sums per wave, another summation order)."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    rng = np.random.default_rng(77)
    tgt = rng.uniform(-1, 1, size=(60000, 3)).astype(np.float32)
    tgt[1000:1040] = tgt[1000]                                          # duplicate vertices: index ties
    src = (tgt[rng.permutation(60000)[:45000]] + rng.normal(size=(45000, 3)) * 2e-3).astype(np.float32)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.03, -0.02, 0.025]), [0.02, -0.015, 0.01])
    eye = np.identity(4, dtype=np.float32)
    kw = dict(thresh=0.2, target_d=1e-9, use_target=True)

    def steps(modes):
        out = []
        with IcpEngine(0) as e:
            e.set_target(tgt)
            e.set_source(src)
            e.set_matrices(mxa, eye)
            for m in modes:
                e.set_search_mode(m)
                M, st = e.iterate(**kw)
                out.append((M.copy(), st["K"], st["mean_dist"]))
            idx, d2, _ = e.nn_search()
        return out, idx, d2

    ref, ridx, rd2 = steps(["brute"] * 8)
    for modes in (["grid", "brute", "bvh", "brute", "brute", "grid", "bvh", "grid"],
                  ["bvh", "bvh", "grid", "grid", "brute", "bvh", "grid", "brute"],
                  ["auto"] * 8):
        got, idx, d2 = steps(modes)
        for k, (a, b) in enumerate(zip(ref, got)):
            assert a[1] == b[1], (modes, k)
            if "bvh" in modes:
                assert np.abs(a[0] - b[0]).max() < 1e-12 and abs(a[2] - b[2]) <= 1e-12 * abs(a[2]), (modes, k)
            else:
                assert np.array_equal(a[0], b[0]) and a[2] == b[2], (modes, k)
        assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)


@pytest.mark.parametrize("surface", [False, True])
def test_tree_then_grid_turns(surface):
    """Shards between the late and the early tree / grid limits get both searches enqueued every iteration and
    DevState::tree_turn (solve kernel: did the pose still move by more than a tenth of a cell?) picks one on the
    device.  Whatever the sequence of turns, every step equals the brute-force run bit for bit -- also with the turns
    switched off (OA_SEARCH_TURNS=0) and with the tree keeping its turn for ever (OA_TURN_FRAC=0)."""
    import os
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    tv, tt = synth.bumpy_icosphere_mesh(5)                              # 10k vertices, 20k triangles
    ns = 19_000 if surface else 6_000                                   # inside (late, early] for either mode
    src = synth.bunny_surface(ns, offset=0.41).astype(np.float32)
    src *= np.float32(np.linalg.norm(tv, axis=1).mean() / np.linalg.norm(src, axis=1).mean())
    pose = synth.rigid4(synth.rotation_from_rotvec([0.05, -0.04, 0.06]), [0.03, -0.02, 0.025])
    eye = np.identity(4, dtype=np.float32)
    kw = dict(thresh=0.3, target_d=1e-12, use_target=True)

    def steps(mode, env=None):
        old = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            out = []
            with IcpEngine(0) as e:
                e.set_search_mode(mode)
                if surface:
                    e.set_target_mesh(tv, tt)
                else:
                    e.set_target(tv)
                e.set_source(src)
                e.set_matrices(pose, eye)
                for _ in range(12):
                    M, st = e.iterate(**kw)
                    out.append((M.copy(), st["K"], st["mean_dist"]))
            return out
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    ref = steps("brute")
    assert ref[0][1] > 0.9 * ns and ref[-1][2] < ref[0][2]              # a real, converging run
    for env in (None, {"OA_SEARCH_TURNS": "0"}, {"OA_TURN_FRAC": "0"}, {"OA_TURN_FRAC": "1e9"}):
        got = steps("auto", env)
        for k, (a, b) in enumerate(zip(ref, got)):      # (the whole-shard tree search sums per wave: another order than the canonical rows)
            assert a[1] == b[1] and abs(a[2] - b[2]) <= 1e-12 * abs(a[2]) and np.abs(a[0] - b[0]).max() < 1e-12, (env, k)


# ---------------------------------------------------------------------------------------------------------------------
# round 3: the closable parity holes of VERDICT r2
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("how", ["auto", "multi8"])
def test_c5_full_size_with_normal_angle_rejection(orc, how):
    """BASELINE config 5 at FULL size WITH its normal-angle leg (an extension, SURVEY D3: pinned against the oracle's
    restatement only): 10M source points with normals, the seeded 10 % cap excluded through the `icp_exclude` semantics,
    2M target vertices with normals, max angle 45 degrees, ten iterations -- one context in AUTO mode and 8 shards through a
    multi-device context -- against orc.icp_run(normals=...).  A third of the source normals are bent so that the test
    really rejects pairs: K per iteration exact (and smaller than without the test), M to 1e-9, Frobenius 1e-5."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    from object_alignment_amd.operators.icp_align import vlist_from_weights
    src, sn = synth.bunny_surface_with_normals(10_000_000, 0.5)
    tgt, tn = synth.bunny_surface_with_normals(2_000_000, 0.0)
    rng = np.random.default_rng(501)
    bent = rng.random(len(src)) < 0.33
    sn = sn.copy()
    sn[bent] += rng.normal(size=(int(bent.sum()), 3)).astype(np.float32) * np.float32(0.9)
    sn /= np.maximum(np.linalg.norm(sn, axis=1, keepdims=True), np.float32(1e-12))
    sn = sn.astype(np.float32)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.003, -0.002, 0.004]), [0.002, -0.001, 0.0015])
    eye = np.identity(4, dtype=np.float32)
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    h = src.astype(np.float64) @ axis
    cap = np.nonzero(h > np.quantile(h, 0.9))[0]
    vlist = np.array(vlist_from_weights(len(src), exclude=[(int(v), 1.0) for v in cap]), dtype=np.int64)
    iters = 10
    kw = dict(iters=iters, thresh=0.5, target_d=0.01, use_target=True, early_exit=False)
    eng = IcpEngine(devices=[0] * 8) if how == "multi8" else IcpEngine(0)
    try:
        eng.set_target(tgt)
        eng.set_source(src, vlist=vlist, stride=1)
        eng.set_normals(sn, tn, 45.0)
        eng.set_matrices(mxa, eye)
        res = eng.run(**kw)
    finally:
        eng.close()
    if "c5n" not in _REF_CACHE:
        _REF_CACHE["c5n"] = orc.icp_run(src, tgt, mxa, eye, iters=iters, sample=1, thresh=0.5, target_d=1e-300, use_target=True,
                                        vlist=vlist, kd=orc.KDTree(tgt), normals=(sn, tn), max_angle_deg=45.0)
    ref = _REF_CACHE["c5n"]
    assert 0.5 * len(vlist) < ref["step_K"][0] < 0.95 * len(vlist)       # the angle test bites
    _assert_loop_equals_oracle(res, ref, iters)


@pytest.mark.gpu
def test_fuzz_parity_slice():
    """A fixed-seed slice of tools/fuzz_parity.py (the builder's campaigns run thousands of trials; this is the part the
    driver executes): 200 random trials -- cloud families, sizes, scales, offsets, mirrored and non-uniformly scaled
    matrices, vertex and surface targets -- nn_search and make_pairs bit-exact against the oracle in brute / grid / bvh
    mode, three-iteration loops against the oracle's loop."""
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "200", "20260929"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0 and "200 trials, 0 mismatches" in p.stdout, p.stdout[-3000:]


@pytest.mark.gpu
def test_fuzz_modes_large_slice():
    """A fixed-seed slice of tools/fuzz_modes_large.py: grid and tree against the brute-force kernels at sizes the CPU
    oracle cannot reach in seconds (0.2-3M targets, multi-million-cell grids, long hand-over lists, surface targets with
    huge triangles) -- indices, distances, pairs and three-iteration loops bitwise."""
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_modes_large.py"), "4", "929"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0 and "4 trials, 0 mismatches" in p.stdout, p.stdout[-3000:]


@pytest.mark.gpu
def test_make_pairs_sees_in_place_edits_of_the_source():
    """The reference reads vertices[i].co on every call (functions/general.py:284).  GpuBVH keeps the source on the GPU
    between calls and re-uploads when its content hash changes: an in-place PERMUTATION of the coordinates (every sum the
    round-2 fingerprint looked at stays the same) or of the vertex list must give the pairs of the new geometry."""
    from object_alignment_amd import synth
    from object_alignment_amd.functions import make_pairs, GpuBVH, AlignObject
    rng = np.random.default_rng(3)
    tgt = synth.bumpy_icosphere(4)
    xyz = (tgt[rng.permutation(len(tgt))[:1500]] * np.float32(1.01)).astype(np.float32)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.03, -0.02, 0.04]), [0.02, -0.01, 0.015])
    base = AlignObject(tgt)
    align = AlignObject(xyz.copy(), mxa)
    bvh = GpuBVH.FromObject(base)
    vlist = list(range(0, 1500, 2))
    A0, B0, _ = make_pairs(align, base, bvh, vlist, 0.5)
    perm = rng.permutation(1500)
    align.xyz[:] = align.xyz[perm]                                  # in place: same array object, same sums
    A1, B1, _ = make_pairs(align, base, bvh, vlist, 0.5)
    fresh = AlignObject(xyz[perm].copy(), mxa)
    A2, B2, _ = make_pairs(fresh, base, GpuBVH.FromObject(base), vlist, 0.5)
    assert np.array_equal(A1, A2) and np.array_equal(B1, B2)
    assert not np.array_equal(A0, A1)
    vl = np.array(vlist, dtype=np.int64)
    vl2 = vl.copy()
    vl2[:] = vl[rng.permutation(len(vl))]                           # same length, same sum, other order
    A3, _, _ = make_pairs(fresh, base, bvh, vl, 0.5)
    A4, _, _ = make_pairs(fresh, base, bvh, vl2, 0.5)
    assert np.array_equal(A3, A2) and not np.array_equal(A4, A3) and np.array_equal(np.sort(A4, axis=1), np.sort(A3, axis=1))
    # a Python LIST edited in place -- what the operators hand over -- at a position round 3's 32-element probe never looked at
    # (VERDICT r3); and a duck-typed mesh edited in place at a vertex its 64-vertex probe never looked at
    import types
    pyl = list(range(0, 1500))
    A5, _, _ = make_pairs(fresh, base, bvh, pyl, 0.5)
    pyl[777] = 3                                                    # vertex 777 leaves, vertex 3 is listed twice
    A6, _, _ = make_pairs(fresh, base, bvh, pyl, 0.5)
    A7, _, _ = make_pairs(fresh, base, bvh, np.array(pyl, dtype=np.int64), 0.5)
    assert np.array_equal(A6, A7) and not np.array_equal(A5, A6)
    verts = [types.SimpleNamespace(co=[float(c) for c in row]) for row in fresh.xyz]
    duck = types.SimpleNamespace(data=types.SimpleNamespace(vertices=verts), matrix_world=mxa)
    A8, B8, _ = make_pairs(duck, base, bvh, vl, 0.5)
    assert np.array_equal(A8, A3)
    moved = (fresh.xyz[501] * np.float32(1.002)).astype(np.float32)
    verts[501].co[:] = [float(c) for c in moved]
    A9, _, _ = make_pairs(duck, base, bvh, np.arange(1500, dtype=np.int64), 0.5)
    assert np.array_equal(A9[:, 501], moved.astype(np.float64))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["settled", "far_start", "half_target", "lanes4"])
def test_grid_paths_give_the_same_bits(orc, case, monkeypatch):
    """The loop's grid search has two forms (docs/HISTORY.md 4.4): FAST -- it finishes the queries its rings did not settle through
    the tree itself and accumulates in its epilogue -- and SAFE -- grid search, tree search of the hand-over list,
    k_pair_accumulate_canon.  The host picks per iteration from what the device last reported, i.e. timing dependent, so
    the two must leave the same bits: forced fast, forced safe and adaptive runs are compared bitwise, and against the
    oracle's loop (K exact, M to 1e-9) -- settled poses, a start far from the target (most queries leave the rings), a
    target cut in half (half the queries have no partner in range), and a shard small enough for four lanes per query."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    n = 60_000 if case != "lanes4" else 9_000
    src, tgt, mxa, mxb = synth.c2_bunny_pair(n)
    thresh, iters = 0.5, 6
    if case == "far_start":
        mxa = synth.rigid4(synth.rotation_from_rotvec([0.2, -0.1, 0.15]), [0.3, -0.2, 0.25])
        thresh = 1.0
    if case == "half_target":
        tgt = np.ascontiguousarray(tgt[tgt[:, 0] > 0.0])
    out = {}
    for tag, env in (("safe", {"OA_GRID_PATH": "safe"}), ("fast", {"OA_GRID_PATH": "fast"}), ("adaptive", {}),
                     ("plain", {"OA_FUSED_ACC": "0"})):
        monkeypatch.delenv("OA_GRID_PATH", raising=False)
        monkeypatch.delenv("OA_FUSED_ACC", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with IcpEngine(0) as e:
            e.set_search_mode("grid")
            e.set_target(tgt)
            e.set_source(src, stride=1)
            e.set_matrices(mxa, mxb)
            r1 = e.run(iters=iters, thresh=thresh, early_exit=False)
            r2 = e.run(iters=3, thresh=thresh, early_exit=False)          # a second loop continues on warm seeds
            out[tag] = (r1, r2)
    ref = orc.icp_run(src, tgt, mxa, mxb, iters=iters, sample=1, thresh=thresh, target_d=1e-300, use_target=True, kd=orc.KDTree(tgt))
    for tag in ("fast", "adaptive", "plain"):
        for a, b in zip(out["safe"], out[tag]):
            assert np.array_equal(a.step_K, b.step_K), (case, tag)
            assert np.array_equal(a.step_M, b.step_M) and np.array_equal(a.matrix_world, b.matrix_world), (case, tag)
    assert np.array_equal(out["safe"][0].step_K, ref["step_K"])
    assert np.abs(out["safe"][0].step_M - ref["step_M"]).max() < 1e-9


@pytest.mark.gpu
def test_identity_base_matrix_shortcut_is_bit_exact(monkeypatch):
    """pair_eval skips `mx2 @ v` when the base object's matrix_world is exactly the identity (for finite v the product
    returns v bit for bit, up to the sign of a zero that neither the distance nor b can see).  With and without the
    shortcut (OA_NO_IDENTITY_PATH=1): the same pairs, bit for bit -- including coordinates that are exactly +-0 -- and
    the same loop."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    rng = np.random.default_rng(9)
    tgt = rng.uniform(-1, 1, size=(20000, 3)).astype(np.float32)
    tgt[:300, 0] = 0.0
    tgt[300:600, 1] = -0.0
    src = (tgt[rng.permutation(20000)[:15000]] + rng.normal(size=(15000, 3)).astype(np.float32) * np.float32(1e-3)).astype(np.float32)
    src[:200] = tgt[:200]                                           # exact hits, zeros included
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.01, -0.02, 0.015]), [0.004, -0.003, 0.002])
    eye = np.identity(4, dtype=np.float32)
    out = []
    for off in ("0", "1"):
        monkeypatch.setenv("OA_NO_IDENTITY_PATH", off)
        for mode in ("grid", "brute"):
            with IcpEngine(0) as e:
                e.set_search_mode(mode)
                e.set_target(tgt); e.set_source(src); e.set_matrices(mxa, eye)
                A, B, st = e.make_pairs(0.05, calc_stats=True)
                e.set_matrices(mxa, eye)
                r = e.run(iters=5, thresh=0.05, early_exit=False)
                out.append((A, B, np.array(st), r.step_M, r.step_K, r.matrix_world))
    for o in out[1:]:
        for a, b in zip(out[0], o):
            assert np.array_equal(a, b)


@pytest.mark.gpu
def test_wave_reduction_keeps_the_bits_of_the_shuffle_form():
    """The pair sums of every accumulating kernel go through wave_reduce_scatter (oa_kernels.hpp): v_permlane32_swap /
    v_permlane16_swap halvings and DPP butterflies.  tools/reduce_check.hip holds the form it replaced (ds_bpermute shuffles
    with selects) and compares the two bit for bit on random doubles of mixed magnitude, and the new form against a long
    double host sum: the rows of partials -- and with them every loop fixture -- must not depend on which one is built."""
    import subprocess
    import __graft_entry__ as entry
    exes = entry.build_tools()
    exe = [e for e in exes if e.endswith("reduce_check.exe")]
    assert exe, "tools/reduce_check.hip did not build"
    out = subprocess.run([exe[0], "2048"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "identical" in out.stdout and "0 of" in out.stdout, out.stdout


# ---------------------------------------------------------------------------------------------------------------------
# round 4: what is closable of f1 -- the tie rule never shows in co1
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", ["ico", "lattice", "ties"])
def test_surface_pairs_do_not_depend_on_the_triangle_order(orc, case):
    """`BVHTree.find_nearest` (/root/reference/functions/general.py:297-304) returns the closest surface point co1; which of
    several equidistant triangles Blender's traversal reports is unknowable here, and this build's rule is "lowest
    triangle index".  Only co1 is consumed (`n`, `face_index` are not, :297-304), so the rule matters exactly where two
    triangles at the same float32 distance have DIFFERENT closest points.  Shuffle the triangle order: every triangle keeps
    its corners and its arithmetic, only the indices change.  Then
      - the float32 distances must be bitwise the same (the minimum over the same set of values),
      - the winners may differ only where the distances tie exactly,
      - make_pairs' A and B -- co1 mapped to align-local space -- are compared bit for bit, at the initial pose and at the poses
        of four iterations; the queries where they differ are counted and characterised: each must be an exact tie between
        triangles whose closest points differ -- either the same geometric point of a shared edge / corner, rounded differently
        by the two triangles' float32 arithmetic (a few ulp), or two distinct equidistant points (symmetric data only),
      - and where there is no such query, the whole loop ends with bitwise the same matrices; else within 1e-4."""
    from object_alignment_amd.engine import IcpEngine
    verts, tris, q = _surface_cases()[case]
    rng = np.random.default_rng(17)
    perm = rng.permutation(len(tris))                              # new triangle k is old triangle perm[k]
    tris2 = np.ascontiguousarray(tris[perm])
    eye = np.identity(4, dtype=np.float32)
    n_swapped = n_moved = n_rounding = n_distinct = 0
    max_rounding_ulp = 0.0
    with IcpEngine(0) as e0, IcpEngine(0) as e1:
        e0.set_target_mesh(verts, tris)
        e1.set_target_mesh(verts, tris2)
        for e in (e0, e1):
            e.set_source(q)
        pose = np.identity(4, dtype=np.float32)
        for it in range(5):                                        # the initial pose and the poses of four iterations
            out = []
            for e in (e0, e1):
                e.set_matrices(pose, eye)
                idx, d2, _ = e.nn_search()
                A, B, _ = e.make_pairs(1e30)
                out.append((idx, d2, A, B))
            (i0, d0, A0, B0), (i1, d1, A1, B1) = out
            assert np.array_equal(d0, d1), it                       # same set of per-triangle distances, same minimum
            same_winner = perm[i1] == i0
            assert A0.shape[1] == A1.shape[1] == len(q) and np.array_equal(A0, A1)
            differs = np.flatnonzero((B0 != B1).any(axis=0))
            assert not same_winner[differs].any()                   # co1 can only move where another triangle won ...
            qw = _cofind(orc, q, pose, eye)
            for qi in differs:
                # ... at the same float32 distance, with a different closest point (the oracle's, on ONE triangle each)
                _, pa, da = orc.nn_tri_brute(qw[qi:qi + 1], verts, tris[i0[qi]][None])
                _, pb, db = orc.nn_tri_brute(qw[qi:qi + 1], verts, tris2[i1[qi]][None])
                assert da[0] == db[0] == d0[qi]
                assert not np.array_equal(pa, pb)
                # two kinds: the SAME geometric point -- on an edge or at a corner the two triangles share -- rounded
                # differently by the two triangles' float32 arithmetic (a few ulp apart), or two genuinely different points at
                # the same float32 distance (the query on the medial axis of the two: symmetric data only)
                gap = float(np.abs(pa.astype(np.float64) - pb.astype(np.float64)).max())
                ulp = float(np.spacing(np.float32(max(1e-30, np.abs(pa).max(), np.abs(qw[qi]).max()))))
                if gap <= 8.0 * ulp:
                    n_rounding += 1
                    max_rounding_ulp = max(max_rounding_ulp, gap / ulp)
                else:
                    n_distinct += 1
            n_swapped += int((~same_winner).sum())
            n_moved += len(differs)
            # the next pose: one iteration of the ORIGINAL order's loop
            e0.set_matrices(pose, eye)
            e0.iterate(thresh=1e30)
            pose = e0.matrix_world()
        # the two loops, each on its own: bitwise the same matrices unless some query's co1 moved on the way
        res = []
        for e in (e0, e1):
            e.set_matrices(np.identity(4, dtype=np.float32), eye)
            res.append(e.run(iters=4, thresh=1e30, early_exit=False))
    total = 5 * len(q)
    print("surface tie rule, case %s: over 5 poses x %d queries, %d answered with another triangle after the shuffle (exact float32 "
          "ties); co1 differed for %d (%.4f %%): %d the same point of a shared edge / corner rounded differently (<= %.1f ulp apart), "
          "%d two distinct equidistant points" % (case, len(q), n_swapped, n_moved, 100.0 * n_moved / total, n_rounding,
                                                  max_rounding_ulp, n_distinct))
    if case != "ties":
        assert n_moved <= 2e-3 * total                             # generic data: (next to) never ...
        assert n_distinct == 0                                     # ... and then only a rounding of the same point
    if n_moved == 0:
        assert np.array_equal(res[0].step_M, res[1].step_M) and np.array_equal(res[0].matrix_world, res[1].matrix_world)
    else:
        assert np.abs(res[0].matrix_world - res[1].matrix_world).max() <= 1e-4

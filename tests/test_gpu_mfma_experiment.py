"""OA_NN_MFMA=1 (experiment, off by default): the first filter level of the brute-force nearest-vertex search on the
matrix cores (object_alignment_amd/csrc/oa_mfma.hpp).  Like every filter level it may only ever PROVE losers, so the
answers must stay bit-identical to the oracle's brute force and to the default kernel.  The kernel takes shards of
>= 65536 points against targets of > 65536 vertices (four points per thread, 1024-vertex tiles); the cases below are sized
for it and assert that it really ran (OA_STAT_BRUTE_KERNEL == 2)."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NS, NT = 66_000, 70_000


def _cofind(orc, src, mxa, mxb):
    """co_find = imx2 @ (mx1 @ co) with the oracle's float32 arithmetic (functions/general.py:287)."""
    imx2 = orc.mat4_inverted(mxb)
    return np.array([orc.mat4_mul_vec3(imx2, orc.mat4_mul_vec3(mxa, p)) for p in src], np.float32)


def _engine(monkeypatch, mfma):
    from object_alignment_amd.engine import IcpEngine
    monkeypatch.setenv("OA_NN_MFMA", "1" if mfma else "0")
    e = IcpEngine(0, experiments=True)
    e.set_search_mode("brute")
    return e


def _case(case):
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    if case == "uniform":
        tgt = rng.uniform(-1, 1, size=(NT, 3)); src = rng.uniform(-1, 1, size=(NS, 3))
    elif case == "far_offset":                    # extent 1 at distance ~2000 from the origin
        off = np.array([1000.0, -2000.0, 500.0])
        tgt = rng.uniform(-0.5, 0.5, size=(NT, 3)) + off; src = rng.uniform(-0.5, 0.5, size=(NS, 3)) + off
    elif case == "lattice_jitter":                # near-ties everywhere
        tgt = rng.integers(-20, 21, size=(NT, 3)) * 0.05 + rng.normal(0, 1e-6, size=(NT, 3))
        src = rng.integers(-20, 20, size=(NS, 3)) * 0.05 + 0.025
    elif case == "exact_ties":                    # duplicates: lowest index must win
        tgt = rng.integers(-12, 13, size=(NT, 3)).astype(np.float64)
        src = rng.integers(-13, 14, size=(NS, 3)) + 0.5
    elif case == "query_outside":                 # most queries far outside the target's box (columns not representable)
        tgt = rng.uniform(-1, 1, size=(NT, 3)); src = rng.normal(size=(NS, 3)) * 50.0
    elif case == "query_far_mixed":               # a few hundred extents away, and a cluster inside
        tgt = rng.uniform(-1, 1, size=(NT, 3)); src = rng.normal(size=(NS, 3)) * 400.0
        src[::3] = rng.uniform(-1, 1, size=(len(src[::3]), 3))
    elif case == "flat_plane":                    # zero-thickness box, duplicates
        tgt = rng.uniform(-1, 1, size=(NT, 3)); tgt[:, 2] = 0.25; tgt[::7] = tgt[3]
        src = rng.uniform(-1, 1, size=(NS, 3))
    elif case == "tiny_scale":
        tgt = rng.uniform(-1, 1, size=(NT, 3)) * 1e-12; src = rng.uniform(-1, 1, size=(NS, 3)) * 1e-12
    elif case == "huge_scale":
        tgt = rng.uniform(-1, 1, size=(NT, 3)) * 1e12; src = rng.uniform(-1, 1, size=(NS, 3)) * 1e12
    elif case == "clusters":                      # tight clusters far apart: scores cancel to many digits
        c = rng.uniform(-100, 100, size=(40, 3))
        tgt = c[rng.integers(0, 40, NT)] + rng.normal(0, 1e-3, size=(NT, 3))
        src = c[rng.integers(0, 40, NS)] + rng.normal(0, 1e-3, size=(NS, 3))
    elif case == "one_axis_line":                 # the target is a line: two axes of zero extent
        tgt = np.zeros((NT, 3)); tgt[:, 0] = rng.uniform(-5, 5, NT)
        src = rng.normal(size=(NS, 3))
    elif case == "non_finite_queries":
        tgt = rng.uniform(-1, 1, size=(NT, 3)); src = rng.uniform(-1, 1, size=(NS, 3))
        src[5] = [np.nan, 0, 0]; src[77] = [np.inf, 1, 1]; src[4097] = [0, -np.inf, 0]; src[-1] = [np.nan] * 3
    else:
        raise KeyError(case)
    return src.astype(np.float32), tgt.astype(np.float32)


CASES = ["uniform", "far_offset", "lattice_jitter", "exact_ties", "query_outside", "query_far_mixed", "flat_plane",
         "tiny_scale", "huge_scale", "clusters", "one_axis_line", "non_finite_queries"]


@pytest.mark.parametrize("case", CASES)
def test_mfma_prefilter_bit_identical(orc, case, monkeypatch):
    src, tgt = _case(case)
    eye = np.identity(4, dtype=np.float32)
    mxa = eye.copy()
    mxa[:3, 3] = np.float32(0.003) * np.abs(tgt).max()
    with _engine(monkeypatch, True) as e:
        e.set_target(tgt)
        e.set_source(src)
        e.set_matrices(mxa, eye)
        assert e.stat("brute_kernel") == 2.0
        idx, d2, _ = e.nn_search()                # unseeded
        # a full iteration leaves seeds; the second search starts from them under a moved pose
        e.iterate(thresh=1e30)
        mx2 = e.matrix_world()
        mx2[:3, 3] += np.float32(0.001) * np.abs(tgt).max()
        e.set_matrices(mx2, eye)
        idx2, d22, _ = e.nn_search()
    q = _cofind(orc, src, mxa, eye)
    ridx, rd2 = orc.nn_brute(q, tgt)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(d2, rd2)
    with _engine(monkeypatch, False) as e:        # the seeded search against the default kernel on the same inputs
        e.set_target(tgt)
        e.set_source(src)
        e.set_matrices(mxa, eye)
        assert e.stat("brute_kernel") == 3.0
        e.iterate(thresh=1e30)
        e.set_matrices(mx2, eye)
        ridx2, rd22, _ = e.nn_search()
    assert np.array_equal(idx2, ridx2)
    assert np.array_equal(d22, rd22)


def test_mfma_prefilter_loop_identical(monkeypatch):
    """A 12-iteration alignment: every iteration's matrix and pair count, bit for bit, with and without the experiment."""
    from object_alignment_amd import synth
    rng = np.random.default_rng(2024)
    tgt = rng.normal(size=(90_000, 3)).astype(np.float32)
    src = (tgt[rng.integers(0, len(tgt), 80_000)] + rng.normal(0, 0.01, size=(80_000, 3))).astype(np.float32)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.02, -0.03, 0.01]), [0.02, 0.01, -0.015])
    eye = np.identity(4, dtype=np.float32)
    out = []
    for mfma in (True, False):
        with _engine(monkeypatch, mfma) as e:
            e.set_target(tgt)
            e.set_source(src)
            e.set_matrices(mxa, eye)
            assert e.stat("brute_kernel") == (2.0 if mfma else 3.0)
            r = e.run(iters=12, thresh=0.5, early_exit=False)
            out.append((r.matrix_world.copy(), r.iters_done, r.step_K.copy(), r.step_M.copy(), r.step_new.copy()))
    assert out[0][1] == out[1][1] == 12
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


def test_mfma_experiment_is_off_by_default(monkeypatch):
    from object_alignment_amd.engine import IcpEngine
    monkeypatch.delenv("OA_NN_MFMA", raising=False)
    rng = np.random.default_rng(1)
    with IcpEngine(0, experiments=True) as e:
        e.set_search_mode("brute")
        e.set_target(rng.normal(size=(NT, 3)).astype(np.float32))
        e.set_source(rng.normal(size=(NS, 3)).astype(np.float32))
        assert e.stat("brute_kernel") == 3.0

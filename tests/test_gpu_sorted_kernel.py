"""k_nn_search_sorted (the headline's kernel): the target in the order of its longest axis, a 1-D first filter level.

The order of the target and the filter levels may only change the SPEED: every answer must be bit-identical to the
oracle's brute force and to the kernel it replaced (OA_NN_SORT=0: k_nn_search_filtered).  The cases are chosen for
what is new -- slabs of the sorted order (wide, degenerate, of very different scales), radii far above and below the
slab width, huge offsets, original indices behind the order.  Both LDS tile sizes run (OA_NN_BIGTILE)."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KERNEL_OF = {"0": 1.0, "1": 3.0}        # OA_NN_SORT -> OA_STAT_BRUTE_KERNEL
CASES = ["uniform", "outlier_slabs", "equal_u", "two_blobs", "duplicates", "lattice_ties", "dynamic_range",
         "line_along_u", "query_far_along_u", "tiny_cluster_in_big_box", "nt_1", "nt_5", "nt_1025", "denormal_u",
         "exact_copies", "far_offset", "planes_along_u", "two_scales", "radius_above_half_range"]


def _case(case):
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    nt, ns = 30000, 5000
    if case == "uniform":
        tgt = rng.uniform(-1, 1, size=(nt, 3))
        src = tgt[rng.permutation(nt)[:ns]] + rng.normal(0, 4e-3, size=(ns, 3))
    elif case == "outlier_slabs":            # a few far outliers stretch the first / last slabs of the order
        tgt = rng.uniform(-1, 1, size=(nt, 3))
        tgt[:40] *= 3000.0
        src = np.concatenate([rng.uniform(-1, 1, size=(ns - 200, 3)), rng.uniform(-3000, 3000, size=(200, 3))])
    elif case == "equal_u":                  # every vertex has the same coordinate along the longest axis' runner-up ... and
        tgt = rng.uniform(-1, 1, size=(nt, 3)) * [4.0, 1.0, 1.0]   # the longest axis itself is quantised to 3 values
        tgt[:, 0] = rng.integers(-1, 2, size=nt) * 4.0
        src = rng.uniform(-1, 1, size=(ns, 3)) * [4.5, 1.0, 1.0]
    elif case == "two_blobs":
        tgt = np.concatenate([rng.normal(0, 0.05, size=(nt // 2, 3)) + [40.0, 0, 0], rng.normal(0, 0.05, size=(nt // 2, 3)) - [40.0, 0, 0]])
        src = np.concatenate([rng.normal(0, 0.05, size=(ns // 2, 3)) + [40.0, 0, 0], rng.normal(0, 30.0, size=(ns // 2, 3))])
    elif case == "duplicates":               # ties on the distance: the lowest ORIGINAL index must win, whatever the order
        base = rng.uniform(-1, 1, size=(500, 3))
        tgt = base[rng.integers(0, 500, size=nt)]
        src = base[rng.integers(0, 500, size=ns)] + rng.normal(0, 1e-3, size=(ns, 3))
    elif case == "lattice_ties":
        tgt = rng.integers(-12, 13, size=(nt, 3)) * 0.125
        src = rng.integers(-12, 12, size=(ns, 3)) * 0.125 + 0.0625
    elif case == "dynamic_range":            # 1e-3 .. 1e+4 along the sorted axis in one cloud
        mag = 10.0 ** rng.uniform(-3, 4, size=(nt, 1))
        tgt = rng.normal(size=(nt, 3)) * mag
        mag = 10.0 ** rng.uniform(-3, 4, size=(ns, 1))
        src = rng.normal(size=(ns, 3)) * mag
    elif case == "line_along_u":             # one-dimensional cloud: the first level does all the work
        t = rng.uniform(-5, 5, size=(nt, 1))
        tgt = t * [1.0, 0.0, 0.0] + rng.normal(0, 1e-7, size=(nt, 3))
        s = rng.uniform(-6, 6, size=(ns, 1))
        src = s * [1.0, 0.0, 0.0] + rng.normal(0, 1e-3, size=(ns, 3))
    elif case == "query_far_along_u":
        tgt = rng.uniform(-1, 1, size=(nt, 3))
        src = rng.uniform(-1, 1, size=(ns, 3)) + [70000.0, 0, 0]
        src[::2] -= [140000.0, 0, 0]
    elif case == "tiny_cluster_in_big_box":  # 99 % of the vertices in a ball of 1e-4 inside a box of 100
        tgt = np.concatenate([rng.normal(0, 1e-4, size=(nt - 300, 3)) + [3.0, 1.0, -2.0], rng.uniform(-50, 50, size=(300, 3))])
        src = np.concatenate([rng.normal(0, 1e-4, size=(ns - 300, 3)) + [3.0, 1.0, -2.0], rng.uniform(-50, 50, size=(300, 3))])
    elif case.startswith("nt_"):
        n = int(case[3:])
        tgt = rng.uniform(-1, 1, size=(n, 3))
        src = rng.uniform(-1, 1, size=(ns, 3))
    elif case == "exact_copies":             # best = 0 for most points: the smallest radii the filter ever sees
        tgt = rng.uniform(-1, 1, size=(nt, 3))
        src = tgt[rng.permutation(nt)[:ns]].copy()
        src[::5] += rng.normal(0, 1e-7, size=src[::5].shape)
    elif case == "far_offset":               # extent 1 at distance 1e3 .. 2e3 from the origin
        tgt = rng.uniform(-0.5, 0.5, size=(nt, 3)) + [1000.0, -2000.0, 500.0]
        src = rng.uniform(-0.5, 0.5, size=(ns, 3)) + [1000.0, -2000.0, 500.0]
    elif case == "planes_along_u":           # 7 planes of constant u: every tile is a slab of zero width
        tgt = rng.uniform(-1, 1, size=(nt, 3))
        tgt[:, 0] = rng.integers(-3, 4, size=nt) * 0.75
        src = rng.uniform(-1, 1, size=(ns, 3)) * [3.0, 1.0, 1.0]
    elif case == "two_scales":               # the same shape at scale 1 and at scale 1e-3, interleaved along u
        a = rng.uniform(-1, 1, size=(nt // 2, 3))
        tgt = np.concatenate([a, a[: nt - nt // 2] * 1e-3 + [0.25, 0.0, 0.0]])
        s = rng.uniform(-1, 1, size=(ns // 2, 3))
        src = np.concatenate([s, s[: ns - ns // 2] * 1e-3 + [0.25, 0.0, 0.0]])
    elif case == "radius_above_half_range":  # seeds so bad that the scaled radius leaves the half-precision range
        tgt = rng.uniform(-1, 1, size=(nt, 3)) * [1.0, 1e-3, 1e-3]
        src = rng.uniform(-1, 1, size=(ns, 3)) * [1.0, 300.0, 300.0]
    elif case == "denormal_u":
        tgt = rng.uniform(-1, 1, size=(nt, 3)) * [1e-40, 1e-42, 1e-44]
        src = rng.uniform(-1, 1, size=(ns, 3)) * [1e-40, 1e-42, 1e-44]
    else:
        raise KeyError(case)
    return tgt.astype(np.float32), src.astype(np.float32)


@pytest.mark.parametrize("sort,bigtile", [("0", "0"), ("1", "0"), ("1", "1")])
@pytest.mark.parametrize("case", CASES)
def test_sorted_kernel_answers_are_the_oracles(orc, case, sort, bigtile, monkeypatch):
    from object_alignment_amd.engine import IcpEngine
    monkeypatch.setenv("OA_NN_SORT", sort)
    monkeypatch.setenv("OA_NN_BIGTILE", bigtile)         # 1: the 1024-vertex LDS tile of large targets, also for these small ones
    tgt, src = _case(case)
    eye = np.identity(4, dtype=np.float32)
    exp = sort == "0"                                    # the predecessor lives in liboa_icp_exp.so; the sorted kernel is tested in the default library
    with IcpEngine(0, experiments=exp) as e:
        e.set_search_mode("brute")
        e.set_target(tgt)
        e.set_source(src)
        e.set_matrices(eye, eye)
        idx, d2, _ = e.nn_search()
        assert e.stat("brute_kernel") == (KERNEL_OF[sort] if len(tgt) >= 2 else (1.0 if exp else 0.0))    # (one vertex: nothing to order)
        ridx, rd2 = orc.nn_brute(src, tgt)
        assert np.array_equal(idx, ridx), (case, sort)
        assert np.array_equal(d2, rd2), (case, sort)
        # seeds (every point's correspondence of this pass) + a moved pose: the seeded search
        e.make_pairs(1e30)
        m = eye.copy()
        m[:3, 3] = np.float32(0.01) * np.abs(tgt).max()
        e.set_matrices(m, eye)
        idx2, d22, _ = e.nn_search()
    moved = np.array([orc.mat4_mul_vec3(m, p) for p in src], np.float32)
    r2, rd22 = orc.nn_brute(moved, tgt)
    assert np.array_equal(idx2, r2) and np.array_equal(d22, rd22), (case, sort, "seeded")


@pytest.mark.parametrize("home", ["1", "0"])
@pytest.mark.parametrize("R,splits", [(1, 0), (2, 3), (8, 0), (4, 7), (4, 41)])
def test_sorted_kernel_geometry_variants(orc, R, splits, home, monkeypatch):
    """Points per thread, target splits (a seed's owner is seed index mod splits) and the two forms of an unseeded search
    (k_nn_seed_sorted + a launch seeded from keys; one unseeded launch) -- same answers."""
    from object_alignment_amd.engine import IcpEngine
    monkeypatch.setenv("OA_NN_HOME_PASS", home)
    monkeypatch.setenv("OA_NN_R", str(R))
    if splits:
        monkeypatch.setenv("OA_NN_SPLITS", str(splits))
    rng = np.random.default_rng(R * 100 + splits)
    tgt = rng.normal(size=(70000, 3)).astype(np.float32)
    src = rng.normal(size=(4100, 3)).astype(np.float32)
    eye = np.identity(4, dtype=np.float32)
    with IcpEngine(0, experiments=(R == 8)) as e:        # (8 points per thread is an OA_EXPERIMENTS instantiation)
        e.set_search_mode("brute")
        e.set_target(tgt)
        e.set_source(src)
        e.set_matrices(eye, eye)
        idx, d2, _ = e.nn_search()
        assert e.stat("brute_kernel") == 3.0
        e.make_pairs(1e30)
        idx2, d22, _ = e.nn_search()                # seeded, nothing moved: every seed is its own answer
    ridx, rd2 = orc.nn_brute(src, tgt)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)
    assert np.array_equal(idx2, ridx) and np.array_equal(d22, rd2)


def test_sorted_and_filtered_loops_agree_bitwise(monkeypatch):
    """A 12-iteration loop on the sorted kernel and on its predecessor: the same matrices, bit for bit."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    src, tgt, mxa, mxb = synth.c3_random_pair(120000, seed=77)[:4]
    out = {}
    for sort in ("1", "0"):
        monkeypatch.setenv("OA_NN_SORT", sort)
        with IcpEngine(0, experiments=(sort == "0")) as e:  # the sorted kernel of the default library against its predecessor in liboa_icp_exp.so
            e.set_search_mode("brute")
            e.set_target(tgt)
            e.set_source(src)
            e.set_matrices(mxa, mxb)
            r = e.run(iters=12, thresh=0.5, early_exit=False)
            out[sort] = (r.matrix_world.copy(), r.iters_done, e.stat("brute_kernel"))
    assert out["1"][2] == 3.0 and out["0"][2] == 1.0
    assert out["1"][1] == out["0"][1] == 12
    assert np.array_equal(out["1"][0], out["0"][0])


def test_sorted_images_are_built_for_the_mode_that_uses_them(orc):
    """A target uploaded under AUTO (grid / tree searches) does not pay for the sorted images; switching to brute force builds
    them, and the answers are the oracle's either way."""
    from object_alignment_amd.engine import IcpEngine
    rng = np.random.default_rng(5)
    tgt = rng.uniform(-1, 1, size=(20000, 3)).astype(np.float32)
    src = rng.uniform(-1, 1, size=(3000, 3)).astype(np.float32)
    eye = np.identity(4, dtype=np.float32)
    ridx, rd2 = orc.nn_brute(src, tgt)
    with IcpEngine(0) as e:
        e.set_target(tgt)                            # AUTO
        e.set_source(src)
        e.set_matrices(eye, eye)
        assert e.stat("brute_kernel") == 0.0         # no sorted images yet: brute force would be the plain k_nn_search (liboa_icp_exp.so: k_nn_search_filtered)
        idx, d2, _ = e.nn_search()
        assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)
        e.set_search_mode("brute")
        assert e.stat("brute_kernel") == 3.0
        idx, d2, _ = e.nn_search()
        assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)
        e.set_target(tgt[::-1].copy())               # a new target under brute: built with the upload
        assert e.stat("brute_kernel") == 3.0
        idx, d2, _ = e.nn_search()
    r2, rd22 = orc.nn_brute(src, tgt[::-1].copy())
    assert np.array_equal(idx, r2) and np.array_equal(d2, rd22)


@pytest.mark.parametrize("wave_order", ["1", "0"])
def test_wave_order_changes_nothing_but_speed(orc, wave_order, monkeypatch):
    """k_sorted_wave_order decides which slot a lane takes as its point r (the wave's slots in the order of u at the current
    pose); every slot is still searched exactly once: same answers with it and without, for a rotated pose as well."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    monkeypatch.setenv("OA_NN_WAVE_ORDER", wave_order)
    monkeypatch.setenv("OA_NN_R", "4")
    rng = np.random.default_rng(21)
    tgt = rng.uniform(-1, 1, size=(70000, 3)).astype(np.float32)
    src = (tgt[rng.permutation(70000)[:9000]] + rng.normal(0, 3e-3, size=(9000, 3))).astype(np.float32)
    src[::97] = np.float32(np.nan)                       # (keys that do not order: still a permutation)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.4, -0.9, 0.3]), [0.05, -0.02, 0.01])
    eye = np.identity(4, dtype=np.float32)
    with IcpEngine(0) as e:
        e.set_search_mode("brute")
        e.set_target(tgt)
        e.set_source(src)
        e.set_matrices(mxa, eye)
        idx, d2, _ = e.nn_search()
        e.make_pairs(1e30)
        idx2, d22, _ = e.nn_search()
    moved = np.array([orc.mat4_mul_vec3(mxa, p) for p in src], np.float32)
    ridx, rd2 = orc.nn_brute(moved, tgt)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2, equal_nan=True)
    assert np.array_equal(idx2, ridx) and np.array_equal(d22, rd2, equal_nan=True)


@pytest.mark.parametrize("persist,n_src", [("4", 9000), ("1", 9000), ("4", 300), ("0", 9000)])
def test_work_queue_changes_nothing_but_speed(orc, persist, n_src, monkeypatch):
    """k_nn_search_sorted's work queue (long launches: as many workgroups as the chip holds take (split, block) items off eight
    queues, oa_kernels.hpp) forced onto a small launch -- every item is still searched exactly once: unseeded (home pass + pass 2)
    and seeded answers are the oracle's, with 1024 / 256 workgroups (more than items, fewer than items), one block, and without."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    monkeypatch.setenv("OA_NN_PERSIST", persist)
    monkeypatch.setenv("OA_NN_QUEUE_MIN_ITEMS", "0")
    monkeypatch.setenv("OA_NN_R", "4")
    rng = np.random.default_rng(77)
    tgt = rng.uniform(-1, 1, size=(140000, 3)).astype(np.float32)
    src = (tgt[rng.permutation(len(tgt))[:n_src]] + rng.normal(0, 3e-3, size=(n_src, 3))).astype(np.float32)
    src[::89] = np.float32(np.nan)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.3, 0.7, -0.2]), [0.02, -0.03, 0.01])
    eye = np.identity(4, dtype=np.float32)
    with IcpEngine(0) as e:
        e.set_search_mode("brute")
        e.set_target(tgt)
        e.set_source(src)
        e.set_matrices(mxa, eye)
        idx, d2, _ = e.nn_search()
        queued_first = e.stat("brute_queue_wgs")
        e.make_pairs(1e30)                               # winner records = seeds
        idx2, d22, _ = e.nn_search()
        assert (e.stat("brute_queue_wgs") > 0) == (persist != "0") and (queued_first > 0) == (persist != "0")
    moved = np.array([orc.mat4_mul_vec3(mxa, p) for p in src], np.float32)
    ridx, rd2 = orc.nn_brute(moved, tgt)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2, equal_nan=True)
    assert np.array_equal(idx2, ridx) and np.array_equal(d22, rd2, equal_nan=True)


def test_sorted_kernel_on_a_2M_vertex_target(orc):
    """BASELINE config 5's target size through the sorted images (1954 LDS tiles, the seeded launch's short splits): an unseeded
    and a seeded search of 6000 points against 2M surface vertices, indices and distances the oracle's brute force."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    tgt = synth.bunny_surface(2_000_000, 0.0)
    rng = np.random.default_rng(55)
    src = (tgt[rng.permutation(len(tgt))[:6000]] + rng.normal(0, 2e-3, size=(6000, 3))).astype(np.float32)
    eye = np.identity(4, dtype=np.float32)
    m = synth.rigid4(synth.rotation_from_rotvec([0.004, -0.003, 0.002]), [0.003, -0.002, 0.001])
    with IcpEngine(0) as e:
        e.set_search_mode("brute")
        e.set_target(tgt)
        e.set_source(src)
        e.set_matrices(eye, eye)
        idx, d2, _ = e.nn_search()
        assert e.stat("brute_kernel") == 3.0
        e.make_pairs(1e30)                               # winner records = seeds
        e.set_matrices(m, eye)
        idx2, d22, _ = e.nn_search()
    ridx, rd2 = orc.nn_brute(src, tgt)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)
    moved = np.array([orc.mat4_mul_vec3(m, p) for p in src], np.float32)
    r2, rd22 = orc.nn_brute(moved, tgt)
    assert np.array_equal(idx2, r2) and np.array_equal(d22, rd22)

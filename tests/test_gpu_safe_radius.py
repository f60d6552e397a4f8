"""The vertex grid search takes a seed on its SAFE RADIUS (oa_grid.hpp: k_grid_safe_radius, docs/HISTORY.md 4.4 "safe radii"): a
query closer to its seed than half the seed's distance to its nearest other target needs no scan.  The answers must stay
those of the exhaustive search (/root/reference/functions/general.py:297, SURVEY D2's vertex rule): index and float32 d2
bit for bit -- also for queries placed ON the rule's edge, next to duplicates, and in crowded cells."""
import re

import numpy as np
import pytest


def _adversarial_cloud(seed):
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    base = rng.uniform(-1, 1, size=(30_000, 3)).astype(np.float32)
    dup = base[rng.integers(0, len(base), 200)]                                   # exact duplicates: S = 0
    near = (base[rng.integers(0, len(base), 200)].astype(np.float64) * (1.0 + 2e-7)).astype(np.float32)   # one or two ulps away
    crowd = (np.float32([0.3, -0.2, 0.1]) + rng.normal(size=(3000, 3)).astype(np.float32) * np.float32(2e-4))   # a crowded block of cells
    tgt = np.ascontiguousarray(np.concatenate([base, dup, near, crowd]).astype(np.float32))
    t64 = tgt.astype(np.float64)
    dd, ii = cKDTree(t64).query(t64, k=2)
    S, other = dd[:, 1], ii[:, 1]
    pick = rng.permutation(len(tgt))[:24_000]
    # factors of S / 2 around 1: the rule accepts below sqrt(1 - 1e-4) ~ 1 - 5e-5
    fs = np.array([0.5, 0.9, 0.999, 0.9999, 0.99994, 0.99995, 0.99996, 0.99999, 0.999999, 1.0, 1.000001, 1.00001, 1.0001, 1.01])
    f = fs[np.arange(len(pick)) % len(fs)]
    t, o = t64[pick], t64[other[pick]]
    toward = t + (o - t) * (0.5 * f)[:, None]                                       # on the segment to the nearest other target
    u = rng.normal(size=(len(pick), 3))
    u /= np.linalg.norm(u, axis=1)[:, None]
    anyway = t + u * (0.5 * f * S[pick])[:, None]                                   # the same distances, any direction
    src = np.concatenate([toward[: len(pick) // 2], anyway[len(pick) // 2:], t64[pick[:2000]]]).astype(np.float32)   # + exact hits
    return np.ascontiguousarray(src), tgt


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", ["1", "2", "4"])
def test_safe_radius_edge_queries_equal_brute_force(lanes, monkeypatch):
    """Seeds from a first search at a slightly shifted pose, then searches at poses that put the queries at 0.5 ... 1.01
    times half the nearest-neighbour distance of a target -- on the segment towards that neighbour (the bisector plane,
    where the two targets tie) and in random directions; duplicates (radius 0), one-ulp neighbours and a crowded block
    (radius withheld) are among the targets.  Grid search with the rule, grid search without it and brute force: the same
    (index, d2) bits at every pose -- and the whole-shard tree search, which takes seeds on the same radii."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    monkeypatch.setenv("OA_GRID_LANES", lanes)
    src, tgt = _adversarial_cloud(5)
    eye = np.identity(4, dtype=np.float32)
    poses = [synth.rigid4(None, [1e-5, -2e-5, 1.5e-5]), eye, synth.rigid4(None, [-3e-6, 2e-6, 1e-6]), eye,
             synth.rigid4(synth.rotation_from_rotvec([1e-5, -1e-5, 2e-5]), [0.0, 0.0, 0.0]), eye]
    out = {}
    for tag, mode, env in (("safe", "grid", "2"), ("nosafe", "grid", "0"), ("tree_safe", "bvh", "2"), ("tree_nosafe", "bvh", "0"),
                           ("brute", "brute", "2")):
        monkeypatch.setenv("OA_GRID_SAFE", env)                 # 2: radii built with the grid (the default builds them after 8 loop iterations)
        with IcpEngine(0) as e:
            e.set_search_mode(mode)
            e.set_target(tgt)
            e.set_source(src, stride=1)
            res = []
            for mxa in poses:
                e.set_matrices(mxa, eye)
                idx, d2, _ = e.nn_search()
                res.append((idx.copy(), d2.copy()))
            out[tag] = res
    for k in range(len(poses)):
        for tag in ("safe", "nosafe", "tree_safe", "tree_nosafe"):
            assert np.array_equal(out[tag][k][0], out["brute"][k][0]), (tag, k, int((out[tag][k][0] != out["brute"][k][0]).sum()))
            assert np.array_equal(out[tag][k][1].view(np.uint32), out["brute"][k][1].view(np.uint32)), (tag, k)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["settled", "far_start", "half_target", "lanes4", "dups", "tree"])
def test_safe_radius_loops_same_bits_as_without(orc, case, monkeypatch):
    """Loops with and without the rule (OA_GRID_SAFE=0), fast and safe grid paths: bitwise the same steps and matrices --
    and the oracle's K per step."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    n = {"lanes4": 9_000, "tree": 3_000}.get(case, 60_000)          # ("tree": the accumulating whole-shard tree search of a small shard)
    src, tgt, mxa, mxb = synth.c2_bunny_pair(n)
    thresh, iters = 0.5, 6
    if case == "far_start":
        mxa = synth.rigid4(synth.rotation_from_rotvec([0.2, -0.1, 0.15]), [0.3, -0.2, 0.25])
        thresh = 1.0
    if case == "half_target":
        tgt = np.ascontiguousarray(tgt[tgt[:, 0] > 0.0])
    if case == "dups":
        tgt = np.ascontiguousarray(np.concatenate([tgt, tgt[::7], tgt[::11] * np.float32(1.0 + 1e-7)]).astype(np.float32))
    out = {}
    for tag, env in (("safe_on", {"OA_GRID_SAFE": "2"}), ("safe_off", {"OA_GRID_SAFE": "0"}),
                     ("on_fast", {"OA_GRID_SAFE": "2", "OA_GRID_PATH": "fast"}), ("on_safe_path", {"OA_GRID_SAFE": "2", "OA_GRID_PATH": "safe"})):
        monkeypatch.delenv("OA_GRID_PATH", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with IcpEngine(0) as e:
            e.set_search_mode("bvh" if case == "tree" else "grid")
            e.set_target(tgt)
            e.set_source(src, stride=1)
            e.set_matrices(mxa, mxb)
            r1 = e.run(iters=iters, thresh=thresh, early_exit=False)
            r2 = e.run(iters=3, thresh=thresh, early_exit=False)          # a second loop continues on warm seeds (and their radii)
            out[tag] = (r1, r2)
    for tag in ("safe_off", "on_fast", "on_safe_path"):
        for a, b in zip(out["safe_on"], out[tag]):
            assert np.array_equal(a.step_K, b.step_K), (case, tag)
            assert np.array_equal(a.step_M, b.step_M) and np.array_equal(a.matrix_world, b.matrix_world), (case, tag)
    ref = orc.icp_run(src, tgt, mxa, mxb, iters=iters, sample=1, thresh=thresh, target_d=1e-300, use_target=True, kd=orc.KDTree(tgt))
    assert np.array_equal(out["safe_on"][0].step_K, ref["step_K"])
    assert np.abs(out["safe_on"][0].step_M - ref["step_M"]).max() < 1e-9


@pytest.mark.gpu
def test_safe_radius_is_taken_once_the_pose_settles(capfd, monkeypatch):
    """The instrumented build (OA_GRID_STATS=1) reports how many queries the rule settled: most of a noisy copy of the
    target once the pose has converged (OA_GRID_SAFE=2: radii built with the grid), the same from the tenth iteration on
    with the default (radii built after eight iterations: short calls never pay for them), none with
    OA_GRID_SAFE=0 -- and the three runs end with the same bits."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    src, tgt, mxa, mxb = synth.c3_random_pair(300_000, seed=77)
    monkeypatch.setenv("OA_GRID_STATS", "1")
    monkeypatch.setenv("OA_GRID_PATH", "fast")
    monkeypatch.setenv("OA_GRID_LANES", "1")                         # (the instrumented kernel is the one-lane-per-query form)
    got = {}
    for env in ("2", "1", "0"):
        monkeypatch.setenv("OA_GRID_SAFE", env)                 # 2: radii built with the grid (the default builds them after 8 loop iterations)
        capfd.readouterr()
        with IcpEngine(0, experiments=True) as e:               # (the instrumented launch is an OA_EXPERIMENTS kernel)
            e.set_search_mode("grid")
            e.set_target(tgt)
            e.set_source(src, stride=1)
            e.set_matrices(mxa, mxb)
            r = e.run(iters=14, thresh=0.5, early_exit=False)
        err = capfd.readouterr().err
        counts = [(int(a), int(b)) for a, b in re.findall(r"vertex grid phases: (\d+) of (\d+) queries settled by the seed's safe radius", err)]
        got[env] = (r, counts)
    on, lazy, off = got["2"][1], got["1"][1], got["0"][1]
    assert len(on) >= 14 and len(lazy) >= 14 and len(off) >= 14, (on, lazy, off)
    assert on[0][0] == 0                                            # a cold start has no seeds
    assert on[5][0] > 0.8 * on[5][1], on
    # the default builds the radii once the target has seen 8 iterations; the search behind that one records them, the next uses them
    assert all(c == 0 for c, _ in lazy[:8]) and lazy[-1][0] > 0.8 * lazy[-1][1], lazy
    assert all(c == 0 for c, _ in off), off
    for tag in ("1", "0"):
        assert np.array_equal(got["2"][0].step_M, got[tag][0].step_M) and np.array_equal(got["2"][0].matrix_world, got[tag][0].matrix_world)

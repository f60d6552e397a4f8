"""Surface mode, round 5: queries settled by their seed triangle and its neighbours (oa_tri_ring.hpp) give the bits the rings of
cells give -- which are the oracle's brute force over all triangles.  Needs a real MI355X: run with `pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cofind(orc, src, mxa, mxb):
    imx2 = orc.mat4_inverted(mxb)
    return np.array([orc.mat4_mul_vec3(imx2, orc.mat4_mul_vec3(mxa, p)) for p in src], np.float32)


def _on_surface(rng, v, t, n, lift):
    """n points on random triangles of the mesh (uniform barycentric), lifted off it by up to `lift` along a random direction"""
    f = rng.integers(0, len(t), size=n)
    w = rng.dirichlet([1.0, 1.0, 1.0], size=n)
    p = (v[t[f, 0]].astype(np.float64) * w[:, :1] + v[t[f, 1]] * w[:, 1:2] + v[t[f, 2]] * w[:, 2:3])
    return (p + rng.normal(size=(n, 3)) * lift).astype(np.float32)


def _meshes():
    from object_alignment_amd import synth
    rng = np.random.default_rng(505)
    out = {}
    v, t = synth.bumpy_icosphere_mesh(4)                                   # 5120 triangles, valence 5-6: every ring fits
    out["ico"] = (v, t, _on_surface(rng, v, t, 6000, 0.002))
    v, t = synth.lattice_surface_mesh(90, 180)                             # thin triangles towards the poles
    out["lattice"] = (v, t, _on_surface(rng, v, t, 9000, 0.001))
    # two sheets a fraction of an edge apart: the clearance is the gap, queries sit between them and on either side
    g = np.linspace(-1.0, 1.0, 41, dtype=np.float32)
    X, Y = np.meshgrid(g, g, indexing="ij")
    def sheet(z):
        return np.stack([X.ravel(), Y.ravel(), np.full(X.size, z, np.float32) + np.float32(0.01) * np.sin(3 * X.ravel())], 1).astype(np.float32)
    i, j = np.meshgrid(np.arange(40), np.arange(40), indexing="ij")
    a = (i * 41 + j).ravel(); b = a + 1; c = a + 41; d = a + 42
    quad = np.concatenate([np.stack([a, c, b], 1), np.stack([b, c, d], 1)]).astype(np.int32)
    v = np.concatenate([sheet(0.0), sheet(0.012)])
    t = np.concatenate([quad, quad + 41 * 41]).astype(np.int32)
    q = _on_surface(rng, v, t, 8000, 0.004)
    out["sheets"] = (v, t, q)
    # the icosphere with every triangle on its own three vertices (no shared indices: neighbours are found by geometry)
    v0, t0 = synth.bumpy_icosphere_mesh(3)
    v = v0[t0.ravel()].astype(np.float32)
    t = np.arange(len(v), dtype=np.int32).reshape(-1, 3)
    out["unwelded"] = (v, t, _on_surface(rng, v, t, 4000, 0.003))
    # needles and slivers among ordinary triangles, duplicated triangles, a zero-area triangle
    v, t = synth.bumpy_icosphere_mesh(3)
    v = v.copy()
    extra_v = np.array([[0.0, 0.0, 1.5], [1e-4, 0.0, 1.5], [0.5, 1e-7, 1.9], [0.25, 5e-8, 1.7]], np.float32)
    nv = len(v)
    extra_t = np.array([[nv, nv + 1, nv + 2], [nv, nv + 2, nv + 1], [nv, nv + 3, nv + 2], [0, 1, nv + 2], [5, 5, 7]], np.int32)
    v = np.concatenate([v, extra_v]); t = np.concatenate([t, extra_t, t[:50]])
    out["odd"] = (v, t, np.concatenate([_on_surface(rng, v[:nv], t[:-55], 4000, 0.004),
                                        np.array([[0.2, 0.0, 1.6], [0.0, 0.0, 1.5], [0.3, 0.0, 1.7], [0.25, 0.0, 1.7]], np.float32)]))
    # integer lattice: exact ties everywhere, degenerate triangles
    v = rng.integers(-5, 6, size=(400, 3)).astype(np.float32)
    t = rng.integers(0, 400, size=(3000, 3)).astype(np.int32)
    out["ties"] = (v, t, (rng.integers(-5, 5, size=(2000, 3)) + rng.choice([0.0, 0.5, 0.25], size=(2000, 3))).astype(np.float32))
    return out


def _poses(rng, k):
    from object_alignment_amd import synth
    out = [np.identity(4, dtype=np.float32)]
    for s in (3e-3, 1e-3, 3e-4, 1e-4, 0.0)[:k]:
        out.append(synth.rigid4(synth.rotation_from_rotvec(rng.normal(size=3) * s), rng.normal(size=3) * s).astype(np.float32))
    return out


@pytest.mark.parametrize("lanes", ["0", "1", "2", "4"])
@pytest.mark.parametrize("case", ["ico", "lattice", "sheets", "unwelded", "odd", "ties"])
def test_ring_settled_queries_are_bit_exact(orc, case, lanes, monkeypatch):
    """Seeded searches at a sequence of nearby poses, neighbour lists built with the grid: index and float32 distance of every
    query equal the oracle's brute force over all triangles; and the lists do settle queries (the diagnostic count)."""
    from object_alignment_amd.engine import IcpEngine
    if lanes != "0" and case not in ("ico", "sheets", "odd"):
        pytest.skip("lane variants on three meshes")
    monkeypatch.setenv("OA_TRI_RING", "2")
    if lanes != "0":
        monkeypatch.setenv("OA_GRID_LANES", lanes)
    v, t, q = _meshes()[case]
    rng = np.random.default_rng(11)
    eye = np.identity(4, dtype=np.float32)
    settled = []
    with IcpEngine(0, experiments=True) as e:
        e.set_search_mode("grid")
        e.set_target_mesh(v, t)
        assert e.stat("tri_ring") == 1.0
        e.set_source(q)
        for k, mx in enumerate(_poses(rng, 5)):
            e.set_matrices(mx, eye)
            idx, d2, _ = e.nn_search()                         # the first one unseeded, the others on the last pose's answers
            e.make_pairs(1e3)                                   # (plants the seeds: one-shot searches do not)
            w = _cofind(orc, q, mx, eye)
            face, _, rd2 = orc.nn_tri_brute(w, v, t)
            assert np.array_equal(d2, rd2), (case, k, int((d2 != rd2).sum()))
            assert np.array_equal(idx, face), (case, k, int((idx != face).sum()))
            settled.append(e.stat("tri_ring_accepts"))
    print("ring %s: queries the lists would settle after each search: %s of %d" % (case, [int(s) for s in settled], len(q)))
    if case in ("ico", "lattice", "unwelded"):
        assert settled[-1] > 0.5 * len(q), settled
    if case == "sheets":
        assert settled[-1] > 0


@pytest.mark.parametrize("surface_case", ["ico", "lattice"])
def test_ring_loop_equals_loop_without(orc, surface_case, monkeypatch):
    """A whole loop with the lists (lazy: built after four searches; and built with the grid) leaves the matrices, the pair
    counts and the statistics of a loop without them, bit for bit -- and of the oracle's loop."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    v, t, _ = _meshes()[surface_case]
    src = synth.bunny_surface(30000, 0.5) if surface_case == "lattice" else (synth.bumpy_icosphere(5)[::3]).astype(np.float32)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.03, -0.02, 0.025]), [0.01, -0.015, 0.01]).astype(np.float32)
    eye = np.identity(4, dtype=np.float32)
    got = {}
    for ring in ("0", "1", "2"):
        monkeypatch.setenv("OA_TRI_RING", ring)
        with IcpEngine(0, experiments=True) as e:
            e.set_search_mode("grid")
            e.set_target_mesh(v, t)
            e.set_source(src)
            e.set_matrices(mxa, eye)
            res = e.run(iters=14, thresh=0.5, target_d=1e-12)
            got[ring] = (res, e.stat("tri_ring"), e.stat("tri_ring_accepts"))
    assert got["0"][1] == 0.0 and got["1"][1] == 1.0 and got["2"][1] == 1.0
    assert got["1"][2] > 0.5 * len(src) and got["2"][2] > 0.5 * len(src), (got["1"][2], got["2"][2])
    for ring in ("1", "2"):
        a, b = got["0"][0], got[ring][0]
        assert np.array_equal(a.step_K, b.step_K)
        assert np.array_equal(a.step_M, b.step_M)
        assert np.array_equal(a.matrix_world, b.matrix_world)
        assert np.array_equal(a.step_stats, b.step_stats)
    ref = orc.icp_run(src, v, mxa, eye, iters=14, sample=1, tris=t, target_d=1e-12)
    assert got["2"][0].iters_done == ref["iters_done"] == 14
    assert np.array_equal(got["2"][0].step_K, ref["step_K"])
    assert np.abs(got["2"][0].step_M - ref["step_M"]).max() < 1e-9

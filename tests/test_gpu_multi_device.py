"""The multi-device context (oa_create_multi): one process drives every GPU, the per-iteration
exchange of the 24 sums lives inside liboa_icp.so.  On the one-GPU test box the device list names device 0 several
times -- the same exchange code (k_reduce_post / k_gather_solve_update, or RCCL with a world of one)."""
import os
import socket
import subprocess
import sys
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32_ULP = 2.5e-7
LOOPS = ["icp_loop_ico_10", "icp_loop_bumpy_converge", "icp_loop_include", "icp_loop_exclude", "icp_loop_bumpy_scale"]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def _run_fixture(g, eng, mode="auto"):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_parity import _settings_from
    from object_alignment_amd.operators import IcpAlign
    eng.set_search_mode(mode)
    return IcpAlign(_settings_from(g), engine=eng).run(g["src"], g["tgt"], g["mx_align"], g["mx_base"], vlist=g["vlist"])


@pytest.mark.parametrize("n_dev", [1, 2, 8])
@pytest.mark.parametrize("name", LOOPS)
def test_multi_context_reproduces_the_reference_loops(golden_dir, name, n_dev):
    """The fixtures of the reference's own execute() through n_dev shards and the in-library exchange: iteration
    count, convergence flag and K per iteration exact, per-iteration M 1e-9, final float32 matrix within 1 ulp --
    the bar the single-device loop is held to."""
    from object_alignment_amd.engine import IcpEngine
    g = _load(golden_dir, name)
    with IcpEngine(devices=[0] * n_dev) as eng:
        assert eng.multi and len(eng.devices) == n_dev
        res = _run_fixture(g, eng)
    assert res.iters_done == int(g["iters_done"]) and res.converged == bool(g["converged"])
    assert np.array_equal(res.step_K, g["step_K"])
    assert np.abs(res.step_M - g["step_M"]).max() < 1e-9
    assert np.abs(res.matrix_world - g["final_world"]).max() <= F32_ULP


def test_multi_context_equals_hand_driven_shards(golden_dir):
    """Same shards, same fixed-order reduction, sums added in rank order: the library's exchange must give the very
    bits the split-phase loop gives when the caller adds the two ranks' sums itself."""
    import torch
    from object_alignment_amd import _capi
    from object_alignment_amd.engine import IcpEngine
    g = _load(golden_dir, "icp_loop_bumpy_converge")
    kw = dict(iters=int(g["iters_done"]), thresh=0.5, target_d=0.01, use_target=True, early_exit=True)
    dev = torch.device("cuda:0")
    engs = [IcpEngine(0) for _ in range(2)]
    try:
        sums = [torch.zeros(_capi.OA_NSUMS, dtype=torch.float64, device=dev) for _ in range(2)]
        for r, e in enumerate(engs):
            e.set_stream(torch.cuda.current_stream().cuda_stream)
            e.set_target(g["tgt"])
            e.set_source(g["src"], stride=1, shard_index=r, shard_count=2)
            e.set_matrices(g["mx_align"], g["mx_base"])
            e.run_begin(**kw)
        for _ in range(kw["iters"]):
            for r, e in enumerate(engs):
                e.iter_partial(sums[r].data_ptr())
            total = sums[0] + sums[1]
            for e in engs:
                e.iter_finish(total.data_ptr())
        hand = [e.run_end() for e in engs][0]
    finally:
        for e in engs:
            e.close()
    with IcpEngine(devices=[0, 0]) as m:
        m.set_target(g["tgt"])
        m.set_source(g["src"], stride=1)
        assert m.n_selected == len(g["src"])
        m.set_matrices(g["mx_align"], g["mx_base"])
        res = m.run(**kw)
    assert res.iters_done == hand.iters_done and res.converged == hand.converged
    assert np.array_equal(res.step_K, hand.step_K)
    assert np.array_equal(res.step_M, hand.step_M)
    assert np.array_equal(res.matrix_world, hand.matrix_world)


@pytest.mark.parametrize("surface", [False, True])
def test_multi_iterate_equals_run(surface):
    """The modal step on a multi-device context walks the same iterations as its oa_run."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    v, t = synth.bumpy_icosphere_mesh(4)
    src = (synth.bumpy_icosphere(4)[::2] * np.float32(1.01)).astype(np.float32)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.03, -0.02, 0.04]), [0.02, -0.01, 0.015])
    eye = np.identity(4, dtype=np.float32)
    with IcpEngine(devices=[0, 0, 0]) as m:
        if surface:
            m.set_target_mesh(v, t)
        else:
            m.set_target(v)
        m.set_source(src)
        m.set_matrices(mxa, eye)
        res = m.run(iters=6, thresh=0.5, early_exit=False)
        m.set_matrices(mxa, eye)
        steps = [m.iterate(thresh=0.5)[0] for _ in range(6)]
        mw = m.matrix_world()
    assert np.array_equal(np.stack(steps), res.step_M)
    assert np.array_equal(mw, res.matrix_world)


@pytest.mark.parametrize("surface", [False, True])
@pytest.mark.parametrize("n_dev", [2, 5, 16])
def test_multi_per_point_outputs_in_vlist_order(surface, n_dev):
    """make_pairs (functions/general.py:257-329) and nn_search on a multi-device context: every shard answers for its
    points, the library merges the answers back into the caller's order -- the same arrays as one device gives, pair
    for pair, with a vlist that is neither sorted nor duplicate-free and a stride."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    rng = np.random.default_rng(100 + n_dev)
    v, t = synth.bumpy_icosphere_mesh(4)
    src = (synth.bumpy_icosphere(4) * np.float32(1.02)).astype(np.float32)
    src[::11] += np.float32(0.4)                                    # some points beyond the threshold
    vlist = rng.permutation(len(src))[: len(src) * 3 // 4]
    vlist = np.concatenate([vlist, vlist[:37]])                     # the reference appends per membership: duplicates stay
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.03, -0.02, 0.04]), [0.02, -0.01, 0.015])
    eye = np.identity(4, dtype=np.float32)
    out = []
    for devs in (None, [0] * n_dev):
        with (IcpEngine(0) if devs is None else IcpEngine(devices=devs)) as e:
            if surface:
                e.set_target_mesh(v, t)
            else:
                e.set_target(v)
            e.set_source(src, vlist=vlist, stride=2)
            e.set_matrices(mxa, eye)
            A, B, st = e.make_pairs(0.12, calc_stats=True)
            idx, d2, _ = e.nn_search()
            out.append((A, B, np.array(st), idx, d2))
    assert 0 < out[0][0].shape[1] < (len(vlist) + 1) // 2           # the threshold cuts, so compaction matters
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.allclose(out[0][2], out[1][2], rtol=1e-12, atol=0)
    assert np.array_equal(out[0][3], out[1][3]) and np.array_equal(out[0][4], out[1][4])


@pytest.mark.parametrize("n_dev", [2, 8])
def test_make_pairs_golden_on_a_multi_device_engine(golden_dir, n_dev):
    """The reference's make_pairs fixtures (generated by its own code) through GpuBVH on a multi-device engine."""
    from object_alignment_amd.engine import IcpEngine
    from object_alignment_amd.functions import make_pairs, GpuBVH, AlignObject
    g = _load(golden_dir, "make_pairs")
    with IcpEngine(devices=[0] * n_dev) as eng:
        for i in range(int(g["n_cases"])):
            p = "c%02d_" % i
            align = AlignObject(g[p + "src"], g[p + "mx_align"])
            base = AlignObject(g[p + "tgt"], g[p + "mx_base"])
            bvh = GpuBVH.FromObject(base, None, engine=eng)
            A, B, ds = make_pairs(align, base, bvh, g[p + "vlist"].tolist(), float(g[p + "thresh"]),
                                  int(g[p + "sample"]), calc_stats=bool(g[p + "calc_stats"]))
            name = str(g[p + "name"])
            assert np.array_equal(A, g[p + "A"]), name
            assert np.array_equal(B, g[p + "B"]), name
            if bool(g[p + "calc_stats"]):
                assert np.allclose(ds, g[p + "d_stats"], rtol=1e-9, atol=1e-13), name


def test_threaded_uploads_give_the_same_context(golden_dir, monkeypatch):
    """On different GPUs the children of a multi-device context upload and build their indices at the same time, one
    host thread each.  This box has one GPU (children then share its stream and stay sequential): OA_MULTI_THREADS=1
    forces the threaded path, which must leave every child exactly as the sequential one does -- same loop, bit for bit,
    same pairs -- and must report a child's failure."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    v, t = synth.bumpy_icosphere_mesh(4)
    src = (synth.bumpy_icosphere(4) * np.float32(1.02)).astype(np.float32)
    nrm = (src / np.linalg.norm(src, axis=1, keepdims=True)).astype(np.float32)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.03, -0.02, 0.04]), [0.02, -0.01, 0.015])
    eye = np.identity(4, dtype=np.float32)
    out = []
    for threads in ("0", "1"):
        monkeypatch.setenv("OA_MULTI_THREADS", threads)
        with IcpEngine(devices=[0] * 6) as m:
            for surface in (False, True):
                if surface:
                    m.set_target_mesh(v, t)
                else:
                    m.set_target(v)
                m.set_source(src, vlist=np.arange(0, len(src), 3))
                m.set_normals(nrm, None if surface else (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32), 60.0)
                m.set_matrices(mxa, eye)
                r = m.run(iters=5, thresh=0.5, early_exit=False)
                A, B, _ = m.make_pairs(0.2)
                out.append((r.step_M.copy(), r.step_K.copy(), r.matrix_world.copy(), A, B))
            with pytest.raises(Exception):                          # a bad triangle index fails in every child
                m.set_target_mesh(v, np.array([[0, 1, len(v) + 5]], np.int32))
    n = len(out) // 2
    for a, b in zip(out[:n], out[n:]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_multi_more_devices_than_points(orc):
    """Empty shards (16 children, 10 selected points) post zero sums and do not disturb the others."""
    from object_alignment_amd.engine import IcpEngine
    rng = np.random.default_rng(4)
    tgt = rng.uniform(-1, 1, size=(500, 3)).astype(np.float32)
    src = (tgt[:10] + np.float32(0.01)).astype(np.float32)
    eye = np.identity(4, dtype=np.float32)
    with IcpEngine(devices=[0] * 16) as m:
        m.set_target(tgt)
        m.set_source(src)
        assert m.n_selected == 10
        m.set_matrices(eye, eye)
        res = m.run(iters=3, thresh=0.5, use_target=True, early_exit=False)
    ref = orc.icp_run(src, tgt, eye, eye, iters=3, sample=1, thresh=0.5, target_d=1e-300, use_target=True)
    assert np.array_equal(res.step_K, ref["step_K"])
    assert np.abs(res.step_M - ref["step_M"]).max() < 1e-9


@pytest.mark.parametrize("surface", [False, True])
def test_multi_normals_and_too_few_pairs(orc, surface):
    """The normal-angle extension and the K < 3 error travel through a multi-device context like through a single one."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    v, t = synth.bumpy_icosphere_mesh(4)
    src, nrm = synth.bunny_surface_with_normals(3000, 0.3)
    src = (src * np.float32(np.linalg.norm(v, axis=1).mean() / np.linalg.norm(src, axis=1).mean())).astype(np.float32)
    tn = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.03, -0.02, 0.04]), [0.02, -0.01, 0.015])
    eye = np.identity(4, dtype=np.float32)
    out = []
    for devs in (None, [0, 0, 0]):
        with (IcpEngine(0) if devs is None else IcpEngine(devices=devs)) as e:
            if surface:
                e.set_target_mesh(v, t)
            else:
                e.set_target(v)
            e.set_source(src)
            e.set_normals(nrm, None if surface else tn, 30.0)
            e.set_matrices(mxa, eye)
            out.append(e.run(iters=4, thresh=0.5, early_exit=False))
            e.set_matrices(mxa, eye)
            with pytest.raises(ValueError) as ei:                                  # nothing within 1e-9: K = 0
                e.run(iters=4, thresh=1e-9, early_exit=False)
            assert getattr(ei.value, "partial", None) is not None and ei.value.partial.iters_done == 0
    assert np.array_equal(out[0].step_K, out[1].step_K)
    assert np.abs(out[0].step_M - out[1].step_M).max() < 1e-9
    assert np.abs(out[0].matrix_world - out[1].matrix_world).max() <= F32_ULP


def test_rccl_exchange_world_of_one(golden_dir):
    """OA_EXCHANGE_RCCL on one GPU: librccl is loaded by the library, ncclCommInitAll builds a communicator of one
    and ncclAllReduce runs on the context's stream every iteration.  Identity for one rank: same bits as the mailbox."""
    from object_alignment_amd.engine import IcpEngine
    g = _load(golden_dir, "icp_loop_bumpy_converge")
    with IcpEngine(devices=[0], exchange="rccl") as e:
        res = _run_fixture(g, e)
    with open("/proc/self/maps") as f:
        assert "librccl" in f.read()
    with IcpEngine(devices=[0]) as e:
        ref = _run_fixture(g, e)
    assert res.iters_done == int(g["iters_done"]) and res.converged
    assert np.array_equal(res.step_M, ref.step_M) and np.array_equal(res.matrix_world, ref.matrix_world)
    assert np.abs(res.matrix_world - g["final_world"]).max() <= F32_ULP


def test_rccl_exchange_refuses_duplicate_devices():
    from object_alignment_amd import _capi
    from object_alignment_amd.engine import IcpEngine
    with IcpEngine(devices=[0, 0]) as e:
        with pytest.raises(_capi.OaError) as ei:
            e.set_exchange("rccl")
        assert ei.value.code == _capi.OA_E_RCCL


def test_torch_distributed_rccl_world_of_one(golden_dir):
    """The one-process-per-GPU path's collective: torch.distributed backend "nccl" (= RCCL) with a world of one, the
    all-reduce forced inside run_sharded -- the call the driver's multi-GPU bench issues, executed for real."""
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from object_alignment_amd.engine import IcpEngine
from object_alignment_amd.distributed import EngineShard, run_sharded, new_sums_tensor
g = np.load(%r)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
with IcpEngine(0) as e:
    e.set_target(g["tgt"]); e.set_source(g["src"], stride=1); e.set_matrices(g["mx_align"], g["mx_base"])
    sums = new_sums_tensor(torch.device("cuda:0"))
    res = run_sharded(EngineShard(e, iters=30, thresh=0.5, target_d=0.01, use_target=True, early_exit=True), 30, sums,
                      world_size=1, force_collective=True)
dist.barrier(); dist.destroy_process_group()
assert "librccl" in open("/proc/self/maps").read()
assert res.iters_done == int(g["iters_done"]) and res.converged
assert np.abs(res.matrix_world - g["final_world"]).max() <= 2.5e-7
print("RCCL_WORLD1_OK")
''' % (ROOT, os.path.join(golden_dir, "icp_loop_bumpy_converge.npz"))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_WORLD1_OK" in p.stdout, p.stdout[-3000:]


def test_operator_execute_on_a_multi_device_engine(golden_dir, orc):
    """OBJECT_OT_icp_align.execute with the `devices` setting: the reference's single call (operators/icp_align.py:
    47-161), sharded over the device list inside the library; vertex groups and m_ objects as in the fixture."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_parity import _duck_scene, _settings_from
    from object_alignment_amd.operators import OBJECT_OT_icp_align, icp_align
    g = _load(golden_dir, "icp_loop_exclude")
    prefs = icp_align.get_addon_preferences()
    try:
        for k, v in _settings_from(g).__dict__.items():
            setattr(prefs, k, v)
        prefs.devices = [0, 0, 0, 0]
        ctx, align, m_obj = _duck_scene(g, orc)
        op = OBJECT_OT_icp_align()
        assert op.execute(ctx) == {"FINISHED"}
        assert op.last_result.iters_done == int(g["iters_done"])
        assert np.array_equal(op.last_result.step_K, g["step_K"])
        got = np.array([[align.matrix_world[r][c] for c in range(4)] for r in range(4)], np.float32)
        assert np.abs(got - g["final_world"]).max() <= F32_ULP
    finally:
        for k, v in icp_align.IcpSettings().__dict__.items():
            setattr(prefs, k, v)


def test_device_memory_returns_after_destroy():
    """Default settings: the allocation cache holds at most 256 MiB while a context lives and nothing once the last
    context is gone -- device memory is back where it was (the library is a guest in the host application)."""
    import torch
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    from object_alignment_amd import _capi
    from object_alignment_amd.functions.general import close_default_engines
    import gc
    torch.cuda.init()
    close_default_engines()                                    # engines other tests left alive keep the cache alive
    gc.collect()
    _capi.load().oa_release_cached_memory()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    src, tgt, mxa, mxb = synth.c3_random_pair(400_000)
    for _ in range(2):
        with IcpEngine(0) as e:
            e.set_target(tgt)
            e.set_source(src)
            e.set_matrices(mxa, mxb)
            e.run(iters=3, thresh=0.5, early_exit=False)
            e.set_target(tgt[:300_000])                        # a re-upload leaves released blocks in the cache
            free_live, _ = torch.cuda.mem_get_info(0)
            assert free0 - free_live > 20 << 20                # the context really holds device memory
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(0)
    assert free0 - free1 <= 8 << 20, (free0, free1)            # back to the baseline (allowing for runtime-internal pools)


def test_bench_self_launches_multi_gpu():
    """`python bench.py --gpus 2` with no launcher: one process, the multi-device context; one JSON line, n_gpus 2."""
    env = dict(os.environ, OA_BENCH_SAME_DEVICE="1")
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--n-source", "120000",
           "--n-target", "100000", "--no-cpu-baseline", "--no-surface", "--c5-source", "300000", "--c5-target", "100000",
           "--c5-steps", "3"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["unit"] == "iterations/s" and d["value"] > 0
    assert "in-library" in d["config"]["parallelism"]
    # the BASELINE config 5 leg of a multi-GPU run (masked source + normal-angle test, AUTO search), at a reduced size here
    c5 = d["c5_path"]
    assert "error" not in c5, c5
    assert c5["n_gpus"] == 2 and c5["steps"] == 3 and c5["ms_per_step"] > 0 and c5["n_selected"] == 270000
    assert c5["exchange"].startswith("mailbox") and c5["rccl_ranks"] == 0
    assert 0 < c5["ms_per_nn_search_per_device"]["min"] <= c5["ms_per_nn_search_per_device"]["max"]
    assert c5["predicted_ms_per_step_design_4_7"] is None           # (the prediction is for the full-size configuration)


# ---------------------------------------------------------------------------------------------------------------------
# round 3: the exchange protocol under real concurrency, fault injection, host threads, the partition-once upload
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("box", ["device", "host"])
@pytest.mark.parametrize("n_dev", [2, 8])
@pytest.mark.parametrize("name", LOOPS)
def test_mailbox_protocol_with_concurrent_producers(golden_dir, monkeypatch, name, n_dev, box):
    """OA_MULTI_OWN_STREAMS=1: children that share the GPU get their own streams, so every k_gather_solve_update really
    spins on its mailbox while the other children's searches and k_reduce_post launches run beside it -- the sequence
    words, the system-scope release / acquire and the two-parity reuse of the slots execute against concurrent producers
    (round 2 only ever ran them in stream order).  Both placements of the mailboxes: peer-mapped device memory (what
    distinct GPUs use: remote writes over xGMI) and the pinned host box (the fallback).  Same bar as the single-device
    loop: K per iteration exact, M to 1e-9, final float32 matrix within 1 ulp."""
    from object_alignment_amd.engine import IcpEngine
    monkeypatch.setenv("OA_MULTI_OWN_STREAMS", "1")
    monkeypatch.setenv("OA_MAILBOX", box)
    monkeypatch.setenv("OA_EXCHANGE_TIMEOUT_S", "20")
    g = _load(golden_dir, name)
    with IcpEngine(devices=[0] * n_dev) as eng:
        info = eng.exchange_info()
        assert info["exchange"].startswith("mailbox") and ("device" in info["exchange"]) == (box == "device")
        assert info["rccl_ranks"] == 0
        res = _run_fixture(g, eng)
        again = _run_fixture(g, eng)                                # a second loop on the same mailboxes: nothing is cleared
    assert res.iters_done == int(g["iters_done"]) and res.converged == bool(g["converged"])
    assert np.array_equal(res.step_K, g["step_K"])
    assert np.abs(res.step_M - g["step_M"]).max() < 1e-9
    assert np.abs(res.matrix_world - g["final_world"]).max() <= F32_ULP
    assert np.array_equal(again.step_M, res.step_M) and np.array_equal(again.matrix_world, res.matrix_world)


def test_own_streams_match_shared_stream_bitwise(golden_dir, monkeypatch):
    """The rank-ordered sum does not depend on who arrives first: concurrent children give the bits of sequential ones."""
    from object_alignment_amd.engine import IcpEngine
    g = _load(golden_dir, "icp_loop_bumpy_converge")
    out = []
    for own in ("0", "1"):
        monkeypatch.setenv("OA_MULTI_OWN_STREAMS", own)
        with IcpEngine(devices=[0] * 5) as eng:
            out.append(_run_fixture(g, eng))
    assert np.array_equal(out[0].step_M, out[1].step_M) and np.array_equal(out[0].matrix_world, out[1].matrix_world)


@pytest.mark.parametrize("box", ["device", "host"])
def test_a_rank_that_never_posts_ends_the_loop_with_an_error(golden_dir, monkeypatch, box):
    """Fault injection (OA_FAULT_SKIP_POST_RANK): rank 1's sums never reach the mailboxes.  Every device's gather kernel
    gives up after OA_EXCHANGE_TIMEOUT_S, the loop ends with OA_E_RCCL -- it does not hang -- and the process can go on
    using the GPU afterwards."""
    import time
    from object_alignment_amd import _capi
    from object_alignment_amd.engine import IcpEngine
    g = _load(golden_dir, "icp_loop_bumpy_converge")
    monkeypatch.setenv("OA_MULTI_OWN_STREAMS", "1")
    monkeypatch.setenv("OA_MAILBOX", box)
    monkeypatch.setenv("OA_EXCHANGE_TIMEOUT_S", "1.5")
    monkeypatch.setenv("OA_FAULT_SKIP_POST_RANK", "1")
    with IcpEngine(devices=[0, 0, 0]) as eng:
        eng.set_target(g["tgt"])
        eng.set_source(g["src"], stride=1)
        eng.set_matrices(g["mx_align"], g["mx_base"])
        t0 = time.perf_counter()
        with pytest.raises(_capi.OaError) as ei:
            eng.run(iters=20, thresh=0.5, early_exit=True)
        dt = time.perf_counter() - t0
        assert ei.value.code == _capi.OA_E_RCCL
        assert 1.0 < dt < 15.0, dt
        with pytest.raises(_capi.OaError):                          # the modal step reports it too
            eng.iterate(thresh=0.5)
    monkeypatch.delenv("OA_FAULT_SKIP_POST_RANK")
    with IcpEngine(devices=[0, 0, 0]) as eng:                       # a healthy context right after
        res = _run_fixture(g, eng)
    assert res.iters_done == int(g["iters_done"]) and np.array_equal(res.step_K, g["step_K"])


def test_one_host_thread_per_child(golden_dir, monkeypatch):
    """OA_MULTI_THREADS=1 on the one-GPU box: every child is driven by its own persistent host thread, as children on
    distinct GPUs are -- uploads, the whole oa_run loop and the modal step.  Two children on their own streams; the
    result is the sequential one, bit for bit."""
    from object_alignment_amd.engine import IcpEngine
    g = _load(golden_dir, "icp_loop_bumpy_converge")
    monkeypatch.setenv("OA_EXCHANGE_TIMEOUT_S", "10")
    out = []
    for threads, own in (("0", "0"), ("1", "1")):
        monkeypatch.setenv("OA_MULTI_THREADS", threads)
        monkeypatch.setenv("OA_MULTI_OWN_STREAMS", own)
        with IcpEngine(devices=[0, 0]) as eng:
            assert eng.exchange_info()["host_threads"] == (2 if threads == "1" else 1)
            res = _run_fixture(g, eng)
            eng.set_matrices(g["mx_align"], g["mx_base"])
            steps = np.stack([eng.iterate(thresh=0.5)[0] for _ in range(4)])
            out.append((res, steps, eng.stat("enqueue_us")))
    assert np.array_equal(out[0][0].step_M, out[1][0].step_M) and np.array_equal(out[0][0].matrix_world, out[1][0].matrix_world)
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][1], out[0][0].step_M[:4])
    assert out[0][2] > 0 and out[1][2] > 0


def test_partition_once_equals_per_device_partition(monkeypatch):
    """oa_set_source on a multi-device context sorts the selection once and deals out ranges (round 2: every device
    sorted the whole selection).  The children end up exactly as the per-device path leaves them: same loop, bit for
    bit, same pairs in the same order, same nearest neighbours -- with an unsorted, duplicated vlist and a stride."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    rng = np.random.default_rng(7)
    v, t = synth.bumpy_icosphere_mesh(5)
    src = (synth.bumpy_icosphere(5) * np.float32(1.02)).astype(np.float32)
    vlist = rng.permutation(len(src))[: len(src) * 2 // 3]
    vlist = np.concatenate([vlist, vlist[:50]])
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.03, -0.02, 0.04]), [0.02, -0.01, 0.015])
    eye = np.identity(4, dtype=np.float32)
    out = []
    for once in ("1", "0"):
        monkeypatch.setenv("OA_PARTITION_ONCE", once)
        with IcpEngine(devices=[0] * 7) as e:
            e.set_target(v)
            e.set_source(src, vlist=vlist, stride=2)
            e.set_matrices(mxa, eye)
            r = e.run(iters=6, thresh=0.5, early_exit=False)
            e.set_matrices(mxa, eye)
            A, B, st = e.make_pairs(0.1, calc_stats=True)
            idx, d2, _ = e.nn_search()
            out.append((r.step_M, r.step_K, r.matrix_world, A, B, np.array(st), idx, d2))
    for a, b in zip(*out):
        assert np.array_equal(a, b)


def test_device_tensors_as_uploads_on_a_multi_device_context():
    """on_device uploads through oa_create_multi: the library stages a device pointer to the device that needs it
    (here all children sit on GPU 0, the pointer is local; on distinct GPUs it is a peer copy)."""
    import torch
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    v = synth.bumpy_icosphere(4)
    src = (v[::2] * np.float32(1.01)).astype(np.float32)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.03, -0.02, 0.04]), [0.02, -0.01, 0.015])
    eye = np.identity(4, dtype=np.float32)
    out = []
    for dev_arrays in (False, True):
        with IcpEngine(devices=[0, 0, 0]) as e:
            e.set_target(torch.from_numpy(v).cuda() if dev_arrays else v)
            e.set_source(torch.from_numpy(src).cuda() if dev_arrays else src)
            e.set_matrices(mxa, eye)
            out.append(e.run(iters=5, thresh=0.5, early_exit=False))
    assert np.array_equal(out[0].step_M, out[1].step_M)


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
def test_calls_between_modal_steps_end_the_sequence(devices):
    """oa_iterate's sequence ends with every call that re-stages the device state (include/oa_icp.h): a search, a
    make_pairs, a new target or source in between.  The next oa_iterate starts a new sequence from the current pose --
    it must neither continue on the one-shot's device state nor on the old target's filter constants."""
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    v = synth.bumpy_icosphere(4)
    v2 = (synth.bumpy_icosphere(4) * np.float32(1.7) + np.float32(0.3)).astype(np.float32)
    src = (v[::2] * np.float32(1.01)).astype(np.float32)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.03, -0.02, 0.04]), [0.02, -0.01, 0.015])
    eye = np.identity(4, dtype=np.float32)

    def fresh(target):
        e = IcpEngine(0) if devices is None else IcpEngine(devices=devices)
        e.set_target(target); e.set_source(src); e.set_matrices(mxa, eye)
        return e

    # reference: three plain steps; then, from the pose they reach, two more as a NEW sequence
    with fresh(v) as e:
        first = [e.iterate(thresh=0.5)[0] for _ in range(3)]
        pose = e.matrix_world()
    with fresh(v) as e:
        e.set_matrices(pose, eye)
        restart = [e.iterate(thresh=0.5) for _ in range(2)]
    with fresh(v2) as e:
        e.set_matrices(pose, eye)
        other = [e.iterate(thresh=0.5)[0] for _ in range(2)]

    for between in ("nn_search", "make_pairs", "reset_seeds", "set_source", "set_target"):
        with fresh(v) as e:
            got = [e.iterate(thresh=0.5)[0] for _ in range(3)]
            assert np.array_equal(np.stack(got), np.stack(first)), between
            if between == "nn_search":
                e.nn_search()
            elif between == "make_pairs":
                e.make_pairs(0.3, calc_stats=True)
            elif between == "reset_seeds":
                e.reset_seeds()
            elif between == "set_source":
                e.set_source(src)
            else:
                e.set_target(v2)
            nxt = [e.iterate(thresh=0.5) for _ in range(2)]
            want = other if between == "set_target" else [m for m, _ in restart]
            assert np.array_equal(np.stack([m for m, _ in nxt]), np.stack(want)), between
            if between != "set_target":
                assert [s["K"] for _, s in nxt] == [s["K"] for _, s in restart], between


def test_history_of_a_modal_sequence_is_its_last_steps():
    """oa_get_history after oa_iterate: the LAST max_n iterations, oldest first -- also before the 64-entry ring wraps."""
    import ctypes as C
    from object_alignment_amd import synth, _capi as capi
    from object_alignment_amd.engine import IcpEngine
    v = synth.bumpy_icosphere(4)
    src = (v[::2] * np.float32(1.01)).astype(np.float32)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.03, -0.02, 0.04]), [0.02, -0.01, 0.015])
    for devs in (None, [0, 0]):
        with (IcpEngine(0) if devs is None else IcpEngine(devices=devs)) as e:
            e.set_target(v); e.set_source(src); e.set_matrices(mxa, np.identity(4, dtype=np.float32))
            steps = [e.iterate(thresh=0.5)[0] for _ in range(7)]
            sM = np.zeros((3, 4, 4))
            got = e._L.oa_get_history(e._h, 3, capi.dptr(sM), None, None, None, None)
            assert got == 3 and np.array_equal(sM, np.stack(steps[-3:]))


def test_bench_two_shards_reports_its_exchange(monkeypatch):
    """`python bench.py --gpus 2` on the one-GPU box with the shards on their own streams: the line names the exchange
    that ran, how many ranks RCCL saw (none here: the device is listed twice, so AUTO takes the mailbox), and the measured
    host enqueue time per iteration."""
    env = dict(os.environ, OA_BENCH_SAME_DEVICE="1", OA_MULTI_OWN_STREAMS="1")
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--n-source", "120000",
           "--n-target", "100000", "--no-cpu-baseline", "--no-surface"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    cfg = d["config"]
    assert cfg["exchange"].startswith("mailbox") and cfg["rccl_ranks"] == 0
    assert cfg["host_enqueue_us_per_iteration"] > 0
    # what makes a first run on real multi-GPU hardware explain itself
    mg = d["multi_gpu"]
    assert 0 < mg["search_ms_per_device"]["min"] <= mg["search_ms_per_device"]["max"]
    assert mg["exchange_us_per_iteration"] > 0                        # GPU-side stamps: own sums ready -> the world's sums in hand
    assert mg["host_enqueue_us_per_iteration"] > 0
    assert mg["exchange"].startswith("mailbox") and mg["rccl_ranks"] == 0 and mg["rccl_fallbacks"] == 0
    assert "more than once" in mg["exchange_note"]                     # why AUTO did not take RCCL here
    assert mg["predicted_ms_per_step_design_4_7"] is None             # (predictions exist for 1 and 8 GPUs at full size)
    # ... and the headline's own diagnostics
    r = d["roofline"]
    assert r["launch_ms"]["n"] == 3 and r["launch_ms"]["min"] <= r["launch_ms"]["median"] <= r["launch_ms"]["max"]
    c = r["measured_issue_ceiling"]
    assert 20.0 < c["used"] < 90.0 and 800.0 < c["after_timed_loop"]["shader_clock_mhz"] < 3000.0, c
    assert 0.0 < r["frac_of_measured_ceiling"] < 1.2
    assert 10.0 < c["after_timed_loop"]["tlaneops_min3"] < c["after_timed_loop"]["tlaneops"]     # the half-rate class, measured
    assert r["kernel"] == "k_nn_search_sorted" and 0.2 < r["half_rate_instruction_share"] < 0.5
    # round 6: the line says what frac is, gives the pairs per second and SURVEY 8d's formula with a flag when it is void; the two
    # mix-ceiling fields of round 5 (a fraction of a ceiling that read above 1) are gone
    assert "issue-slot utilisation" in r["what_frac_is"] and r["pairs_per_s"] > 1e11
    a8 = r["frac_8d_algorithmic"]
    assert a8["value"] > 0.0 and a8["void"] == (a8["value"] > 1.0) and a8["effective_tflops"] > 0.0
    assert "frac_of_measured_mix_ceiling" not in r and "mix_ceiling_at_the_search_clock" not in r
    assert 800.0 < r["shader_clock_mhz_during_search"] < 3000.0


# ---------------------------------------------------------------------------------------------------------------------
# round 4: the host threads agree on how many iterations they enqueue; the wait for the streams is bounded in RCCL mode
# ---------------------------------------------------------------------------------------------------------------------
def _small_job(eng, g):
    eng.set_target(g["tgt"])
    eng.set_source(g["src"], stride=1)


def _agree_stress(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "agree_stress.py")] + [str(a) for a in args], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-3000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("n_dev", [2, 4])
def test_host_threads_agree_on_the_enqueued_count(n_dev):
    """One persistent host thread per child (OA_MULTI_THREADS=1 + OA_MULTI_OWN_STREAMS=1: what distinct GPUs get), early
    exit on, a 2562-point mesh -- an iteration takes ~20 us, so host and device race for real.  Every thread stops on what
    ITS device reports, at its own time; the number of iterations each enqueues must still be the same on all of them,
    because in RCCL mode every enqueued iteration holds a collective (round 3: each thread for itself).
    250 loops; each must also be the fixture's loop.  (tools/agree_stress.py, its own process: every child's stream needs
    a hardware queue of its own on the one-GPU box.)"""
    d = _agree_stress("--children", n_dev, "--reps", 250)
    assert d["host_threads"] == n_dev and d["reps"] == 250
    assert d["loops_with_differing_counts"] == 0 and d["loops_with_wrong_result"] == 0, d
    for key in d["enqueued_counts_histogram"]:
        assert d["iterations_executed"] <= int(key) <= 30, d
    print(d)


def test_a_lagging_host_thread_hits_the_window_and_the_agreement_closes_it():
    """OA_FAULT_LAG_GROUP: host thread 1 sleeps 300 us before every look at its halt flag.  Its device then halts before it
    has enqueued anything ahead, while thread 0 is `lag` iterations ahead of the GPU -- the window of VERDICT r03 item 1,
    hit on every loop.  With the agreement switched off (OA_MULTI_AGREE=0, honoured on the mailbox only -- round 3's code)
    the two threads enqueue different numbers of iterations; with it, the same number.  The results are the fixture's
    either way (on the mailbox a surplus iteration is an empty launch; with RCCL it would be a collective nobody joins)."""
    old = _agree_stress("--children", 2, "--reps", 20, "--lag-group", 1, "--lag-us", 300, "--agree", 0)
    new = _agree_stress("--children", 2, "--reps", 20, "--lag-group", 1, "--lag-us", 300, "--agree", 1)
    assert old["loops_with_wrong_result"] == 0 and new["loops_with_wrong_result"] == 0
    assert old["loops_with_differing_counts"] >= 10, old   # the window is real (and the test would notice if the hook stopped hitting it)
    assert new["loops_with_differing_counts"] == 0, new


def test_children_build_their_safe_radii_without_allocating_inside_a_loop(golden_dir, monkeypatch):
    """OA_GRID_SAFE=1 builds the radii lazily, once a target has served 8 loop iterations.  Round 4 allocated the array at that
    moment -- and inside a multi-device group's loop an allocation can wait for a sibling's stream whose gather kernel waits for
    the post this very thread has not enqueued yet (the lagging-thread test above ended with OA_E_RCCL when it did), so children
    only built between loops.  Round 5: the array comes with the grid and the build is ONE launch on the child's own stream, so a
    child builds where a context of its own does -- between two iterations of the running loop, here with every child on its own
    stream and host thread -- and every loop before, across and after the build gives the fixture's matrix."""
    from object_alignment_amd.engine import IcpEngine
    g = _load(golden_dir, "icp_loop_bumpy_converge")
    monkeypatch.setenv("OA_MULTI_THREADS", "1")
    monkeypatch.setenv("OA_MULTI_OWN_STREAMS", "1")
    monkeypatch.setenv("OA_GRID_SAFE", "1")
    with IcpEngine(devices=[0, 0]) as eng:
        eng.set_search_mode("grid")
        _small_job(eng, g)
        assert int(eng.stat("safe_radii")) == 0                      # not with the grid: lazily
        seen = []
        for _ in range(4):
            eng.set_matrices(g["mx_align"], g["mx_base"])
            res = eng.run(iters=30, thresh=0.5, target_d=0.01, use_target=True, early_exit=True)
            seen.append(int(eng.stat("safe_radii")))
            assert res.iters_done == int(g["iters_done"]) and np.abs(res.matrix_world - g["final_world"]).max() <= 2.5e-7
        # a loop enqueues 7-9 iterations per child: built inside loop 1 or 2, and it stays built
        assert seen[1] == 1 and seen[-1] == 1 and seen == sorted(seen), seen


def test_rccl_watchdog_turns_a_stalled_collective_into_an_error(golden_dir, monkeypatch):
    """include/oa_icp.h promises OA_E_RCCL when a device's sums do not arrive within OA_EXCHANGE_TIMEOUT_S -- in RCCL mode
    too.  OA_FAULT_STALL_RANK stops rank 0's stream ahead of the collective of iteration 2 (a bounded spin: what a rank
    whose peers never enter the collective looks like from the host).  The watchdog sees no device finish an iteration for
    1.5 s, aborts the communicators (ncclCommAbort), the streams drain, oa_run fails with OA_E_RCCL -- no hang -- and the
    same context, asked for RCCL again, builds new communicators and runs the fixture's loop."""
    import time
    from object_alignment_amd import _capi
    from object_alignment_amd.engine import IcpEngine
    g = _load(golden_dir, "icp_loop_bumpy_converge")
    monkeypatch.setenv("OA_EXCHANGE_TIMEOUT_S", "1.5")
    monkeypatch.setenv("OA_FAULT_STALL_RANK", "0")
    with IcpEngine(devices=[0], exchange="rccl") as eng:
        assert eng.exchange_info()["rccl_ranks"] == 1
        _small_job(eng, g)
        eng.set_matrices(g["mx_align"], g["mx_base"])
        t0 = time.perf_counter()
        with pytest.raises(_capi.OaError) as ei:
            eng.run(iters=30, thresh=0.5, target_d=0.01, use_target=True, early_exit=True)
        dt = time.perf_counter() - t0
        assert ei.value.code == _capi.OA_E_RCCL and "abort" in ei.value.msg
        assert 1.0 < dt < 12.0, dt
        assert eng.stat("watchdog_aborts") == 1
        eng.set_exchange("rccl")                                     # new communicators (handshake included)
        assert eng.exchange_info()["rccl_ranks"] == 1
        res = _run_fixture(g, eng)
        assert eng.stat("watchdog_aborts") == 1
    assert res.iters_done == int(g["iters_done"]) and np.abs(res.matrix_world - g["final_world"]).max() <= F32_ULP


def test_auto_finishes_through_the_mailboxes_when_rccl_lets_the_loop_down(golden_dir, monkeypatch):
    """AUTO resolves to RCCL on distinct devices, and that path has never run on distinct GPUs.  When a loop AUTO
    started on RCCL ends in the watchdog, the library aborts the communicators and runs the SAME loop again through the
    mailboxes: the caller gets the fixture's alignment, not OA_E_RCCL (an explicit request for RCCL still gets the error:
    test_rccl_watchdog_turns_a_stalled_collective_into_an_error).  One GPU: OA_AUTO_RCCL_ANY lets AUTO take RCCL for a world of
    one, OA_FAULT_STALL_RANK stalls its first loop's collective."""
    import time
    from object_alignment_amd.engine import IcpEngine
    g = _load(golden_dir, "icp_loop_bumpy_converge")
    monkeypatch.setenv("OA_EXCHANGE_TIMEOUT_S", "1.5")
    monkeypatch.setenv("OA_FAULT_STALL_RANK", "0")
    monkeypatch.setenv("OA_AUTO_RCCL_ANY", "1")
    with IcpEngine(devices=[0]) as eng:                                  # exchange: AUTO
        t0 = time.perf_counter()
        res = _run_fixture(g, eng)
        dt = time.perf_counter() - t0
        assert 1.0 < dt < 15.0, dt
        assert eng.stat("watchdog_aborts") == 1
        info = eng.exchange_info()
        assert info["rccl_ranks"] == 0 and "mailbox" in info["exchange"], info
        res2 = _run_fixture(g, eng)                                      # and stays there: no second abort
        assert eng.stat("watchdog_aborts") == 1
    for r in (res, res2):
        assert r.iters_done == int(g["iters_done"]) and np.abs(r.matrix_world - g["final_world"]).max() <= F32_ULP


@pytest.mark.parametrize("exchange", ["mailbox", "rccl"])
def test_a_failing_host_thread_ends_the_loop_everywhere(golden_dir, monkeypatch, exchange):
    """OA_FAULT_FAIL_GROUP: one host thread's enqueue fails at iteration 3.  The other threads stop committing to new
    iterations at once (they do not enqueue on for seconds), the streams are waited for with a bound -- mailbox: the gather
    kernels' own time limit; RCCL: the communicators are aborted right away, the ranks' collective counts may differ -- the
    call returns the thread's error, and the context runs the next loop."""
    import time
    from object_alignment_amd import _capi
    from object_alignment_amd.engine import IcpEngine
    g = _load(golden_dir, "icp_loop_bumpy_converge")
    devices = [0] if exchange == "rccl" else [0, 0, 0]
    monkeypatch.setenv("OA_MULTI_THREADS", "1")
    monkeypatch.setenv("OA_MULTI_OWN_STREAMS", "1")
    monkeypatch.setenv("OA_EXCHANGE_TIMEOUT_S", "1.5")
    monkeypatch.setenv("OA_FAULT_FAIL_GROUP", "0" if exchange == "rccl" else "1")
    monkeypatch.setenv("OA_FAULT_FAIL_ITER", "3")
    with IcpEngine(devices=devices, exchange=exchange) as eng:
        _small_job(eng, g)
        eng.set_matrices(g["mx_align"], g["mx_base"])
        t0 = time.perf_counter()
        with pytest.raises(_capi.OaError) as ei:
            eng.run(iters=30, thresh=0.5, target_d=0.01, use_target=True, early_exit=True)
        dt = time.perf_counter() - t0
        assert ei.value.code == _capi.OA_E_HIP and "injected" in ei.value.msg
        assert dt < 10.0, dt
        assert eng.stat("watchdog_aborts") == (1 if exchange == "rccl" else 0)
        if exchange == "rccl":
            eng.set_exchange("rccl")
        res = _run_fixture(g, eng)
    assert res.iters_done == int(g["iters_done"]) and np.abs(res.matrix_world - g["final_world"]).max() <= F32_ULP

"""Two ranks (one process each) driving the sharded bench path on ONE GPU over gloo: exercises bench.py's
torch.distributed flow, oa_set_source sharding and the per-iteration all-reduce end to end with the real kernels.
(RCCL itself needs >= 2 GPUs and is exercised by the driver's scaling run.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env):
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-3000:]                 # rank 0 prints exactly one JSON line
    return json.loads(lines[0])


def test_bench_two_ranks_gloo_same_gpu():
    env = dict(os.environ)
    common = ["--steps", "4", "--warmup", "1", "--n-source", "120000", "--n-target", "100000", "--no-cpu-baseline"]
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + common, env)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env2 = dict(env, OA_BENCH_BACKEND="gloo", OA_BENCH_SAME_DEVICE="1")
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2"] + common, env2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    for d in (one, two):
        assert d["metric"].startswith("ICP iterations/sec") and d["unit"] == "iterations/s" and d["steps"] == 4
        assert d["roofline"]["frac"] > 0 and d["grid_path"]["final_matrix_bitwise_equal_to_brute_force"] is True
    # one process per GPU: every rank's search time and its GPU-side wait for the all-reduced sums are on rank 0's line
    mg = two["multi_gpu"]
    assert one["multi_gpu"] is None
    assert 0 < mg["search_ms_per_device"]["min"] <= mg["search_ms_per_device"]["max"]
    assert mg["exchange_us_per_iteration"] > 0 and "torch.distributed" in mg["exchange"]
    # same job, different sharding: the per-iteration sums differ only by fp64 summation order
    assert one["result"]["last_K"] == two["result"]["last_K"]
    assert abs(one["result"]["final_translation"] - two["result"]["final_translation"]) < 1e-12
    assert abs(one["result"]["mean_dist"] - two["result"]["mean_dist"]) < 1e-12


def test_c_program_drives_the_library_on_the_gpu(tmp_path):
    """tests/c/abi_smoke.c: a pure C consumer of include/oa_icp.h aligns a small cloud on the GPU."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_cpu import _build_abi_smoke
    exe = _build_abi_smoke(tmp_path)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout
    assert "ABI_SMOKE_OK device" in p.stdout, p.stdout

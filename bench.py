#!/usr/bin/env python3
"""bench.py -- ICP iterations/s on BASELINE.json's 1M <-> 1M workload, one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one ICP iteration over the whole workload: nearest-vertex search for every source point against the
whole target cloud, the fused threshold/accumulate pass, (N > 1: one all-reduce of 24 doubles over RCCL) and the
Kabsch solve + matrix_world update.  Inputs are resident in HBM before the timed region; the early-exit of the
convergence test is disabled so exactly K full iterations execute.

N = 1: BASELINE config "1M <-> 1M random point clouds with 5% Gaussian noise, 50 iters, 1xMI355X".
N > 1: BASELINE config "1M <-> 1M, source sharded across N GPUs with RCCL covariance all-reduce" (strong scaling:
       the whole-job work per iteration is fixed, each rank holds 1/N of the source and the whole target).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (k_nn_search): it is fp32-VALU bound
(SURVEY.md 8d / DESIGN.md), so the primary roofline is 8 flop per (source, target) pair against the 157.3 TFLOP/s
fp32 vector peak (numerically also the dense fp32 MFMA peak); the HBM view the north-star asks for is reported
next to it in `roofline_hbm`.  `cpu_baseline` times the CPU oracle (KD-tree + Kabsch; OpenMP on all host cores) on
a bounded sample of the same workload, rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FP32_VECTOR_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: peak FP32 (vector) == peak FP32 (matrix, f32-in MFMA)
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FLOP_PER_PAIR = 8                    # 3 sub, 3 mul, 2 add (difference-form squared distance), SURVEY.md 8d


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-source", type=int, default=1_000_000)
    ap.add_argument("--n-target", type=int, default=1_000_000)
    ap.add_argument("--cpu-iters", type=int, default=3, help="iterations of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-surface", action="store_true", help="skip the surface-mode leg (1M points vs a 2M-triangle mesh)")
    return ap.parse_args()


def cpu_baseline(src, tgt, mxa, mxb, iters, gpu_step_M):
    """The oracle's loop (KD-tree nearest vertex + Kabsch) on the host cores; also a live parity check."""
    from oracle import oracle as orc
    orc.build()
    t0 = time.perf_counter()
    kd = orc.KDTree(tgt)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref = orc.icp_run(src, tgt, mxa, mxb, iters=iters, sample=1, thresh=0.5, target_d=1e-300, use_target=True, kd=kd)
    t_loop = time.perf_counter() - t0
    n = min(iters, len(gpu_step_M))
    dM = float(np.abs(ref["step_M"][:n] - gpu_step_M[:n]).max()) if n else float("nan")
    return {
        "value": iters / t_loop, "unit": "iterations/s", "cores": orc.max_threads(), "kind": "port",
        "sample": "first %d of the ICP iterations of the same %d<->%d workload; KD-tree nearest vertex (OpenMP) + "
                  "Kabsch; tree build %.2f s excluded like the GPU's one-time upload" % (iters, len(src), len(tgt), t_build),
        "ms_per_iteration": 1e3 * t_loop / iters,
        "parity_max_abs_dM_vs_gpu": dM,
    }


def cpu_tiers(src, tgt, mxa, mxb):
    """T1 = the reference's cost structure (per-vertex interpreter loop), T3 = like-for-like
    brute force in C/OpenMP.  Both on bounded samples; the `cpu_baseline` object above is tier T2."""
    from oracle import oracle as orc
    out = {}
    kd = orc.KDTree(tgt)
    n1 = 4000
    t0 = time.perf_counter()
    A, B, _ = orc.make_pairs_python_loop(src[:n1], tgt, mxa, mxb, 0.5, kd, calc_stats=True)
    t1 = time.perf_counter() - t0
    A2, B2, _ = orc.make_pairs(src[:n1], tgt, mxa, mxb, 0.5, calc_stats=True, kd=kd)
    out["T1_reference_style_python_loop"] = {
        "us_per_vertex": 1e6 * t1 / n1, "cores": 1, "sample": "%d source vertices of the workload" % n1,
        "extrapolated_s_per_iteration": t1 / n1 * len(src), "same_pairs_as_vectorised_oracle": bool(np.array_equal(A, A2) and np.array_equal(B, B2)),
        "note": "interpreter loop + float32 4x4 transforms on Python objects + one tree query per vertex, as "
                "functions/general.py:280-321 does (its per-vertex print omitted)"}
    n3 = min(len(src), 20000)
    t0 = time.perf_counter()
    orc.nn_brute(src[:n3], tgt)
    t3 = time.perf_counter() - t0
    out["T3_bruteforce_c_openmp"] = {
        "gpairs_per_s": n3 * len(tgt) / t3 / 1e9, "cores": orc.max_threads(),
        "sample": "%d x %d pairs, same fp32 metric" % (n3, len(tgt)),
        "extrapolated_s_per_iteration": t3 / n3 * len(src)}
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:                          # before the HIP runtime comes up: RCCL needs dmabuf IPC on this host driver
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the oa_icp engine has no CPU fallback")
    # test hooks (not used by the driver): run N ranks on ONE GPU over gloo to exercise the sharded path end to end
    backend = os.environ.get("OA_BENCH_BACKEND", "nccl")
    if os.environ.get("OA_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from object_alignment_amd import synth
    from object_alignment_amd.distributed import EngineShard, new_sums_tensor, run_sharded
    from object_alignment_amd.engine import IcpEngine

    src, tgt, mxa, mxb = synth.c3_random_pair(args.n_source, seed=1234, n_target=args.n_target)
    eng = IcpEngine(local_rank)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_search_mode("brute")          # the north-star kernel: LDS-tiled brute force (grid path measured below)
    t0 = time.perf_counter()
    eng.set_target(tgt)
    eng.set_source(src, stride=1, shard_index=rank, shard_count=world)
    torch.cuda.synchronize()
    upload_s = time.perf_counter() - t0
    sums = new_sums_tensor(dev)
    kw = dict(thresh=0.5, target_d=0.01, use_target=True, with_scale=False, early_exit=False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(steps, warmup):
        """W untimed + exactly `steps` timed iterations from the initial pose; max over ranks."""
        if warmup > 0:
            eng.set_matrices(mxa, mxb)
            run_sharded(EngineShard(eng, iters=warmup, **kw), warmup, sums, world_size=world)
        eng.set_matrices(mxa, mxb)
        barrier()
        t0 = time.perf_counter()
        r = run_sharded(EngineShard(eng, iters=steps, **kw), steps, sums, world_size=world)
        barrier()
        dt = time.perf_counter() - t0
        ms = r.nn_ms_total / max(1, steps)
        if world > 1:
            t = torch.tensor([dt, ms], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, ms = float(t[0]), float(t[1])
        return r, dt, ms

    res, elapsed, nn_ms = timed(args.steps, args.warmup)

    # SURVEY 8f rank 2 ("next" row, reported beside the headline, never instead of it): the same run with the
    # uniform-grid exact search.  Correspondences are identical, so the final matrix must be bitwise the same.
    grid = None
    try:
        eng.set_search_mode("grid")
        g_steps = max(args.steps, 200)
        gres, g_elapsed, g_nn_ms = timed(g_steps, max(args.warmup, 5))
        gcheck, _, _ = timed(args.steps, 0)
        grid = (g_steps, g_elapsed, g_nn_ms, bool(np.array_equal(gcheck.matrix_world, res.matrix_world)))
    except Exception as exc:                                  # never lose the headline line
        grid = ("error: %r" % (exc,),)

    # SURVEY 8f rank 1 ("next" row, reported beside the headline): surface mode -- closest point on the base mesh's
    # triangles, what the reference's BVHTree.find_nearest returns -- on a 1M-point cloud against a ~1M-vertex /
    # ~2M-triangle mesh of the synthetic bunny surface, library-default search (grid + tree).  N = 1 only.
    surf = None
    if world == 1 and not args.no_surface:
        try:
            sv, st = synth.lattice_surface_mesh(700, 1400)
            ssrc = synth.bunny_surface(args.n_source, offset=0.37)
            s_mxa = synth.rigid4(synth.rotation_from_rotvec([0.02, -0.015, 0.025]), [0.01, -0.008, 0.012])
            eye4 = np.identity(4, dtype=np.float32)
            with IcpEngine(local_rank) as se:
                se.set_target_mesh(sv, st)
                se.set_source(ssrc, stride=1)
                se.set_matrices(s_mxa, eye4)
                se.run(iters=3, thresh=0.05, early_exit=False)             # warm-up (module load, first launches)
                se.set_matrices(s_mxa, eye4)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sres = se.run(iters=30, thresh=0.05, early_exit=False)
                s_dt = time.perf_counter() - t0
            surf = {"what": "SURVEY 8f rank 1 (next row): surface mode, closest point on triangles (k_tri_search_grid + "
                            "k_bvh_search), bit-identical to the oracle's brute force over all triangles (tests)",
                    "n_source": int(len(ssrc)), "n_target_vertices": int(len(sv)), "n_target_triangles": int(len(st)),
                    "steps": 30, "value": 30 / s_dt, "unit": "iterations/s", "ms_per_step": 1e3 * s_dt / 30,
                    "ms_per_nn_search": sres.nn_ms_total / 30, "last_K": sres.last_K, "mean_dist": sres.mean_dist}
        except Exception as exc:                                  # never lose the headline line
            surf = {"error": repr(exc)}

    if rank == 0:
        assert res.iters_done == args.steps, (res.iters_done, args.steps)
        ns_local = eng.n_selected
        pairs = float(ns_local) * float(args.n_target)                      # per launch of k_nn_search on one GPU
        achieved_tflops = FLOP_PER_PAIR * pairs / (nn_ms * 1e-3) / 1e12
        algo_bytes = 16.0 * ns_local + 12.0 * args.n_target + 8.0 * ns_local  # source float4 + target SoA + keys
        out = {
            "metric": "ICP iterations/sec + ms/NN-search, 1M<->1M verts",
            "value": args.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_nn_search": nn_ms,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32", "accumulate_dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "1M<->1M uniform [-1,1]^3 clouds, sigma = 5%% of mean spacing, seed 1234, "
                                   "thresh 0.5, stride 1, %d iterations, early-exit off" % args.steps,
                       "n_source": args.n_source, "n_target": args.n_target,
                       "parallelism": "source sharded x%d, target replicated, all-reduce of 24 f64 per iteration" % world
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "valu", "achieved": achieved_tflops, "peak": FP32_VECTOR_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved_tflops / FP32_VECTOR_PEAK_TFLOPS, "traffic": None,
                         "kernel": "k_nn_search", "flop_per_pair": FLOP_PER_PAIR, "pairs_per_launch": pairs,
                         "avg_launch_ms": nn_ms,
                         "note": "fp32 vector-ALU bound brute-force search; 157.3 TFLOP/s is both the fp32 VALU peak "
                                 "and the dense f32-input MFMA peak.  `achieved` credits the ALGORITHMIC 8 flop per "
                                 "(source, target) pair (SURVEY 8d); the kernel's conservative two-level filter proves "
                                 "most pairs losers with 2 fma + 1 min, so it issues ~3 VALU instructions per pair "
                                 "(PMC: profiles/) and runs at the chip's measured v_fma issue rate -- a fraction near "
                                 "1.0 means the issue limit is reached, not that 8 flops per pair were executed"},
            "roofline_hbm": {"bound": "hbm", "achieved": algo_bytes / (nn_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": algo_bytes / (nn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": algo_bytes, "traffic": None},
            "result": {"final_translation": res.last_translation, "last_K": res.last_K,
                       "mean_dist": res.mean_dist, "std_dist": res.std_dist},
            "upload_ms": 1e3 * upload_s,
            "loop_ms_hipevents": res.loop_ms,
        }
        if grid is not None and len(grid) == 4:
            g_steps, g_elapsed, g_nn_ms, same = grid
            g_bytes = 40.0 * ns_local + 16.0 * args.n_target        # source float4 + winner record + key, sorted target image
            out["grid_path"] = {
                "what": "SURVEY 8f rank 2 (next row): k_nn_search_grid, exact uniform-grid search, same correspondences",
                "value": g_steps / g_elapsed, "unit": "iterations/s", "steps": g_steps,
                "ms_per_step": 1e3 * g_elapsed / g_steps, "ms_per_nn_search": g_nn_ms,
                "final_matrix_bitwise_equal_to_brute_force": same,
                "roofline": {"bound": "hbm", "achieved": g_bytes / (g_nn_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": g_bytes / (g_nn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": g_bytes, "traffic": None,
                             "note": "latency-bound dependent lookups (cell range -> vertices); not a streaming kernel"},
            }
        elif grid is not None:
            out["grid_path"] = {"error": grid[0]}
        if surf is not None:
            out["surface_path"] = surf
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(prof):
            try:
                tr = json.load(open(prof))
                key = "%dx%d_n%d" % (args.n_source, args.n_target, world)
                if key in tr:
                    out["roofline"]["traffic"] = tr[key]["bytes_per_launch"]
                    out["roofline_hbm"]["traffic"] = tr[key]["bytes_per_launch"]
                    if "valu_instructions_per_pair" in tr[key]:
                        out["roofline"]["valu_instructions_per_pair_pmc"] = tr[key]["valu_instructions_per_pair"]
                gkey = "grid_" + key
                if gkey in tr and "roofline" in out.get("grid_path", {}):
                    out["grid_path"]["roofline"]["traffic"] = tr[gkey]["bytes_per_launch"]
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline and args.cpu_iters > 0:
            out["cpu_baseline"] = cpu_baseline(src, tgt, mxa, mxb, min(args.cpu_iters, args.steps), res.step_M)
            try:
                out["cpu_baseline_tiers"] = cpu_tiers(src, tgt, mxa, mxb)
            except Exception as exc:
                out["cpu_baseline_tiers"] = {"error": repr(exc)}
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

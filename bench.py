#!/usr/bin/env python3
"""bench.py -- ICP iterations/s on BASELINE.json's 1M <-> 1M workload.

  python bench.py [--gpus N] [--steps K] [--warmup W]                 # N >= 1: ONE process drives all N GPUs
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W                       # one process per GPU over RCCL

A "step" is one ICP iteration over the whole workload: nearest-vertex search for every source point against the
whole target cloud, the fused threshold/accumulate pass, the exchange of 24 doubles between the GPUs (N > 1) and the
Kabsch solve + matrix_world update.  Inputs are resident in HBM before the timed region; the early exit of the
convergence test is disabled so exactly K full iterations execute; the timed run starts COLD (initial pose, no
correspondence seeds from the warm-up: `oa_reset_seeds`), as a real K-iteration alignment does.

N = 1: BASELINE config "1M <-> 1M random point clouds with 5% Gaussian noise, 50 iters, 1xMI355X".
N > 1: BASELINE config "1M <-> 1M, source sharded across N GPUs with ... covariance all-reduce" (strong scaling: the
       whole-job work per iteration is fixed, each GPU holds 1/N of the source and the whole target).
       Launched plainly, one process holds a multi-device context (oa_create_multi): the exchange lives inside
       liboa_icp.so (mailbox all-gather; OA_EXCHANGE=rccl: ncclAllReduce).  Launched under torch.distributed.run, every
       rank holds one context and the 24 sums go through torch.distributed's all_reduce (backend "nccl" = RCCL).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel of the headline, the brute-force search
k_nn_search_sorted.  It is bound by fp32 vector-ALU issue (SURVEY.md 8d / DESIGN.md 3.1), so

    roofline.achieved = executed VALU lane-ops/s = (SQ_INSTS_VALU per launch x 64 lanes) / average launch time
    roofline.peak     = 78.6e12 lane-ops/s       = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz
    roofline.frac     = achieved / peak          (<= 1 by construction): VALU ISSUE-SLOT UTILISATION -- how full the pipe is,
                                                  not how good the kernel is (`what_frac_is`)

with the instruction count from the committed PMC pass (profiles/hbm_traffic.json, stamped with the kernel name and
the commit it was collected at) and the launch time measured live with hipEvents on the kernel's stream.  The chip does
not hold 2.4 GHz under every load, so the line also carries every launch's time (`launch_ms`: min / median / max), what
v_add_f32 (full rate) and v_min3_f32 (half rate) issue on THIS box right before and right after the timed loop
(`measured_issue_ceiling`, with the shader clock under that load), `frac_of_measured_ceiling` (instructions counted alike),
`shader_clock_mhz_during_search`, `pairs_per_s`, and SURVEY 8d's algorithmic formula as written -- 8 flop per (source, target)
pair / time / 157.3 TFLOP/s -- as `frac_8d_algorithmic` {value, void}: the kernel's conservative filter proves most pairs losers
in 1.5 instructions, so that model counts arithmetic nobody executes; above 1 it is flagged void and is NOT a roofline fraction.
`cpu_baseline` times the CPU oracle (KD-tree + Kabsch; OpenMP on all host cores) on a
bounded sample of the same workload, rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

VALU_PEAK_TLANEOPS = 78.6            # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (MI355X_MICROARCH.md), T lane-ops/s
FP32_VECTOR_PEAK_TFLOPS = 157.3      # the same peak counting an FMA as 2 flop
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense f16 / bf16 MFMA (the experiment leg only)
FLOP_PER_PAIR = 8                    # 3 sub, 3 mul, 2 add (difference-form squared distance), SURVEY.md 8d
# hot loops, instructions per pair (the fallback when profiles/ holds no PMC pass for the kernel that ran):
#   k_nn_search_sorted   (256 sub + 127 min3 + min + cmp) / 256 pairs = 1.5, + the blocks going on to level 1
#   k_nn_search_filtered (64 fma + 16 min3 + 4 cmp) / 32 pairs
VALU_PER_PAIR_ISA = {"k_nn_search_sorted": 1.7, "k_nn_search_filtered": 2.7}
# share of half-rate instructions (v_min_f32 / v_min3_f32 / v_cmp_*_f32, tools/valu_rates.hip) in the hot loop, from the ISA:
#   k_nn_search_sorted   129 of 385 per point and block of 256 vertices (256 v_sub | 127 v_min3 + v_min + v_cmp)
#   k_nn_search_filtered 20 of 84 per 32 pairs (64 v_fma | 16 v_min3 + 4 v_cmp)
HALF_RATE_SHARE = {"k_nn_search_sorted": 129.0 / 385.0, "k_nn_search_filtered": 20.0 / 84.0}
BRUTE_KERNELS = {3.0: "k_nn_search_sorted", 1.0: "k_nn_search_filtered", 2.0: "k_nn_search_mfma", 0.0: "k_nn_search"}
KERNELS = {"brute": "k_nn_search_sorted", "grid": "k_nn_search_grid", "surface_grid": "k_tri_search_grid",
           "surface_tree": "k_bvh_search"}


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
    ap.add_argument("--no-grid", action="store_true", help="skip the grid-search leg")
    ap.add_argument("--no-mfma", action="store_true", help="skip the OA_NN_MFMA=1 experiment leg")
    ap.add_argument("--no-whole-call", action="store_true", help="skip the whole-call leg (operator calls from host arrays, upload included)")
    ap.add_argument("--no-c5", action="store_true", help="skip the BASELINE config 5 leg of a multi-GPU run")
    ap.add_argument("--c5", action="store_true", help="run the config 5 leg at N = 1 too")
    ap.add_argument("--c5-source", type=int, default=10_000_000)
    ap.add_argument("--c5-target", type=int, default=2_000_000)
    ap.add_argument("--c5-steps", type=int, default=20)
    return ap.parse_args()


def csrc_sha16():
    """Fingerprint of the kernel sources: a PMC figure collected from other sources is marked stale."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "object_alignment_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_entry(key, kernel):
    """profiles/hbm_traffic.json[key] if it was collected for `kernel`; (entry or None, stamp dict)."""
    prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        e = json.load(open(prof)).get(key)
    except Exception:
        e = None
    if not e:
        return None, {"source": None, "note": "no PMC figure committed for this configuration"}
    stamp = {"source": e.get("source"), "kernel": e.get("kernel"), "commit": e.get("commit"),
             "csrc_sha16": e.get("csrc_sha16"), "stale": e.get("csrc_sha16") != csrc_sha16()}
    if not str(e.get("kernel", "")).startswith(kernel):
        stamp["note"] = "the committed figure is for another kernel (%s): not reported" % e.get("kernel")
        return None, stamp
    return e, stamp


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
    tm = {}
    A, B, _ = orc.make_pairs_python_loop(src[:n1], tgt, mxa, mxb, 0.5, kd, calc_stats=True, timing=tm)
    t1 = tm["loop_s"]              # the function body only: the duck-typed objects it is handed are built outside the clock
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


def whole_call_leg(local_rank, src, tgt, mxa, mxb):
    """Whole operator calls from HOST arrays -- upload over PCIe + index builds + a 50-iteration run with the reference's early
    exit (operators/icp_align.py:96-151) -- on a context that is kept between calls: what the add-on's user waits for.  Never
    `value` (the metric excludes the one-time upload); reported so that DESIGN.md's whole-call figures have a driver-run twin.
    Best of 5 calls each; the 1M <-> 1M pair is the bench workload, the 100k pair BASELINE config 2; the two `surface_*` cases are
    the call the reference's operator really makes -- a base MESH with faces, closest point on its triangles (set_target_mesh)."""
    import time
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    out = {"what": "host arrays -> aligned matrix: set_target + set_source + run(iters=50, early exit); PCIe-inclusive, "
                   "AUTO search; best (`ms`, with its parts), median and slowest of 5 calls on a warm context; the copies' own durations: "
                   "profiles/r05*_whole_call_traces.txt (rocprofv3 --memory-copy-trace)", "unit": "ms"}
    cases = {"c3_1M_1M": (src, tgt, mxa, mxb)}
    s2, t2, a2, b2 = synth.c2_bunny_pair(100_000)
    cases["c2_100k_100k"] = (s2, t2, a2, b2)
    with IcpEngine(local_rank) as e:
        e.set_search_mode("auto")
        for name, (s_, t_, a_, b_) in cases.items():
            best, parts, res, walls = 1e9, None, None, []
            for _ in range(5):
                t0 = time.perf_counter()
                e.set_target(t_)
                t1 = time.perf_counter()
                e.set_source(s_, stride=1)
                e.set_matrices(a_, b_)
                t2_ = time.perf_counter()
                r = e.run(iters=50, thresh=0.5, target_d=0.01, use_target=True, early_exit=True)
                t3 = time.perf_counter()
                walls.append(t3 - t0)
                if t3 - t0 < best:
                    best, parts, res = t3 - t0, (t1 - t0, t2_ - t1, t3 - t2_), r
            out[name] = {"ms": 1e3 * best, "median_ms": 1e3 * float(np.median(walls)), "max_ms": 1e3 * max(walls), "set_target_ms": 1e3 * parts[0], "set_source_ms": 1e3 * parts[1], "run_ms": 1e3 * parts[2],
                         "iterations": int(res.iters_done), "converged": bool(res.converged), "last_K": int(res.last_K)}
        # what the reference's operator really runs (operators/icp_align.py:47-161 on a base object WITH faces): surface mode --
        # host vertices + triangles -> set_target_mesh (triangle images, tree, grid) + set_source + the early-exit loop
        mesh_cases = {"surface_200k_82ktri": (synth.lattice_surface_mesh(143, 288), 200_000),
                      "surface_1M_2Mtri": (synth.lattice_surface_mesh(700, 1400), 1_000_000)}
        s_mxa = synth.rigid4(synth.rotation_from_rotvec([0.02, -0.015, 0.025]), [0.01, -0.008, 0.012])
        eye4 = np.identity(4, dtype=np.float32)
        for name, ((verts, tris), n_src) in mesh_cases.items():
            pts = synth.bunny_surface(n_src, offset=0.37)
            best, parts, res, walls = 1e9, None, None, []
            for _ in range(5):
                t0 = time.perf_counter()
                e.set_target_mesh(verts, tris)
                t1 = time.perf_counter()
                e.set_source(pts, stride=1)
                e.set_matrices(s_mxa, eye4)
                t2_ = time.perf_counter()
                r = e.run(iters=50, thresh=0.05, target_d=0.01, use_target=True, early_exit=True)
                t3 = time.perf_counter()
                walls.append(t3 - t0)
                if t3 - t0 < best:
                    best, parts, res = t3 - t0, (t1 - t0, t2_ - t1, t3 - t2_), r
            out[name] = {"ms": 1e3 * best, "median_ms": 1e3 * float(np.median(walls)), "max_ms": 1e3 * max(walls), "set_target_mesh_ms": 1e3 * parts[0],
                         "set_source_ms": 1e3 * parts[1], "run_ms": 1e3 * parts[2], "n_source": int(n_src), "n_target_vertices": int(len(verts)),
                         "n_target_triangles": int(len(tris)), "iterations": int(res.iters_done), "converged": bool(res.converged), "last_K": int(res.last_K),
                         "mean_dist": float(res.mean_dist)}
    return out


def surface_leg(args, local_rank):
    """SURVEY 8f rank 1 ("next" row, reported beside the headline): surface mode -- closest point on the base mesh's
    triangles, what the reference's BVHTree.find_nearest returns -- on a 1M-point cloud against a ~1M-vertex /
    ~2M-triangle mesh of the synthetic bunny surface, library-default search (AUTO: grid + tree).  Two timings from
    the same cold start: the first 5 iterations (what an early-exit operator call lives in) and 30 iterations."""
    import torch
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    sv, st = synth.lattice_surface_mesh(700, 1400)
    ssrc = synth.bunny_surface(args.n_source, offset=0.37)
    s_mxa = synth.rigid4(synth.rotation_from_rotvec([0.02, -0.015, 0.025]), [0.01, -0.008, 0.012])
    eye4 = np.identity(4, dtype=np.float32)
    legs = {}
    with IcpEngine(local_rank) as se:
        se.set_target_mesh(sv, st)
        se.set_source(ssrc, stride=1)
        se.set_matrices(s_mxa, eye4)
        se.run(iters=3, thresh=0.05, early_exit=False)             # warm-up (module load, first launches)
        entries, cells = se.stat("tri_grid_entries"), se.stat("tri_grid_cells")
        for iters in (5, 30):
            se.set_matrices(s_mxa, eye4)
            se.reset_seeds()                                           # cold start: no correspondences from the last run
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sres = se.run(iters=iters, thresh=0.05, early_exit=False)
            dt = time.perf_counter() - t0
            legs[iters] = (dt, sres)
    dt30, r30 = legs[30]
    dt5, r5 = legs[5]
    ns, ntri = float(len(ssrc)), float(len(st))
    # one search reads every query (16 B) and its seed (4 B), writes its key (8 B), and touches every triangle image
    # (36 B of coordinates), cell-list entry (20 B: sphere record + triangle id) and cell offset (4 B) at least once
    algo = 28.0 * ns + 36.0 * ntri + 20.0 * entries + 4.0 * cells
    nn30 = r30.nn_ms_total / 30
    e, stamp = pmc_entry("surface_%dx%d_n1" % (len(ssrc), len(st)), KERNELS["surface_grid"])
    return {"what": "SURVEY 8f rank 1 (next row): surface mode, closest point on triangles (k_tri_search_grid + "
                    "k_bvh_search, OA_SEARCH_AUTO), bit-identical to the oracle's brute force over all triangles (tests)",
            "n_source": int(ns), "n_target_vertices": int(len(sv)), "n_target_triangles": int(ntri),
            "steps": 30, "value": 30 / dt30, "unit": "iterations/s", "ms_per_step": 1e3 * dt30 / 30,
            "ms_per_nn_search": nn30, "last_K": r30.last_K, "mean_dist": r30.mean_dist,
            "cold_5_iterations": {"ms_per_step": 1e3 * dt5 / 5, "ms_per_nn_search": r5.nn_ms_total / 5,
                                  "note": "same cold start, first 5 iterations only: stale seeds, long reach -- the "
                                          "regime a typical early-exit operator call (5-10 iterations) runs in"},
            "roofline": {"bound": "hbm", "achieved": algo / (nn30 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": algo / (nn30 * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_search": algo,
                         "traffic": e["bytes_per_launch"] if e else None, "traffic_profile": stamp,
                         "formula": "(28 N_s + 36 N_tris + 20 cell-list entries + 4 cells) bytes / search time "
                                    "(GPU-side stamps: end of the previous iteration -> start of k_pair_accumulate), "
                                    "30-iteration average; latency-bound dependent lookups, not a streaming kernel",
                         "cell_list_entries": entries, "cells": cells}}


# DESIGN.md 4 (derivation: docs/HISTORY.md 4.7)'s table, for the line that sits next to the measurement (ms per iteration, AUTO search, masked + normal-angle
# test, first 10 iterations; the 1-GPU figure is measured -- profiles/r03e_baseline_configs.txt --, the 8-GPU one predicted from
# the per-shard search time + ~30 us of reduce / exchange / solve)
C5_PREDICTED_MS_PER_ITERATION = {1: 2.06, 8: 0.35}
# the headline workload (BASELINE config 4 at N = 8), brute-force kernel: one GPU measured (round 6, k_nn_search_sorted through its work queue: 26.4 ms),
# 8 GPUs PREDICTED from the 125k-point shard's search time on one GPU (3.93 ms, profiles/r06z_baseline_configs.txt) + the
# exchange (derivation: docs/HISTORY.md 4.7)
C4_PREDICTED_MS_PER_ITERATION = {1: 26.4, 8: 3.96}


def c5_leg(args, n_gpus, in_process, world, rank, local_rank, devices, dev, backend):
    """BASELINE config 5 beside the headline of a multi-GPU run: 10M source points on the synthetic surface, a seeded 10 %
    cap excluded (the icp_exclude mask -> vlist, operators/icp_align.py:67-76), 2M target vertices, the normal-angle test
    (45 degrees; an extension), library-default search (AUTO), source sharded over the N GPUs.  Bounded: `--c5-steps`
    iterations from a cold start after a 2-iteration warm-up.  Returns the dict for `c5_path` (rank 0) or None."""
    import torch
    import torch.distributed as dist
    from object_alignment_amd import synth
    from object_alignment_amd.distributed import EngineShard, new_sums_tensor, run_sharded
    from object_alignment_amd.engine import IcpEngine
    from object_alignment_amd.operators.icp_align import vlist_from_weights
    src, src_n = synth.bunny_surface_with_normals(args.c5_source, 0.5)
    tgt, tgt_n = synth.bunny_surface_with_normals(args.c5_target, 0.0)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.003, -0.002, 0.004]), [0.002, -0.001, 0.0015])
    eye = np.identity(4, dtype=np.float32)
    rng = np.random.default_rng(500)
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    h = src.astype(np.float64) @ axis
    cap = np.nonzero(h > np.quantile(h, 0.9))[0]                                  # the seeded 10 % cap
    vlist = np.array(vlist_from_weights(len(src), exclude=[(int(v), 1.0) for v in cap]), dtype=np.int64)
    eng = IcpEngine(devices=devices) if in_process else IcpEngine(local_rank)
    try:
        if not in_process:
            eng.set_stream(torch.cuda.current_stream().cuda_stream)
        t0 = time.perf_counter()
        eng.set_target(tgt)
        if in_process or world == 1:
            eng.set_source(src, vlist=vlist, stride=1)
        else:
            eng.set_source(src, vlist=vlist, stride=1, shard_index=rank, shard_count=world)
        eng.set_normals(src_n, tgt_n, 45.0)
        for d in set(devices):
            torch.cuda.synchronize(d)
        upload_s = time.perf_counter() - t0
        kw = dict(thresh=0.5, target_d=0.01, use_target=True, with_scale=False, early_exit=False)
        sums = None if (in_process or world == 1) else new_sums_tensor(dev)

        def loop(iters):
            if in_process or world == 1:
                return eng.run(iters=iters, **kw)
            return run_sharded(EngineShard(eng, iters=iters, **kw), iters, sums, world_size=world)

        def barrier():
            if world > 1:
                dist.barrier()
            for d in set(devices):
                torch.cuda.synchronize(d)

        eng.set_matrices(mxa, eye)
        loop(2)
        eng.set_matrices(mxa, eye)
        eng.reset_seeds()
        barrier()
        t0 = time.perf_counter()
        r = loop(args.c5_steps)
        barrier()
        dt = time.perf_counter() - t0
        nn = r.nn_ms_total / max(1, args.c5_steps)
        nn_min = nn_max = nn
        if in_process:
            nn_min, nn_max = eng.stat("nn_ms_min") / args.c5_steps, eng.stat("nn_ms_max") / args.c5_steps
            xinfo = eng.exchange_info()
            exchange, rccl_ranks = xinfo["exchange"], xinfo["rccl_ranks"]
        elif world > 1:
            t = torch.tensor([dt, nn, -nn], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, nn_max, nn_min = float(t[0]), float(t[1]), -float(t[2])
            exchange = "torch.distributed all_reduce (%s)" % ("nccl = RCCL" if backend == "nccl" else backend)
            rccl_ranks = world if backend == "nccl" else 0
        else:
            exchange, rccl_ranks = None, 0
        if rank != 0:
            return None
        ms = 1e3 * dt / args.c5_steps
        pred = C5_PREDICTED_MS_PER_ITERATION.get(n_gpus) if (args.c5_source, args.c5_target) == (10_000_000, 2_000_000) else None
        return {"what": "BASELINE config 5: 10M <-> 2M, 10 % of the source masked out (icp_exclude -> vlist), normal-angle "
                        "rejection at 45 degrees (extension), OA_SEARCH_AUTO, source sharded over the GPUs",
                "n_source": int(len(src)), "n_selected": int(len(vlist)), "n_target": int(len(tgt)), "n_gpus": n_gpus,
                "steps": args.c5_steps, "value": args.c5_steps / dt, "unit": "iterations/s", "ms_per_step": ms,
                "ms_per_nn_search_per_device": {"min": nn_min, "max": nn_max},
                "exchange": exchange, "rccl_ranks": rccl_ranks, "last_K": r.last_K, "mean_dist": r.mean_dist,
                "upload_ms": 1e3 * upload_s,
                "predicted_ms_per_step_design_4_7": pred,
                "measured_over_predicted": (ms / pred) if pred else None,
                "prediction_note": "DESIGN.md 4 (derivation: docs/HISTORY.md 4.7): 1 GPU measured in round 3 (2.06 ms), 8 GPUs PREDICTED (0.35 ms = shard search "
                                   "0.32 + ~0.03 reduce / exchange / solve); no figure for 2 and 4 GPUs"}
    finally:
        eng.close()


def mfma_leg(args, local_rank, src, tgt, mxa, mxb, kw, ref_matrix):
    """EXPERIMENT, reported beside the headline and never instead of it (BASELINE.json's north-star describes the
    brute-force search without MFMA): the same cold run with OA_NN_MFMA=1 -- the first filter level of the brute-force
    search as one v_mfma_f32_32x32x16_f16 per 32 targets x 32 points (object_alignment_amd/csrc/oa_mfma.hpp)."""
    from object_alignment_amd.engine import IcpEngine
    os.environ["OA_NN_MFMA"] = "1"
    try:
        with IcpEngine(local_rank, experiments=True) as e:      # liboa_icp_exp.so: the default library does not carry the experiment
            e.set_search_mode("brute")
            e.set_target(tgt)
            e.set_source(src, stride=1)
            if e.stat("brute_kernel") != 2.0:
                return {"error": "k_nn_search_mfma does not take this shard (needs >= 65536 points and > 65536 target vertices)"}
            e.set_matrices(mxa, mxb)
            e.run(iters=1, **kw)
            e.set_matrices(mxa, mxb)
            e.reset_seeds()
            t0 = time.perf_counter()
            r = e.run(iters=args.steps, **kw)
            dt = time.perf_counter() - t0
    finally:
        os.environ.pop("OA_NN_MFMA", None)
    nn_ms = r.nn_ms_total / max(1, args.steps)
    pairs = float(args.n_source) * float(args.n_target)
    mfma_tflops = 32.0 * pairs / (nn_ms * 1e-3) / 1e12           # 2 x K = 16 flop per pair and MFMA, all executed
    return {
        "what": "EXPERIMENT (liboa_icp_exp.so with OA_NN_MFMA=1; not in the default library): k_nn_search_mfma, filter levels 1-2 of the brute-force search "
                "on the matrix cores (binary16 hi/lo split, sign test); same correspondences",
        "value": args.steps / dt, "unit": "iterations/s", "steps": args.steps, "ms_per_step": 1e3 * dt / args.steps,
        "ms_per_nn_search": nn_ms, "final_matrix_bitwise_equal_to_default_kernel": bool(np.array_equal(r.matrix_world, ref_matrix)),
        "roofline": {"bound": "mfma", "achieved": mfma_tflops, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": mfma_tflops / MFMA_F16_PEAK_TFLOPS, "kernel": "k_nn_search_mfma",
                     "formula": "32 flop (2 x K = 16) x N_s x N_t / search time / 2500 TFLOP/s (dense f16 MFMA peak, "
                                "MI355X_MICROARCH.md)",
                     "note": "the loop issues 8 v_or3_b32 per MFMA to look at the signs of its 1024 results; those half-rate "
                             "VALU instructions, not the matrix pipe, bound it (tools/mfma_microbench.hip: MFMA alone 17 ms per "
                             "1e12 pairs, with the sign test 25.5 ms)"},
    }


def main():
    args = parse()
    os.environ.pop("OA_NN_MFMA", None)                            # the headline is the north-star kernel, never the experiment
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL needs dmabuf IPC on this host driver
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    import torch
    import torch.distributed as dist

    if world > 1 and args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the oa_icp engine has no CPU fallback")
    in_process = world == 1 and args.gpus > 1                     # launched plainly: ONE process drives all the GPUs
    n_gpus = args.gpus if in_process else world
    same_device = os.environ.get("OA_BENCH_SAME_DEVICE") == "1"  # test hook: all ranks / shards on GPU 0
    if in_process and not same_device and torch.cuda.device_count() < args.gpus:
        raise SystemExit("--gpus %d but only %d visible" % (args.gpus, torch.cuda.device_count()))
    backend = os.environ.get("OA_BENCH_BACKEND", "nccl")          # test hook: gloo
    if same_device:
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
    if in_process:
        devices = [0] * n_gpus if same_device else list(range(n_gpus))
        eng = IcpEngine(devices=devices)
        xinfo = eng.exchange_info()                  # resolves AUTO: RCCL on distinct devices, else the mailbox
        exchange, rccl_ranks, host_threads = xinfo["exchange"], xinfo["rccl_ranks"], xinfo["host_threads"]
    else:
        devices = [local_rank]
        eng = IcpEngine(local_rank)
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        exchange = ("torch.distributed all_reduce (%s)" % ("nccl = RCCL" if backend == "nccl" else backend)) if world > 1 else None
        rccl_ranks = dist.get_world_size() if (world > 1 and backend == "nccl") else 0
        host_threads = 1
    eng.set_search_mode("brute")          # the north-star kernel: LDS-tiled brute force (grid path measured below)
    t0 = time.perf_counter()
    eng.set_target(tgt)
    if in_process:
        eng.set_source(src, stride=1)
    else:
        eng.set_source(src, stride=1, shard_index=rank, shard_count=world)
    for d in set(devices):
        torch.cuda.synchronize(d)
    upload_s = time.perf_counter() - t0
    sums = None if in_process else new_sums_tensor(dev)
    kw = dict(thresh=0.5, target_d=0.01, use_target=True, with_scale=False, early_exit=False)

    def barrier():
        if world > 1:
            dist.barrier()
        for d in set(devices):
            torch.cuda.synchronize(d)

    def loop(iters):
        if in_process or world == 1:
            return eng.run(iters=iters, **kw)                    # the whole loop inside the library (oa_run): one GPU, or all of them
        return run_sharded(EngineShard(eng, iters=iters, **kw), iters, sums, world_size=world)

    def timed(steps, warmup):
        """W untimed + exactly `steps` timed iterations, both from the initial pose; the timed run starts cold (no
        correspondence seeds left over from the warm-up); max over ranks."""
        if warmup > 0:
            eng.set_matrices(mxa, mxb)
            loop(warmup)
        eng.set_matrices(mxa, mxb)
        eng.reset_seeds()
        barrier()
        t0 = time.perf_counter()
        r = loop(steps)
        barrier()
        dt = time.perf_counter() - t0
        ms = r.nn_ms_total / max(1, steps)
        # per device: search time (fastest / slowest), the wait for the world's sums (GPU-side stamps), host enqueue time
        diag = {"search_ms_per_device": {"min": ms, "max": ms}, "exchange_us_per_iteration": None, "host_enqueue_us_per_iteration": None}
        try:
            diag["exchange_us_per_iteration"] = eng.stat("exchange_us")
            if in_process and n_gpus > 1:
                diag["search_ms_per_device"] = {"min": eng.stat("nn_ms_min") / max(1, steps), "max": eng.stat("nn_ms_max") / max(1, steps)}
                diag["host_enqueue_us_per_iteration"] = eng.stat("enqueue_us")
        except Exception:
            pass
        if world > 1:
            t = torch.tensor([dt, ms, -ms, diag["exchange_us_per_iteration"] or 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, ms = float(t[0]), float(t[1])
            diag["search_ms_per_device"] = {"min": -float(t[2]), "max": ms}
            diag["exchange_us_per_iteration"] = float(t[3])
        timed.diag = diag
        return r, dt, ms

    def ceiling():
        """What the vector ALUs issue right now (5 ms of v_add_f32 / v_min3_f32 on every SIMD, oa_measure_valu_ceiling) and the shader clock
        under that load: taken right before and right after the timed loop, it tells a throttling box from a slower kernel."""
        try:
            return eng.valu_ceiling(5.0)
        except Exception as exc:                                  # never lose the headline line
            return {"error": repr(exc)}

    exchange_note = None
    ceil_before = ceiling()
    try:
        res, elapsed, nn_ms = timed(args.steps, args.warmup)
    except Exception as exc:
        # in-process multi-GPU run on the in-library RCCL exchange (never run on distinct GPUs before a first hardware run):
        # if its loop ends with OA_E_RCCL -- the watchdog aborted a stalled collective -- the line is still measured, on the
        # mailbox exchange (peer-mapped device memory), and says so
        if not (in_process and n_gpus > 1 and isinstance(exchange, str) and exchange.startswith("rccl")):
            raise
        exchange_note = "RCCL exchange failed (%r); measured on the mailbox exchange" % (exc,)
        eng.set_exchange("mailbox")
        xinfo = eng.exchange_info()
        exchange, rccl_ranks, host_threads = xinfo["exchange"], xinfo["rccl_ranks"], xinfo["host_threads"]
        res, elapsed, nn_ms = timed(args.steps, args.warmup)
    ceil_after = ceiling()
    try:
        brute_kind = eng.stat("brute_kernel")                     # which brute-force kernel the timed loop launched
    except Exception:
        brute_kind = 3.0
    try:
        search_clock_mhz = eng.stat("search_clock_mhz")           # the shader clock during the last launch of the timed loop
    except Exception:
        search_clock_mhz = None
    try:
        launch_ms = eng.search_ms()
    except Exception:
        launch_ms = np.zeros(0)
    head_diag = dict(getattr(timed, "diag", {}))
    xinfo_after = eng.exchange_info() if in_process else {}
    if in_process and xinfo_after.get("rccl_fallbacks"):
        # AUTO began on RCCL and the library finished the loop through the mailboxes (multi_run): say so, with RCCL's reason
        exchange_note = "the library fell back to the mailbox exchange: %s" % xinfo_after.get("note")
        exchange, rccl_ranks = xinfo_after["exchange"], xinfo_after["rccl_ranks"]

    # SURVEY 8f rank 2 ("next" row, reported beside the headline, never instead of it): the same run with the
    # uniform-grid exact search.  Correspondences are identical, so the final matrix must be bitwise the same.
    grid = None
    if not args.no_grid:
        try:
            eng.set_search_mode("grid")
            g_steps = max(args.steps, 200)
            gres, g_elapsed, g_nn_ms = timed(g_steps, max(args.warmup, 5))
            gcheck, _, _ = timed(args.steps, 0)
            grid = (g_steps, g_elapsed, g_nn_ms, bool(np.array_equal(gcheck.matrix_world, res.matrix_world)), bool(eng.stat("safe_radii")))
        except Exception as exc:                                  # never lose the headline line
            grid = ("error: %r" % (exc,),)

    mfma = None
    if n_gpus == 1 and not args.no_mfma:
        try:
            mfma = mfma_leg(args, local_rank, src, tgt, mxa, mxb, kw, res.matrix_world)
        except Exception as exc:                                  # never lose the headline line
            mfma = {"error": repr(exc)}

    surf = None
    if n_gpus == 1 and not args.no_surface:
        try:
            surf = surface_leg(args, local_rank)
        except Exception as exc:                                  # never lose the headline line
            surf = {"error": repr(exc)}

    whole = None
    if n_gpus == 1 and not args.no_whole_call:
        try:
            whole = whole_call_leg(local_rank, src, tgt, mxa, mxb)
        except Exception as exc:                                  # never lose the headline line
            whole = {"error": repr(exc)}

    c5 = None
    if (n_gpus > 1 and not args.no_c5) or (n_gpus == 1 and args.c5):
        try:
            c5 = c5_leg(args, n_gpus, in_process, world, rank, local_rank, devices, dev, backend)
        except Exception as exc:                                  # never lose the headline line
            c5 = {"error": repr(exc)}
            if world > 1:
                raise                                             # (a rank that stays behind would hang the others' barrier)

    if rank == 0:
        assert res.iters_done == args.steps, (res.iters_done, args.steps)
        ns_local = -(-args.n_source // n_gpus)                              # points per GPU (the largest shard)
        pairs = float(ns_local) * float(args.n_target)                      # per launch of the search kernel on one GPU
        key = "%dx%d_n%d" % (args.n_source, args.n_target, n_gpus)
        brute_kernel = BRUTE_KERNELS.get(float(brute_kind), KERNELS["brute"])   # the kernel that RAN (OA_STAT_BRUTE_KERNEL)
        e, stamp = pmc_entry(key, brute_kernel)
        per_pair = e.get("valu_instructions_per_pair") if e else None
        per_pair_src = "pmc" if per_pair else "isa-count"
        if e and e.get("valu_instructions_per_pair_cold") and e.get("valu_instructions_per_pair_seeded") and args.steps >= 1:
            # the first launch of a loop searches without seeds and executes more; the PMC pass holds both kinds: weigh them
            # as THIS run's K launches are (1 cold, K - 1 seeded)
            per_pair = (e["valu_instructions_per_pair_cold"] + (args.steps - 1) * e["valu_instructions_per_pair_seeded"]) / args.steps
            per_pair_src = "pmc (1 unseeded + %d seeded launches)" % (args.steps - 1)
        if not per_pair:
            per_pair = VALU_PER_PAIR_ISA.get(brute_kernel, 3.0)
        laneops = per_pair * pairs / (nn_ms * 1e-3) / 1e12                  # T lane-ops/s actually executed
        eff_tflops = FLOP_PER_PAIR * pairs / (nn_ms * 1e-3) / 1e12
        algo_bytes = 16.0 * ns_local + 12.0 * args.n_target + 8.0 * ns_local  # source float4 + target SoA + keys
        traffic = e["bytes_per_launch"] if e else None
        if n_gpus == 1:
            par = "single GPU"
        elif in_process:
            par = ("one process, %d GPUs (oa_create_multi): source sharded x%d, target replicated, in-library exchange of "
                   "24 f64 per iteration (%s)" % (n_gpus, n_gpus, exchange))
        else:
            par = ("one process per GPU: source sharded x%d, target replicated, all-reduce of 24 f64 per iteration (%s)"
                   % (world, exchange))
        # the burn's issue rate PER CLOCK (it runs at whatever clock the box holds for 5 ms: 2090-2420 MHz seen), priced at the clock one
        # workgroup of the search stamped DURING the search -- the lower of the two burns at their own clocks when that stamp is missing.
        # (Round 6: the lower burn alone read 0.99-1.005 of itself once the work queue had taken the idle out of the launch: a burn at
        #  2087 MHz is no ceiling for a search at 2350.)
        ceil_t = [c["tlaneops"] for c in (ceil_before, ceil_after) if "tlaneops" in c and c["tlaneops"] > 0]
        per_clock = [c["tlaneops"] * 1e6 / c["shader_clock_mhz"] for c in (ceil_before, ceil_after)
                     if c.get("tlaneops", 0) > 0 and c.get("shader_clock_mhz", 0) > 0]                 # lane-ops per shader cycle, whole chip
        if per_clock and search_clock_mhz:
            ceil_now = max(per_clock) * search_clock_mhz * 1e-6
        else:
            ceil_now = min(ceil_t) if ceil_t else None
        half_share = HALF_RATE_SHARE.get(brute_kernel)
        out = {
            "metric": "ICP iterations/sec + ms/NN-search, 1M<->1M verts",
            "value": args.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_nn_search": nn_ms,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32", "accumulate_dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "1M<->1M uniform [-1,1]^3 clouds, sigma = 5%% of mean spacing, seed 1234, "
                                   "thresh 0.5, stride 1, %d iterations from a cold start (no seeds), early-exit off" % args.steps,
                       "n_source": args.n_source, "n_target": args.n_target, "search": "brute force (north-star kernel)",
                       "parallelism": par, "exchange": exchange, "exchange_note": exchange_note, "rccl_ranks": rccl_ranks, "host_threads_per_process": host_threads,
                       "host_enqueue_us_per_iteration": (eng.stat("enqueue_us") if in_process else None)},
            # a first run on real multi-GPU hardware should explain itself: per-device search times, the GPU-side wait for the
            # world's sums, what the exchange resolved to and why, and DESIGN.md 4 (derivation: docs/HISTORY.md 4.7)'s prediction for this N beside the measurement
            "multi_gpu": ({**head_diag,
                           "exchange": exchange, "exchange_note": exchange_note or (xinfo_after.get("note") or None),
                           "rccl_ranks": rccl_ranks,
                           "rccl_ranks_of_the_last_communicator": (xinfo_after.get("rccl_ranks_last") if in_process else rccl_ranks),
                           "rccl_fallbacks": xinfo_after.get("rccl_fallbacks", 0),
                           "predicted_ms_per_step_design_4_7": C4_PREDICTED_MS_PER_ITERATION.get(n_gpus),
                           "measured_over_predicted": ((1e3 * elapsed / args.steps) / C4_PREDICTED_MS_PER_ITERATION[n_gpus])
                           if n_gpus in C4_PREDICTED_MS_PER_ITERATION and (args.n_source, args.n_target) == (1_000_000, 1_000_000) else None}
                          if n_gpus > 1 else None),
            "roofline": {"bound": "valu", "achieved": laneops, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s",
                         "frac": laneops / VALU_PEAK_TLANEOPS, "traffic": traffic,
                         "kernel": brute_kernel, "valu_instructions_per_pair": per_pair,
                         "valu_instructions_per_pair_source": per_pair_src, "pairs_per_launch": pairs,
                         "avg_launch_ms": nn_ms, "traffic_profile": stamp,
                         # every launch of the timed loop (hipEvent pairs on the kernel's stream): a throttled box shows here
                         "launch_ms": ({"min": float(launch_ms.min()), "median": float(np.median(launch_ms)), "max": float(launch_ms.max()),
                                        "n": int(len(launch_ms))} if len(launch_ms) else None),
                         # ... and here: what v_fma_f32 issues on this box right before / right after the timed loop, and the
                         # shader clock under that load (nominal: 78.6 T lane-ops/s at 2400 MHz)
                         "frac_of_nominal": laneops / VALU_PEAK_TLANEOPS,
                         "measured_issue_ceiling": {"unit": "Tlane-op/s", "instruction": "v_add_f32 (two register sources: one wave-instruction per SIMD every two cycles = 32 lanes per clock), "
                                                                   "16 independent chains, 8 waves per SIMD, ~2.5 ms; v_min3_f32 (`tlaneops_min3`: the half-rate class) beside it",
                                                    "before_timed_loop": ceil_before, "after_timed_loop": ceil_after, "used": ceil_now,
                                                    "used_is": "the higher of the two burns' lane-ops per shader cycle x shader_clock_mhz_during_search "
                                                               "(nominal: 32768 lane-ops per cycle); without that clock: the lower burn as measured",
                                                    "lane_ops_per_cycle": (max(per_clock) if per_clock else None)},
                         "frac_of_measured_ceiling": (laneops / ceil_now) if ceil_now else None,
                         "half_rate_instruction_share": half_share,
                         "shader_clock_mhz_during_search": search_clock_mhz,
                         "what_frac_is": "VALU issue-slot utilisation: executed vector-ALU instructions (PMC) x 64 lanes / launch time, against the "
                                         "nominal 78.6 T lane-ops/s.  It says how full the pipe is, NOT how good the kernel is: a third of these "
                                         "instructions are of the half-rate class (v_min3_f32 / v_cmp: at 4 cycles each the pipe would be full at "
                                         "0.75-0.78 -- round 5's reading of 0.79; since the work queue of round 6 the launch reaches 0.82-0.83, so "
                                         "that class costs less in this mix than a burn of it measures and round 5's 'full pipe' had idle in it), "
                                         "and most of them prove losers the images' order already implies (DESIGN.md 3.1: a block's [u_min, u_max] "
                                         "would replace level 0 -- which turns brute force into the slab search of row f2).  The loop is frozen "
                                         "(VERDICT r5); round 6 changed who runs which piece of it (DESIGN.md 3.1, the work queue).",
                         "formula": "frac = valu_instructions_per_pair x pairs_per_launch / avg_launch_ms / 78.6e12 lane-ops/s "
                                    "(256 CU x 4 SIMD x 32 lanes x 2.4 GHz); valu_instructions_per_pair = SQ_INSTS_VALU x 64 / "
                                    "pairs from the committed PMC pass (profiles/), avg_launch_ms = hipEvent pairs around "
                                    "every launch of this run",
                         "pairs_per_s": pairs / (nn_ms * 1e-3),
                         # SURVEY 8d's primary formula, as written: 8 flop per (source, target) pair / launch time / 157.3 TFLOP/s.
                         # Above 1 the model counts arithmetic the kernel never executes (level 0 proves ~97 % of the pairs losers with
                         # one v_sub_f32 + half a v_min3_f32): then it is void as a roofline and says so.
                         "frac_8d_algorithmic": {"value": eff_tflops / FP32_VECTOR_PEAK_TFLOPS, "void": bool(eff_tflops / FP32_VECTOR_PEAK_TFLOPS > 1.0),
                                                 "effective_tflops": eff_tflops,
                                                 "formula": "8 flop x N_s x N_t / avg_launch_ms / 157.3 TFLOP/s (SURVEY 8d)"}},
            "roofline_hbm": {"bound": "hbm", "achieved": algo_bytes / (nn_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": algo_bytes / (nn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": algo_bytes, "traffic": traffic},
            "result": {"final_translation": res.last_translation, "last_K": res.last_K,
                       "mean_dist": res.mean_dist, "std_dist": res.std_dist},
            "upload_ms": 1e3 * upload_s,
            "loop_ms_hipevents": res.loop_ms,
        }
        if grid is not None and len(grid) == 5:
            g_steps, g_elapsed, g_nn_ms, same, safe_radii = grid
            g_bytes = 40.0 * ns_local + 16.0 * args.n_target        # source float4 + winner record + key, sorted target image
            ge, gstamp = pmc_entry("grid_" + key, KERNELS["grid"])
            out["grid_path"] = {
                "what": "SURVEY 8f rank 2 (next row): k_nn_search_grid, exact uniform-grid search, same correspondences",
                "value": g_steps / g_elapsed, "unit": "iterations/s", "steps": g_steps,
                "ms_per_step": 1e3 * g_elapsed / g_steps, "ms_per_nn_search": g_nn_ms,
                "final_matrix_bitwise_equal_to_brute_force": same,
                "safe_radii": safe_radii,                              # docs/HISTORY.md 4.4: seeds inside their safe radius settle their query without a scan
                "roofline": {"bound": "hbm", "achieved": g_bytes / (g_nn_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": g_bytes / (g_nn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": g_bytes, "traffic": ge["bytes_per_launch"] if ge else None,
                             "traffic_profile": gstamp,
                             "formula": "(40 N_s,local + 16 N_t) bytes / search time (GPU-side stamps)",
                             "note": "latency-bound dependent lookups (cell range -> vertices); not a streaming kernel"},
            }
        elif grid is not None:
            out["grid_path"] = {"error": grid[0]}
        if mfma is not None:
            out["mfma_experiment"] = mfma
        if surf is not None:
            out["surface_path"] = surf
        if whole is not None:
            out["whole_call"] = whole
        if c5 is not None:
            out["c5_path"] = c5
        if n_gpus == 1 and not args.no_cpu_baseline and args.cpu_iters > 0:
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

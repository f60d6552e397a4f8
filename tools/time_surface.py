#!/usr/bin/env python3
"""GPU box: surface-mode (closest point on triangles) timing at scale, next to vertex mode on the same meshes.
Usage: python tools/time_surface.py [nu] [nv] [n_source]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 700
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 1400
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
partial = os.environ.get("PARTIAL", "0") == "1"      # keep only the z > 0 half of the target: half the queries are far
if os.environ.get("MESH", "lattice") == "cubed":          # no poles: six patches of nu x nu quads
    tgt, tris = synth.cubed_surface_mesh(nu)
else:
    tgt, tris = synth.lattice_surface_mesh(nu, nv)
if partial:
    keep_v = tgt[:, 2] > 0.0
    keep_t = keep_v[tris].all(axis=1)
    remap = np.cumsum(keep_v) - 1
    tgt, tris = tgt[keep_v], remap[tris[keep_t]].astype(np.int32)
src = synth.bunny_surface(ns, offset=0.37)
mxa = synth.rigid4(synth.rotation_from_rotvec([0.02, -0.015, 0.025]), [0.01, -0.008, 0.012])
mxb = np.identity(4, dtype=np.float32)
print("target %d vertices, %d triangles; source %d points" % (len(tgt), len(tris), ns), flush=True)
combos = [("surface", "grid"), ("surface", "bvh"), ("vertex", "grid"), ("vertex", "bvh")]
if os.environ.get("ONLY"):                               # e.g. ONLY=surface:grid
    combos = [tuple(os.environ["ONLY"].split(":"))]
for mode, search in combos:
    with IcpEngine(0, experiments=os.environ.get("EXP", "0") == "1") as e:   # EXP=1: liboa_icp_exp.so (OA_GRID_STATS, OA_TRI_RING ...)
        e.set_search_mode(search)
        t0 = time.perf_counter()
        if mode == "surface":
            e.set_target_mesh(tgt, tris)
        else:
            e.set_target(tgt)
        e.set_source(src, stride=1)
        up = time.perf_counter() - t0
        for it in [int(a) for a in os.environ.get("ITERS", "5,30").split(",")]:     # ITERS=1,3: other loop lengths from the same cold start
            e.set_matrices(mxa, mxb)
            if os.environ.get("WG_LOG_DIR"): os.environ["OA_WG_LOG"] = os.path.join(os.environ["WG_LOG_DIR"], "it%d.bin" % it)   # (instrumented builds only)
            t0 = time.perf_counter()
            r = e.run(iters=it, thresh=0.05, early_exit=False)
            wall = time.perf_counter() - t0
            print("%-8s %-5s upload+build %.1f ms; iters %d: wall %.3f ms/iter, device loop %.3f ms/iter, nn %.3f ms/iter, K %d, "
                  "mean dist %.3e%s" % (mode, search, 1e3 * up, it, 1e3 * wall / it, r.loop_ms / it, r.nn_ms_total / it, r.last_K,
                                      r.mean_dist, (", neighbour lists built, would settle %d queries now" % e.stat("tri_ring_accepts"))
                                      if mode == "surface" and e.stat("tri_ring") else ""), flush=True)

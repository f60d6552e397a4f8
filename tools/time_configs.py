#!/usr/bin/env python3
"""GPU box: every BASELINE.json configuration on ONE GPU -- whole runs for C1-C3, one rank's shard for C4 / C5 (the
8-GPU jobs are the driver's; a shard's time is what each of the 8 ranks spends between all-reduces).
Brute force = the north-star kernel; auto = the library default (grid / tree, same correspondences)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine
from object_alignment_amd.operators.icp_align import vlist_from_weights


def run(name, src, tgt, mxa, mxb, iters, modes, shard=(0, 1), vlist=None, normals=None, thresh=0.5):
    for mode in modes:
        with IcpEngine(0) as e:
            e.set_search_mode(mode)
            e.set_target(tgt)
            e.set_source(src, vlist=vlist, stride=1, shard_index=shard[0], shard_count=shard[1])
            if normals is not None:
                e.set_normals(normals[0], normals[1], 45.0)
            e.set_matrices(mxa, mxb)
            e.run(iters=2, thresh=thresh, early_exit=False)
            e.set_matrices(mxa, mxb)
            t0 = time.perf_counter()
            r = e.run(iters=iters, thresh=thresh, early_exit=False)
            dt = time.perf_counter() - t0
            print("%-44s %-5s %3d iterations: %9.3f ms total, %8.3f ms / iteration, search %8.3f ms, K %d" % (
                name, mode, iters, 1e3 * dt, 1e3 * dt / iters, r.nn_ms_total / iters, r.last_K), flush=True)


eye = np.identity(4, dtype=np.float32)
run("C1  2562 <-> 2562 icospheres", *synth.c1_icospheres(), 10, ("brute", "auto"))
run("C2  100k <-> 100k bunny", *synth.c2_bunny_pair(100_000), 50, ("brute", "auto"))
run("C3  1M <-> 1M random, 1 GPU", *synth.c3_random_pair(1_000_000), 50, ("brute", "auto"))
run("C4  1M <-> 1M, shard 1 of 8", *synth.c3_random_pair(1_000_000), 50, ("brute", "auto"), shard=(0, 8))
# C5: 10M source / 2M target on the bunny surface, 10 % cap excluded (icp_exclude -> vlist), normal-angle test 45 deg
src, sn = synth.bunny_surface_with_normals(10_000_000, 0.5)
tgt, tn = synth.bunny_surface_with_normals(2_000_000, 0.0)
cap = np.nonzero(src[:, 2] > np.quantile(src[:, 2], 0.9))[0]
keep = np.ones(len(src), bool); keep[cap] = False
vlist = np.nonzero(keep)[0].astype(np.int64)
mxa = synth.rigid4(synth.rotation_from_rotvec([0.03, -0.02, 0.04]), [0.02, -0.01, 0.015])
run("C5  10M <-> 2M masked + normals, shard 1 of 8", src, tgt, mxa, eye, 10, ("brute", "auto"), shard=(0, 8), vlist=vlist, normals=(sn, tn))
run("C5  10M <-> 2M masked + normals, whole on 1 GPU", src, tgt, mxa, eye, 10, ("auto",), vlist=vlist, normals=(sn, tn))

#!/usr/bin/env python3
"""GPU box: what an iteration costs by path (library-default search, AUTO):
  plain     OA_FUSED_ACC=0      search -> k_pair_accumulate[_canon] -> reduce + solve (round 2's sequence)
  safe      OA_GRID_PATH=safe   grid search -> tree search of the list -> k_pair_accumulate_canon -> reduce + solve
  fast      OA_GRID_PATH=fast   the grid search finishes its leftovers and accumulates itself -> reduce + solve
  adaptive  (default)           safe / fast per iteration, from what the host last heard about the hand-over list
on the BASELINE configurations, plus cases that stress the hand-over (cold starts, a start far from the target, a target
cut in half).  safe / fast / adaptive must give bitwise the same matrices."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine

VARIANTS = [("plain", {"OA_FUSED_ACC": "0"}), ("safe", {"OA_GRID_PATH": "safe"}), ("fast", {"OA_GRID_PATH": "fast"}), ("adaptive", {})]


def run(name, src, tgt, mxa, mxb, iters, cold=False, thresh=0.5, mode="auto"):
    out = {}
    for tag, env in VARIANTS:
        for k in ("OA_FUSED_ACC", "OA_GRID_PATH"):
            os.environ.pop(k, None)
        os.environ.update(env)
        with IcpEngine(0) as e:
            e.set_search_mode(mode)
            e.set_target(tgt)
            e.set_source(src, stride=1)
            e.set_matrices(mxa, mxb)
            e.run(iters=3, thresh=thresh, early_exit=False)
            best = None
            for _ in range(3):
                e.set_matrices(mxa, mxb)
                if cold:
                    e.reset_seeds()
                t0 = time.perf_counter()
                r = e.run(iters=iters, thresh=thresh, early_exit=False)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out[tag] = (best, r)
    ref = out["safe"][1]
    same = all(np.array_equal(out[t][1].step_M, ref.step_M) and np.array_equal(out[t][1].matrix_world, ref.matrix_world) for t in ("fast", "adaptive"))
    close = np.array_equal(out["plain"][1].step_K, ref.step_K) and float(np.abs(out["plain"][1].step_M - ref.step_M).max()) < 1e-9
    print("%-44s %4d it | us/it: %s | safe = fast = adaptive bitwise: %s | plain: K equal, |dM| < 1e-9: %s" % (
        name, iters, "  ".join("%s %8.2f" % (t, 1e6 * out[t][0] / iters) for t, _ in VARIANTS), same, close), flush=True)


eye = np.identity(4, dtype=np.float32)
run("C1  2562 <-> 2562 icospheres", *synth.c1_icospheres(), 200)
run("C2  100k <-> 100k bunny", *synth.c2_bunny_pair(100_000), 200)
run("C2  cold, 5 iterations", *synth.c2_bunny_pair(100_000), 5, cold=True)
s, t, a, b = synth.c3_random_pair(1_000_000)
run("C3  1M <-> 1M random", s, t, a, b, 200)
run("C3  cold, 5 iterations", s, t, a, b, 5, cold=True)
run("C3  cold, 50 iterations", s, t, a, b, 50, cold=True)
run("C4  shard-size 125k <-> 1M", s[:125_000], t, a, b, 200)
run("12k <-> 1M (whole-shard tree)", s[:12_000], t, a, b, 200)
run("24k <-> 1M (tree / grid turns)", s[:24_000], t, a, b, 200)
far = synth.rigid4(synth.rotation_from_rotvec([0.2, -0.1, 0.15]), [0.3, -0.2, 0.25])
s2, t2, a2, _ = synth.c2_bunny_pair(100_000)
run("100k bunny, far start, cold 10 iterations", s2, t2, far, eye, 10, cold=True, thresh=1.0)
run("100k bunny vs target cut in half", s2, t2[t2[:, 0] > 0.0], a2, eye, 50)
for k in ("OA_FUSED_ACC", "OA_GRID_PATH"):
    os.environ.pop(k, None)

#!/usr/bin/env python3
"""GPU box: wall time of one whole alignment call from host arrays -- set_target (+ index build) + set_source +
set_matrices + a 50-iteration run with early exit -- on a context that is kept between calls (steady state: device
blocks come from the library's cache)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine

cases = {}
s, t, a, b = synth.c1_icospheres(); cases["C1 2562 <-> 2562 (vertex)"] = (s, t, None, a, b, 0.5)
s, t, a, b = synth.c2_bunny_pair(100_000); cases["C2 100k <-> 100k (vertex)"] = (s, t, None, a, b, 0.5)
s, t, a, b = synth.c3_random_pair(1_000_000); cases["C3 1M <-> 1M (vertex)"] = (s, t, None, a, b, 0.5)
tv, tt = synth.bumpy_icosphere_mesh(6)
pose = synth.rigid4(synth.rotation_from_rotvec([0.02, -0.015, 0.025]), [0.01, -0.008, 0.012])
eye = np.identity(4, dtype=np.float32)
cases["41k points on an 82k-triangle mesh (surface)"] = (synth.bunny_surface(41_000, offset=0.37), tv, tt, pose, eye, 0.05)
cases["200k points on an 82k-triangle mesh (surface)"] = (synth.bunny_surface(200_000, offset=0.37), tv, tt, pose, eye, 0.05)

only = os.environ.get("ONLY")                                        # substring of a case name (profiling one case); REPS = repetitions
if only:
    cases = {k: v for k, v in cases.items() if only in k}
with IcpEngine(0) as e:
    for name, (src, tgt, tris, mxa, mxb, thresh) in cases.items():
        best, parts, res, walls = 1e9, None, None, []
        for rep in range(int(os.environ.get("REPS", "8"))):
            t0 = time.perf_counter()
            if tris is None:
                e.set_target(tgt)
            else:
                e.set_target_mesh(tgt, tris)
            t1 = time.perf_counter()
            e.set_source(src, stride=1)
            e.set_matrices(mxa, mxb)
            t2 = time.perf_counter()
            r = e.run(iters=50, thresh=thresh, target_d=0.01, use_target=True, early_exit=True)
            t3 = time.perf_counter()
            walls.append(t3 - t0)
            if t3 - t0 < best:
                best, parts, res = t3 - t0, (t1 - t0, t2 - t1, t3 - t2), r
        # (uploads: the target's and the source's coordinates, and the triangle indices of a mesh -- for tools/trace_whole_call.py)
        up = [tgt.nbytes, src.nbytes] + ([np.asarray(tris, np.int32).nbytes] if tris is not None else [])
        print("%-48s best %7.2f ms, median %7.2f  (target %.2f, source %.2f, run %.2f: %d iterations, converged %s, K %d) uploads_bytes=%s" % (
            name, 1e3 * best, 1e3 * float(np.median(walls)), 1e3 * parts[0], 1e3 * parts[1], 1e3 * parts[2], res.iters_done, res.converged,
            res.last_K, ",".join(str(int(x)) for x in up)), flush=True)

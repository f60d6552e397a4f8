#!/usr/bin/env python3
"""GPU box: fixed cost of one oa_run call (begin: synchronise, stage the loop state; end: fetch the state and the history)
against its per-iteration cost: wall time of run(iters = k) for k = 1 .. 8, least-squares line."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine

for name, (src, tgt, mxa, mxb) in (("C1 2562", synth.c1_icospheres()), ("C2 100k", synth.c2_bunny_pair(100_000))):
    with IcpEngine(0) as e:
        e.set_target(tgt); e.set_source(src, stride=1); e.set_matrices(mxa, mxb)
        e.run(iters=20, thresh=0.5, early_exit=False)
        ks, ts = [], []
        for k in (1, 2, 3, 4, 6, 8, 12, 16):
            best = 1e9
            for _ in range(20):
                t0 = time.perf_counter()
                e.run(iters=k, thresh=0.5, early_exit=False)
                best = min(best, time.perf_counter() - t0)
            ks.append(k); ts.append(best * 1e6)
        slope, icpt = np.polyfit(ks, ts, 1)
        print("%-8s run(k): %s us  ->  %.1f us fixed + %.1f us per iteration" % (name, ["%.0f" % t for t in ts], icpt, slope))
        import ctypes as C
        from object_alignment_amd import _capi as capi
        st = e._settings(4, 0.5, 0.01, True, False, False)
        rep = capi.Report()
        best = 1e9
        for _ in range(50):
            t0 = time.perf_counter(); e._L.oa_run(e._h, C.byref(st), C.byref(rep)); best = min(best, time.perf_counter() - t0)
        print("         bare oa_run(4 iterations) through ctypes: %.1f us (the rest is Python: history arrays, RunResult)" % (best * 1e6))

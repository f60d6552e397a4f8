#!/usr/bin/env python3
"""run_cfg.py <c1|c2|c3|c4> <iters> [mode]: one warm AUTO-mode loop on a BASELINE configuration (profiling target)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine
cfg, iters = sys.argv[1], int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "auto"
if cfg == "c1":
    src, tgt, mxa, mxb = synth.c1_icospheres()
elif cfg == "c2":
    src, tgt, mxa, mxb = synth.c2_bunny_pair(100_000)
else:
    src, tgt, mxa, mxb = synth.c3_random_pair(1_000_000)
    if cfg == "c4":
        src = src[:125_000]
with IcpEngine(0) as e:
    e.set_search_mode(mode)
    e.set_target(tgt); e.set_source(src, stride=1); e.set_matrices(mxa, mxb)
    e.run(iters=5, thresh=0.5, early_exit=False)
    e.set_matrices(mxa, mxb)
    r = e.run(iters=iters, thresh=0.5, early_exit=False)
    print(cfg, mode, iters, "iterations", r.loop_ms / iters * 1e3, "us per iteration (hipEvents)")

#!/usr/bin/env python3
"""GPU box: brute-force search time of small / mid-size problems vs the number of workgroups requested."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine
for n in (2562, 20000, 100000, 300000):
    src, tgt, mxa, mxb = synth.c3_random_pair(n)
    row = []
    for blocks in (128, 256, 512, 1024, 2048, 4096, 8192, 16384):
        os.environ["OA_NN_TARGET_BLOCKS"] = str(blocks)
        with IcpEngine(0) as e:
            e.set_search_mode("brute"); e.set_target(tgt); e.set_source(src); e.set_matrices(mxa, mxb)
            e.iterate(thresh=0.5)
            ts = [e.nn_search(want_output=False)[2] for _ in range(12)]
        row.append("%d:%.0f" % (blocks, 1e3 * min(ts)))
    print("n=%-7d ideal %.0f us | blocks:us  %s" % (n, 3.0 * n * n / 59e12 * 1e6, "  ".join(row)), flush=True)

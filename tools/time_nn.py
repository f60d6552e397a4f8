#!/usr/bin/env python3
"""GPU box: seeded k_nn_search_filtered time at 1M<->1M (brute mode), min of N launches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # time shard 0 of `shards` (BASELINE config 4: 8)
src, tgt, mxa, mxb = synth.c3_random_pair(n)
with IcpEngine(0) as e:
    e.set_search_mode("brute")
    e.set_target(tgt); e.set_source(src, shard_index=0, shard_count=shards); e.set_matrices(mxa, mxb)
    e.iterate(thresh=0.5)
    ts = [e.nn_search(want_output=False)[2] for _ in range(5)]
print("n=%d shard 1/%d seeded nn_search ms: min %.3f  all %s  -> %.1f TFLOP/s" % (n, shards, min(ts), ["%.2f" % t for t in ts], 8.0 * (n / shards) * n / min(ts) / 1e9))

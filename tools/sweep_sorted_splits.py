#!/usr/bin/env python3
"""GPU box: target splits of k_nn_search_sorted (OA_NN_SPLITS) on the headline workload and on one shard of eight.
A split is a contiguous range of slabs; a workgroup's points only reach levels 1-3 in the split that holds their own slab,
so few splits per row of workgroups mean uneven workgroups, many mean short ones (prologue, epilogue atomics)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from object_alignment_amd import synth
    from object_alignment_amd.engine import IcpEngine
    n = int(os.environ.get("N", "1000000"))
    src, tgt, mxa, mxb = synth.c3_random_pair(n)[:4]
    for label, shard in (("%d x %d" % (n, n), (0, 1)), ("shard 1 of 8", (0, 8))):
        for splits in [int(a) for a in sys.argv[1:]] or [0, 8, 16, 24, 32, 48, 64, 96, 136, 192]:
            if splits:
                os.environ["OA_NN_SPLITS"] = str(splits)
            else:
                os.environ.pop("OA_NN_SPLITS", None)
            with IcpEngine(0) as e:
                e.set_search_mode("brute")
                e.set_target(tgt)
                e.set_source(src, shard_index=shard[0], shard_count=shard[1])
                e.set_matrices(mxa, mxb)
                r = e.run(iters=12, thresh=0.5, early_exit=False)
                ms = e.search_ms()
            print("%-26s OA_NN_SPLITS=%-4s first %7.3f ms   seeded min %7.3f  median %7.3f ms" % (label, splits or "auto", ms[0], ms[1:].min(), np.median(ms[1:])), flush=True)


if __name__ == "__main__":
    main()

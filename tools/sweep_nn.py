#!/usr/bin/env python3
"""Sweep k_nn_search launch geometry on the GPU box: points per thread (OA_NN_R) x target blocks.
Prints kernel ms and fp32 TFLOP/s (8 flop per pair) for each; used to pick the defaults in oa_icp.hip."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    nt = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
    rng = np.random.default_rng(0)
    tgt = rng.uniform(-1, 1, size=(nt, 3)).astype(np.float32)
    src = rng.uniform(-1, 1, size=(ns, 3)).astype(np.float32)
    eye = np.identity(4, dtype=np.float32)
    from object_alignment_amd.engine import IcpEngine
    for filt in (1, 0):
        for R in (4, 8):
            for blocks in (4096, 8192, 16384, 32768):
                os.environ["OA_NN_R"] = str(R)
                os.environ["OA_NN_TARGET_BLOCKS"] = str(blocks)
                os.environ["OA_NN_FILTER"] = str(filt)
                with IcpEngine(0) as e:
                    e.set_search_mode("brute")
                    e.set_target(tgt)
                    e.set_source(src)
                    e.set_matrices(eye, eye)
                    cold = e.nn_search(want_output=False)[2]          # unseeded
                    e.iterate(thresh=10.0)                            # stores seeds
                    e.set_matrices(eye, eye)
                    ms = min(e.nn_search(want_output=False)[2] for _ in range(3))
                print("filter=%d R=%d target_blocks=%5d  seeded %8.3f ms  %6.1f TFLOP/s (8 flop/pair)  %.2f Gpairs/s   unseeded %8.3f ms"
                      % (filt, R, blocks, ms, 8.0 * ns * nt / ms / 1e9, ns * nt / ms / 1e6, cold), flush=True)


if __name__ == "__main__":
    main()

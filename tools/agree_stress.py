"""Early-exit loops on a multi-device context with one host thread per child (OA_MULTI_THREADS=1 + OA_MULTI_OWN_STREAMS=1):
do the threads enqueue the same number of iterations?  (docs/HISTORY.md 4.7, "the invariant")

One GPU, the device listed `--children` times; every child gets its own stream and host thread, as children on distinct
GPUs do.  GPU_MAX_HW_QUEUES is raised so that every child's stream has a hardware queue of its own (HIP multiplexes
streams over 4 by default: two children sharing one could put a gather in front of the post it waits for -- a one-GPU
artefact, distinct GPUs have a queue set each).  Prints one JSON line.

    python tools/agree_stress.py --children 4 --reps 250
    python tools/agree_stress.py --children 2 --reps 20 --lag-group 1 --lag-us 300 --agree 0     # round 3's behaviour
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--children", type=int, default=4)
    ap.add_argument("--reps", type=int, default=250)
    ap.add_argument("--fixture", default="icp_loop_bumpy_converge")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--lag-group", type=int, default=-1)
    ap.add_argument("--lag-us", type=int, default=0)
    ap.add_argument("--agree", type=int, default=1)
    ap.add_argument("--exchange", default="mailbox")
    a = ap.parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(8, 2 * a.children + 2)))
    os.environ["OA_MULTI_THREADS"] = "1"
    os.environ["OA_MULTI_OWN_STREAMS"] = "1"
    os.environ.setdefault("OA_EXCHANGE_TIMEOUT_S", "20")
    os.environ["OA_MULTI_AGREE"] = str(a.agree)
    if a.lag_group >= 0:
        os.environ["OA_FAULT_LAG_GROUP"] = str(a.lag_group)
        os.environ["OA_FAULT_LAG_US"] = str(a.lag_us)
    import numpy as np
    from object_alignment_amd.engine import IcpEngine
    g = np.load(os.path.join(ROOT, "tests", "golden", a.fixture + ".npz"), allow_pickle=False)
    done = int(g["iters_done"])
    hist, differing, wrong = {}, 0, 0
    with IcpEngine(devices=[0] * a.children, exchange=a.exchange) as eng:
        threads = eng.exchange_info()["host_threads"]
        eng.set_target(g["tgt"])
        eng.set_source(g["src"], stride=1)
        for _ in range(a.reps):
            eng.set_matrices(g["mx_align"], g["mx_base"])
            res = eng.run(iters=a.iters, thresh=0.5, target_d=0.01, use_target=True, early_exit=True)
            per_child = eng.enqueued_iterations()
            differing += len(set(per_child)) != 1
            key = ",".join(str(v) for v in sorted(set(per_child)))
            hist[key] = hist.get(key, 0) + 1
            ok = (res.iters_done == done and bool(res.converged) == bool(g["converged"])
                  and np.abs(res.matrix_world - g["final_world"]).max() <= 2.5e-7)
            wrong += not ok
    print(json.dumps({"children": a.children, "host_threads": threads, "reps": a.reps, "agree": a.agree, "lag_us": a.lag_us,
                      "iterations_executed": done, "loops_with_differing_counts": differing, "loops_with_wrong_result": wrong,
                      "enqueued_counts_histogram": hist}))


if __name__ == "__main__":
    main()

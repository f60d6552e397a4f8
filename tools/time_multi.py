#!/usr/bin/env python3
"""GPU box: what the in-library exchange of a multi-device context costs per iteration.  One GPU is all this box has,
so the device list names device 0 several times: the shards then run one after the other on one stream -- the search
time is the single-GPU time, and what is measured is the EXTRA cost per iteration of driving N children and joining
their sums (launches, mailbox posts / waits, or the ncclAllReduce of a communicator of one).
Usage: python tools/time_multi.py [n_points] [iters]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
src, tgt, mxa, mxb = synth.c3_random_pair(n)
kw = dict(iters=iters, thresh=0.5, target_d=0.01, use_target=True, early_exit=False)


def run(label, mode, env=None, **ekw):
    for k in ("OA_MAILBOX", "OA_MULTI_OWN_STREAMS", "OA_MULTI_THREADS"):
        os.environ.pop(k, None)
    os.environ.update(env or {})
    with IcpEngine(**ekw) as e:
        e.set_search_mode(mode)
        e.set_target(tgt)
        e.set_source(src)
        e.set_matrices(mxa, mxb)
        e.run(iters=5, thresh=0.5, early_exit=False)
        best = 1e9
        for _ in range(3):
            e.set_matrices(mxa, mxb)
            t0 = time.perf_counter()
            r = e.run(**kw)
            best = min(best, time.perf_counter() - t0)
        info = e.exchange_info() if e.multi else {"exchange": None, "host_threads": 1}
        enq = e.stat("enqueue_us") if e.multi else float("nan")
    print("%-58s %-5s  %8.1f us / iteration (wall), search %7.1f us, host enqueue %5.1f us / iteration and device, %d host thread(s), exchange: %s" % (
        label, mode, 1e6 * best / iters, 1e3 * r.nn_ms_total / iters, enq, info["host_threads"], info["exchange"]), flush=True)
    return best / iters


for mode in ("auto",):
    base = run("one context (oa_create)", mode, device=0)
    m1 = run("multi-device context, 1 device, mailbox (device memory)", mode, devices=[0], exchange="mailbox")
    mh = run("multi-device context, 1 device, mailbox (pinned host)", mode, env={"OA_MAILBOX": "host"}, devices=[0], exchange="mailbox")
    r1 = run("multi-device context, 1 device, RCCL", mode, devices=[0], exchange="rccl")
    for k in (2, 4, 8):
        run("%d shards on this GPU, one stream, mailbox" % k, mode, devices=[0] * k)
    for k in (2, 8):
        run("%d shards on this GPU, own streams, mailbox" % k, mode, env={"OA_MULTI_OWN_STREAMS": "1"}, devices=[0] * k)
    run("2 shards, own streams, one host thread per shard", mode, env={"OA_MULTI_OWN_STREAMS": "1", "OA_MULTI_THREADS": "1"}, devices=[0] * 2)
    run("4 shards, own streams, one host thread per shard", mode, env={"OA_MULTI_OWN_STREAMS": "1", "OA_MULTI_THREADS": "1"}, devices=[0] * 4)
    print("host-memory mailbox against device-memory mailbox on one device: %+.1f us per iteration" % (1e6 * (mh - m1)))
    print("exchange cost per iteration on one device: mailbox %+.1f us, RCCL all-reduce %+.1f us" % (1e6 * (m1 - base), 1e6 * (r1 - base)))

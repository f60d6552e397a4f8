"""world_size-2 gloo test of the sharded loop (object_alignment_amd.distributed.run_sharded) on CPU.

The loop logic, the shard partition and the 24-double sums layout are the product's; the per-shard
arithmetic is supplied by a CPU stand-in built on the oracle (tests may use the oracle as a checker/stand-in).
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShard:
    """ShardBackend: partial sums for one contiguous shard, computed with the oracle."""

    def __init__(self, orc, src, tgt, mxa, mxb, rank, world, thresh=0.5, target_d=0.01, iters=10):
        from object_alignment_amd.engine import shard_bounds
        self.orc, self.tgt = orc, tgt
        b, e = shard_bounds(len(src), rank, world)
        self.src = src[b:e]
        self.pivot = src[0].astype(np.float64)             # first selected vertex of the WHOLE selection
        self.mx1, self.mx2 = mxa.copy(), mxb
        self.thresh, self.target_d, self.iters = thresh, target_d, iters
        self.kd = orc.KDTree(tgt)

    def begin(self):
        self.n, self.converged, self.ring = 0, False, [self.target_d * 2] * 5
        self.halt = False

    def partial(self, sums):
        import torch
        s = np.zeros(24)
        if not self.halt and len(self.src):
            A, B, ds, = self.orc.make_pairs(self.src, self.tgt, self.mx1, self.mx2, self.thresh, calc_stats=False,
                                            kd=self.kd)
            a = A - self.pivot[:, None]
            b = B - self.pivot[:, None]
            s[0:3], s[3:6] = a.sum(1), b.sum(1)
            s[6:15] = (b @ a.T).reshape(9)
            s[15], s[16], s[17] = (a * a).sum(), (b * b).sum(), A.shape[1]
        sums.copy_(torch.from_numpy(s))

    def finish(self, sums):
        if self.halt:
            return
        s = sums.numpy()
        K = s[17]
        ca, cb = s[0:3] / K, s[3:6] / K
        H = s[6:15].reshape(3, 3) - K * np.outer(cb, ca)
        u, _, vh = np.linalg.svd(H)
        R = u @ vh
        if np.linalg.det(R) < 0:
            R -= np.outer(u[:, 2], vh[2, :] * 2.0)
        M = np.identity(4)
        M[:3, :3] = R
        M[:3, 3] = (cb + self.pivot) - R @ (ca + self.pivot)
        new_mat = M.astype(np.float32)
        self.mx1 = self.orc.mat4_mul(self.mx1, new_mat)
        self.ring[self.n % 5] = self.orc.vec3_length(new_mat[:3, 3])
        self.n += 1
        if all(d < self.target_d for d in self.ring):
            self.converged = True
        if self.converged or self.n >= self.iters:
            self.halt = True

    def end(self):
        return dict(matrix_world=self.mx1, iters_done=self.n, converged=self.converged)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from object_alignment_amd import synth
    from object_alignment_amd.distributed import run_sharded
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bumpy = synth.bumpy_icosphere(4)
    mxa = synth.rigid4(synth.rotation_from_rotvec([0.06, -0.05, 0.08]), [0.03, -0.02, 0.025])
    mxb = np.identity(4, dtype=np.float32)
    sums = torch.zeros(24, dtype=torch.float64)
    res = run_sharded(OracleShard(orc, bumpy, bumpy, mxa, mxb, rank, world, iters=30), 30, sums)
    np.savez(out_path % rank, mw=res["matrix_world"], n=res["iters_done"], c=res["converged"])
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_loop_world2_gloo(tmp_path, orc, golden_dir):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    assert np.array_equal(r0["mw"], r1["mw"])                       # every rank ends with the identical matrix
    g = np.load(os.path.join(golden_dir, "icp_loop_bumpy_converge.npz"))
    assert int(r0["n"]) == int(g["iters_done"]) and bool(r0["c"]) == bool(g["converged"])
    assert np.abs(r0["mw"] - g["final_world"]).max() <= 2.5e-7

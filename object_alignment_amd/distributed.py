"""One process per GPU: source cloud sharded across ranks, one all-reduce of OA_NSUMS doubles per iteration.

The path shards naturally: source points are independent given the current transform, the
only coupling is the 24 sums feeding the 3x3 solve.  Every rank keeps the whole target and the running 4x4,
all-reduces the sums over RCCL (torch.distributed backend "nccl" on ROCm) and performs the identical solve, so no
broadcast is needed.  Nothing here synchronises with the host inside the loop: nn-search, accumulate, all-reduce
and solve of all iterations are enqueued back to back.

`ShardBackend` is the interface the loop drives; `EngineShard` is the real one (liboa_icp.so).  Tests drive the
same loop with a CPU stand-in over gloo.
"""
from __future__ import annotations

from typing import Protocol

from . import _capi as capi


class ShardBackend(Protocol):
    def begin(self) -> None: ...
    def partial(self, sums) -> None: ...      # write this shard's OA_NSUMS partial sums into `sums` (a tensor)
    def finish(self, sums) -> None: ...       # consume the all-reduced sums: solve + update
    def end(self): ...


class EngineShard:
    """ShardBackend over an IcpEngine whose source was set with (shard_index=rank, shard_count=world)."""

    def __init__(self, engine, **loop_kwargs):
        self.engine = engine
        self.kw = loop_kwargs

    def begin(self):
        import torch
        self.engine.set_stream(torch.cuda.current_stream().cuda_stream)
        self.engine.run_begin(**self.kw)

    def partial(self, sums):
        self.engine.iter_partial(sums.data_ptr())

    def finish(self, sums):
        self.engine.iter_finish(sums.data_ptr())

    def end(self):
        return self.engine.run_end()


def new_sums_tensor(device):
    import torch
    return torch.zeros(capi.OA_NSUMS, dtype=torch.float64, device=device)


def run_sharded(backend: ShardBackend, iters: int, sums, group=None, world_size: int | None = None,
                force_collective: bool = False):
    """Drive `iters` iterations: partial -> all_reduce(SUM) -> finish.  Returns backend.end().
    force_collective: issue the all-reduce even in a world of one (how the RCCL call itself is exercised on a
    one-GPU box)."""
    import torch.distributed as dist
    if world_size is None:
        world_size = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    reduce = world_size > 1 or (force_collective and dist.is_available() and dist.is_initialized())
    backend.begin()
    for _ in range(int(iters)):
        backend.partial(sums)
        if reduce:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        backend.finish(sums)
    return backend.end()

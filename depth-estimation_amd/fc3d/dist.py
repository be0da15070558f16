"""Data-parallel plumbing: one process per GPU, RCCL through torch.distributed
(backend "nccl" on ROCm), gloo for CPU tests.

The image formation shards over the batch with no communication at all; the
only collective of a training step is the gradient all-reduce of the two
U-Nets (17.27 M fp32 parameters).  FlatGradBucket keeps every .grad as a view
into ONE contiguous buffer, so that all-reduce is a single call on a single
69 MB message (the right shape for xGMI's point-to-point links: few, large
transfers) with no flatten/unflatten copies."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None, device: torch.device | None = None):
    """(rank, world, local_rank); initialises the process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, world, local


def max_over_ranks(x: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_seed(base_seed: int, rank: int) -> int:
    """Each rank simulates its own scenes: disjoint seed ranges per rank."""
    return base_seed + 100 * rank


class FlatGradBucket:
    """All parameters' gradients live in one flat buffer; average() all-reduces it."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.handle = None

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4

    def zero(self):
        self.flat.zero_()

    def average(self, async_op: bool = False):
        """Mean over ranks (sum all-reduce, then one scale).  With async_op the call returns
        at once on the communication stream; wait() before the optimizer step."""
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        self.handle = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        if not async_op:
            self.flat.div_(dist.get_world_size())
            self.handle = None

    def wait(self):
        if self.handle is not None:
            self.handle.wait()
            self.flat.div_(dist.get_world_size())
            self.handle = None

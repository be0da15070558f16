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


def broadcast_module_state(modules, src: int = 0) -> None:
    """Parameters and buffers (BatchNorm statistics) of `modules` from rank `src` to every rank, so
    that data-parallel replicas start equal whatever each process's seed was.  No-op outside a
    process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    with torch.no_grad():
        for m in modules:
            for t in list(m.parameters()) + list(m.buffers()):
                dist.broadcast(t.data, src)


class _SumOverRanks(torch.autograd.Function):
    """Sum of a tensor over ranks whose backward passes the gradient through unchanged: every rank
    then differentiates the same global quantity w.r.t. its own share, and the SUM of the ranks'
    parameter gradients is the gradient of that quantity."""

    @staticmethod
    def forward(ctx, x):
        y = x.detach().clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM)
        return y

    @staticmethod
    def backward(ctx, g):
        return g


def sum_over_ranks(x: torch.Tensor) -> torch.Tensor:
    """See _SumOverRanks; identity outside a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return x
    return _SumOverRanks.apply(x)


class FlatGradBucket:
    """All parameters' gradients live in one flat buffer, cut into `n_buckets` contiguous ranges.

    overlap=False: average() all-reduces the whole buffer in one call after backward.
    overlap=True: every parameter reports when its gradient has been accumulated; as soon as all
    parameters of a range have reported, that range's all-reduce is started on the communication
    stream while autograd keeps working on earlier layers.  Ranges are filled from the back of the
    buffer, the order in which backward reaches the parameters.  wait() then blocks on the pending
    transfers and scales.  Few large messages suit the point-to-point xGMI links of an 8-GPU node:
    the default is three ranges of about 23 MB for the two U-Nets."""

    def __init__(self, params, n_buckets: int = 1, overlap: bool = False, average: bool = True):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.mean = average
        offs, off = [], 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            offs.append(off)
            off += p.numel()
        self.handle = None
        # bucket k covers parameters [first_k, first_{k+1}): cut where the running size passes k/n_buckets
        n_buckets = max(1, min(int(n_buckets), len(self.params)))
        cuts, k = [0], 1
        for i, o in enumerate(offs):
            if k < n_buckets and o >= k * n / n_buckets:
                cuts.append(i)
                k += 1
        cuts.append(len(self.params))
        self.ranges = []                      # (start element, end element, number of parameters)
        self._bucket_of = {}
        for b in range(len(cuts) - 1):
            lo, hi = cuts[b], cuts[b + 1]
            if hi == lo:
                continue
            end = offs[hi] if hi < len(offs) else n
            self.ranges.append((offs[lo], end, hi - lo))
            for i in range(lo, hi):
                self._bucket_of[i] = len(self.ranges) - 1
        self.overlap = bool(overlap)
        self._pending = [0] * len(self.ranges)
        self._handles = []
        self._hooks = []
        if self.overlap:
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(self._bucket_of[i])))

    def _active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _make_hook(self, b):
        def hook(_param):
            if self._pending[b] <= 0:
                # a second backward without zero(): this range's transfer is already in flight (or done) and
                # would be reduced again on top of the summed values
                raise RuntimeError("FlatGradBucket(overlap=True): a gradient arrived for a range whose all-reduce "
                                   "has been started already; call zero() before every backward (gradient "
                                   "accumulation over several backward passes needs overlap=False)")
            self._pending[b] -= 1
            if self._pending[b] == 0 and self._active():
                lo, hi, _ = self.ranges[b]
                self._handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        return hook

    def close(self):
        """Removes the autograd hooks (the gradients stay views of the flat buffer)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4

    def zero(self):
        """Clears the gradients and re-arms the per-range counters; call before every backward."""
        self.flat.zero_()
        self._pending = [r[2] for r in self.ranges]
        self._handles = []

    def average(self, async_op: bool = False):
        """Mean (or sum, with average=False) over ranks.  With overlap=True the transfers were
        started during backward and this only starts what is left (ranges with a parameter that
        received no gradient); otherwise one all-reduce per range is issued now.  With async_op
        the call returns at once; wait() before the optimizer step."""
        if not self._active():
            return
        for b, (lo, hi, _) in enumerate(self.ranges):
            if self.overlap and self._pending[b] == 0:
                continue                                          # already in flight
            self._pending[b] = 0
            self._handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        if not async_op:
            self.wait()

    def wait(self):
        if not self._handles:
            return
        for h in self._handles:
            h.wait()
        self._handles = []
        if self.mean:
            self.flat.div_(dist.get_world_size())

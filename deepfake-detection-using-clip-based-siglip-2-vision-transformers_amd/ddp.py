"""Data-parallel gradient exchange for the encoder path: one process per GPU, RCCL over xGMI
(``torch.distributed`` backend ``"nccl"`` is RCCL on ROCm), gradient buckets all-reduced while the backward
of the blocks below is still running.

The reference has no working distributed path (an un-initialised FSDP wrapper only); the
exchange pattern here is designed for the MI355X node: images are independent, so the only
collective is the sum-all-reduce of parameter gradients.  The encoder's backward hands over one flat fp32
bucket per transformer block (≈61 MB for so400m) the moment that block's gradients are complete
(``encoder._EncoderFn.backward``); ``reduce_bucket`` launches the all-reduce asynchronously on the process
group's own stream, and ``finish`` makes the compute stream wait for all of them and applies the 1/world
average.  Buckets are large (tens of MB) so each ring step moves big messages over the point-to-point xGMI
links rather than many small ones.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, process_group: Optional["dist.ProcessGroup"] = None, average: bool = True):
        self.pg = process_group
        self.average = average
        self._pending: list[tuple[torch.Tensor, object]] = []

    # ---- wiring ------------------------------------------------------------------------------------------
    def attach(self, encoder_module) -> "GradBucketReducer":
        """Make ``encoder_module`` (a ``SiglipVisionModelHIP``) call back into this reducer per bucket."""
        encoder_module._grad_reducer = self
        return self

    def world_size(self) -> int:
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.pg)

    # ---- called from the encoder's backward -----------------------------------------------------------------
    def reduce_bucket(self, flat: torch.Tensor) -> None:
        if self.world_size() == 1:
            return
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._pending.append((flat, work))

    def finish(self) -> None:
        """Order every outstanding all-reduce before whatever the current stream does next (no host sync on
        the NCCL/RCCL backend) and turn sums into means."""
        if not self._pending:
            return
        scale = 1.0 / self.world_size()
        for flat, work in self._pending:
            work.wait()
            if self.average:
                flat.mul_(scale)
        self._pending.clear()

    # ---- everything outside the encoder (heads, decoder): one bucket after backward ---------------------------
    def reduce_grads(self, params: Iterable[torch.nn.Parameter]) -> None:
        ps = [p for p in params if p.grad is not None]
        if not ps or self.world_size() == 1:
            return
        flat = torch.cat([p.grad.reshape(-1).float() for p in ps])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg)
        if self.average:
            flat.mul_(1.0 / self.world_size())
        off = 0
        for p in ps:
            n = p.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Rank ``src``'s parameters and buffers to every rank (one flat message per dtype)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    by_dtype: dict[torch.dtype, list[torch.Tensor]] = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for ts in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


def shard_batch(global_batch: int, rank: int, world: int) -> tuple[int, int]:
    """[begin, end) of this rank's images; the global batch is split as evenly as possible and whole clips /
    images never straddle ranks."""
    base, rem = divmod(global_batch, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)

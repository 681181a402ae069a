"""Data-parallel gradient exchange for the encoder path: one process per GPU, RCCL over xGMI
(``torch.distributed`` backend ``"nccl"`` is RCCL on ROCm), gradients all-reduced while the backward of the
blocks below is still running.

The reference has no working distributed path (an un-initialised FSDP wrapper only); the
exchange pattern here is designed for the MI355X node: images are independent, so the only
collective is the sum of parameter gradients.

* **Few, large messages.**  The encoder's backward produces one flat fp32 gradient bucket per transformer block
  (61 MB for so400m); consecutive blocks share one allocation (``SiglipVisionModelHIP._grad_buckets`` carves a *chunk* of
  ``ceil(groups / max_buckets)`` blocks out of one tensor), and the chunk is handed to ``reduce_bucket`` the moment its
  last block is complete: 8 collectives of ≈0.2 GB per step for so400m instead of 29 of 61 MB.  xGMI is point-to-point
  (7 links × ≈153 GB/s per GPU), so a ring step is bound by one link: big messages amortise the per-step latency.
* **Wire format.**  ``wire="fp32"`` (default): one in-place sum all-reduce per chunk, exact.  ``wire="bf16"``: the
  full-mesh form of SURVEY.md §8e — every rank sends shard *j* of its bf16-rounded gradients straight to rank *j*
  (``all_to_all_single``: 7 peers, 7 different links), sums the ``world`` shards it received **in fp32**, and the reduced
  shard goes back as bf16 (``all_gather_into_tensor``): half the bytes of the fp32 ring, one rounding of the inputs and one
  of the mean, no bf16 accumulation chain.
* **1/world.**  ``average=True`` scales all chunks with ONE ``torch._foreach_mul_`` in ``finish()``;
  ``average="defer"`` leaves sums in ``.grad`` and the consumer applies the factor: ``FusedAdamW(grad_scale=1/world)``
  folds it into the clip coefficient it already multiplies every gradient by (no extra pass over the gradients).
* ``no_sync()`` suppresses the exchange for gradient-accumulation micro-steps (the reference accumulates:
  Siglip2sidafrozen.py:1390 ``loss / current_grad_accum``).  The micro-step that leaves the context must exchange the
  ACCUMULATED gradient, not just its own contribution: when a trainable encoder parameter already carries a ``.grad`` the
  in-backward hand-off is skipped and ``reduce_accumulated`` runs from an autograd-engine callback after every
  ``AccumulateGrad`` of the pass has fired, one collective per chunk over the memory ``.grad`` lives in.
* ``reduce_grads`` (everything outside the encoder: decoder, heads) is asynchronous and completed by ``finish()`` as well.
* Chunk sizes taper (``SiglipVisionModelHIP._bucket_layout``): the chunk that completes last holds one block (+ the
  embeddings), so the exchange that cannot overlap anything is 67 MB, not 244 MB.
* ``all_gather_eval`` collects per-rank logits / labels for epoch metrics (Siglip2sidafrozen.py:1424-1548).
* RCCL's channel kernels take CUs from GEMMs that occupy every CU; ``NCCL_MAX_NCHANNELS`` (read by RCCL at
  ``init_process_group``; ``bench.py --rccl-channels``) bounds how many.

Nothing here has been timed on more than one GPU by the builder (the 8-GPU node is the driver's): correctness is covered by
world-2 gloo tests, the scaling curve is whatever the driver's SCALE run measures.
"""
from __future__ import annotations

import contextlib
from typing import Iterable, Optional, Union

import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, process_group: Optional["dist.ProcessGroup"] = None, average: Union[bool, str] = True,
                 wire: str = "fp32", max_buckets: int = 8):
        if wire not in ("fp32", "bf16"):
            raise ValueError("wire must be 'fp32' or 'bf16'")
        if average not in (True, False, "defer"):
            raise ValueError("average must be True, False or 'defer'")
        self.pg = process_group
        self.average = average
        self.wire = wire
        self.max_buckets = int(max_buckets)
        self._sync = True
        self._pending: list = []          # (kind, flat, work, extra)
        self._heads: list = []            # (flat, params) of reduce_grads calls in flight
        self.collectives_issued = 0       # statistics for tests / bench
        self.time_exposed = False         # bench: HIP events around finish()'s waits = communication nothing overlapped
        self._exposed: list = []

    # ---- wiring ------------------------------------------------------------------------------------------
    def attach(self, encoder_module) -> "GradBucketReducer":
        """Make ``encoder_module`` (a ``SiglipVisionModelHIP``) call back into this reducer per gradient chunk."""
        encoder_module._grad_reducer = self
        encoder_module._bucket_cache = {}
        return self

    def world_size(self) -> int:
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.pg)

    @property
    def grad_scale(self) -> float:
        """What the consumer must multiply gradients by (1/world when average='defer', else 1)."""
        return 1.0 / self.world_size() if self.average == "defer" else 1.0

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient-accumulation micro-steps: backward runs without any collective; the first backward after the context
        exchanges the ACCUMULATED ``.grad`` (``reduce_accumulated``, called by the encoder's autograd formula), and
        ``reduce_grads`` does the same for everything outside the encoder."""
        prev, self._sync = self._sync, False
        try:
            yield self
        finally:
            self._sync = prev

    def syncing(self) -> bool:
        """Will a chunk handed to ``reduce_bucket`` really be exchanged?"""
        return self._sync and self.world_size() > 1

    def exposed_ms(self) -> list:
        """Milliseconds the launch stream spent blocked in each ``finish()`` since the last call (``time_exposed``)."""
        out = [a.elapsed_time(b) for a, b in self._exposed]
        self._exposed.clear()
        return out

    # ---- called from the encoder's backward -----------------------------------------------------------------
    def reduce_bucket(self, flat: torch.Tensor) -> None:
        """Start the exchange of one flat fp32 gradient chunk (asynchronous; completed by ``finish``)."""
        world = self.world_size()
        if world == 1 or not self._sync:
            return
        self.collectives_issued += 1
        if self.wire == "fp32":
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._pending.append(("ar", flat, work, None))
            return
        # bf16 wire, fp32 accumulate: all-to-all of bf16 shards -> local fp32 sum -> all-gather of the bf16 result
        n = flat.numel()
        shard = (n + world - 1) // world
        send = torch.zeros(world * shard, dtype=torch.bfloat16, device=flat.device)
        send[:n].copy_(flat)
        recv = torch.empty_like(send)
        work = dist.all_to_all_single(recv, send, group=self.pg, async_op=True)
        self._pending.append(("a2a", flat, work, (send, recv, shard)))

    def finish(self) -> None:
        """Order every outstanding collective before whatever the current stream does next (no host sync on the
        NCCL/RCCL backend) and, with average=True, turn sums into means with one multi-tensor launch."""
        if not self._pending and not self._heads:
            return
        world = self.world_size()
        ev = None
        if self.time_exposed and torch.cuda.is_available():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        done: list[torch.Tensor] = []
        gathers = []
        for kind, flat, work, extra in self._pending:
            work.wait()
            if kind == "ar":
                done.append(flat)
                continue
            send, recv, shard = extra
            red = recv.view(world, shard).float().sum(0).to(torch.bfloat16)       # fp32 accumulation of the world shards
            out = send                                                                # reuse as the gather destination
            gathers.append((flat, out, dist.all_gather_into_tensor(out, red, group=self.pg, async_op=True), red))
        for flat, out, work, _red in gathers:
            work.wait()
            flat.copy_(out[:flat.numel()])
            done.append(flat)
        self._pending.clear()
        for flat, params, work in self._heads:
            work.wait()
            done.append(flat)
        if ev is not None:
            ev[1].record()
            self._exposed.append(ev)
        if self.average is True and done:
            torch._foreach_mul_(done, 1.0 / world)
        for flat, params, _ in self._heads:
            off = 0
            for p in params:
                n = p.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        self._heads.clear()

    # ---- gradient accumulation: the micro-step after no_sync() exchanges what .grad holds ---------------------
    def reduce_accumulated(self, chunks) -> None:
        """``chunks`` = [(total_elems, [(param, offset, numel)])] in the encoder's chunk layout.  Called once autograd has
        accumulated this pass into ``.grad``.  When a chunk's gradients still sit at their offsets inside one allocation
        (the flat tensor an earlier micro-step's backward wrote: ``AccumulateGrad`` adopts the views and later adds in
        place) that memory is exchanged directly, one collective per chunk as in the overlapped case; otherwise the chunk
        goes through a packed copy.  Completed (and averaged) by the ``finish()`` at the end."""
        if not self.syncing():
            return
        for total, entries in chunks:
            entries = [(p, off, n) for p, off, n in entries if p.grad is not None]
            if not entries:
                continue
            flat = _common_flat(entries, total)
            if flat is not None:
                self.reduce_bucket(flat)
            else:
                self.reduce_grads([p for p, _, _ in entries], async_op=True)
        self.finish()

    # ---- everything outside the encoder (heads, decoder): one bucket after backward ---------------------------
    def reduce_grads(self, params: Iterable[torch.nn.Parameter], async_op: bool = False) -> None:
        """Sum (and average) the ``.grad`` of parameters outside the encoder as ONE flat message.  ``async_op=True``
        returns right after launching the collective; ``finish()`` completes it and writes the results back."""
        ps = [p for p in params if p.grad is not None]
        if not ps or self.world_size() == 1 or not self._sync:
            return
        flat = torch.cat([p.grad.reshape(-1).float() for p in ps])
        self.collectives_issued += 1
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._heads.append((flat, ps, work))
        if not async_op:
            self.finish()


def _common_flat(entries, total) -> Optional[torch.Tensor]:
    """The fp32 tensor of ``total`` elements whose [off, off+n) ranges ARE the ``.grad`` of ``entries``' parameters, or
    None when the gradients do not sit in one allocation at those offsets."""
    p0, off0, _ = entries[0]
    g0 = p0.grad
    if g0.dtype != torch.float32:
        return None
    st = g0.untyped_storage()
    start = g0.data_ptr() - 4 * off0
    if start < st.data_ptr() or start + 4 * total > st.data_ptr() + st.nbytes():
        return None
    for p, off, n in entries:
        g = p.grad
        if (g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != n or g.data_ptr() != start + 4 * off
                or g.untyped_storage().data_ptr() != st.data_ptr()):
            return None
    return torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, (start - st.data_ptr()) // 4, (total,))


def all_gather_eval(*tensors: torch.Tensor, group=None):
    """Validation-loop collective (the reference's validate() computes epoch metrics over every sample,
    Siglip2sidafrozen.py:1424-1548): concatenate each rank's per-sample tensors (logits, labels, ...) along dim 0, in rank
    order, on every rank.  Ranks may hold different numbers of samples (the last shard of ``shard_batch``); trailing
    dimensions must agree.  Returns one tensor per argument (a single tensor for a single argument)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tensors[0] if len(tensors) == 1 else tuple(tensors)
    world = dist.get_world_size(group)
    outs = []
    dev = tensors[0].device
    counts = torch.tensor([t.shape[0] for t in tensors], dtype=torch.int64, device=dev)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    for ti, t in enumerate(tensors):
        ns = [int(c[ti]) for c in all_counts]
        cap = max(ns)
        pad = t.new_zeros((cap, *t.shape[1:]))
        pad[:t.shape[0]] = t
        got = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(got, pad.contiguous(), group=group)
        outs.append(torch.cat([g[:n] for g, n in zip(got, ns)], dim=0))
    return outs[0] if len(outs) == 1 else tuple(outs)


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

"""Optimizer step tail on the GPU: the reference's per-step pair

    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)     Siglip2sidafrozen.py:1396
    optimizer.step()        # torch.optim.AdamW(model.parameters(), lr=args.lr, weight_decay=args.wd)   :1241-1244,1398

as two HIP launches over a table of tensors (``csrc/optimizer.hip``), without the host round trip the reference pays
for the gradient norm (``.item()``, Siglip2sidafrozen.py:1391).  ``FusedAdamW`` keeps ``torch.optim.AdamW``'s
constructor arguments, parameter groups and ``state_dict`` layout (``exp_avg`` / ``exp_avg_sq`` / ``step``), so a
checkpoint written by either loads into the other.  fp32 CUDA parameters only; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as _lib


def _is_dense(t: torch.Tensor) -> bool:
    """Storage holds exactly the tensor's elements, in some permutation (model.to(memory_format=channels_last) gives the
    patch convolution's weight such a layout, Siglip2sidafrozen.py:1191)."""
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


class FusedAdamW(torch.optim.Optimizer):
    """``torch.optim.AdamW`` semantics (decoupled weight decay, bias correction, same operation order) in one launch
    for all tensors.  ``max_grad_norm`` > 0 folds ``clip_grad_norm_(all parameters, max_grad_norm)`` into the step: the
    gradients themselves are left untouched, the clip coefficient is applied as they are read.  After ``step()``,
    ``last_grad_norm`` is a 0-d CUDA tensor holding the pre-clip global norm (no synchronisation).

    In the same pass over each parameter the kernel can also write what the update invalidates (kernel work-list k11):

    * ``attach_encoder(enc)``: the bf16 (or strict-fp32) weight shadows of a ``SiglipVisionModelHIP`` and their
      transposes, so the next forward does no re-cast (the reference's autocast casts every weight every step,
      Siglip2sidafrozen.py:1375);
    * ``attach_ema(ema)``: ``ExponentialMovingAverage.update()`` (cifake_binary_classifier.py:222-225);
    * ``grad_scale``: gradients are rank SUMS and this is 1/world (``GradBucketReducer(average="defer")``): folded into
      the clip coefficient, no pass of its own.

    Learning rate and weight decay travel with each launch (per parameter group), so a per-step LR scheduler does not
    re-upload the device table."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None,
                 grad_scale: float = 1.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) > 16:
            raise ValueError("FusedAdamW supports at most 16 parameter groups")
        self.max_grad_norm = max_grad_norm
        self.grad_scale = float(grad_scale)
        self.last_grad_norm = None
        self._table_key = None
        self._aux_key = None
        self._plans = {}
        self._bufs = {}
        self._encoders = []
        self._ema = None

    def attach_encoder(self, encoder) -> "FusedAdamW":
        """Write ``encoder``'s weight shadows in the AdamW pass (it must own some of this optimizer's parameters)."""
        import weakref
        self._encoders.append(weakref.ref(encoder))
        self._aux_key = None
        return self

    def attach_ema(self, ema) -> "FusedAdamW":
        """Fold ``ema.update()`` into the step (``ema`` = ``ExponentialMovingAverage``; do not call update() yourself)."""
        self._ema = ema
        self._aux_key = None
        return self

    # -- helpers -----------------------------------------------------------------------------------------------
    def _collect(self):
        ents = []
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize"):  # e.g. after load_state_dict of a torch AdamW state
                raise RuntimeError("FusedAdamW implements plain AdamW only (amsgrad / maximize are not supported)")
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise RuntimeError("FusedAdamW handles fp32 CUDA parameters and gradients only (no CPU path)")
                if not _is_dense(p):
                    raise RuntimeError("FusedAdamW needs dense (contiguous or channels_last) parameters")
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdamW does not support sparse gradients")
                if p.grad.stride() != p.stride():   # the update is elementwise over storage: same layout everywhere
                    p.grad = torch.empty_like(p).copy_(p.grad)
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                ents.append((p, st, group, gi))
        return ents

    def _device_tables(self, lib, ents, dev):
        """(table, aux, chunk counts): pointers only — hyper-parameters are launch arguments."""
        key = tuple((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), gi)
                    for p, st, _, gi in ents)
        encs = [e() for e in self._encoders]
        encs = [e for e in encs if e is not None and e._shadow is not None and e._shadow.device == dev]
        ema_ptrs = ()
        if self._ema is not None:
            by_param = {id(p): self._ema.shadow[n] for n, p in self._ema.model.named_parameters()
                        if n in self._ema.shadow}
            ema_ptrs = tuple(by_param[id(p)].data_ptr() if id(p) in by_param else 0 for p, _, _, _ in ents)
        aux_key = (key, tuple((id(e), e._shadow.data_ptr(), id(e._weights_struct)) for e in encs), ema_ptrs)
        if key != self._table_key or aux_key != self._aux_key:
            n = len(ents)
            arr = (_lib.SglAdamwTensor * n)()
            aux = (_lib.SglAdamwAux * n)()
            for i, (e, k) in enumerate(zip(arr, key)):
                e.p, e.g, e.m, e.v, e.n = k[:5]
                e.lr, e.weight_decay = 0.0, 0.0
                aux[i].group = k[5]
                if ema_ptrs and ema_ptrs[i]:
                    sh = by_param[id(ents[i][0])]
                    if sh.dtype != torch.float32 or sh.stride() != ents[i][0].stride():
                        raise RuntimeError("EMA shadows must be fp32 with the parameter's layout")
                    aux[i].ema = ema_ptrs[i]
            bound = set()
            for enc in encs:
                # only row-major masters can be tiled (channels_last patch weights keep the encoder's own re-cast)
                lib.sgl_adamw_bind_shadows(enc._ctx, C.byref(enc._weights_struct), enc._shadow.data_ptr(), arr, aux, n)
            chunks = []
            for i, (p, _, _, _) in enumerate(ents):
                if (aux[i].dst or aux[i].dst_t) and not p.is_contiguous():
                    aux[i].dst = aux[i].dst_t = None
                if aux[i].dst or aux[i].dst_t:
                    chunks.append(((aux[i].rows + 63) // 64) * ((aux[i].cols + 63) // 64) * 4096)
                    bound.add(p.data_ptr())
                else:
                    chunks.append(p.numel())
                    if aux[i].dst_f32:
                        bound.add(p.data_ptr())
            self._bufs["table"] = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            self._bufs["aux"] = torch.frombuffer(bytearray(bytes(aux)), dtype=torch.uint8).to(dev)
            self._bufs["chunks"] = tuple(chunks)
            self._bufs["bound"] = bound
            self._table_key, self._aux_key = key, aux_key
        return self._bufs["table"], self._bufs["aux"], self._bufs["chunks"], encs

    def _plan(self, lib, numel, members, dev):
        """Block map over the tensors listed in ``members`` (indices into the table), cached."""
        key = (numel, members)
        hit = self._plans.get(key)
        if hit is None:
            sub = (C.c_uint64 * len(members))(*[numel[i] for i in members])
            nb = lib.sgl_adamw_plan(sub, len(members), None, 0)
            bm = (C.c_int32 * (2 * max(nb, 1)))()
            lib.sgl_adamw_plan(sub, len(members), bm, nb)
            pairs = torch.frombuffer(bytearray(bytes(bm)), dtype=torch.int32).view(-1, 2).clone()
            pairs[:, 0] = torch.tensor(members, dtype=torch.int32)[pairs[:, 0].long()] if nb else 0
            hit = (pairs.contiguous().view(-1).to(dev), nb)
            if len(self._plans) > 16:
                self._plans.clear()
            self._plans[key] = hit
        return hit

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ents = self._collect()
        if not ents:
            return loss
        lib = _lib.load()
        dev = ents[0][0].device
        if any(p.device != dev for p, _, _, _ in ents):
            raise RuntimeError("FusedAdamW: all parameters must live on one device")
        table, aux, chunks, encs = self._device_tables(lib, ents, dev)
        numel = tuple(p.numel() for p, _, _, _ in ents)
        stream = torch.cuda.current_stream(dev).cuda_stream
        in_sync = [enc._units_in_sync() for enc in encs]
        # one AdamW launch per distinct (betas, eps, step): a single one for every trainer of the reference (one param
        # group, every trainable tensor receives a gradient every step); torch tracks the step per tensor, so do we
        launches = {}
        for i, (_, st, g, _) in enumerate(ents):
            launches.setdefault((g["betas"][0], g["betas"][1], g["eps"], int(st["step"].item())), []).append(i)
        hyper = (C.c_float * (2 * len(self.param_groups)))()
        for gi, g in enumerate(self.param_groups):
            hyper[2 * gi], hyper[2 * gi + 1] = float(g["lr"]), float(g["weight_decay"])
        norm_ptr = None
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        with torch.cuda.device(dev):
            if clip or self.grad_scale != 1.0:
                bmap, nb = self._plan(lib, numel, tuple(range(len(ents))), dev)
                if "norm" not in self._bufs or self._bufs["partials"].numel() < max(nb, 1):
                    self._bufs["partials"] = torch.empty(max(nb, 1), device=dev, dtype=torch.float32)
                    self._bufs["norm"] = torch.zeros(2, device=dev, dtype=torch.float32)
                _lib.check(lib.sgl_op_grad_norm_scaled(table.data_ptr(), bmap.data_ptr(), nb,
                                                       float(self.max_grad_norm) if clip else 0.0, self.grad_scale,
                                                       self._bufs["partials"].data_ptr(), self._bufs["norm"].data_ptr(),
                                                       stream), "sgl_op_grad_norm_scaled")
                norm_ptr = self._bufs["norm"].data_ptr()
                self.last_grad_norm = self._bufs["norm"][0]
            decay = float(self._ema.decay) if self._ema is not None else 0.0
            for (beta1, beta2, eps, step0), members in launches.items():
                bmap, nb = self._plan(lib, chunks, tuple(members), dev)
                _lib.check(lib.sgl_op_adamw_ex(table.data_ptr(), aux.data_ptr(), bmap.data_ptr(), nb, float(beta1),
                                               float(beta2), float(eps), step0 + 1, norm_ptr, hyper,
                                               len(self.param_groups), decay, stream), "sgl_op_adamw_ex")
        for _, st, _, _ in ents:
            st["step"] += 1
        # the kernel wrote the parameters behind autograd's back: bump their version counters, which is what everything
        # that caches on `p._version` keys on (the encoder's bf16 weight shadows, saved-tensor checks)
        torch.autograd.graph.increment_version([p for p, _, _, _ in ents])
        for enc, ok in zip(encs, in_sync):
            enc._adopt_written_shadows(ok, self._bufs["bound"])
        return loss


def global_grad_norm(parameters) -> torch.Tensor:
    """||grad||_2 over ``parameters`` as a 0-d CUDA tensor, computed by the same two launches as the fused clip and
    without a host synchronisation (what the reference logs per step, Siglip2sidafrozen.py:1386-1391)."""
    params = [p for p in parameters if p.grad is not None]
    if not params:
        raise ValueError("no gradients")
    lib = _lib.load()
    dev = params[0].device
    arr = (_lib.SglAdamwTensor * len(params))()
    for e, p in zip(arr, params):
        if not p.grad.is_cuda or p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
            raise RuntimeError("global_grad_norm handles contiguous fp32 CUDA gradients only")
        e.p, e.g, e.m, e.v, e.n, e.lr, e.weight_decay = 0, p.grad.data_ptr(), 0, 0, p.numel(), 0.0, 0.0
    numel = (C.c_uint64 * len(params))(*[p.numel() for p in params])
    nb = lib.sgl_adamw_plan(numel, len(params), None, 0)
    bm = (C.c_int32 * (2 * max(nb, 1)))()
    lib.sgl_adamw_plan(numel, len(params), bm, nb)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    bmap = torch.frombuffer(bytearray(bytes(bm)), dtype=torch.int32).to(dev)
    partials = torch.empty(max(nb, 1), device=dev, dtype=torch.float32)
    out = torch.zeros(2, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.check(lib.sgl_op_grad_norm(table.data_ptr(), bmap.data_ptr(), nb, 0.0, partials.data_ptr(), out.data_ptr(),
                                        torch.cuda.current_stream(dev).cuda_stream), "sgl_op_grad_norm")
    return out[0]


class ExponentialMovingAverage:
    """Weight EMA with the surface of the CiFake trainer's helper (cifake_binary_classifier.py:211-236: ``shadow`` — what
    it stores as ``checkpoint['ema_state_dict']`` — ``update()``, ``apply_shadow()``, ``restore()``), laid out for the GPU:
    all averages live in ONE flat fp32 buffer (``shadow[name]`` are views into it), ``update()`` is one HIP launch over
    every trainable tensor (12 B/parameter) or, after ``FusedAdamW.attach_ema(self)``, no launch at all (the AdamW kernel
    updates the average while the new parameter is still in registers), and the eval-time swap exchanges storage
    pointers instead of copying weights."""

    def __init__(self, model, decay=0.9999):
        self.model = model
        self.decay = decay
        tracked = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        sizes = [(p.numel() + 3) // 4 * 4 for _, p in tracked]          # 16-byte aligned slots
        dev = tracked[0][1].device if tracked else "cpu"
        self._flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        self.shadow, off = {}, 0
        for (n, p), sz in zip(tracked, sizes):
            view = torch.as_strided(self._flat, p.shape, p.stride(), off) if _is_dense(p) else \
                self._flat[off:off + p.numel()].view(p.shape)
            view.copy_(p.detach())
            self.shadow[n] = view
            off += sz
        self._live = None      # name -> the training weights' storage while the averages are swapped in
        self._key = None
        self._bufs = None

    def _plan(self, lib):
        ents = [(p, self.shadow[n]) for n, p in self.model.named_parameters() if n in self.shadow]
        key = tuple((p.data_ptr(), s.data_ptr(), p.numel()) for p, s in ents)
        if key != self._key:
            for p, s in ents:
                if not (p.is_cuda and s.is_cuda and p.dtype == torch.float32 and _is_dense(p)
                        and s.stride() == p.stride()):
                    raise RuntimeError("ExponentialMovingAverage handles dense fp32 CUDA parameters only")
            dev = ents[0][0].device
            arr = (_lib.SglAdamwTensor * len(ents))()
            for e, (pp, sp, n) in zip(arr, key):
                e.p, e.g, e.m, e.v, e.n, e.lr, e.weight_decay = pp, 0, sp, 0, n, 0.0, 0.0
            numel = (C.c_uint64 * len(ents))(*[k[2] for k in key])
            nb = lib.sgl_adamw_plan(numel, len(ents), None, 0)
            bm = (C.c_int32 * (2 * max(nb, 1)))()
            lib.sgl_adamw_plan(numel, len(ents), bm, nb)
            self._bufs = (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev),
                          torch.frombuffer(bytearray(bytes(bm)), dtype=torch.int32).to(dev), nb, dev)
            self._key = key
        return self._bufs

    @torch.no_grad()
    def update(self):
        """average <- average*decay + weight*(1-decay) for every tracked tensor, one launch."""
        if not self.shadow:
            return
        lib = _lib.load()
        table, bmap, nb, dev = self._plan(lib)
        with torch.cuda.device(dev):
            _lib.check(lib.sgl_op_ema(table.data_ptr(), bmap.data_ptr(), nb, float(self.decay),
                                      torch.cuda.current_stream(dev).cuda_stream), "sgl_op_ema")

    def apply_shadow(self):
        """Evaluate with the averaged weights: every tracked parameter points at its average until ``restore()``."""
        if self._live is not None:
            raise RuntimeError("apply_shadow() called twice without restore()")
        self._live = {}
        for n, p in self.model.named_parameters():
            if n in self.shadow:
                self._live[n] = p.data
                p.data = self.shadow[n]

    def restore(self):
        if self._live is None:
            return
        for n, p in self.model.named_parameters():
            if n in self._live:
                p.data = self._live[n]
        self._live = None

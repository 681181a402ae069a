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
    ``last_grad_norm`` is a 0-d CUDA tensor holding the pre-clip global norm (no synchronisation)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = max_grad_norm
        self.last_grad_norm = None
        self._table_key = None
        self._plans = {}
        self._bufs = {}

    # -- helpers -----------------------------------------------------------------------------------------------
    def _collect(self):
        ents = []
        for group in self.param_groups:
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
                ents.append((p, st, group))
        return ents

    def _device_table(self, lib, ents, dev):
        key = tuple((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                     float(g["lr"]), float(g["weight_decay"])) for p, st, g in ents)
        if key != self._table_key:
            arr = (_lib.SglAdamwTensor * len(ents))()
            for e, k in zip(arr, key):
                e.p, e.g, e.m, e.v, e.n, e.lr, e.weight_decay = k
            self._bufs["table"] = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            self._table_key = key
        return self._bufs["table"]

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
        if any(p.device != dev for p, _, _ in ents):
            raise RuntimeError("FusedAdamW: all parameters must live on one device")
        table = self._device_table(lib, ents, dev)
        numel = tuple(p.numel() for p, _, _ in ents)
        stream = torch.cuda.current_stream(dev).cuda_stream
        # one AdamW launch per distinct (betas, eps, step): a single one for every trainer of the reference (one param
        # group, every trainable tensor receives a gradient every step); torch tracks the step per tensor, so do we
        launches = {}
        for i, (_, st, g) in enumerate(ents):
            launches.setdefault((g["betas"][0], g["betas"][1], g["eps"], int(st["step"].item())), []).append(i)
        norm_ptr = None
        with torch.cuda.device(dev):
            if self.max_grad_norm is not None and self.max_grad_norm > 0:
                bmap, nb = self._plan(lib, numel, tuple(range(len(ents))), dev)
                if "norm" not in self._bufs or self._bufs["partials"].numel() < max(nb, 1):
                    self._bufs["partials"] = torch.empty(max(nb, 1), device=dev, dtype=torch.float32)
                    self._bufs["norm"] = torch.zeros(2, device=dev, dtype=torch.float32)
                _lib.check(lib.sgl_op_grad_norm(table.data_ptr(), bmap.data_ptr(), nb, float(self.max_grad_norm),
                                                self._bufs["partials"].data_ptr(), self._bufs["norm"].data_ptr(),
                                                stream), "sgl_op_grad_norm")
                norm_ptr = self._bufs["norm"].data_ptr()
                self.last_grad_norm = self._bufs["norm"][0]
            for (beta1, beta2, eps, step0), members in launches.items():
                bmap, nb = self._plan(lib, numel, tuple(members), dev)
                _lib.check(lib.sgl_op_adamw(table.data_ptr(), bmap.data_ptr(), nb, float(beta1), float(beta2),
                                            float(eps), step0 + 1, norm_ptr, stream), "sgl_op_adamw")
        for _, st, _ in ents:
            st["step"] += 1
        # the kernel wrote the parameters behind autograd's back: bump their version counters, which is what everything
        # that caches on `p._version` keys on (the encoder's bf16 weight shadows, saved-tensor checks)
        torch.autograd.graph.increment_version([p for p, _, _ in ents])
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
    """The CiFake trainer's weight EMA (cifake_binary_classifier.py:211-236) with ``update()`` as ONE HIP launch over
    all trainable tensors instead of three elementwise kernels per tensor.  Same attributes and methods: ``shadow``
    (name -> tensor, what the reference stores as ``checkpoint['ema_state_dict']``), ``update()``, ``apply_shadow()``,
    ``restore()``."""

    def __init__(self, model, decay=0.9999):
        self.model = model
        self.decay = decay
        self.shadow = {}
        self.backup = {}
        for name, param in model.named_parameters():
            if param.requires_grad:
                self.shadow[name] = param.data.clone()
        self._key = None
        self._bufs = None

    def _plan(self, lib):
        ents = [(p, self.shadow[n]) for n, p in self.model.named_parameters() if p.requires_grad]
        key = tuple((p.data_ptr(), s.data_ptr(), p.numel()) for p, s in ents)
        if key != self._key:
            for p, s in ents:
                if not (p.is_cuda and s.is_cuda and p.dtype == torch.float32 and s.dtype == torch.float32
                        and _is_dense(p) and s.stride() == p.stride()):
                    raise RuntimeError("ExponentialMovingAverage handles contiguous fp32 CUDA parameters only")
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
        if not self.shadow:
            return
        lib = _lib.load()
        table, bmap, nb, dev = self._plan(lib)
        with torch.cuda.device(dev):
            _lib.check(lib.sgl_op_ema(table.data_ptr(), bmap.data_ptr(), nb, float(self.decay),
                                      torch.cuda.current_stream(dev).cuda_stream), "sgl_op_ema")

    def apply_shadow(self):
        for name, param in self.model.named_parameters():
            if param.requires_grad:
                self.backup[name] = param.data.clone()
                param.data = self.shadow[name]

    def restore(self):
        for name, param in self.model.named_parameters():
            if param.requires_grad:
                param.data = self.backup[name]

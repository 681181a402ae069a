"""Reference-owned heads, decoder and losses that sit on the encoder's outputs.

These are the small PyTorch modules the reference defines itself around the third-party ViT.  They consume the
HIP encoder's outputs through autograd; parameter names match the reference's modules so its checkpoints load
(`decoder.projs.N.proj.weight`, `cls_head.weight`, `se.0.weight`, `classifier.N.weight`, `mlp.0.weight`, …).
Parity is pinned by `tests/golden/heads_*.npz`, produced by `oracle/gen_golden_heads.py`, which executes the
reference's own class definitions (lifted from its source text in the build container) on seeded weights/inputs.

The decoder is the only piece with real HBM traffic; here the 1x1 `head` conv is applied BEFORE
the bilinear up-sampling — exact in real arithmetic because bilinear weights sum to one — so the
(B, E, img, img) tensor of the reference (151 MB per image at E=512, 384²) is never materialised.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .config import isqrt_exact


# ---------------------------------------------------------------------------------------------------------
# SID_Set multi-task model  (Siglip2sidafrozen.py:693-803)
# ---------------------------------------------------------------------------------------------------------
class _TapProj(nn.Module):
    """`LinearProj` (:693-696): keeps the `.proj` sub-module name of the reference checkpoints."""

    def __init__(self, in_dim: int, out_dim: int):
        super().__init__()
        self.proj = nn.Linear(in_dim, out_dim)

    def forward(self, x):
        return _linear_tokens(x, self.proj.weight, self.proj.bias)


_W_CACHE: dict = {}       # id(parameter) -> (weakref(parameter), version, data_ptr, bf16 W, bf16 Wᵀ or None)
_SCRATCH: dict = {}       # device -> 64 MiB split-K scratch, reused by every call (stream-ordered)


def _cached_bf16(weight: torch.Tensor, want_t: bool):
    """bf16 copy (and, on demand, transposed copy) of a weight, re-made only when the parameter changed (optimizer steps
    bump ``_version``): the decoder has ~40 Linear layers and used to re-cast + re-transpose each one on every call.
    Keyed on the parameter OBJECT (through a weak reference), never on its address alone: the caching allocator hands a
    freed parameter's address to the next model's weights."""
    import weakref
    base = weight._base if weight._base is not None else weight      # conv.weight.view(out, in) is a fresh view per call
    key = id(base)
    hit = _W_CACHE.get(key)
    if (hit is None or hit[0]() is not base or hit[1] != base._version or hit[2] != weight.data_ptr()
            or hit[3].shape != weight.shape):
        if len(_W_CACHE) > 256:
            for k in [k for k, v in _W_CACHE.items() if v[0]() is None]:
                del _W_CACHE[k]
            if len(_W_CACHE) > 256:
                _W_CACHE.clear()
        hit = (weakref.ref(base), base._version, weight.data_ptr(), weight.detach().to(torch.bfloat16).contiguous(), None)
        _W_CACHE[key] = hit
    if want_t and hit[4] is None:
        hit = hit[:4] + (hit[3].t().contiguous(),)
        _W_CACHE[key] = hit
    return hit[3], hit[4]


def _split_scratch(dev):
    buf = _SCRATCH.get(dev)
    if buf is None:
        buf = _SCRATCH[dev] = torch.empty(64 << 20, device=dev, dtype=torch.uint8)
    return buf


class _HipLinearFn(torch.autograd.Function):
    """y = x Wᵀ + b on bf16 token-major activations through the encoder's own MFMA GEMM kernels (sgl_op_gemm_nt /
    sgl_op_gemm_tn): the decoder's tall-skinny shapes (46656 x 512 x 1152, 46656 x 512 x 512 ...) are where the
    library GEMM picks 110-240 TFLOP/s kernels.  Used under autocast only (bf16 operands, fp32 accumulate: the
    arithmetic autocast's F.linear does); fp32 callers keep F.linear."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from . import lib as _lib
        lib = _lib.load()
        M, K = x.shape
        N = weight.shape[0]
        xb = x.to(torch.bfloat16).contiguous()
        wb, _ = _cached_bf16(weight, False)
        bf = None if bias is None else bias.detach().float().contiguous()
        y = torch.empty(M, N, device=x.device, dtype=torch.bfloat16)
        _lib.check(lib.sgl_op_gemm_nt(_lib.SGL_DTYPE_BF16, xb.data_ptr(), K, wb.data_ptr(), K, M, N, K, _lib.EPI_STORE,
                                      y.data_ptr(), N, None, 0, _lib.ptr(bf), None, 0, None, 0, None, 1, 1, 1, 8, 8, 1,
                                      _lib.current_stream_handle()), "sgl_op_gemm_nt")
        ctx.save_for_backward(xb, wb)
        ctx.weight = weight
        ctx.has_bias = bias is not None
        ctx.wdtype = weight.dtype
        ctx.xdtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import lib as _lib
        lib = _lib.load()
        xb, wb = ctx.saved_tensors
        M, K = xb.shape
        N = wb.shape[0]
        dyb = dy.to(torch.bfloat16).contiguous()
        stream = _lib.current_stream_handle()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:   # dX[M,K] = dY[M,N] · W[N,K]: NT form with the K x N transpose of W as "B"
            cur, cur_t = _cached_bf16(ctx.weight, True)
            wt = cur_t if cur is wb else wb.t().contiguous()      # the cached pair only if it is still this forward's weight
            dx = torch.empty(M, K, device=xb.device, dtype=torch.bfloat16)
            _lib.check(lib.sgl_op_gemm_nt(_lib.SGL_DTYPE_BF16, dyb.data_ptr(), N, wt.data_ptr(), N, M, K, N,
                                          _lib.EPI_STORE, dx.data_ptr(), K, None, 0, None, None, 0, None, 0, None, 1, 1,
                                          1, 8, 8, 1, stream), "sgl_op_gemm_nt(dX)")
            dx = dx.to(ctx.xdtype)
        if ctx.needs_input_grad[1]:   # dW[N,K] = dYᵀ · X
            dw = torch.empty(N, K, device=xb.device, dtype=torch.float32)
            scratch = _split_scratch(xb.device)   # split-K slabs: deterministic sum
            _lib.check(lib.sgl_op_gemm_tn_ws(_lib.SGL_DTYPE_BF16, dyb.data_ptr(), N, xb.data_ptr(), K, M, N, K, 0,
                                             dw.data_ptr(), K, 0, scratch.data_ptr(), scratch.numel(), stream),
                       "sgl_op_gemm_tn_ws(dW)")
            dw = dw.to(ctx.wdtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dyb.float().sum(0) if M < 64 else _hip_colsum(lib, _lib, dyb, M, N, stream)
        return dx, dw, db


@torch.compiler.disable
def _hip_linear(x2d, weight, bias):
    return _HipLinearFn.apply(x2d, weight, bias)


@torch.compiler.disable
def _hip_dwconv(x, weight, bias):
    return _DepthwiseConv3x3Fn.apply(x, weight, bias)


def _hip_colsum(lib, _lib, t, M, N, stream):
    out = torch.empty(N, device=t.device, dtype=torch.float32)
    nbytes = ((M + 511) // 512 if (M + 511) // 512 < 256 else 256) * N * 4
    scratch = torch.empty(max(nbytes, 4), device=t.device, dtype=torch.uint8)
    _lib.check(lib.sgl_op_colsum(_lib.SGL_DTYPE_BF16, t.data_ptr(), N, M, N, out.data_ptr(), 0, scratch.data_ptr(),
                                 scratch.numel(), stream), "sgl_op_colsum")
    return out


# developer A/B switch: "hip" (default) every token-major Linear / 1x1 convolution of the decoder under bf16 autocast runs on
# this repo's MFMA GEMMs (256x256-tile kernels for large shapes, the 128x128-tile kernels for narrow ones: E = 256
# projections, the gate's bottleneck, the 1-channel head); "wide": only shapes with >= 512 columns (round 2's rule, the
# rest on the vendor GEMM behind F.linear); "torch": F.linear everywhere
_HIP_LINEAR = __import__("os").environ.get("SGL_HEADS_LINEAR", "hip")


def _linear_tokens(x: torch.Tensor, weight: torch.Tensor, bias) -> torch.Tensor:
    """F.linear on (..., K) token-major data; under CUDA bf16 autocast it runs on the HIP GEMMs (bf16 operands, fp32
    accumulate: the arithmetic autocast's F.linear does).  fp32 callers (strict parity runs, CPU) keep F.linear."""
    K, N = x.shape[-1], weight.shape[0]
    if (_HIP_LINEAR != "torch" and x.is_cuda and torch.is_autocast_enabled()
            and torch.get_autocast_dtype("cuda") == torch.bfloat16 and K % 8 == 0 and x.numel() // K >= 64):
        wide = N % 8 == 0 and N >= 512 and K >= 512 and x.numel() // K >= 2048
        if wide or _HIP_LINEAR == "hip":
            if N % 8:   # the 1-channel mask head: pad the output columns to the GEMM's 8-column granularity
                pad = 8 - N % 8
                weight = F.pad(weight, (0, 0, 0, pad))
                bias = None if bias is None else F.pad(bias, (0, pad))
            y = _hip_linear(x.reshape(-1, K), weight, bias)
            return y.reshape(*x.shape[:-1], y.shape[-1])[..., :N]
    return F.linear(x, weight, bias)


class _DepthwiseConv3x3Fn(torch.autograd.Function):
    """Depthwise 3x3 (padding 1) on channels-last (B, gh, gw, E) CUDA tensors through the HIP kernels of
    csrc/decoder.hip: forward, data gradient (same stencil, flipped taps) and the two-stage weight/bias gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from . import lib as _lib
        lib = _lib.load()
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        x = x.contiguous()
        B, gh, gw, E = x.shape
        w = weight.detach().float().reshape(E, 9).t().contiguous()      # tap-major [9][E] (see siglip_hip.h)
        b = None if bias is None else bias.detach().float().contiguous()
        y = torch.empty_like(x)
        dt = _lib.SGL_DTYPE_BF16 if x.dtype == torch.bfloat16 else _lib.SGL_DTYPE_F32
        _lib.check(lib.sgl_op_dwconv3x3(x.data_ptr(), dt, w.data_ptr(), _lib.ptr(b), y.data_ptr(), B, gh, gw, E, 0,
                                        _lib.current_stream_handle()), "sgl_op_dwconv3x3")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.wdtype = weight.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import lib as _lib
        lib = _lib.load()
        x, w = ctx.saved_tensors
        dy = dy.to(x.dtype).contiguous()
        B, gh, gw, E = x.shape
        dt = _lib.SGL_DTYPE_BF16 if x.dtype == torch.bfloat16 else _lib.SGL_DTYPE_F32
        stream = _lib.current_stream_handle()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(lib.sgl_op_dwconv3x3(dy.data_ptr(), dt, w.data_ptr(), None, dx.data_ptr(), B, gh, gw, E, 1, stream),
                       "sgl_op_dwconv3x3(flip)")
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw10 = torch.empty(10, E, device=x.device, dtype=torch.float32)   # nine tap rows + the bias row
            nbytes = lib.sgl_op_dwconv3x3_wgrad_scratch_bytes(B, gh, gw, E)
            scratch = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
            _lib.check(lib.sgl_op_dwconv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dt, dw10.data_ptr(), 0,
                                                  scratch.data_ptr(), nbytes, B, gh, gw, E, stream),
                       "sgl_op_dwconv3x3_wgrad")
            dw = dw10[:9].t().reshape(E, 1, 3, 3).to(ctx.wdtype)
            db = dw10[9] if ctx.has_bias else None
        return dx, dw, db


class _GateMulFn(torch.autograd.Function):
    """y = sigmoid(g) * x in one HBM pass (csrc/decoder_tail.hip), backward (dg w.r.t. the PRE-sigmoid gate, dx) in one
    more: the SE-style gate of the SID decoder (`gate * x`, Siglip2sidafrozen.py:741-742) on (B*N, E*K) activations."""

    @staticmethod
    def forward(ctx, g, x):
        from . import lib as _lib
        lib = _lib.load()
        dt = g.dtype if g.dtype in (torch.float32, torch.bfloat16) else torch.float32
        g2, x2 = g.to(dt).contiguous(), x.to(dt).contiguous()
        y = torch.empty_like(x2)
        code = _lib.SGL_DTYPE_BF16 if dt == torch.bfloat16 else _lib.SGL_DTYPE_F32
        _lib.check(lib.sgl_op_gate_mul(g2.data_ptr(), x2.data_ptr(), y.data_ptr(), g2.numel(), code,
                                       _lib.current_stream_handle()), "sgl_op_gate_mul")
        ctx.save_for_backward(g2, x2)
        ctx.code, ctx.gdt, ctx.xdt = code, g.dtype, x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import lib as _lib
        lib = _lib.load()
        g2, x2 = ctx.saved_tensors
        dy2 = dy.to(g2.dtype).contiguous()
        dg = torch.empty_like(g2) if ctx.needs_input_grad[0] else None
        dx = torch.empty_like(x2) if ctx.needs_input_grad[1] else None
        _lib.check(lib.sgl_op_gate_mul_bwd(dy2.data_ptr(), g2.data_ptr(), x2.data_ptr(), _lib.ptr(dg), _lib.ptr(dx),
                                           g2.numel(), ctx.code, _lib.current_stream_handle()), "sgl_op_gate_mul_bwd")
        return (None if dg is None else dg.to(ctx.gdt)), (None if dx is None else dx.to(ctx.xdt))


@torch.compiler.disable
def _gate_mul(gate_pre: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    nv = 8 if gate_pre.dtype == torch.bfloat16 else 4
    if gate_pre.is_cuda and gate_pre.shape == x.shape and gate_pre.numel() % nv == 0 and gate_pre.numel() >= 4096:
        return _GateMulFn.apply(gate_pre, x)
    return torch.sigmoid(gate_pre) * x


class _SegLossFromLowresFn(torch.autograd.Function):
    """`bce_dice_loss(F.interpolate(logit_lr, (S,S), 'bilinear'), masks)` over the images flagged in `sel`, without
    ever forming the (B,1,S,S) logits: csrc/decoder_tail.hip evaluates every output pixel from its four low-res logits in
    registers (forward: per-image partial sums; backward: transposed interpolation gathered per low-res pixel, fixed order).
    No host synchronisation: an empty selection gives 0 (the reference skips the term, Siglip2sidafrozen.py:1380-1389)."""

    @staticmethod
    def forward(ctx, logit_lr, masks, sel, bce_w, dice_w, eps):
        from . import lib as _lib
        lib = _lib.load()
        B, g = logit_lr.shape[0], logit_lr.shape[-1]
        S = masks.shape[-1]
        lr = logit_lr.detach().reshape(B, g, g).float().contiguous()
        t = masks.reshape(B, S, S).float().contiguous()
        chunks = lib.sgl_op_seg_loss_chunks(S)
        partial = torch.empty(B, chunks, 4, device=lr.device, dtype=torch.float32)
        _lib.check(lib.sgl_op_seg_loss_fwd(lr.data_ptr(), t.data_ptr(), partial.data_ptr(), B, g, S,
                                           _lib.current_stream_handle()), "sgl_op_seg_loss_fwd")
        sums = partial.sum(1)                                   # (B, 4), fixed order
        w = sel.to(torch.float32)
        n = w.sum()
        nz = (n > 0).to(torch.float32)
        n1 = n.clamp(min=1.0)
        bce = (sums[:, 0] * w).sum() / (n1 * float(S * S))
        dice_b = 2.0 * sums[:, 1] / (sums[:, 2] + sums[:, 3] + eps)
        dice = 1.0 - (dice_b * w).sum() / n1
        loss = (bce_w * bce + dice_w * dice) * nz
        ctx.save_for_backward(lr, t, sums, w, n1, nz)
        ctx.cfg = (B, g, S, bce_w, dice_w, eps, logit_lr.shape, logit_lr.dtype)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        from . import lib as _lib
        lib = _lib.load()
        lr, t, sums, w, n1, nz = ctx.saved_tensors
        B, g, S, bce_w, dice_w, eps, shape, dtype = ctx.cfg
        up = dloss.float() * nz
        coef = torch.stack([up * bce_w * w / (n1 * float(S * S)), -up * dice_w * w / n1], dim=1).contiguous()
        dlr = torch.empty_like(lr)
        _lib.check(lib.sgl_op_seg_loss_bwd(lr.data_ptr(), t.data_ptr(), sums.contiguous().data_ptr(), coef.data_ptr(),
                                           dlr.data_ptr(), B, g, S, float(eps), _lib.current_stream_handle()),
                   "sgl_op_seg_loss_bwd")
        return dlr.reshape(shape).to(dtype), None, None, None, None, None


@torch.compiler.disable
def bce_dice_loss_from_lowres(logit_lr: torch.Tensor, masks: torch.Tensor, has_mask: torch.Tensor = None,
                              bce_w: float = 1.0, dice_w: float = 0.5, eps: float = 1e-6) -> torch.Tensor:
    """`bce_dice_loss(upsample(logit_lr)[has_mask], masks[has_mask])` (Siglip2sidafrozen.py:174-181,743) from the (B,1,g,g)
    logits of `SegFormerMaskDecoder(..., return_lowres=True)`; CUDA only (HIP kernels), fp32 statistics."""
    if not logit_lr.is_cuda:
        raise RuntimeError("bce_dice_loss_from_lowres runs on the GPU (HIP kernels); use bce_dice_loss on CPU tensors")
    if has_mask is None:
        has_mask = torch.ones(logit_lr.shape[0], dtype=torch.bool, device=logit_lr.device)
    return _SegLossFromLowresFn.apply(logit_lr, masks, has_mask, float(bce_w), float(dice_w), float(eps))


class SegFormerMaskDecoder(nn.Module):
    """SegFormer-style mask decoder (`SegFormerStrongDecoder`, Siglip2sidafrozen.py:698-745).

    per tap: Linear(D→E) → (B,E,g,g) → depthwise 3x3 → 1x1 → GELU; concat K taps → channel gate
    (1x1 E·K→E·K/4 → GELU → 1x1 → sigmoid) ⊙ → 1x1 E·K→E → [1x1 E→1 ∘ bilinear up to target]."""

    def __init__(self, in_dims: Sequence[int], embed_dim: int = 256, dropout_rate: float = 0.0,
                 head_before_upsample: bool = True):
        super().__init__()
        k = len(in_dims)
        self.projs = nn.ModuleList([_TapProj(d, embed_dim) for d in in_dims])
        smooth = []
        for _ in in_dims:
            layers: List[nn.Module] = [nn.Conv2d(embed_dim, embed_dim, 3, padding=1, groups=embed_dim),
                                       nn.Conv2d(embed_dim, embed_dim, 1), nn.GELU()]
            if dropout_rate > 0:
                layers.append(nn.Dropout2d(p=dropout_rate))
            smooth.append(nn.Sequential(*layers))
        self.smooth = nn.ModuleList(smooth)
        self.fuse_attn = nn.Sequential(nn.Conv2d(embed_dim * k, (embed_dim * k) // 4, 1), nn.GELU(),
                                       nn.Conv2d((embed_dim * k) // 4, embed_dim * k, 1), nn.Sigmoid())
        fuse: List[nn.Module] = [nn.Conv2d(embed_dim * k, embed_dim, 1)]
        if dropout_rate > 0:
            fuse.append(nn.Dropout2d(p=dropout_rate))
        self.fuse = nn.Sequential(*fuse)
        self.head = nn.Conv2d(embed_dim, 1, 1)
        self.head_before_upsample = head_before_upsample

    @staticmethod
    def _pointwise(conv: nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
        """A 1x1 convolution on token-major data (..., C_in) -> (..., C_out): a plain GEMM (hipBLASLt) instead of a
        MIOpen convolution; the parameters keep their Conv2d shapes, so checkpoints are unchanged."""
        return _linear_tokens(x, conv.weight.view(conv.out_channels, conv.in_channels), conv.bias)

    @staticmethod
    def _depthwise3x3(conv: nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
        """Depthwise 3x3, zero padding 1, on channels-last data (B, gh, gw, E): nine shifted multiply-adds (the same
        arithmetic as nn.Conv2d(E, E, 3, padding=1, groups=E); MIOpen has only a naive fp32 NHWC solver for it)."""
        e = x.shape[-1]
        if x.is_cuda and e % 8 == 0 and e <= 1024 and 256 % (e // 8) == 0 and 256 % (e // 4) == 0:
            return _hip_dwconv(x, conv.weight, conv.bias)   # one HBM pass (csrc/decoder.hip)
        w = conv.weight                      # (E, 1, 3, 3)
        xp = F.pad(x, (0, 0, 1, 1, 1, 1))    # pad gw and gh by one
        gh, gw = x.shape[1], x.shape[2]
        out = None
        for dy in range(3):
            for dx in range(3):
                term = xp[:, dy:dy + gh, dx:dx + gw, :] * w[:, 0, dy, dx]
                out = term if out is None else out + term
        return out + conv.bias if conv.bias is not None else out

    def forward(self, hidden_list: Sequence[torch.Tensor], grid_hw: Tuple[int, int], target_size: int = 448,
                return_lowres: bool = False):
        """Everything up to the 1-channel logit map runs token-major / channels-last as GEMMs and elementwise ops (same
        math as the reference's NCHW convolutions, Siglip2sidafrozen.py:726-745); only the final bilinear up-sample of
        the (B,1,g,g) logits uses an NCHW tensor.  `return_lowres=True` stops before that up-sample and returns the
        (B,1,gh,gw) logits: `bce_dice_loss_from_lowres` consumes them without ever materialising (B,1,S,S)."""
        gh, gw = grid_hw
        feats = []
        for proj, smooth, h in zip(self.projs, self.smooth, hidden_list):
            x = proj(h)                                              # (B, N, E)
            b, _, e = x.shape
            x = self._depthwise3x3(smooth[0], x.reshape(b, gh, gw, e))
            x = F.gelu(self._pointwise(smooth[1], x))
            for extra in list(smooth)[3:]:                           # Dropout2d when configured: acts on channels
                x = extra(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
            feats.append(x)
        x = torch.cat(feats, dim=-1)                                 # (B, gh, gw, E*K)
        gate_pre = self._pointwise(self.fuse_attn[2], F.gelu(self._pointwise(self.fuse_attn[0], x)))
        x = self._pointwise(self.fuse[0], _gate_mul(gate_pre, x))       # sigmoid(gate) * x in one pass on the GPU
        for extra in list(self.fuse)[1:]:
            x = extra(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        if self.head_before_upsample or return_lowres:
            logit = self._pointwise(self.head, x).permute(0, 3, 1, 2)   # (B, 1, gh, gw)
            if return_lowres:
                return logit
            return F.interpolate(logit, size=(target_size, target_size), mode="bilinear", align_corners=False)
        x = F.interpolate(x.permute(0, 3, 1, 2), size=(target_size, target_size), mode="bilinear", align_corners=False)
        return self.head(x)


class SigLIP2MTL(nn.Module):
    """3-class (real / synthetic / tampered) + mask-localisation model (`SigLIP2_MTL`,
    Siglip2sidafrozen.py:750-803) on the HIP encoder.  `forward(pixel_values) -> (cls_logit (B,3), seg_logits
    (B,1,S,S))`.  Only the taps the decoder needs are requested from the encoder."""

    def __init__(self, encoder: nn.Module, seg_layers: Sequence[int] = (2, 6, 10, -1), embed_dim: int = 256,
                 dropout_rate: float = 0.0, freeze_below: Optional[int] = None):
        super().__init__()
        self.encoder = encoder
        hid = encoder.config.hidden_size
        if freeze_below is not None:  # frozen variant (:754-768): embeddings + blocks < freeze_below
            for p in encoder.vision_model.embeddings.parameters():
                p.requires_grad = False
            for i, layer in enumerate(encoder.vision_model.encoder.layers):
                for p in layer.parameters():
                    p.requires_grad = i >= freeze_below
        self.cls_head = (nn.Sequential(nn.Dropout(p=dropout_rate), nn.Linear(hid, 3)) if dropout_rate > 0
                         else nn.Linear(hid, 3))
        self.seg_layers = tuple(seg_layers)
        self.decoder = SegFormerMaskDecoder([hid] * len(self.seg_layers), embed_dim=embed_dim,
                                            dropout_rate=dropout_rate)

    def forward(self, pixel_values, return_lowres: bool = False):
        """(cls_logit (B,3), seg_logits (B,1,S,S)) as the reference; `return_lowres=True` gives the (B,1,g,g) logit map
        instead (the training path of `training_loss`, which never forms the up-sampled logits)."""
        n_layers = self.encoder.config.num_hidden_layers
        idxs = [(i + 1 if i >= 0 else n_layers) for i in self.seg_layers]   # hs = [emb, h1..hL] (:790-793)
        out = self.encoder(pixel_values=pixel_values, hidden_state_ids=idxs, interpolate_pos_encoding=True)
        pooled = out.pooler_output if out.pooler_output is not None else out.last_hidden_state.mean(1)
        cls_logit = self.cls_head(pooled).squeeze(1)
        feats = list(out.hidden_states)
        g = isqrt_exact(feats[0].shape[1])
        seg_logits = self.decoder(feats, (g, g), target_size=int(pixel_values.shape[-1]), return_lowres=return_lowres)
        return cls_logit, seg_logits

    def training_loss(self, pixel_values, y_class, masks, has_mask, lam_seg: float = 1.0, bce_w: float = 1.0,
                      dice_w: float = 0.5):
        """The SID train-step loss (Siglip2sidafrozen.py:1375-1389: CE + lam * BCE/Dice on the samples with a mask) with the
        decoder tail fused for HBM: low-res logits -> loss directly.  ``bce_w`` / ``dice_w`` are the epoch-scheduled
        ``current_bce_w`` / ``current_dice_w`` the reference passes to ``bce_dice_loss`` (Siglip2sidafrozen.py:1340-1351,
        1389).  Returns (loss, cls_logit, seg_logits_lowres)."""
        cls_logit, seg_lr = self.forward(pixel_values, return_lowres=True)
        loss = F.cross_entropy(cls_logit.float(), y_class) + lam_seg * bce_dice_loss_from_lowres(
            seg_lr, masks, has_mask, bce_w=bce_w, dice_w=dice_w)
        return loss, cls_logit, seg_lr


# ---------------------------------------------------------------------------------------------------------
# binary heads on the pooled embedding
# ---------------------------------------------------------------------------------------------------------
def l2_normalize(f: torch.Tensor, eps: float = 0.0) -> torch.Tensor:
    """`f / f.norm(dim=-1, keepdim=True)` (cifake_binary_classifier.py:728, hidf_video_classifier.py:308);
    eps=1e-6 is the `train_fusion_head_only.py:106` form."""
    return f / (f.norm(dim=-1, keepdim=True) + eps)


class SingleTokenAttention(nn.Module):
    """`LightweightAttention` (cifake_binary_classifier.py:574-595) applied to a length-1 sequence."""

    def __init__(self, dim: int, num_heads: int = 4):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        b, n, c = x.shape
        qkv = self.qkv(x).reshape(b, n, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        attn = ((qkv[0] @ qkv[1].transpose(-2, -1)) * self.scale).softmax(dim=-1)
        return self.proj((attn @ qkv[2]).transpose(1, 2).reshape(b, n, c))


class CifakeBinaryHead(nn.Module):
    """Head of `FastBinaryClassifier` (cifake_binary_classifier.py:640-698,727-749): L2-norm → LayerNorm →
    (length-1 attention) → size-dependent MLP → scalar logit."""

    def __init__(self, feature_dim: int, model_size: str = "small", dropout_rate: float = 0.1,
                 use_lightweight_attention: bool = True):
        super().__init__()
        if use_lightweight_attention and model_size in ("tiny", "small"):
            self.attention: Optional[nn.Module] = SingleTokenAttention(feature_dim, num_heads=4)
        elif model_size == "large":
            self.attention = nn.MultiheadAttention(feature_dim, min(8, feature_dim // 64), dropout=dropout_rate,
                                                   batch_first=True)
        else:
            self.attention = None
        d = feature_dim
        if model_size == "tiny":
            self.classifier = nn.Sequential(nn.Dropout(dropout_rate * 0.5), nn.Linear(d, 1))
        elif model_size == "small":
            self.classifier = nn.Sequential(nn.Linear(d, d // 4), nn.GELU(), nn.Dropout(dropout_rate),
                                            nn.Linear(d // 4, 1))
        else:
            self.classifier = nn.Sequential(nn.Linear(d, d // 2), nn.GELU(), nn.Dropout(dropout_rate),
                                            nn.Linear(d // 2, d // 4), nn.GELU(), nn.Dropout(dropout_rate * 0.5),
                                            nn.Linear(d // 4, 1))
        self.layer_norm = nn.LayerNorm(d)

    def forward(self, features, return_features: bool = False):
        f = self.layer_norm(l2_normalize(features))
        if self.attention is not None:
            f = f.unsqueeze(1)
            f = self.attention(f) if isinstance(self.attention, SingleTokenAttention) else self.attention(f, f, f)[0]
            f = f.squeeze(1)
        return f if return_features else self.classifier(f).squeeze(-1)


class _L2NormTemporalMeanFn(torch.autograd.Function):
    """(B*T, D) frame embeddings -> per-frame L2-norm -> mean over the T frames of a clip -> (B, D), one HIP launch forward
    and one backward (csrc/preprocess.hip) instead of norm / div / view / mean and their four backward kernels
    (hidf_video_classifier.py:308-316)."""

    @staticmethod
    def forward(ctx, f, batch_size):
        from . import lib as _lib
        lib = _lib.load()
        f32 = f.float().contiguous()
        BT, D = f32.shape
        T = BT // batch_size
        out = torch.empty(batch_size, D, device=f.device, dtype=torch.float32)
        inv = torch.empty(BT, device=f.device, dtype=torch.float32)
        _lib.check(lib.sgl_op_l2norm_tmean_fwd(f32.data_ptr(), out.data_ptr(), inv.data_ptr(), batch_size, T, D,
                                               _lib.current_stream_handle()), "sgl_op_l2norm_tmean_fwd")
        ctx.save_for_backward(f32, inv)
        ctx.dims, ctx.dtype = (batch_size, T, D), f.dtype
        return out.to(f.dtype)

    @staticmethod
    def backward(ctx, dout):
        from . import lib as _lib
        lib = _lib.load()
        f32, inv = ctx.saved_tensors
        B, T, D = ctx.dims
        g = dout.float().contiguous()
        df = torch.empty_like(f32)
        _lib.check(lib.sgl_op_l2norm_tmean_bwd(f32.data_ptr(), inv.data_ptr(), g.data_ptr(), df.data_ptr(), B, T, D,
                                               _lib.current_stream_handle()), "sgl_op_l2norm_tmean_bwd")
        return df.to(ctx.dtype), None


@torch.compiler.disable
def l2norm_temporal_mean(frame_features: torch.Tensor, batch_size: int) -> torch.Tensor:
    """`l2_normalize(f).view(B, T, D).mean(1)`; fused HIP kernels on CUDA tensors, PyTorch ops elsewhere."""
    if frame_features.is_cuda and frame_features.dim() == 2 and frame_features.shape[0] % batch_size == 0:
        return _L2NormTemporalMeanFn.apply(frame_features, batch_size)
    return l2_normalize(frame_features).view(batch_size, -1, frame_features.shape[-1]).mean(dim=1)


class VideoBinaryHead(nn.Module):
    """Tail of `BinaryVideoClassifier` (hidf_video_classifier.py:276-320): per-frame L2-norm → mean over T →
    LayerNorm / MLP → one logit per clip.  Input: per-frame embeddings (B*T, D)."""

    def __init__(self, feature_dim: int, num_frames: int = 4, dropout_rate: float = 0.3):
        super().__init__()
        d = feature_dim
        self.num_frames = num_frames
        self.binary_classifier = nn.Sequential(
            nn.LayerNorm(d), nn.Dropout(dropout_rate), nn.Linear(d, d // 2), nn.ReLU(),
            nn.Dropout(dropout_rate * 0.67), nn.Linear(d // 2, d // 4), nn.ReLU(), nn.Dropout(dropout_rate * 0.33),
            nn.Linear(d // 4, 1))

    def forward(self, frame_features, batch_size: int):
        return self.binary_classifier(l2norm_temporal_mean(frame_features, batch_size)).squeeze(-1)


class SEBinaryHead(nn.Module):
    """SE gate + MLP of `BinaryClassifier` (train_fusion_head_only.py:84-109): f·σ(W2 relu(W1 f)) → MLP."""

    def __init__(self, dim: int = 1024):
        super().__init__()
        self.se = nn.Sequential(nn.Linear(dim, dim // 16), nn.ReLU(), nn.Linear(dim // 16, dim), nn.Sigmoid())
        self.classifier = nn.Sequential(nn.LayerNorm(dim), nn.Dropout(0.3), nn.Linear(dim, dim // 2), nn.GELU(),
                                        nn.Dropout(0.2), nn.Linear(dim // 2, dim // 4), nn.GELU(),
                                        nn.Linear(dim // 4, 1))

    def forward(self, pooled):
        f = l2_normalize(pooled, eps=1e-6)
        return self.classifier(f * self.se(f)).squeeze(-1)


class ImageBinaryHead(nn.Module):
    """Head of the HiDF image-track classifier (``BinaryClassifier.classifier`` + the L2-norm in front of it,
    simple_classifier.py:141-148,159-164): f/||f|| -> LayerNorm -> Dropout(0.3) -> Linear(D, D/2) -> GELU -> Dropout(0.2) ->
    Linear(D/2, 1) -> (B,).  Parameter names equal the reference's (``classifier.0.weight`` ...)."""

    def __init__(self, dim: int = 1024):
        super().__init__()
        self.classifier = nn.Sequential(nn.LayerNorm(dim), nn.Dropout(0.3), nn.Linear(dim, dim // 2), nn.GELU(),
                                        nn.Dropout(0.2), nn.Linear(dim // 2, 1))

    def forward(self, pooled):
        return self.classifier(l2_normalize(pooled)).squeeze(-1)


# ---------------------------------------------------------------------------------------------------------
# fusion / calibration  (train_fusion_head_only.py:230-317, appv3.py:1497-1510,1573-1578,3147-3182, coral.py:300-322)
# ---------------------------------------------------------------------------------------------------------
class TemperatureScaler(nn.Module):
    def __init__(self):
        super().__init__()
        self.T = nn.Parameter(torch.tensor(1.0))

    def forward(self, logits):
        return logits / (self.T + 1e-6)


class AdaptiveFusionHead(nn.Module):
    """w = softmax(MLP([zf, zs, |zf − zs|])); z = (w0·zf + w1·zs) / (T + 1e-6)."""

    def __init__(self, hidden_dim: int = 32):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(3, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, 2))
        self.temp = TemperatureScaler()

    def forward(self, z_freq, z_sig):
        w = F.softmax(self.mlp(torch.stack([z_freq, z_sig, (z_freq - z_sig).abs()], dim=-1)), dim=-1)
        return self.temp(w[..., 0] * z_freq + w[..., 1] * z_sig)


class LinearFusionHead(nn.Module):
    """The shipped 2→1 `FusionHead` (appv3.py:1573-1578; `siglip/fusion_head.safetensors`)."""

    def __init__(self):
        super().__init__()
        self.fc = nn.Linear(2, 1)

    def forward(self, x):
        return self.fc(x)


class _FeatureNormalizer(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.register_buffer("mean", torch.zeros(dim))
        self.register_buffer("std", torch.ones(dim))

    def forward(self, x):
        return (x - self.mean) / (self.std + 1e-6)


class _ContrastScaler(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(dim))
        self.beta = nn.Parameter(torch.zeros(dim))

    def forward(self, x):
        return torch.tanh(self.alpha * x + self.beta)


class _BandGating(nn.Module):
    def __init__(self, dim, num_bands=4):
        super().__init__()
        assert dim % num_bands == 0
        self.band_dim, self.num_bands = dim // num_bands, num_bands
        self.gates = nn.Parameter(torch.zeros(num_bands))

    def forward(self, x):
        g = torch.sigmoid(self.gates).repeat_interleave(self.band_dim)
        return x * g


class _ResidualMLPBlock(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return x + self.fc2(F.gelu(self.fc1(self.norm(x))))


class FreqMLPv5(nn.Module):
    """v5 frequency-feature classifier (train_fusion_head_only.py:282-301)."""

    def __init__(self, dim: int = 24, hidden: int = 64, num_bands: int = 4):
        super().__init__()
        self.normer = _FeatureNormalizer(dim)
        self.contrast = _ContrastScaler(dim)
        self.band = _BandGating(dim, num_bands)
        self.blocks = nn.ModuleList([_ResidualMLPBlock(dim, hidden), _ResidualMLPBlock(dim, hidden)])
        self.head = nn.Linear(dim, 1)
        self.temp = TemperatureScaler()

    def forward(self, x):
        x = self.band(self.contrast(self.normer(x)))
        for blk in self.blocks:
            x = blk(x)
        return self.temp(self.head(x).squeeze(-1))


class _SafeLayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.eps = eps

    def forward(self, x):
        mean = x.mean(dim=-1, keepdim=True)
        var = x.var(dim=-1, unbiased=False, keepdim=True)
        return (x - mean) / torch.sqrt(var + self.eps) * self.weight + self.bias


class FreqMLPApp(nn.Module):
    """The shipped app FreqMLP (appv3.py:1497-1510; `siglip/freq_mlp.safetensors`: net.0 LN(24), net.1 24→64,
    net.3 64→1).  The reference adds 0.001·randn jitter in eval mode (:1508-1509); `jitter=False` gives the
    deterministic function used for the known-answer test."""

    def __init__(self, in_dim: int = 24, hid: int = 64):
        super().__init__()
        self.net = nn.Sequential(_SafeLayerNorm(in_dim), nn.Linear(in_dim, hid), nn.GELU(), nn.Linear(hid, 1))

    def forward(self, x, jitter: bool = False):
        if jitter and not self.training:
            x = x + 0.001 * torch.randn_like(x)
        return self.net(x).squeeze(-1)


def _logit(p: float) -> float:
    p = min(max(p, 1e-6), 1 - 1e-6)
    return math.log(p / (1 - p))


class CoralCalibrator:
    """Ordinal 5-bin calibration (appv3.py:3154-3182): g_k = σ(z − c_k); p0 = 1−g0, p_k = g_{k−1}−g_k, p4 = g3."""

    def __init__(self, cutpoints: Optional[dict] = None, logit_cuts: Optional[Sequence[float]] = None):
        if logit_cuts is not None:
            cuts = list(logit_cuts)
        elif cutpoints:
            cuts = [_logit(cutpoints[k]) for k in ("q25", "q50", "q75", "max")]
        else:
            cuts = [_logit(0.32), _logit(0.47), _logit(0.61), _logit(0.75)]
        self.c = torch.tensor(cuts, dtype=torch.float32)

    @torch.no_grad()
    def probs(self, z_scaled):
        g = torch.sigmoid(torch.as_tensor(z_scaled, dtype=torch.float32) - self.c)
        p = torch.cat([1.0 - g[:1], g[:-1] - g[1:], g[-1:]])
        return p / (p.sum() + 1e-8)

    @torch.no_grad()
    def predict(self, z_scaled):
        p = self.probs(z_scaled)
        return int(torch.argmax(p).item()), p

    @torch.no_grad()
    def probs_batch(self, z_scaled: torch.Tensor) -> torch.Tensor:
        """Vectorised over a batch of fused logits (the app's 9-crop + rotated-view pattern, appv3.py:3221-3247)."""
        g = torch.sigmoid(z_scaled.float().unsqueeze(-1) - self.c.to(z_scaled.device))
        p = torch.cat([1.0 - g[..., :1], g[..., :-1] - g[..., 1:], g[..., -1:]], dim=-1)
        return p / (p.sum(dim=-1, keepdim=True) + 1e-8)


@torch.no_grad()
def core_signals_batched(z_sigs: torch.Tensor, crop_weights: torch.Tensor, z_freqs: torch.Tensor, z_rot: torch.Tensor,
                         fusion_head: nn.Module, coral: "CoralCalibrator", freq_temp: float = 1.25,
                         coral_temp: float = 1.0) -> dict:
    """The inference-side tail of the app's `detect_core` (appv3.py:3221-3300) for a BATCH of images, entirely on the
    device: weighted multi-crop logits, the 90-degree dual-view stabiliser, the 2->1 fusion head on probabilities, the
    temperature-scaled raw probability, the CORAL ordinal distribution with its Gaussian-smoothed probability and entropy,
    and the conservative blend.  The app does this per image with ~12 `.item()` round trips; here every quantity is a
    (B,) / (B,5) tensor and nothing synchronises.

    z_sigs, z_freqs: (B, C) per-crop logits of the SigLIP classifier and the frequency MLP (C = 9 crops,
    make_multicrops); crop_weights: (C,); z_rot: (B,) logit of the rotated view."""
    dev = z_sigs.device
    w = crop_weights.to(dev, torch.float32)
    z_sig0 = (z_sigs.float() * w).sum(-1)
    z_freq = (z_freqs.float() * w).sum(-1)
    visual_prob = 0.6 * torch.sigmoid(z_sig0) + 0.4 * torch.sigmoid(z_rot.float())
    pc = visual_prob.clamp(1e-6, 1 - 1e-6)
    z_sig = torch.log(pc / (1 - pc))                                     # _logit (appv3.py:3150-3152)
    p_freq = torch.sigmoid(z_freq / freq_temp)
    z = fusion_head(torch.stack([visual_prob, p_freq], dim=-1)).reshape(-1)
    z_scaled = z / max(float(coral_temp), 1e-3)
    p_fake_raw = torch.sigmoid(z_scaled)
    risk_probs = coral.probs_batch(z_scaled)                             # (B, 5)
    risk_idx = risk_probs.argmax(-1)
    risk_vec = torch.arange(5, dtype=torch.float32, device=dev)
    mu = (risk_probs * risk_vec).sum(-1)
    var = (risk_probs * (risk_vec - mu[:, None]) ** 2).sum(-1)
    p_coral = (mu / 4.0 + 0.5 * var).clamp(0.0, 1.0)
    entropy = -(risk_probs * torch.log(risk_probs + 1e-8)).sum(-1)
    p_blend = (0.70 * p_fake_raw + 0.30 * p_coral).clamp(0.0, 1.0)
    return {"z_sig": z_sig, "z_freq": z_freq, "visual_prob": visual_prob, "p_freq": p_freq, "z": z, "z_scaled": z_scaled,
            "p_fake_raw": p_fake_raw, "risk_probs": risk_probs, "risk_idx": risk_idx, "p_fake_coral": p_coral,
            "coral_entropy": entropy, "p_blend": p_blend}


RISK_NAMES = ["REAL", "LEAN_REAL", "BORDERLINE", "LEAN_FAKE", "FAKE"]


def fit_coral_cutpoints(logits: torch.Tensor, labels: torch.Tensor = None, num_classes: int = 5) -> List[float]:
    """15/35/55/75-percentile cut-points of the fused logits (coral.py:300-322: value at index int(q·n) of the
    ascending sort)."""
    s = np.sort(np.asarray(logits.detach().cpu().numpy() if torch.is_tensor(logits) else logits))
    return [float(s[int(q * len(s))]) for q in (0.15, 0.35, 0.55, 0.75)]


# ---------------------------------------------------------------------------------------------------------
# losses / metrics  (Siglip2sidafrozen.py:69-189, cifake_binary_classifier.py:238-251,788-792)
# ---------------------------------------------------------------------------------------------------------
def focal_loss(logits, targets, alpha: float = 0.25, gamma: float = 2.0):
    p = torch.sigmoid(logits)
    ce = F.binary_cross_entropy_with_logits(logits, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    alpha_t = alpha * targets + (1 - alpha) * (1 - targets)
    return (alpha_t * (1 - p_t) ** gamma * ce).mean()


def _box(x, k):
    return F.conv2d(x, torch.ones(1, 1, k, k, device=x.device, dtype=x.dtype), padding=k // 2)


def boundary_aware_loss(logits, targets, kernel_size: int = 3):
    """Convolution-morphology branch of the reference (Siglip2sidafrozen.py:107-115), the one taken when kornia
    is absent: boundary = dilate − erode by a k×k box; BCE weighted 1 + 3·boundary."""
    s = _box(targets, kernel_size)
    boundary = ((s > 0).float() - (s == kernel_size ** 2).float()).detach()
    bce = F.binary_cross_entropy_with_logits(logits, targets, reduction="none")
    return (bce * (1 + 3 * boundary)).mean()


def morphological_loss(logits, targets, kernel_size: int = 3):
    """Smoothness-penalty fallback branch (Siglip2sidafrozen.py:136-140)."""
    p = torch.sigmoid(logits)
    return (p[:, :, :, 1:] - p[:, :, :, :-1]).abs().mean() + (p[:, :, 1:, :] - p[:, :, :-1, :]).abs().mean()


def iou_loss(logits, targets, smooth: float = 1e-6):
    p = torch.sigmoid(logits)
    inter = (p * targets).sum(dim=(1, 2, 3))
    union = p.sum(dim=(1, 2, 3)) + targets.sum(dim=(1, 2, 3)) - inter + smooth
    return 1 - (inter / union).mean()


def _dice_term(logits, targets, eps):
    p = torch.sigmoid(logits)
    inter = (p * targets).sum(dim=(1, 2, 3))
    denom = p.sum(dim=(1, 2, 3)) + targets.sum(dim=(1, 2, 3)) + eps
    return 1 - (2 * inter / denom).mean()


def bce_dice_loss(logits, targets, bce_w: float = 1.0, dice_w: float = 0.5, eps: float = 1e-6):
    return bce_w * F.binary_cross_entropy_with_logits(logits, targets) + dice_w * _dice_term(logits, targets, eps)


def combined_segmentation_loss(logits, targets, bce_w=0.4, focal_w=0.3, dice_w=0.5, boundary_w=0.4, iou_w=0.4,
                               morph_w=0.2, eps=1e-6):
    return (bce_w * F.binary_cross_entropy_with_logits(logits, targets) + focal_w * focal_loss(logits, targets)
            + dice_w * _dice_term(logits, targets, eps) + boundary_w * boundary_aware_loss(logits, targets)
            + iou_w * iou_loss(logits, targets) + morph_w * morphological_loss(logits, targets))


def dice_iou_from_logits(logits, targets, thr: float = 0.5, eps: float = 1e-6):
    p_bin = (torch.sigmoid(logits) > thr).float()
    inter = (p_bin * targets).sum(dim=(1, 2, 3))
    union = (p_bin + targets - p_bin * targets).sum(dim=(1, 2, 3)) + eps
    dice = 2 * inter / (p_bin.sum(dim=(1, 2, 3)) + targets.sum(dim=(1, 2, 3)) + eps)
    return dice.detach().cpu().tolist(), (inter / union).detach().cpu().tolist(), p_bin


class FocalLoss(nn.Module):
    """cifake_binary_classifier.py:238-251 (pt = exp(−bce) form)."""

    def __init__(self, alpha: float = 1.0, gamma: float = 2.0, pos_weight=None):
        super().__init__()
        self.alpha, self.gamma, self.pos_weight = alpha, gamma, pos_weight

    def forward(self, inputs, targets):
        bce = F.binary_cross_entropy_with_logits(inputs, targets, pos_weight=self.pos_weight, reduction="none")
        return (self.alpha * (1 - torch.exp(-bce)) ** self.gamma * bce).mean()


def label_smoothing_loss(pred, target, smoothing: float = 0.1):
    target = target.float() * (1 - smoothing) + 0.5 * smoothing
    return F.binary_cross_entropy_with_logits(pred, target)


def mtl_loss(cls_logit, seg_logits, y_class, masks, has_mask, lam_seg: float = 1.0, enhanced: bool = False):
    """Train-step loss of the SID script (Siglip2sidafrozen.py:1377-1389): CE + λ·seg-loss on the samples that
    carry a mask."""
    loss = F.cross_entropy(cls_logit, y_class)
    if has_mask.any():
        seg = combined_segmentation_loss if enhanced else bce_dice_loss
        loss = loss + lam_seg * seg(seg_logits[has_mask], masks[has_mask])
    return loss


# ---------------------------------------------------------------------------------------------------------
# composed task models (reference L3 modules: encoder surface O + head)
# ---------------------------------------------------------------------------------------------------------
class _FlatHeadKeys(nn.Module):
    """The reference's task models own their head layers directly (``se.0.weight``, ``classifier.5.bias`` …), ours
    keep them in ``self.head``; checkpoints use the reference's names, so the ``head.`` level is dropped on
    ``state_dict()`` and re-inserted on ``load_state_dict()`` (cifake_binary_classifier.py:2089,
    train_fusion_head_only.py:110-122)."""

    def __init__(self):
        super().__init__()
        self._register_state_dict_hook(self._flat_out)
        self._register_load_state_dict_pre_hook(self._flat_in, with_module=True)

    @staticmethod
    def _flat_out(module, state_dict, prefix, local_metadata):
        ph = prefix + "head."
        for k in [k for k in state_dict if k.startswith(ph)]:
            state_dict[prefix + k[len(ph):]] = state_dict.pop(k)

    @staticmethod
    def _flat_in(module, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        own = {n.split(".")[0] for n, _ in module.head.named_parameters()} | \
              {n.split(".")[0] for n, _ in module.head.named_buffers()}
        for k in [k for k in state_dict if k.startswith(prefix) and k[len(prefix):].split(".")[0] in own]:
            state_dict[prefix + "head." + k[len(prefix):]] = state_dict.pop(k)


class FastBinaryClassifierHIP(_FlatHeadKeys):
    """`FastBinaryClassifier.forward` (cifake_binary_classifier.py:714-749) on the HIP encoder: bilinear resize to
    the model resolution if needed → `backbone.encode_image` → head."""

    def __init__(self, backbone: nn.Module, model_size: str = "small", dropout_rate: float = 0.1,
                 use_lightweight_attention: bool = True):
        super().__init__()
        self.backbone = backbone
        self.resolution = backbone.image_size
        self.feature_dim = backbone.embed_dim
        self.head = CifakeBinaryHead(self.feature_dim, model_size, dropout_rate, use_lightweight_attention)

    def forward(self, x, return_features: bool = False):
        if x.shape[-1] != self.resolution:
            x = F.interpolate(x, size=(self.resolution, self.resolution), mode="bilinear", align_corners=False)
        return self.head(self.backbone.encode_image(x), return_features)


class BinaryVideoClassifierHIP(_FlatHeadKeys):
    """`BinaryVideoClassifier.forward` (hidf_video_classifier.py:299-320): (B,T,C,H,W) → per-frame encoder →
    L2-norm → temporal mean → MLP → (B,) logits."""

    def __init__(self, vision_encoder: nn.Module, num_frames: int = 4, dropout_rate: float = 0.3):
        super().__init__()
        self.vision_encoder = vision_encoder
        self.feature_dim = vision_encoder.embed_dim
        self.num_frames = num_frames
        self.head = VideoBinaryHead(self.feature_dim, num_frames, dropout_rate)

    def forward(self, x):
        b, t, c, h, w = x.shape
        return self.head(self.vision_encoder.encode_image(x.view(b * t, c, h, w)), batch_size=b)


class ImageBinaryClassifierHIP(_FlatHeadKeys):
    """`BinaryClassifier` of the HiDF image track (simple_classifier.py:115-164; BASELINE config 3 = this head on
    so400m-patch14-384): bilinear resize to the model resolution if needed -> `backbone.encode_image` -> L2-norm -> head
    -> (B,) logits.  ``partially_unfreeze_backbone()`` is the script's default fine-tuning recipe (:483-496)."""

    def __init__(self, backbone: nn.Module):
        super().__init__()
        self.backbone = backbone
        self.resolution = backbone.image_size
        self.feature_dim = backbone.embed_dim
        self.head = ImageBinaryHead(self.feature_dim)

    def forward(self, x):
        if x.shape[-1] != self.resolution:
            x = F.interpolate(x, size=(self.resolution, self.resolution), mode="bilinear")
        return self.head(self.backbone.encode_image(x))

    def partially_unfreeze_backbone(self, last_blocks: int = 2) -> int:
        """simple_classifier.py:483-496: freeze the whole backbone, then re-enable every parameter whose open_clip/timm
        NAME contains one of ``blocks.<L-1>`` ... ``blocks.<L-last_blocks>``, ``ln_final``, ``norm`` (the script
        hard-codes ``blocks.23`` / ``blocks.22`` for its 24-block tower).  Substring matching is kept as is: ``norm`` also
        selects ``norm1`` / ``norm2`` of EVERY block, the final norm and the pooling head's norm, and ``blocks.2`` style
        prefixes behave as in the script.  Returns the number of re-enabled parameters (what the script prints)."""
        for prm in self.backbone.parameters():
            prm.requires_grad = False
        depth = self.backbone.visual.config.num_hidden_layers
        keys = [f"blocks.{depth - 1 - i}" for i in range(last_blocks)] + ["ln_final", "norm"]
        n = 0
        for name, prm in self.backbone.named_parameters():
            if any(k in name for k in keys):
                prm.requires_grad = True
                n += prm.numel()
        return n


class SEBinaryClassifierHIP(_FlatHeadKeys):
    """`BinaryClassifier.forward` of the fusion script (train_fusion_head_only.py:101-109): frozen encoder under
    no_grad, nearest-neighbour resize to the model size, SE gate + MLP."""

    def __init__(self, backbone: nn.Module):
        super().__init__()
        self.backbone = backbone
        self.img_size = backbone.image_size
        self.head = SEBinaryHead(backbone.embed_dim)

    def forward(self, x):
        with torch.no_grad():
            if x.shape[-1] != self.img_size:
                x = F.interpolate(x, size=(self.img_size, self.img_size))
            f = self.backbone.encode_image(x)
        return self.head(f)

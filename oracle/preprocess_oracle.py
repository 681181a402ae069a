"""CPU ORACLE for the GPU input pipeline (csrc/preprocess.hip).  TEST INFRASTRUCTURE ONLY.

Restates the reference's per-batch transform (cifake_binary_classifier.py:1791-1797,808-817):
    Resize(S, antialias=True) -> [MixUp lam*x + (1-lam)*x[index]] -> Normalize(mean=0.5, std=0.5)
and the patch gather of the patch-embedding convolution (TF:models/siglip/modeling_siglip.py:175-185: Conv2d(k=p, s=p,
'valid') == a GEMM over rows (b, gy, gx) and columns (c, ky, kx)).

Pin: the resize is torch's own CPU implementation of ``upsample_bilinear2d(antialias=True)`` — the kernel torchvision
``transforms.Resize(antialias=True)`` dispatches to for tensors, i.e. the reference's CPU transform (cifake…:1795-1797).
The reference's GPU transform is kornia's ``K.Resize(antialias=True)`` (a Gaussian pre-blur + interpolate); kornia is not
importable in this image, so equality with it is PARITY UNPINNED.  `triangle_resize_1d` below is the explicit-math form of the
same filter (aten/native/cpu/UpSampleKernel.cpp, _compute_indices_weights_aa), checked against torch in
tests/test_preprocess.py.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def aa_weights(out_size: int, in_size: int):
    """Per output index: (first tap, weights) of the antialiased triangle filter (fp32 arithmetic as in aten)."""
    scale = np.float32(in_size) / np.float32(out_size)
    support = scale if scale >= 1.0 else np.float32(1.0)
    invscale = np.float32(1.0) / scale if scale >= 1.0 else np.float32(1.0)
    taps = []
    for i in range(out_size):
        center = scale * np.float32(i + 0.5)
        lo = max(int(center - support + np.float32(0.5)), 0)
        hi = min(int(center + support + np.float32(0.5)), in_size)
        w = np.array([max(0.0, 1.0 - abs((j - center + 0.5) * invscale)) for j in range(lo, hi)], dtype=np.float64)
        taps.append((lo, (w / w.sum()).astype(np.float32)))
    return taps


def triangle_resize(x: torch.Tensor, size: int) -> torch.Tensor:
    """(B,C,H,W) float -> (B,C,size,size), explicit separable form (rows then columns)."""
    B, C, H, W = x.shape
    ty, tx = aa_weights(size, H), aa_weights(size, W)
    tmp = torch.zeros(B, C, H, size, dtype=torch.float32)
    for o, (lo, w) in enumerate(tx):
        tmp[..., o] = (x[..., lo:lo + len(w)].float() * torch.from_numpy(w)).sum(-1)
    out = torch.zeros(B, C, size, size, dtype=torch.float32)
    for o, (lo, w) in enumerate(ty):
        out[:, :, o, :] = (tmp[:, :, lo:lo + len(w), :] * torch.from_numpy(w)[:, None]).sum(2)
    return out


def gpu_transform(images: torch.Tensor, size: int, mean: float = 0.5, std: float = 0.5, mix_index=None,
                  lam: float = 1.0) -> torch.Tensor:
    """uint8 NHWC or float NCHW [0,1] -> (B,3,size,size) float32, via torch's own antialiased bilinear resize."""
    x = images.permute(0, 3, 1, 2).float() / 255.0 if images.dtype == torch.uint8 else images.float()
    if x.shape[-2:] != (size, size):
        x = F.interpolate(x, size=(size, size), mode="bilinear", antialias=True, align_corners=False)
    if mix_index is not None:
        x = lam * x + (1.0 - lam) * x[mix_index.long()]
    return (x - mean) / std


def patch_operand(pixels: torch.Tensor, patch: int) -> torch.Tensor:
    """(B,3,S,S) -> [B*g*g, round_up(3*p*p, 64)] with k = c*p*p + ky*p + kx, zero padded (g = S // p, trailing pixels dropped)."""
    B, C, S, _ = pixels.shape
    g = S // patch
    x = pixels[:, :, :g * patch, :g * patch].reshape(B, C, g, patch, g, patch).permute(0, 2, 4, 1, 3, 5)
    flat = x.reshape(B * g * g, C * patch * patch)
    kp = math.ceil(C * patch * patch / 64) * 64
    out = torch.zeros(B * g * g, kp, dtype=pixels.dtype)
    out[:, :flat.shape[1]] = flat
    return out

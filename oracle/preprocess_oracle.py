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


# ----------------------------------------------------------------------------------------------------------------
# augmentation branch of the video trainer's GPU transform (hidf_video_classifier.py:2868-2874):
#     K.Resize(S, antialias=True) -> K.RandomHorizontalFlip(p=0.5) -> K.RandomRotation(degrees=5, p=0.3)
#     -> K.ColorJitter(brightness=0.1, contrast=0.1, saturation=0.1, hue=0.05, p=0.3) -> K.Normalize(0.5, 0.5)
# kornia is not importable here, so this is an explicit restatement of the PUBLISHED operator definitions (torchvision's
# functional_tensor formulas for the colour operators, OpenCV / kornia get_rotation_matrix2d + bilinear warp with zero padding
# for the rotation), with the random draws passed in; equality with kornia's own code is PARITY UNPINNED.
# ----------------------------------------------------------------------------------------------------------------
def _grey(x):
    return 0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3]


def _rgb2hsv(img):
    r, g, b = img.unbind(dim=-3)
    maxc = torch.max(img, dim=-3).values
    minc = torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    cr_divisor = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / cr_divisor, (maxc - g) / cr_divisor, (maxc - b) / cr_divisor
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(img):
    h, s, v = img.unbind(dim=-3)
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int32) % 6
    p = torch.clamp(v * (1.0 - s), 0.0, 1.0)
    q = torch.clamp(v * (1.0 - f * s), 0.0, 1.0)
    t = torch.clamp(v * (1.0 - (1.0 - f) * s), 0.0, 1.0)
    mask = i.unsqueeze(dim=-3) == torch.arange(6).view(-1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v), dim=-3)
    a2 = torch.stack((t, v, v, q, p, p), dim=-3)
    a3 = torch.stack((p, p, t, v, v, q), dim=-3)
    a4 = torch.stack((a1, a2, a3), dim=-4)
    return torch.einsum("...ijk, ...xijk -> ...xjk", mask.to(dtype=img.dtype), a4)


def rotate_bilinear_zeros(x, cos_a, sin_a):
    """out(p) = x(M^-1 p), rotation by the angle (counter-clockwise positive, image coordinates) about ((S-1)/2, (S-1)/2),
    bilinear taps, zeros outside.  x: (3, S, S)."""
    S = x.shape[-1]
    ctr = 0.5 * (S - 1)
    ys, xs = torch.meshgrid(torch.arange(S, dtype=torch.float32), torch.arange(S, dtype=torch.float32), indexing="ij")
    dx, dy = xs - ctr, ys - ctr
    sx = cos_a * dx - sin_a * dy + ctr
    sy = sin_a * dx + cos_a * dy + ctr
    x0, y0 = torch.floor(sx), torch.floor(sy)
    fx, fy = sx - x0, sy - y0
    out = torch.zeros_like(x)
    for jy in (0, 1):
        for jx in (0, 1):
            yy, xx = (y0 + jy).long(), (x0 + jx).long()
            w = (fy if jy else 1.0 - fy) * (fx if jx else 1.0 - fx)
            ok = (yy >= 0) & (yy < S) & (xx >= 0) & (xx < S)
            v = x[:, yy.clamp(0, S - 1), xx.clamp(0, S - 1)]
            out = out + torch.where(ok, w, torch.zeros_like(w)) * v
    return out


def augment_transform(images: torch.Tensor, size: int, params, mean: float = 0.5, std: float = 0.5) -> torch.Tensor:
    """params: one dict per sample {flip, cos, sin, brightness, contrast, saturation, hue, order (list of 4 or None)}."""
    x = images.permute(0, 3, 1, 2).float() / 255.0 if images.dtype == torch.uint8 else images.float()
    if x.shape[-2:] != (size, size):
        x = F.interpolate(x, size=(size, size), mode="bilinear", antialias=True, align_corners=False)
    outs = []
    for b, pr in enumerate(params):
        im = x[b]
        if pr["flip"]:
            im = im.flip(-1)
        if not (pr["cos"] == 1.0 and pr["sin"] == 0.0):
            im = rotate_bilinear_zeros(im, pr["cos"], pr["sin"])
        im = im.unsqueeze(0)
        for op in (pr["order"] or []):
            if op == 0:
                im = (im * pr["brightness"]).clamp(0, 1)
            elif op == 1:
                m = _grey(im).mean()
                im = ((im - m) * pr["contrast"] + m).clamp(0, 1)
            elif op == 2:
                g = _grey(im)
                im = ((im - g) * pr["saturation"] + g).clamp(0, 1)
            else:
                hsv = _rgb2hsv(im)
                h = (hsv[:, 0] + pr["hue"]) % 1.0
                im = _hsv2rgb(torch.stack((h, hsv[:, 1], hsv[:, 2]), dim=1))
        outs.append(im[0])
    return (torch.stack(outs) - mean) / std

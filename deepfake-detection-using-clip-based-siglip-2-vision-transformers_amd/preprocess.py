"""GPU input pipeline (first slice): the reference's per-batch GPU transform

    gpu_transform = nn.Sequential(K.Resize(target_resolution, antialias=True),
                                  K.Normalize(mean=[0.5]*3, std=[0.5]*3))            cifake_binary_classifier.py:1791-1794
    images = gpu_transform(images);  [MixUp: lam*images + (1-lam)*images[index]]      cifake_binary_classifier.py:808-817

as ONE HIP kernel (``csrc/preprocess.hip``) that can also write its result directly in the layout the patch-embedding GEMM
reads (``to_patch_operand``), so that the fp32 (B,3,S,S) pixel tensor and the encoder's im2col pass both disappear:

    enc(patches=to_patch_operand(uint8_nhwc_batch, enc.config))      instead of      enc(pixel_values=gpu_transform(x))

Sources: decoded ``uint8`` images in NHWC (B,H,W,3) — what a JPEG decoder hands over — or ``float32`` NCHW in [0,1] (what the
reference's CPU transform ``Resize + ToTensor`` produces).  Resampling is torch's antialiased bilinear filter
(``F.interpolate(mode="bilinear", antialias=True)``, the arithmetic torchvision ``Resize(antialias=True)`` runs in the
reference's CPU transform, cifake…:1795-1797); kornia itself is not installed here, so equality with ``K.Resize`` is
"parity unpinned".  CUDA only: there is no CPU path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from . import lib as _lib
from .config import get_config


@dataclass
class PatchOperand:
    """The patch GEMM's A operand [B*gh*gw, round_up(3*p*p, 64)] in the encoder's compute dtype, plus its geometry."""
    data: torch.Tensor
    batch: int
    height: int
    width: int


def _source(images: torch.Tensor):
    if not images.is_cuda:
        raise RuntimeError("the GPU input pipeline runs on CUDA tensors only (no CPU path)")
    if images.dtype == torch.uint8:
        if images.dim() != 4 or images.shape[-1] != 3:
            raise ValueError(f"uint8 images must be NHWC (B,H,W,3), got {tuple(images.shape)}")
        return images.contiguous(), 1, images.shape[0], images.shape[1], images.shape[2]
    if images.dim() != 4 or images.shape[1] != 3:
        raise ValueError(f"float images must be NCHW (B,3,H,W) in [0,1], got {tuple(images.shape)}")
    return images.float().contiguous(), 0, images.shape[0], images.shape[2], images.shape[3]


def _mix(mix_index, B, dev):
    if mix_index is None:
        return None
    idx = mix_index.to(device=dev, dtype=torch.int32).contiguous()
    if idx.numel() != B:
        raise ValueError("mix_index must have one entry per image")
    return idx


def resize_normalize(images: torch.Tensor, size: int, mean: float = 0.5, std: float = 0.5,
                     mix_index: Optional[torch.Tensor] = None, lam: float = 1.0,
                     dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """(B,3,size,size) = Normalize(mean,std)(MixUp(Resize(size, antialias=True)(images))): the tensor the reference's GPU
    transform returns, in one pass."""
    lib = _lib.load()
    src, is_u8, B, Hs, Ws = _source(images)
    out = torch.empty((B, 3, size, size), device=src.device, dtype=dtype)
    code = _lib.SGL_DTYPE_BF16 if dtype == torch.bfloat16 else _lib.SGL_DTYPE_F32
    idx = _mix(mix_index, B, src.device)
    with torch.cuda.device(src.device):
        _lib.check(lib.sgl_op_preprocess(src.data_ptr(), is_u8, B, Hs, Ws, out.data_ptr(), code, size, 1, 3, 0, float(mean),
                                         float(std), _lib.ptr(idx), float(lam), _lib.current_stream_handle()),
                   "sgl_op_preprocess")
    return out


def to_patch_operand(images: torch.Tensor, config, size: Optional[int] = None, compute_dtype: str = "bf16",
                     mean: float = 0.5, std: float = 0.5, mix_index: Optional[torch.Tensor] = None,
                     lam: float = 1.0) -> PatchOperand:
    """Resize (antialias) + MixUp + Normalize straight into the patch-embedding GEMM's operand for ``config`` (its patch
    size and K padding); feed the result to ``SiglipVisionModelHIP(patches=...)`` / ``encode_image(patches=...)``."""
    lib = _lib.load()
    cfg = get_config(config)
    S = int(size or cfg.image_size)
    P = cfg.patch_size
    g = S // P
    Kp = (3 * P * P + 63) // 64 * 64
    src, is_u8, B, Hs, Ws = _source(images)
    dt = torch.bfloat16 if compute_dtype == "bf16" else torch.float32
    out = torch.empty((B * g * g, Kp), device=src.device, dtype=dt)
    code = _lib.SGL_DTYPE_BF16 if dt == torch.bfloat16 else _lib.SGL_DTYPE_F32
    idx = _mix(mix_index, B, src.device)
    with torch.cuda.device(src.device):
        _lib.check(lib.sgl_op_preprocess(src.data_ptr(), is_u8, B, Hs, Ws, out.data_ptr(), code, S, P, Kp, 1, float(mean),
                                         float(std), _lib.ptr(idx), float(lam), _lib.current_stream_handle()),
                   "sgl_op_preprocess")
    return PatchOperand(out, B, S, S)


# ---- augmentation branch (hidf_video_classifier.py:2868-2874) ----------------------------------------------------------
def sample_augmentation(batch: int, generator: Optional[torch.Generator] = None, p_flip: float = 0.5,
                        degrees: float = 5.0, p_rotation: float = 0.3, brightness: float = 0.1, contrast: float = 0.1,
                        saturation: float = 0.1, hue: float = 0.05, p_jitter: float = 0.3) -> list:
    """Host-side random draws of ``K.RandomHorizontalFlip(p=0.5)``, ``K.RandomRotation(degrees=5, p=0.3)`` and
    ``K.ColorJitter(0.1, 0.1, 0.1, 0.05, p=0.3)``: one dict per sample (the format ``augment_table`` packs and the oracle
    reads).  The RNG is the caller's ``torch.Generator``; nothing random happens on the device."""
    import math
    g = generator
    u = lambda n: torch.rand(n, generator=g)   # noqa: E731
    flip = u(batch) < p_flip
    rot = u(batch) < p_rotation
    ang = (u(batch) * 2 - 1) * degrees
    jit = u(batch) < p_jitter
    fb, fc, fs = (1 + (u(batch) * 2 - 1) * a for a in (brightness, contrast, saturation))
    fh = (u(batch) * 2 - 1) * hue
    out = []
    for b in range(batch):
        a = math.radians(float(ang[b])) if bool(rot[b]) else 0.0
        order = torch.randperm(4, generator=g).tolist() if bool(jit[b]) else None
        out.append(dict(flip=bool(flip[b]), cos=math.cos(a) if a else 1.0, sin=math.sin(a) if a else 0.0,
                        brightness=float(fb[b]), contrast=float(fc[b]), saturation=float(fs[b]), hue=float(fh[b]),
                        order=order))
    return out


def augment_table(params: list, device) -> torch.Tensor:
    """Pack per-sample parameters into the device table of ``sgl_aug_sample`` records (12 x 4 bytes each)."""
    import struct
    buf = bytearray()
    for pr in params:
        order = pr["order"] if pr["order"] else [-1, -1, -1, -1]
        buf += struct.pack("<7f5i", 1.0 if pr["flip"] else 0.0, pr["cos"], pr["sin"], pr["brightness"], pr["contrast"],
                           pr["saturation"], pr["hue"], *order, 0)
    return torch.frombuffer(buf, dtype=torch.uint8).to(device)


def augment_resize_normalize(images: torch.Tensor, size: int, params: list, mean: float = 0.5, std: float = 0.5,
                             dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """(B,3,size,size) = Normalize(ColorJitter(Rotation(Flip(Resize(images))))) with the given per-sample draws: the tensor
    the video trainer's augmenting GPU transform returns, in one pass over the pixels (plus a per-image mean pre-pass for
    the contrast operator)."""
    lib = _lib.load()
    src, is_u8, B, Hs, Ws = _source(images)
    if len(params) != B:
        raise ValueError("one augmentation record per image")
    out = torch.empty((B, 3, size, size), device=src.device, dtype=dtype)
    code = _lib.SGL_DTYPE_BF16 if dtype == torch.bfloat16 else _lib.SGL_DTYPE_F32
    tab = augment_table(params, src.device)
    gm = torch.empty(B, device=src.device, dtype=torch.float32)
    with torch.cuda.device(src.device):
        _lib.check(lib.sgl_op_preprocess_aug(src.data_ptr(), is_u8, B, Hs, Ws, out.data_ptr(), code, size, 1, 3, 0,
                                             float(mean), float(std), tab.data_ptr(), gm.data_ptr(),
                                             _lib.current_stream_handle()), "sgl_op_preprocess_aug")
    return out


def augment_to_patch_operand(images: torch.Tensor, config, params: list, size: Optional[int] = None,
                             compute_dtype: str = "bf16", mean: float = 0.5, std: float = 0.5) -> PatchOperand:
    """The augmenting transform written straight into the patch GEMM's operand (see ``to_patch_operand``)."""
    lib = _lib.load()
    cfg = get_config(config)
    S = int(size or cfg.image_size)
    P = cfg.patch_size
    g = S // P
    Kp = (3 * P * P + 63) // 64 * 64
    src, is_u8, B, Hs, Ws = _source(images)
    if len(params) != B:
        raise ValueError("one augmentation record per image")
    dt = torch.bfloat16 if compute_dtype == "bf16" else torch.float32
    out = torch.empty((B * g * g, Kp), device=src.device, dtype=dt)
    code = _lib.SGL_DTYPE_BF16 if dt == torch.bfloat16 else _lib.SGL_DTYPE_F32
    tab = augment_table(params, src.device)
    gm = torch.empty(B, device=src.device, dtype=torch.float32)
    with torch.cuda.device(src.device):
        _lib.check(lib.sgl_op_preprocess_aug(src.data_ptr(), is_u8, B, Hs, Ws, out.data_ptr(), code, S, P, Kp, 1,
                                             float(mean), float(std), tab.data_ptr(), gm.data_ptr(),
                                             _lib.current_stream_handle()), "sgl_op_preprocess_aug")
    return PatchOperand(out, B, S, S)


class GpuTransform(nn.Module):
    """``nn.Sequential(K.Resize(res, antialias=True), K.Normalize(0.5, 0.5))`` of the reference (cifake…:1791-1794,
    hidf_video_classifier.py:2874-2878) as one module; ``forward(images, mix_index=None, lam=1.0)``."""

    def __init__(self, resolution: int, mean: float = 0.5, std: float = 0.5, data_augmentation: bool = False,
                 generator: Optional[torch.Generator] = None):
        super().__init__()
        self.resolution, self.mean, self.std = int(resolution), float(mean), float(std)
        self.data_augmentation, self.generator = bool(data_augmentation), generator

    @torch.no_grad()
    def forward(self, images, mix_index=None, lam: float = 1.0):
        """``data_augmentation=True`` (hidf_video_classifier.py ``--data_augmentation``, :2866-2874) inserts flip / rotation /
        colour jitter between resize and normalize while the module is in training mode; draws come from ``generator``."""
        if self.data_augmentation and self.training:
            if mix_index is not None:
                raise ValueError("MixUp and the augmentation branch belong to different trainers; use one of them")
            params = sample_augmentation(images.shape[0], self.generator)
            return augment_resize_normalize(images, self.resolution, params, self.mean, self.std)
        return resize_normalize(images, self.resolution, self.mean, self.std, mix_index, lam)

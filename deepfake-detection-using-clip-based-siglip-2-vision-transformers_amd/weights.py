"""Closed-form seeded weight / input generator.

There are no checkpoints on the GPU box and no network, so every parity and bench run regenerates
identical fp32 weights from a seed: ``splitmix64(fnv1a(name) ^ mix(seed) + index)`` → 24-bit uniform in
[-1, 1) → scaled per tensor by fan-in.  Pure numpy integer arithmetic, so the build container and the GPU
box produce bit-identical tensors.  State-dict names are HuggingFace ``SiglipVisionModel`` names
(``TF:models/siglip/modeling_siglip.py:116-135,268-271,315-316,329-331,567,626-630``), which are the keys
the reference saves/loads under ``encoder.*`` (``Siglip2sidafrozen.py:1153-1175``).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .config import SiglipVisionConfig

_U64 = np.uint64
_MASK = (1 << 64) - 1


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _MASK
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + _U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def uniform_pm1(name: str, numel: int, seed: int = 0) -> np.ndarray:
    """numel fp32 values in [-1, 1), a pure function of (name, seed, index)."""
    base = (fnv1a64(name) ^ ((seed * 0xD1342543DE82EF95 + 0x2545F4914F6CDD1D) & _MASK)) & _MASK
    out = np.empty(numel, dtype=np.float32)
    chunk = 1 << 22
    for s in range(0, numel, chunk):
        e = min(numel, s + chunk)
        with np.errstate(over="ignore"):
            idx = np.arange(s, e, dtype=np.uint64) + _U64(base)
        z = _splitmix64(idx)
        u = (z >> _U64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))  # 24-bit, exact in fp32
        out[s:e] = u * np.float32(2.0) - np.float32(1.0)
    return out


def seeded_tensor(name: str, shape, bound: float, seed: int = 0, offset: float = 0.0) -> torch.Tensor:
    numel = int(np.prod(shape))
    v = uniform_pm1(name, numel, seed) * np.float32(bound)
    if offset:
        v = v + np.float32(offset)
    return torch.from_numpy(v.reshape(tuple(shape)))


def param_shapes(cfg: SiglipVisionConfig) -> dict[str, tuple]:
    """HF state-dict name → shape, in HF registration order."""
    d, i, p = cfg.hidden_size, cfg.intermediate_size, cfg.patch_size
    s: dict[str, tuple] = {}
    s["embeddings.patch_embedding.weight"] = (d, 3, p, p)
    s["embeddings.patch_embedding.bias"] = (d,)
    s["embeddings.position_embedding.weight"] = (cfg.num_positions, d)
    for l in range(cfg.num_hidden_layers):
        pre = f"encoder.layers.{l}."
        s[pre + "layer_norm1.weight"] = (d,)
        s[pre + "layer_norm1.bias"] = (d,)
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[pre + f"self_attn.{nm}.weight"] = (d, d)
            s[pre + f"self_attn.{nm}.bias"] = (d,)
        s[pre + "layer_norm2.weight"] = (d,)
        s[pre + "layer_norm2.bias"] = (d,)
        s[pre + "mlp.fc1.weight"] = (i, d)
        s[pre + "mlp.fc1.bias"] = (i,)
        s[pre + "mlp.fc2.weight"] = (d, i)
        s[pre + "mlp.fc2.bias"] = (d,)
    s["post_layernorm.weight"] = (d,)
    s["post_layernorm.bias"] = (d,)
    if cfg.vision_use_head:
        s["head.probe"] = (1, 1, d)
        s["head.attention.in_proj_weight"] = (3 * d, d)
        s["head.attention.in_proj_bias"] = (3 * d,)
        s["head.attention.out_proj.weight"] = (d, d)
        s["head.attention.out_proj.bias"] = (d,)
        s["head.layernorm.weight"] = (d,)
        s["head.layernorm.bias"] = (d,)
        s["head.mlp.fc1.weight"] = (i, d)
        s["head.mlp.fc1.bias"] = (i,)
        s["head.mlp.fc2.weight"] = (d, i)
        s["head.mlp.fc2.bias"] = (d,)
    return s


def _bound_for(name: str, shape: tuple) -> tuple[float, float]:
    """(bound, offset): Linear/Conv weights U(±sqrt(3/fan_in)) (unit-variance-preserving), LayerNorm
    γ = 1 ± 0.1, β = ± 0.05, biases ± 0.05, position table ± 0.1, probe ± 1."""
    if name.endswith("layer_norm1.weight") or name.endswith("layer_norm2.weight") or \
            name.endswith("layernorm.weight"):
        return 0.1, 1.0
    if name.endswith("probe"):
        return 1.0, 0.0
    if name.endswith("position_embedding.weight"):
        return 0.1, 0.0
    if name.endswith("bias"):
        return 0.05, 0.0
    fan_in = int(np.prod(shape[1:]))
    return math.sqrt(3.0 / fan_in), 0.0


def seeded_state_dict(cfg: SiglipVisionConfig, seed: int = 0) -> dict[str, torch.Tensor]:
    sd = {}
    for name, shape in param_shapes(cfg).items():
        bound, offset = _bound_for(name, shape)
        sd[name] = seeded_tensor(name, shape, bound, seed, offset)
    return sd


def seeded_pixels(batch: int, height: int, width: int, seed: int = 1234) -> torch.Tensor:
    """Synthetic ``pixel_values``: 2·U[0,1)−1, the range ``Normalize(0.5, 0.5)`` produces
    (reference ``cifake_binary_classifier.py:1793``, ``Siglip2sidafrozen.py:942-945``)."""
    return seeded_tensor("pixel_values", (batch, 3, height, width), 1.0, seed)

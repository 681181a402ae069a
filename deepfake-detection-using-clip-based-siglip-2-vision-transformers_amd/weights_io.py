"""Weight / checkpoint formats either side of the encoder path.

The reference stores and reloads the encoder under two naming schemes:

  HF        ``encoder.vision_model.encoder.layers.N.self_attn.q_proj.weight`` …   (Siglip2sidafrozen.py:753,1639)
  open_clip ``backbone.visual.trunk.blocks.N.attn.qkv.weight`` …                   (cifake_binary_classifier.py:625-638,
            timm ``VisionTransformer`` + ``AttentionPoolLatent`` names)             2089; train_fusion_head_only.py:110-122)

This module converts between them (fused ``qkv`` / ``kv`` split and merge, position table ``[1,N,D]`` vs ``[N,D]``) so
that ``OpenClipStyleEncoder.state_dict()`` / ``load_state_dict()`` speak the open_clip names and a checkpoint written by
either reference trainer loads here unchanged, and writes/reads the reference's ``.pt`` dictionary layout.

Pinning: the HF names are pinned by the golden fixtures (generated from ``transformers``).  ``timm``/``open_clip`` are
not installed in this image and the reference ships no backbone checkpoint, so the timm-side key list below is
"parity unpinned": it restates timm's ``VisionTransformer`` / ``AttentionPoolLatent`` parameter names as the reference
addresses them (``visual.trunk.blocks.N`` freezing patterns, ``_filter_state_for_model`` shape matching); the tests
pin the round trip and the tensor algebra (split/merge), not the names.

Loading is restricted to formats that execute nothing: safetensors, or ``torch.load(..., weights_only=True)``.
"""
from __future__ import annotations

import re
from collections import OrderedDict

import torch

from .config import SiglipVisionConfig

_BLOCK_MAP = [  # timm suffix, HF suffix
    ("norm1", "layer_norm1"), ("attn.proj", "self_attn.out_proj"), ("norm2", "layer_norm2"),
    ("mlp.fc1", "mlp.fc1"), ("mlp.fc2", "mlp.fc2"),
]
_HEAD_MAP = [
    ("attn_pool.proj", "head.attention.out_proj"), ("attn_pool.norm", "head.layernorm"),
    ("attn_pool.mlp.fc1", "head.mlp.fc1"), ("attn_pool.mlp.fc2", "head.mlp.fc2"),
]


def hf_to_timm(sd: dict, cfg: SiglipVisionConfig, prefix: str = "trunk.") -> "OrderedDict[str, torch.Tensor]":
    """HF ``SiglipVisionTransformer`` names (no ``vision_model.`` prefix) -> timm ``VisionTransformer`` names."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    P = prefix
    out[P + "pos_embed"] = sd["embeddings.position_embedding.weight"].unsqueeze(0)
    out[P + "patch_embed.proj.weight"] = sd["embeddings.patch_embedding.weight"]
    out[P + "patch_embed.proj.bias"] = sd["embeddings.patch_embedding.bias"]
    for l in range(cfg.num_hidden_layers):
        h, t = f"encoder.layers.{l}.", f"{P}blocks.{l}."
        for wb in ("weight", "bias"):
            out[t + f"attn.qkv.{wb}"] = torch.cat([sd[h + f"self_attn.{n}_proj.{wb}"] for n in ("q", "k", "v")], 0)
        for tn, hn in _BLOCK_MAP:
            for wb in ("weight", "bias"):
                out[t + f"{tn}.{wb}"] = sd[h + f"{hn}.{wb}"]
    out[P + "norm.weight"] = sd["post_layernorm.weight"]
    out[P + "norm.bias"] = sd["post_layernorm.bias"]
    if cfg.vision_use_head:
        d = cfg.hidden_size
        out[P + "attn_pool.latent"] = sd["head.probe"]
        w, b = sd["head.attention.in_proj_weight"], sd["head.attention.in_proj_bias"]
        out[P + "attn_pool.q.weight"], out[P + "attn_pool.q.bias"] = w[:d], b[:d]
        out[P + "attn_pool.kv.weight"], out[P + "attn_pool.kv.bias"] = w[d:], b[d:]
        for tn, hn in _HEAD_MAP:
            for wb in ("weight", "bias"):
                out[P + f"{tn}.{wb}"] = sd[f"{hn}.{wb}"]
    return out


def timm_to_hf(sd: dict, cfg: SiglipVisionConfig, prefix: str = "trunk.") -> "OrderedDict[str, torch.Tensor]":
    """Inverse of :func:`hf_to_timm`.  Keys outside ``prefix`` are ignored; missing keys raise ``KeyError``."""
    P = prefix
    d = cfg.hidden_size
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    out["embeddings.patch_embedding.weight"] = sd[P + "patch_embed.proj.weight"]
    out["embeddings.patch_embedding.bias"] = sd[P + "patch_embed.proj.bias"]
    pos = sd[P + "pos_embed"]
    out["embeddings.position_embedding.weight"] = pos.reshape(pos.shape[-2], pos.shape[-1])
    for l in range(cfg.num_hidden_layers):
        h, t = f"encoder.layers.{l}.", f"{P}blocks.{l}."
        out[h + "layer_norm1.weight"] = sd[t + "norm1.weight"]
        out[h + "layer_norm1.bias"] = sd[t + "norm1.bias"]
        for wb in ("weight", "bias"):
            qkv = sd[t + f"attn.qkv.{wb}"]
            if qkv.shape[0] != 3 * d:
                raise ValueError(f"{t}attn.qkv.{wb}: leading dim {qkv.shape[0]} != 3*{d}")
            for j, n in enumerate(("q", "k", "v")):
                out[h + f"self_attn.{n}_proj.{wb}"] = qkv[j * d:(j + 1) * d]
        for tn, hn in _BLOCK_MAP:
            if tn == "norm1":
                continue
            for wb in ("weight", "bias"):
                out[h + f"{hn}.{wb}"] = sd[t + f"{tn}.{wb}"]
    out["post_layernorm.weight"] = sd[P + "norm.weight"]
    out["post_layernorm.bias"] = sd[P + "norm.bias"]
    if cfg.vision_use_head:
        out["head.probe"] = sd[P + "attn_pool.latent"].reshape(1, 1, d)
        out["head.attention.in_proj_weight"] = torch.cat([sd[P + "attn_pool.q.weight"], sd[P + "attn_pool.kv.weight"]], 0)
        out["head.attention.in_proj_bias"] = torch.cat([sd[P + "attn_pool.q.bias"], sd[P + "attn_pool.kv.bias"]], 0)
        for tn, hn in _HEAD_MAP:
            for wb in ("weight", "bias"):
                out[f"{hn}.{wb}"] = sd[P + f"{tn}.{wb}"]
    return out


def detect_format(keys) -> str:
    """'timm' (open_clip image tower), 'hf' (transformers) or 'unknown', from the key names alone."""
    ks = list(keys)
    if any(re.search(r"(^|\.)trunk\.blocks\.\d+\.attn\.qkv\.weight$", k) for k in ks):
        return "timm"
    if any(re.search(r"encoder\.layers\.\d+\.self_attn\.q_proj\.weight$", k) for k in ks):
        return "hf"
    return "unknown"


def encoder_state_from_checkpoint(sd: dict, cfg: SiglipVisionConfig) -> "OrderedDict[str, torch.Tensor]":
    """Pull the vision-encoder tensors out of any of the reference's model state dicts and return them under HF names
    (no prefix): ``encoder.vision_model.*`` (SigLIP2MTL), ``vision_model.*`` (bare HF), ``backbone.visual.trunk.*``
    (CiFake / HiDF / fusion-head trainers; ``backbone.text.*`` is dropped as train_fusion_head_only.py:113-116 does)."""
    fmt = detect_format(sd.keys())
    if fmt == "timm":
        k0 = next(k for k in sd if re.search(r"trunk\.blocks\.0\.attn\.qkv\.weight$", k))
        prefix = k0[: k0.index("blocks.0.")]
        return timm_to_hf(sd, cfg, prefix)
    if fmt == "hf":
        k0 = next(k for k in sd if k.endswith("encoder.layers.0.self_attn.q_proj.weight"))
        prefix = k0[: k0.index("encoder.layers.0.")]
        return OrderedDict((k[len(prefix):], v) for k, v in sd.items() if k.startswith(prefix))
    raise ValueError("state dict holds neither HF SiglipVisionModel nor open_clip/timm vision-tower keys")


def load_state_file(path: str) -> dict:
    """``.safetensors`` or a ``.pt``/``.pth`` written by the reference trainers; nothing in the file is executed.
    A ``.pt`` holding ``{'model_state_dict': …}`` (Siglip2sidafrozen.py:1638-1647, cifake_binary_classifier.py:2087-
    2095) yields that inner dict."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and "model_state_dict" in obj:
        return obj["model_state_dict"]
    return obj


def save_checkpoint(path: str, model: torch.nn.Module, optimizer=None, scheduler=None, **extra) -> None:
    """The reference's checkpoint dictionary (Siglip2sidafrozen.py:1638-1647): ``model_state_dict``,
    ``optimizer_state_dict``, ``scheduler_state_dict`` + scalar extras (``epoch``, ``metrics`` …); a path ending in
    ``.safetensors`` stores the model tensors only (``best_model.safetensors`` of train_fusion_head_only.py:6)."""
    msd = {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    if path.endswith(".safetensors"):
        from safetensors.torch import save_file
        save_file({k: v.clone() for k, v in msd.items()}, path)  # clone: safetensors refuses shared storage
        return
    ck = {"model_state_dict": msd}
    if optimizer is not None:
        ck["optimizer_state_dict"] = optimizer.state_dict()
    if scheduler is not None:
        ck["scheduler_state_dict"] = scheduler.state_dict()
    ck.update(extra)
    torch.save(ck, path)


def load_checkpoint(path: str, model: torch.nn.Module, optimizer=None, scheduler=None, strict: bool = True) -> dict:
    """Inverse of :func:`save_checkpoint`; returns the non-tensor extras (``epoch``, ``metrics`` …)."""
    if path.endswith(".safetensors"):
        model.load_state_dict(load_state_file(path), strict=strict)
        return {}
    ck = torch.load(path, map_location="cpu", weights_only=True)
    model.load_state_dict(ck["model_state_dict"], strict=strict)
    if optimizer is not None and "optimizer_state_dict" in ck:
        optimizer.load_state_dict(ck["optimizer_state_dict"])
    if scheduler is not None and "scheduler_state_dict" in ck:
        scheduler.load_state_dict(ck["scheduler_state_dict"])
    return {k: v for k, v in ck.items() if not k.endswith("_state_dict")}

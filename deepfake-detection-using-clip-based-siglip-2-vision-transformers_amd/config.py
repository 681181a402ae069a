"""Encoder configuration for the MI355X-native SigLIP-2 ViT path.

Mirrors the fields of HuggingFace ``SiglipVisionConfig`` that the reference reads
(``Siglip2sidafrozen.py:771`` uses ``encoder.config.hidden_size``) and the open_clip model names the
reference's scripts select (``cifake_binary_classifier.py:547-572``, ``hidf_video_classifier.py:2810``,
``train_fusion_head_only.py:82``).  Only what the hot path needs lives here.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict


@dataclass
class SiglipVisionConfig:
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    num_channels: int = 3
    image_size: int = 224
    patch_size: int = 16
    layer_norm_eps: float = 1e-6
    hidden_act: str = "gelu_pytorch_tanh"
    attention_dropout: float = 0.0
    vision_use_head: bool = True

    def __post_init__(self):
        if self.hidden_size % self.num_attention_heads:
            raise ValueError(
                f"embed_dim must be divisible by num_heads (got `embed_dim`: {self.hidden_size} and "
                f"`num_heads`: {self.num_attention_heads}).")
        if self.hidden_act != "gelu_pytorch_tanh":
            raise ValueError("only gelu_pytorch_tanh is on the reference's path")
        if self.attention_dropout != 0.0:
            raise ValueError("attention_dropout must be 0.0 (reference config value)")
        if self.num_channels != 3:
            raise ValueError("num_channels must be 3")

    # --- derived ---------------------------------------------------------------------------
    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def native_grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_positions(self) -> int:
        return self.native_grid ** 2

    def grid_for(self, height: int, width: int) -> tuple[int, int]:
        return height // self.patch_size, width // self.patch_size

    def to_dict(self) -> dict:
        return asdict(self)

    # --- algorithmic work (SURVEY.md §8d / BASELINE.md §2) -------------------------------------
    def fwd_flops_per_image(self, height: int | None = None, width: int | None = None) -> float:
        h = height or self.image_size
        w = width or self.image_size
        gh, gw = self.grid_for(h, w)
        n, d, i, l, p = gh * gw, self.hidden_size, self.intermediate_size, self.num_hidden_layers, self.patch_size
        layer = 8 * n * d * d + 4 * n * n * d + 4 * n * d * i
        patch = 2 * n * 3 * p * p * d
        pool = 4 * n * d * d + 4 * d * d + 4 * n * d + 4 * d * i if self.vision_use_head else 0
        return float(l * layer + patch + pool)

    def train_flops_per_image(self, height: int | None = None, width: int | None = None) -> float:
        return 3.0 * self.fwd_flops_per_image(height, width)

    def num_params(self) -> int:
        d, i, l, p = self.hidden_size, self.intermediate_size, self.num_hidden_layers, self.patch_size
        block = 4 * (d * d + d) + 2 * d * i + i + d + 4 * d
        emb = d * 3 * p * p + d + self.num_positions * d
        head = (d + 3 * d * d + 3 * d + d * d + d + 2 * d + 2 * d * i + i + d) if self.vision_use_head else 0
        return l * block + emb + 2 * d + head


# HF checkpoint names (reference: Siglip2sidafrozen.py:1732 default google/siglip2-large-patch16-384) and
# open_clip names (reference: cifake_binary_classifier.py:547-572, hidf_video_classifier.py:2810).
NAMED_CONFIGS: dict[str, dict] = {
    # HF-style ids
    "google/siglip2-base-patch16-224": dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                            num_attention_heads=12, image_size=224, patch_size=16),
    "google/siglip-base-patch16-224": dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                           num_attention_heads=12, image_size=224, patch_size=16),
    "google/siglip2-large-patch16-384": dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                                             num_attention_heads=16, image_size=384, patch_size=16),
    "google/siglip2-so400m-patch14-384": dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27,
                                              num_attention_heads=16, image_size=384, patch_size=14),
    "google/siglip2-so400m-patch16-512": dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27,
                                              num_attention_heads=16, image_size=512, patch_size=16),
    # open_clip-style names
    "ViT-B-16-SigLIP": dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                            num_attention_heads=12, image_size=224, patch_size=16),
    "ViT-B-16-SigLIP-256": dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                num_attention_heads=12, image_size=256, patch_size=16),
    "ViT-B-16-SigLIP-384": dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                num_attention_heads=12, image_size=384, patch_size=16),
    "ViT-L-16-SigLIP-384": dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                                num_attention_heads=16, image_size=384, patch_size=16),
    "ViT-SO400M-14-SigLIP-384": dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27,
                                     num_attention_heads=16, image_size=384, patch_size=14),
    "ViT-SO400M-16-SigLIP2-512": dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27,
                                      num_attention_heads=16, image_size=512, patch_size=16),
    # short aliases used by bench.py / tests
    "so400m-patch14-384": dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27,
                               num_attention_heads=16, image_size=384, patch_size=14),
    "base-patch16-224": dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                             num_attention_heads=12, image_size=224, patch_size=16),
    "large-patch16-384": dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                              num_attention_heads=16, image_size=384, patch_size=16),
    # parity-test shapes (SURVEY.md §8c golden list)
    "tiny": dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4,
                 image_size=32, patch_size=16),
    "hostile": dict(hidden_size=144, intermediate_size=538, num_hidden_layers=2, num_attention_heads=2,
                    image_size=42, patch_size=14),
    "so400m-1layer": dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=1,
                          num_attention_heads=16, image_size=384, patch_size=14),
    "base-1layer": dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=1,
                        num_attention_heads=12, image_size=224, patch_size=16),
}


def get_config(name_or_cfg) -> SiglipVisionConfig:
    if isinstance(name_or_cfg, SiglipVisionConfig):
        return name_or_cfg
    if isinstance(name_or_cfg, dict):
        return SiglipVisionConfig(**name_or_cfg)
    if name_or_cfg not in NAMED_CONFIGS:
        raise KeyError(f"unknown SigLIP vision config '{name_or_cfg}'; known: {sorted(NAMED_CONFIGS)}")
    return SiglipVisionConfig(**NAMED_CONFIGS[name_or_cfg])


def isqrt_exact(n: int) -> int:
    r = math.isqrt(n)
    if r * r != n:
        raise ValueError(f"Cannot reshape {n} tokens into square grid. Try using square input images.")
    return r

"""Host-side mirror of the two call surfaces the reference uses for the encoder, backed by the
hand-written gfx950 kernels in ``libsiglip_hip.so``.

Surface H (HuggingFace-style; ``Siglip2sidafrozen.py:753,757-768,771,787-793``):
    ``SiglipVisionModelHIP.from_pretrained(path_or_name)``, ``model(pixel_values=..., output_hidden_states=True,
    interpolate_pos_encoding=True)`` → ``.pooler_output`` / ``.last_hidden_state`` / ``.hidden_states``;
    ``.config.hidden_size``; ``.vision_model.embeddings`` / ``.vision_model.encoder.layers[i]`` for freezing;
    ``state_dict`` keys equal HF ``SiglipVisionModel`` keys.
Surface O (open_clip-style; ``cifake_binary_classifier.py:625-638,721``, ``hidf_video_classifier.py:259-273,307``):
    ``create_model_and_transforms(name, pretrained, device)`` → ``(model, None, preprocess)``;
    ``model.encode_image(x)`` → un-normalised (B, D); ``model.embed_dim``.

PyTorch is plumbing only (device memory, streams, autograd graph).  All arithmetic of the encoder runs in the
HIP library; there is no CPU fallback — calling the model on a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import lib as _lib
from .config import SiglipVisionConfig, get_config, NAMED_CONFIGS
from .weights import seeded_state_dict


# ---------------------------------------------------------------------------------------------------------
# parameter holders: same module tree / names as HF SiglipVisionModel (TF:modeling_siglip.py:116-135,
# 253-271,309-316,324-331,560-571,622-631) so state_dict keys and the reference's freezing code line up
# ---------------------------------------------------------------------------------------------------------
class _Affine(nn.Module):
    def __init__(self, out_f, in_shape=None, bias=True):
        super().__init__()
        shape = (out_f,) if in_shape is None else (out_f, *in_shape)
        self.weight = nn.Parameter(torch.zeros(shape))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_f))


class _AttnParams(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.k_proj = _Affine(d, (d,))
        self.v_proj = _Affine(d, (d,))
        self.q_proj = _Affine(d, (d,))
        self.out_proj = _Affine(d, (d,))


class _MLPParams(nn.Module):
    def __init__(self, d, i):
        super().__init__()
        self.fc1 = _Affine(i, (d,))
        self.fc2 = _Affine(d, (i,))


class _LayerParams(nn.Module):
    def __init__(self, d, i):
        super().__init__()
        self.layer_norm1 = _Affine(d)
        self.self_attn = _AttnParams(d)
        self.layer_norm2 = _Affine(d)
        self.mlp = _MLPParams(d, i)


class _EncoderParams(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([_LayerParams(cfg.hidden_size, cfg.intermediate_size)
                                     for _ in range(cfg.num_hidden_layers)])


class _EmbeddingParams(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.patch_embedding = _Affine(cfg.hidden_size, (3, cfg.patch_size, cfg.patch_size))
        self.position_embedding = _Affine(cfg.num_positions, (cfg.hidden_size,), bias=False)


class _MHAParams(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.zeros(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = _Affine(d, (d,))


class _HeadParams(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d = cfg.hidden_size
        self.probe = nn.Parameter(torch.zeros(1, 1, d))
        self.attention = _MHAParams(d)
        self.layernorm = _Affine(d)
        self.mlp = _MLPParams(d, cfg.intermediate_size)


@dataclass
class VisionModelOutput:
    """Fields of HF ``BaseModelOutputWithPooling`` the reference reads (``Siglip2sidafrozen.py:788-793``)."""
    last_hidden_state: torch.Tensor
    pooler_output: Optional[torch.Tensor]
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None

    def __getitem__(self, i):
        return (self.last_hidden_state, self.pooler_output, self.hidden_states)[i]


# ---------------------------------------------------------------------------------------------------------
# autograd bridge
# ---------------------------------------------------------------------------------------------------------
class _EncoderFn(torch.autograd.Function):
    """One autograd node for the whole encoder.  Inputs: pixels + every parameter (fixed order, see
    ``SiglipVisionModelHIP._flat_params``).  Outputs: pooled, last_hidden_state, then the requested
    hidden-state taps as separate tensors (so unused taps cost no gradient memory)."""

    @staticmethod
    def forward(ctx, mod, train, interp, want_pooled, tap_ids, pixel_values, *params):
        L, D = mod.config.num_hidden_layers, mod.config.hidden_size
        lib = _lib.load()
        px = pixel_values
        if px.dim() != 4 or px.shape[1] != 3:
            raise ValueError(f"pixel_values must be (B,3,H,W), got {tuple(px.shape)}")
        if px.dtype != torch.float32:
            px = px.float()
        channels_last = 0
        if not px.is_contiguous():
            if px.is_contiguous(memory_format=torch.channels_last):
                channels_last = 1
            else:
                px = px.contiguous()
        B, _, H, W = px.shape
        P = mod.config.patch_size
        if H < P or W < P:
            raise ValueError(f"image size ({H},{W}) is smaller than patch_size {P}")
        gh, gw = H // P, W // P
        N, M = gh * gw, B * gh * gw
        if (gh, gw) != (mod.config.native_grid, mod.config.native_grid) and not interp:
            raise ValueError(f"Input image size ({H}*{W}) doesn't match model native "
                             f"({mod.config.image_size}*{mod.config.image_size}); pass interpolate_pos_encoding=True")
        dev = px.device
        shadow, weights = mod._prepared(dev)
        sizes = mod._sizes(B, H, W, train)
        keep_all = train or len(tap_ids) > 0
        hs_slots = L + 1 if keep_all else 2
        hs = torch.empty((hs_slots, M, D), dtype=torch.float32, device=dev)
        last = torch.empty((M, D), dtype=torch.float32, device=dev)
        pooled = torch.empty((B, D), dtype=torch.float32, device=dev) if want_pooled else None
        saved = torch.empty(sizes[1], dtype=torch.uint8, device=dev) if train else None
        ws = None if train else torch.empty(sizes[2], dtype=torch.uint8, device=dev)
        # first block that can receive a gradient (frozen prefix, Siglip2sidafrozen.py:757-768); 0 when the embeddings train
        first = 0
        if train:
            trainable = {grp for (grp, _), p in zip(mod._flat_names, params) if p.requires_grad}
            if "emb" not in trainable:
                first = min([int(g_[5:]) for g_ in trainable if g_.startswith("layer")] or [L])
        st = lib.sgl_forward_ex(mod._ctx, C.byref(weights), shadow.data_ptr(), px.data_ptr(), channels_last, B, H, W,
                                1 if interp else 0, hs.data_ptr(), hs_slots, last.data_ptr(), _lib.ptr(pooled),
                                _lib.ptr(saved), sizes[1] if train else 0, _lib.ptr(ws), 0 if train else sizes[2],
                                first, _lib.current_stream_handle())
        _lib.check(st, "sgl_forward_ex", mod._ctx)
        if train:
            ctx.mod, ctx.saved, ctx.hs, ctx.geom, ctx.interp = mod, saved, hs, (B, H, W, N, M), interp
            ctx.tap_ids, ctx.want_pooled, ctx.weights, ctx.shadow = tap_ids, want_pooled, weights, shadow
        outs = [pooled if want_pooled else last.new_zeros(()), last.view(B, N, D)]
        outs += [hs[i].view(B, N, D) for i in tap_ids]
        dead = [] if want_pooled else [outs[0]]
        if train:
            # hidden_states[i] only feeds gradient to the embeddings and to blocks < i: with those frozen
            # (Siglip2sidafrozen.py:757-768) the tap's gradient would be computed by the consumer (the SID decoder's
            # tap projections) and then dropped here, so tell autograd not to ask for it
            if "emb" not in trainable:
                dead += [t for i, t in zip(tap_ids, outs[2:]) if i <= first]
        if dead:
            ctx.mark_non_differentiable(*dead)
        return tuple(outs)

    @staticmethod
    def backward(ctx, d_pooled, d_last, *d_taps):
        mod = ctx.mod
        lib = _lib.load()
        cfg = mod.config
        L, D = cfg.num_hidden_layers, cfg.hidden_size
        B, H, W, N, M = ctx.geom
        dev = ctx.hs.device
        needs = ctx.needs_input_grad[6:]
        names = mod._flat_names
        params = mod._flat_params()

        def prep(g):
            if g is None:
                return None
            g = g.float() if g.dtype != torch.float32 else g
            return g.contiguous()

        d_pooled = prep(d_pooled) if ctx.want_pooled else None
        d_last = prep(d_last)
        tap_grads = [None] * (L + 1)
        for i, g in zip(ctx.tap_ids, d_taps):
            if g is not None:
                g = prep(g)
                tap_grads[i] = g if tap_grads[i] is None else tap_grads[i] + g

        # gradient buffers: one flat fp32 bucket per group (embeddings, each block, post-LN + head)
        grads_out = [None] * len(params)
        groups: dict[str, list[int]] = {}
        for idx, (grp, field) in enumerate(names):
            if needs[idx]:
                groups.setdefault(grp, []).append(idx)
        buckets: dict[str, torch.Tensor] = {}
        # q/k/v weight (and bias) gradients back to back: the C side then runs them as one dW GEMM / one column sum
        rank = {"q_w": 0, "k_w": 1, "v_w": 2, "q_b": 3, "k_b": 4, "v_b": 5}
        for grp, idxs in groups.items():
            idxs.sort(key=lambda i: (rank.get(names[i][1], 6), i))
            total = sum((params[i].numel() + 3) // 4 * 4 for i in idxs)   # every tensor 16-byte aligned
            flat = torch.zeros(total, dtype=torch.float32, device=dev)
            off = 0
            for i in idxs:
                n = params[i].numel()
                grads_out[i] = flat[off:off + n].view(params[i].shape)
                off += (n + 3) // 4 * 4
            buckets[grp] = flat

        gl = (_lib.SglLayerPtrs * max(L, 1))()
        g = _lib.SglGrads()
        g.layers = C.cast(gl, C.POINTER(_lib.SglLayerPtrs))
        g.accumulate = 0
        for idx, (grp, field) in enumerate(names):
            p = None if grads_out[idx] is None else grads_out[idx].data_ptr()
            if grp.startswith("layer"):
                setattr(gl[int(grp[5:])], field, p)
            else:
                setattr(g, field, p)

        train_emb = "emb" in groups
        layer_ids = sorted(int(k[5:]) for k in groups if k.startswith("layer"))
        first = layer_ids[0] if layer_ids else L
        stop = 0 if train_emb else first
        sizes = mod._sizes(B, H, W, True)
        ws = torch.empty(sizes[2], dtype=torch.uint8, device=dev)
        stream = _lib.current_stream_handle()
        wts, shadow = ctx.weights, ctx.shadow
        reducer = mod._grad_reducer
        st = lib.sgl_backward_begin(mod._ctx, C.byref(wts), shadow.data_ptr(), C.byref(g), B, H, W, ctx.hs.data_ptr(),
                                    _lib.ptr(d_last), _lib.ptr(d_pooled), _lib.ptr(tap_grads[L]),
                                    ctx.saved.data_ptr(), sizes[1], ws.data_ptr(), sizes[2], stream)
        _lib.check(st, "sgl_backward_begin", mod._ctx)
        if reducer is not None and "head" in buckets:
            reducer.reduce_bucket(buckets["head"])
        for l in range(L - 1, stop - 1, -1):
            need_dx = 1 if (l > stop or train_emb) else 0
            st = lib.sgl_backward_layer(mod._ctx, C.byref(wts), shadow.data_ptr(), C.byref(g), l, B, H, W,
                                        ctx.hs.data_ptr(), _lib.ptr(tap_grads[l]), need_dx, ctx.saved.data_ptr(),
                                        sizes[1], ws.data_ptr(), sizes[2], stream)
            _lib.check(st, f"sgl_backward_layer[{l}]", mod._ctx)
            if reducer is not None and f"layer{l}" in buckets:
                reducer.reduce_bucket(buckets[f"layer{l}"])
        if train_emb:
            st = lib.sgl_backward_embed(mod._ctx, C.byref(wts), C.byref(g), B, H, W, 1 if ctx.interp else 0,
                                        ctx.saved.data_ptr(), sizes[1], ws.data_ptr(), sizes[2], stream)
            _lib.check(st, "sgl_backward_embed", mod._ctx)
            if reducer is not None:
                reducer.reduce_bucket(buckets["emb"])
        if reducer is not None:
            reducer.finish()
        ctx.saved = ctx.hs = None
        return (None, None, None, None, None, None, *grads_out)


# ---------------------------------------------------------------------------------------------------------
# surface H
# ---------------------------------------------------------------------------------------------------------
def _hf_state_dict_hook(module, state_dict, prefix, local_metadata):
    """state_dict(): parameters are registered without the ``vision_model.`` level; add it on the way out."""
    for k in [k for k in state_dict if k.startswith(prefix) and not k.startswith(prefix + "vision_model.")]:
        state_dict[prefix + "vision_model." + k[len(prefix):]] = state_dict.pop(k)


def _hf_load_pre_hook(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
    """load_state_dict(): strip the ``vision_model.`` level (also when this module is nested in a parent)."""
    pv = prefix + "vision_model."
    for k in [k for k in state_dict if k.startswith(pv)]:
        state_dict[prefix + k[len(pv):]] = state_dict.pop(k)


class SiglipVisionModelHIP(nn.Module):
    """Drop-in for ``transformers.SiglipVisionModel`` on the reference's path (see module docstring)."""

    def __init__(self, config, compute_dtype: str = "bf16"):
        super().__init__()
        self.config = get_config(config)
        if compute_dtype not in ("bf16", "fp32"):
            raise ValueError("compute_dtype must be 'bf16' or 'fp32'")
        self.compute_dtype = compute_dtype
        cfg = self.config
        self.embeddings = _EmbeddingParams(cfg)
        self.encoder = _EncoderParams(cfg)
        self.post_layernorm = _Affine(cfg.hidden_size)
        self.use_head = bool(cfg.vision_use_head)
        if self.use_head:
            self.head = _HeadParams(cfg)
        self._ctx = None
        self._shadow = None
        self._shadow_key = None
        self._weights_struct = None
        self._weights_keep = None
        self._weights_key = None
        self._size_cache: dict = {}
        self._grad_reducer = None
        self._gradient_checkpointing = False
        self._flat_names = self._build_names()
        # checkpoints keep transformers' key names (``vision_model.encoder.layers.N…``, Siglip2sidafrozen.py:1639)
        self._register_state_dict_hook(_hf_state_dict_hook)
        self._register_load_state_dict_pre_hook(_hf_load_pre_hook)

    # ---- HF surface ------------------------------------------------------------------------------------
    @property
    def vision_model(self):
        """transformers-4.x layout alias used by the reference's freezing code
        (``Siglip2sidafrozen.py:757,762``): ``encoder.vision_model.embeddings`` / ``.encoder.layers``."""
        return self

    @classmethod
    def from_pretrained(cls, name_or_path: str, compute_dtype: str = "bf16", seed: int = 0):
        """Local directory (``config.json`` + ``model.safetensors``), a ``.safetensors`` file next to a
        ``config.json``, or a known config name.  There is no network: a bare name gives the closed-form
        seeded initialisation of that architecture (``weights.seeded_state_dict``)."""
        if os.path.isdir(name_or_path):
            with open(os.path.join(name_or_path, "config.json")) as f:
                raw = json.load(f)
            raw = raw.get("vision_config", raw)
            fields = SiglipVisionConfig.__dataclass_fields__
            cfg = SiglipVisionConfig(**{k: v for k, v in raw.items() if k in fields})
            model = cls(cfg, compute_dtype)
            from safetensors.torch import load_file
            model.load_state_dict(load_file(os.path.join(name_or_path, "model.safetensors")))
            return model
        if name_or_path in NAMED_CONFIGS:
            model = cls(get_config(name_or_path), compute_dtype)
            model.load_state_dict(seeded_state_dict(model.config, seed))
            return model
        raise OSError(f"{name_or_path} is neither a local checkpoint directory nor a known config name "
                      f"(no network access); known: {sorted(NAMED_CONFIGS)}")

    def gradient_checkpointing_enable(self, **_):
        """Accepted for interface parity (``Siglip2sidafrozen.py:1195-1196``).  Activations for a 64-image
        so400m batch (≈52 GB) fit the 288 GB of HBM3E, so nothing is recomputed."""
        self._gradient_checkpointing = True

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """Accepts HF names with or without the ``vision_model.`` prefix, or an open_clip/timm vision tower
        (``…visual.trunk.blocks.N.attn.qkv.weight``, converted by ``weights_io.timm_to_hf``)."""
        from . import weights_io
        if weights_io.detect_format(state_dict.keys()) == "timm":
            state_dict = weights_io.encoder_state_from_checkpoint(state_dict, self.config)
        return super().load_state_dict(dict(state_dict), strict=strict, **kw)

    @torch.compiler.disable   # the reference torch.compile()s its models (cifake…:1888, hidf…:2922): Dynamo must not
    # trace into the ctypes calls; the encoder is one opaque eager region and the graph breaks cleanly around it
    def forward(self, pixel_values, output_hidden_states: bool = False, interpolate_pos_encoding: bool = False,
                hidden_state_ids=None, **_):
        if pixel_values.device.type != "cuda":
            raise RuntimeError("SiglipVisionModelHIP runs only on an AMD GPU through libsiglip_hip.so "
                               "(no CPU fallback); move the model and pixel_values to 'cuda'")
        L = self.config.num_hidden_layers
        if hidden_state_ids is not None:
            tap_ids = tuple(int(i) % (L + 1) for i in hidden_state_ids)
        elif output_hidden_states:
            tap_ids = tuple(range(L + 1))
        else:
            tap_ids = ()
        params = self._flat_params()
        train = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        outs = _EncoderFn.apply(self, train, bool(interpolate_pos_encoding), self.use_head, tap_ids, pixel_values,
                                *params)
        pooled = outs[0] if self.use_head else None
        hs = tuple(outs[2:]) if tap_ids else None
        if hs is not None and hidden_state_ids is None and output_hidden_states:
            pass
        return VisionModelOutput(last_hidden_state=outs[1], pooler_output=pooled, hidden_states=hs)

    # ---- plumbing ----------------------------------------------------------------------------------------
    def _build_names(self):
        names = [("emb", "patch_w"), ("emb", "patch_b"), ("emb", "pos")]
        for l in range(self.config.num_hidden_layers):
            names += [(f"layer{l}", f) for f in _lib.LAYER_FIELDS]
        names += [("head", "post_ln_w"), ("head", "post_ln_b")]
        if self.use_head:
            names += [("head", f) for f in _lib.HEAD_FIELDS]
        return names

    def _flat_params(self):
        e = self.embeddings
        ps = [e.patch_embedding.weight, e.patch_embedding.bias, e.position_embedding.weight]
        for lyr in self.encoder.layers:
            a, m = lyr.self_attn, lyr.mlp
            ps += [lyr.layer_norm1.weight, lyr.layer_norm1.bias, a.q_proj.weight, a.q_proj.bias, a.k_proj.weight,
                   a.k_proj.bias, a.v_proj.weight, a.v_proj.bias, a.out_proj.weight, a.out_proj.bias,
                   lyr.layer_norm2.weight, lyr.layer_norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias]
        ps += [self.post_layernorm.weight, self.post_layernorm.bias]
        if self.use_head:
            h = self.head
            ps += [h.probe, h.attention.in_proj_weight, h.attention.in_proj_bias, h.attention.out_proj.weight,
                   h.attention.out_proj.bias, h.layernorm.weight, h.layernorm.bias, h.mlp.fc1.weight, h.mlp.fc1.bias,
                   h.mlp.fc2.weight, h.mlp.fc2.bias]
        return ps

    def _ensure_ctx(self):
        if self._ctx is None:
            lib = _lib.load()
            cfg = self.config
            c = _lib.SglConfig(cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.num_attention_heads,
                               cfg.patch_size, cfg.native_grid, cfg.layer_norm_eps,
                               _lib.SGL_DTYPE_BF16 if self.compute_dtype == "bf16" else _lib.SGL_DTYPE_F32,
                               1 if self.use_head else 0)
            ctx = lib.sgl_create(C.byref(c))
            if not ctx:
                raise _lib.SglError(f"sgl_create: unsupported configuration {cfg}")
            self._ctx = ctx
        return self._ctx

    def _sizes(self, B, H, W, train):
        key = (B, H, W, bool(train))
        if key not in self._size_cache:
            lib = _lib.load()
            a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
            st = lib.sgl_query_sizes(self._ensure_ctx(), B, H, W, 1 if train else 0, C.byref(a), C.byref(b), C.byref(c))
            _lib.check(st, "sgl_query_sizes", self._ctx)
            self._size_cache[key] = (a.value, b.value, c.value)
        return self._size_cache[key]

    def _build_weights_struct(self, params):
        L = self.config.num_hidden_layers
        for (grp, field), p in zip(self._flat_names, params):
            # the 4-D patch-conv weight may be channels_last after model.to(memory_format=torch.channels_last)
            # (Siglip2sidafrozen.py:1191): only the shadow refresh reads it, through a contiguous copy (_prepared)
            strided_ok = field == "patch_w" and p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last)
            if p.dtype != torch.float32 or not (p.is_contiguous() or strided_ok):
                raise RuntimeError("encoder master parameters must be contiguous fp32 (the HIP path keeps its own "
                                   "bf16 shadows); do not call .half()/.bfloat16() on the encoder")
        layers = (_lib.SglLayerPtrs * max(L, 1))()
        w = _lib.SglWeights()
        w.layers = C.cast(layers, C.POINTER(_lib.SglLayerPtrs))
        for (grp, field), p in zip(self._flat_names, params):
            if grp.startswith("layer"):
                setattr(layers[int(grp[5:])], field, p.data_ptr())
            else:
                setattr(w, field, p.data_ptr())
        return w, layers

    def _prepared(self, dev):
        """(shadow arena, weights struct), refreshed when any master parameter changed (optimizer step,
        load_state_dict, EMA swap of ``param.data`` — ``cifake_binary_classifier.py:227-236``)."""
        lib = _lib.load()
        self._ensure_ctx()
        params = self._flat_params()
        if params[0].device != dev:
            raise RuntimeError(f"model is on {params[0].device}, input on {dev}")
        ptr_key = tuple(p.data_ptr() for p in params)
        if self._weights_struct is None or self._weights_key != ptr_key:
            self._weights_struct, self._weights_keep = self._build_weights_struct(params)
            self._weights_key = ptr_key
        # one (pointers, versions) key per block and one for everything else: only what changed is re-cast, so a
        # frozen-prefix run (Siglip2sidafrozen.py:757-768) refreshes its 6 trainable blocks, not all 27
        L = self.config.num_hidden_layers
        keys = [[] for _ in range(L + 1)]
        for (grp, _), p in zip(self._flat_names, params):
            keys[int(grp[5:]) if grp.startswith("layer") else L].append((p.data_ptr(), p._version))
        keys = [tuple(k) for k in keys]
        fresh = self._shadow is None or self._shadow.device != dev or self._shadow_key is None
        if fresh or self._shadow_key != keys:
            nbytes = self._sizes(1, self.config.patch_size, self.config.patch_size, False)[0]
            if self._shadow is None or self._shadow.device != dev or self._shadow.numel() < nbytes:
                self._shadow = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                fresh = True
            if fresh:
                dirty, glob = None, 1
            else:
                dirty = bytes(1 if keys[l] != self._shadow_key[l] else 0 for l in range(L))
                glob = 1 if keys[L] != self._shadow_key[L] else 0
            pw = params[[f for _, f in self._flat_names].index("patch_w")]
            if glob and not pw.is_contiguous():
                self._patch_w_dense = pw.detach().contiguous()          # kept alive until the next refresh
                self._weights_struct.patch_w = self._patch_w_dense.data_ptr()
            st = lib.sgl_prepare_weights_dirty(self._ctx, C.byref(self._weights_struct), self._shadow.data_ptr(),
                                               self._shadow.numel(), dirty, glob, _lib.current_stream_handle())
            _lib.check(st, "sgl_prepare_weights_dirty", self._ctx)
            self._shadow_key = keys
        return self._shadow, self._weights_struct

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._shadow = None
        self._shadow_key = None
        self._weights_struct = None
        return out

    def __del__(self):
        try:
            if self._ctx is not None and _lib._lib is not None:
                _lib._lib.sgl_destroy(self._ctx)
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------------
# surface O
# ---------------------------------------------------------------------------------------------------------
class OpenClipStyleEncoder(nn.Module):
    """``open_clip`` image-tower surface: ``encode_image(x)`` returns the attention-pooled, un-normalised
    embedding (callers L2-normalise themselves: ``cifake_binary_classifier.py:728``)."""

    def __init__(self, config, compute_dtype: str = "bf16"):
        super().__init__()
        self.visual = SiglipVisionModelHIP(config, compute_dtype)
        self.embed_dim = self.visual.config.hidden_size
        self.image_size = self.visual.config.image_size
        # checkpoints keep open_clip's key names (``visual.trunk.blocks.N.attn.qkv.weight``): the reference saves and
        # reloads ``backbone.visual.trunk.*`` (cifake_binary_classifier.py:2089, train_fusion_head_only.py:110-122)
        self._register_state_dict_hook(self._timm_state_dict_hook)
        self._register_load_state_dict_pre_hook(self._timm_load_pre_hook, with_module=True)

    @staticmethod
    def _timm_state_dict_hook(module, state_dict, prefix, local_metadata):
        from . import weights_io
        pv = prefix + "visual.vision_model."
        hf = {k[len(pv):]: state_dict.pop(k) for k in [k for k in state_dict if k.startswith(pv)]}
        for k, v in weights_io.hf_to_timm(hf, module.visual.config, "trunk.").items():
            state_dict[prefix + "visual." + k] = v

    @staticmethod
    def _timm_load_pre_hook(module, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                            error_msgs):
        from . import weights_io
        pt = prefix + "visual.trunk."
        if any(k.startswith(pt) for k in state_dict):
            timm = {k[len(prefix + "visual."):]: state_dict[k] for k in state_dict if k.startswith(pt)}
            try:
                hf = weights_io.timm_to_hf(timm, module.visual.config, "trunk.")
            except (KeyError, ValueError) as e:   # incomplete / mis-shaped tower: report it the way torch reports keys
                error_msgs.append(f"open_clip vision tower under '{pt}' cannot be converted: {e!r}")
                return
            for k in list(timm):
                state_dict.pop(prefix + "visual." + k)
            for k, v in hf.items():
                state_dict[prefix + "visual." + k] = v
        for k in [k for k in state_dict if k.startswith(prefix + "text.") or k == prefix + "logit_scale"
                  or k == prefix + "logit_bias"]:
            state_dict.pop(k)  # the text tower is never used on this path (train_fusion_head_only.py:115)

    _HF_TO_TIMM = [("embeddings.patch_embedding.", "trunk.patch_embed.proj."),
                   ("embeddings.position_embedding.weight", "trunk.pos_embed"),
                   ("post_layernorm.", "trunk.norm."), ("head.probe", "trunk.attn_pool.latent"),
                   ("head.attention.out_proj.", "trunk.attn_pool.proj."), ("head.attention.", "trunk.attn_pool."),
                   ("head.layernorm.", "trunk.attn_pool.norm."), ("head.mlp.", "trunk.attn_pool.mlp."),
                   (".layer_norm1.", ".norm1."), (".layer_norm2.", ".norm2."), (".self_attn.out_proj.", ".attn.proj."),
                   (".self_attn.", ".attn."), ("encoder.layers.", "trunk.blocks.")]

    def named_parameters(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True):
        """Parameter NAMES in open_clip/timm style (``visual.trunk.blocks.23.norm1.weight`` …): the reference selects what
        to unfreeze by substring (``'blocks.23'``, ``'norm'``, simple_classifier.py:489-493).  q/k/v stay separate tensors
        (``…attn.q_proj.weight``); the fused ``attn.qkv`` layout exists only in ``state_dict()``."""
        root = prefix + ("." if prefix else "") + "visual."
        for name, p in super().named_parameters(prefix=prefix, recurse=recurse, remove_duplicate=remove_duplicate):
            if name.startswith(root):
                tail = name[len(root):]
                for a, b in self._HF_TO_TIMM:
                    tail = tail.replace(a, b)
                name = root + tail
            yield name, p

    def encode_image(self, x, normalize: bool = False):
        out = self.visual(pixel_values=x, interpolate_pos_encoding=False)
        f = out.pooler_output
        if normalize:
            f = f / f.norm(dim=-1, keepdim=True)
        return f

    def forward(self, image):
        return self.encode_image(image)


def _preprocess_factory(image_size: int):
    """Resize(bilinear) + Normalize(0.5, 0.5) on an already-decoded float tensor in [0,1] (C,H,W)."""
    def preprocess(img: torch.Tensor) -> torch.Tensor:
        x = img.unsqueeze(0) if img.dim() == 3 else img
        if x.shape[-1] != image_size or x.shape[-2] != image_size:
            x = torch.nn.functional.interpolate(x, size=(image_size, image_size), mode="bilinear", align_corners=False)
        x = (x - 0.5) / 0.5
        return x[0] if img.dim() == 3 else x
    return preprocess


def create_model_and_transforms(model_name: str, pretrained: Optional[str] = None, device="cuda",
                                compute_dtype: str = "bf16", seed: int = 0):
    """Signature of ``open_clip.create_model_and_transforms`` as the reference calls it
    (``cifake_binary_classifier.py:625-629``).  ``pretrained`` may be a local checkpoint directory; the
    reference's ``'webli'`` tag needs the network, so it (like ``None``) selects seeded random init."""
    cfg = get_config(model_name)
    model = OpenClipStyleEncoder(cfg, compute_dtype)
    if pretrained and os.path.isdir(pretrained):
        from safetensors.torch import load_file
        model.visual.load_state_dict(load_file(os.path.join(pretrained, "model.safetensors")))
    else:
        model.visual.load_state_dict(seeded_state_dict(cfg, seed))
    model = model.to(device)
    pre = _preprocess_factory(cfg.image_size)
    return model, pre, pre

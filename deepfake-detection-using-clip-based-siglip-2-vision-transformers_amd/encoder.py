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
import warnings
import weakref
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

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
# PyTorch custom ops over the C ABI (SURVEY.md §8b "Who calls it"): torch.ops.siglip_hip.encoder_fwd / encoder_bwd,
# with fake (meta) implementations and a registered autograd formula, so that torch.compile(fullgraph=True) of the
# SURROUNDING model (cifake_binary_classifier.py:1888, hidf_video_classifier.py:2922) traces straight through the
# encoder call without a graph break.  The ops are thin: they allocate outputs through PyTorch and make the ctypes calls.
# ---------------------------------------------------------------------------------------------------------
_MODULES: "weakref.WeakValueDictionary[int, SiglipVisionModelHIP]" = weakref.WeakValueDictionary()
_NEXT_HANDLE = [1]


def _module_of(handle: int) -> "SiglipVisionModelHIP":
    mod = _MODULES.get(int(handle))
    if mod is None:
        raise RuntimeError(f"siglip_hip: encoder handle {handle} is not alive (module was deleted)")
    return mod


def _geometry(mod, pixel_values, layout=0, img_h=0, img_w=0):
    if layout == 2:   # ready patch-major operand [B*N, Kp] (preprocess.to_patch_operand); geometry travels beside it
        P = mod.config.patch_size
        gh, gw = img_h // P, img_w // P
        kp = (3 * P * P + 63) // 64 * 64
        if pixel_values.dim() != 2 or pixel_values.shape[1] != kp or gh * gw == 0 or pixel_values.shape[0] % (gh * gw):
            raise ValueError(f"patch operand must be (B*{gh * gw}, {kp}), got {tuple(pixel_values.shape)}")
        B = pixel_values.shape[0] // (gh * gw)
        return B, img_h, img_w, gh * gw, B * gh * gw, (gh, gw)
    if pixel_values.dim() != 4 or pixel_values.shape[1] != 3:
        raise ValueError(f"pixel_values must be (B,3,H,W), got {tuple(pixel_values.shape)}")
    B, _, H, W = pixel_values.shape
    P = mod.config.patch_size
    if H < P or W < P:
        raise ValueError(f"image size ({H},{W}) is smaller than patch_size {P}")
    gh, gw = H // P, W // P
    return B, H, W, gh * gw, B * gh * gw, (gh, gw)


def _taper(order, max_chunks):
    """Cut the completion-ordered group list into at most ``max_chunks`` runs whose lengths shrink towards the END: the
    exchange of the last chunk overlaps nothing (backward is over when it starts), the first has the whole backward to
    hide behind.  Boundaries, counted from the end, follow j(j+1)/2 (28 groups, 8 chunks -> 6,6,4,4,3,3,1,1); the small
    embeddings group rides with the block it follows."""
    tail = [order[-1]] if len(order) > 1 and order[-1] == "emb" else []
    body = order[:len(order) - len(tail)]
    n, c = len(body), max(1, min(max_chunks, len(body)))
    tri = c * (c + 1) // 2
    cuts = sorted({n - min(n, max(j, round(n * j * (j + 1) / 2 / tri))) for j in range(1, c)} | {0, n})
    runs = [body[a:b] for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    if tail:
        if runs:
            runs[-1] = runs[-1] + tail
        else:
            runs = [tail]
    return runs


@torch.library.custom_op("siglip_hip::encoder_fwd", mutates_args=())
def encoder_fwd(pixel_values: torch.Tensor, params: Sequence[torch.Tensor], handle: int, train: bool, interp: bool,
                want_pooled: bool, tap_ids: Sequence[int], first_trainable: int, layout: int, img_h: int,
                img_w: int) -> List[torch.Tensor]:
    """sgl_forward_slots.  Returns [pooled (B,D) or empty, last_hidden_state (B,N,D), one (B,N,D) tensor per entry of
    tap_ids (distinct, ascending), saved (uint8 activation arena, empty when not training), hs_rest (the hidden-state
    slots nobody asked for: [n, B*N, D])].  No output aliases another."""
    mod = _module_of(handle)
    cfg = mod.config
    L, D = cfg.num_hidden_layers, cfg.hidden_size
    lib = _lib.load()
    px = pixel_values
    channels_last = 0
    if layout == 2:
        want = torch.bfloat16 if mod.compute_dtype == "bf16" else torch.float32
        if px.dtype != want or not px.is_contiguous():
            raise ValueError(f"patch operand must be contiguous {want} (the encoder's compute dtype)")
        channels_last = 2
    else:
        if px.dtype != torch.float32:
            px = px.float()
        if not px.is_contiguous():
            if px.is_contiguous(memory_format=torch.channels_last):
                channels_last = 1
            else:
                px = px.contiguous()
    B, H, W, N, M, grid = _geometry(mod, px, layout, img_h, img_w)
    if grid != (cfg.native_grid, cfg.native_grid) and not interp:
        raise ValueError(f"Input image size ({H}*{W}) doesn't match model native "
                         f"({cfg.image_size}*{cfg.image_size}); pass interpolate_pos_encoding=True")
    dev = px.device
    with torch.cuda.device(dev):
        shadow, weights = mod._prepared(dev)
        sizes = mod._sizes(B, H, W, train)
        taps = [torch.empty((B, N, D), dtype=torch.float32, device=dev) for _ in tap_ids]
        tapset = {int(t): i for i, t in enumerate(tap_ids)}
        n_rest = (L + 1 - len(tapset)) if train else min(2, L + 1 - len(tapset))
        hs_rest = torch.empty((n_rest, M, D), dtype=torch.float32, device=dev)
        slots = (_lib._fp * (L + 1))()
        k = 0
        for l in range(L + 1):
            if l in tapset:
                slots[l] = taps[tapset[l]].data_ptr()
            elif train:
                slots[l] = hs_rest[k].data_ptr()
                k += 1
            else:
                slots[l] = hs_rest[l & 1].data_ptr()     # inference: ping-pong (neighbours differ in parity)
        last = torch.empty((B, N, D), dtype=torch.float32, device=dev)
        pooled = torch.empty((B, D) if want_pooled else (0,), dtype=torch.float32, device=dev)
        saved = torch.empty(sizes[1] if train else 0, dtype=torch.uint8, device=dev)
        ws = None if train else mod._workspace(sizes[2], dev)
        st = lib.sgl_forward_slots(mod._ctx, C.byref(weights), shadow.data_ptr(), px.data_ptr(), channels_last, B, H, W,
                                   1 if interp else 0, slots, last.data_ptr(),
                                   pooled.data_ptr() if want_pooled else None, saved.data_ptr() if train else None,
                                   sizes[1] if train else 0, _lib.ptr(ws), 0 if train else sizes[2],
                                   int(first_trainable), _lib.current_stream_handle())
        _lib.check(st, "sgl_forward_slots", mod._ctx)
    if train:
        mod._note_forward(saved)
    return [pooled, last, *taps, saved, hs_rest]


@encoder_fwd.register_fake
def _(pixel_values, params, handle, train, interp, want_pooled, tap_ids, first_trainable, layout, img_h, img_w):
    mod = _module_of(handle)
    cfg = mod.config
    L, D = cfg.num_hidden_layers, cfg.hidden_size
    B, H, W, N, M, _ = _geometry(mod, pixel_values, layout, img_h, img_w)
    if not all(isinstance(v, int) for v in (B, H, W)):
        raise RuntimeError("siglip_hip::encoder_fwd needs static image shapes under torch.compile (dynamic=False)")
    new = pixel_values.new_empty
    n_rest = (L + 1 - len(tap_ids)) if train else min(2, L + 1 - len(tap_ids))
    saved_bytes = mod._sizes(B, H, W, True)[1] if train else 0
    return [new((B, D) if want_pooled else (0,), dtype=torch.float32), new((B, N, D), dtype=torch.float32),
            *[new((B, N, D), dtype=torch.float32) for _ in tap_ids], new((saved_bytes,), dtype=torch.uint8),
            new((n_rest, M, D), dtype=torch.float32)]


@torch.library.custom_op("siglip_hip::encoder_bwd", mutates_args=())
def encoder_bwd(grads: Sequence[Optional[torch.Tensor]], taps: Sequence[torch.Tensor], saved: torch.Tensor,
                hs_rest: torch.Tensor, params: Sequence[torch.Tensor], handle: int, image_hw: Sequence[int],
                interp: bool, want_pooled: bool, tap_ids: Sequence[int], needs: Sequence[bool]) -> List[torch.Tensor]:
    """sgl_backward_begin_p -> sgl_backward_layer_p (L-1 ... first trainable block) -> sgl_backward_embed.
    grads = [d pooled, d last_hidden_state, d tap...] (None = no gradient).  Returns the flat fp32 gradient chunks of
    ``SiglipVisionModelHIP._bucket_layout(needs)`` (the DDP all-reduce units); the autograd formula slices the
    per-parameter views out of them outside the op, so no output of the op aliases another."""
    mod = _module_of(handle)
    lib = _lib.load()
    cfg = mod.config
    L, D = cfg.num_hidden_layers, cfg.hidden_size
    H, W = int(image_hw[0]), int(image_hw[1])
    dev = saved.device
    B = int(taps[0].shape[0]) if len(taps) else int(hs_rest.shape[1] // ((H // cfg.patch_size) * (W // cfg.patch_size)))
    N = (H // cfg.patch_size) * (W // cfg.patch_size)
    M = B * N
    mod._check_backward(saved)
    names = mod._flat_names
    params = mod._flat_params()

    def prep(g):
        if g is None:
            return None
        g = g.float() if g.dtype != torch.float32 else g
        return g.contiguous()

    d_pooled = prep(grads[0]) if want_pooled else None
    d_last = prep(grads[1])
    tap_grads = [None] * (L + 1)
    hs_ptr = [None] * (L + 1)
    k = 0
    tapset = {int(t): i for i, t in enumerate(tap_ids)}
    for l in range(L + 1):
        if l in tapset:
            hs_ptr[l] = taps[tapset[l]].data_ptr()
            tap_grads[l] = prep(grads[2 + tapset[l]])
        else:
            hs_ptr[l] = hs_rest[k].data_ptr()
            k += 1

    with torch.cuda.device(dev):
        chunks, groups = mod._bucket_layout(needs)
        flats = mod._alloc_buckets(chunks, dev)
        grads_out = [None] * len(params)
        chunk_of, last_of = {}, {}
        for ci, (total, members, entries) in enumerate(chunks):
            for i, off, n in entries:
                grads_out[i] = flats[ci][off:off + n]
            for grp in members:
                chunk_of[grp] = ci
            last_of[ci] = members[-1]
        if d_pooled is None and "head" in groups:     # the C side skips the pooling head (and post-LN when d_last is None too)
            for i in groups["head"]:
                grads_out[i].zero_()
        reducer = mod._grad_reducer
        # gradient accumulation (micro-steps under reducer.no_sync() came before): what must be exchanged is the SUM in
        # .grad, which autograd forms after this op returns, so nothing is handed over from here (see _encoder_backward)
        overlapped = reducer is not None and not mod._accumulating(needs)

        def group_done(grp):
            """Hand a chunk to the reducer once its last group (in completion order) is complete."""
            if overlapped and grp in chunk_of and last_of[chunk_of[grp]] == grp:
                reducer.reduce_bucket(flats[chunk_of[grp]])

        gl = (_lib.SglLayerPtrs * max(L, 1))()
        g = _lib.SglGrads()
        g.layers = C.cast(gl, C.POINTER(_lib.SglLayerPtrs))
        g.accumulate = 0
        for idx, (grp, field) in enumerate(names):
            ptr = None if grads_out[idx] is None else grads_out[idx].data_ptr()
            if grp.startswith("layer"):
                setattr(gl[int(grp[5:])], field, ptr)
            else:
                setattr(g, field, ptr)

        train_emb = "emb" in groups
        layer_ids = sorted(int(k_[5:]) for k_ in groups if k_.startswith("layer"))
        first = layer_ids[0] if layer_ids else L
        stop = 0 if train_emb else first
        sizes = mod._sizes(B, H, W, True)
        ws = mod._workspace(sizes[2], dev)
        stream = _lib.current_stream_handle()
        shadow, wts = mod._shadow, mod._weights_struct
        st = lib.sgl_backward_begin_p(mod._ctx, C.byref(wts), shadow.data_ptr(), C.byref(g), B, H, W, hs_ptr[L],
                                      _lib.ptr(d_last), _lib.ptr(d_pooled), _lib.ptr(tap_grads[L]),
                                      saved.data_ptr(), sizes[1], ws.data_ptr(), sizes[2], stream)
        _lib.check(st, "sgl_backward_begin_p", mod._ctx)
        group_done("head")
        for l in range(L - 1, stop - 1, -1):
            need_dx = 1 if (l > stop or train_emb) else 0
            st = lib.sgl_backward_layer_p(mod._ctx, C.byref(wts), shadow.data_ptr(), C.byref(g), l, B, H, W, hs_ptr[l],
                                          _lib.ptr(tap_grads[l]), need_dx, saved.data_ptr(), sizes[1], ws.data_ptr(),
                                          sizes[2], stream)
            _lib.check(st, f"sgl_backward_layer_p[{l}]", mod._ctx)
            group_done(f"layer{l}")
        if train_emb:
            st = lib.sgl_backward_embed(mod._ctx, C.byref(wts), C.byref(g), B, H, W, 1 if interp else 0,
                                        saved.data_ptr(), sizes[1], ws.data_ptr(), sizes[2], stream)
            _lib.check(st, "sgl_backward_embed", mod._ctx)
            group_done("emb")
        if overlapped:
            reducer.finish()
    return flats


@encoder_bwd.register_fake
def _(grads, taps, saved, hs_rest, params, handle, image_hw, interp, want_pooled, tap_ids, needs):
    chunks, _ = _module_of(handle)._bucket_layout(needs)
    return [saved.new_empty((total,), dtype=torch.float32) for total, _, _ in chunks]


def _encoder_setup_context(ctx, inputs, output):
    pixel_values, params, handle, train, interp, want_pooled, tap_ids, first_trainable, layout, img_h, img_w = inputs
    ctx.set_materialize_grads(False)
    ctx.handle, ctx.interp, ctx.want_pooled, ctx.tap_ids = handle, interp, want_pooled, list(tap_ids)
    ctx.image_hw = [int(img_h), int(img_w)] if layout == 2 else [int(pixel_values.shape[2]), int(pixel_values.shape[3])]
    ctx.ntaps, ctx.nparams, ctx.train = len(tap_ids), len(params), train
    if train:
        # saving the taps (outputs) makes autograd's version counter catch a consumer's in-place edit of a hidden state
        ctx.save_for_backward(*output[2:], *params)
        # hidden_states[i] only feeds gradient to the embeddings and to blocks < i: with those frozen
        # (Siglip2sidafrozen.py:757-768) the tap's gradient would be computed by the consumer (the SID decoder's tap
        # projections) and then dropped here, so tell autograd not to ask for it
        dead = [t for i, t in zip(tap_ids, output[2:2 + len(tap_ids)]) if first_trainable > 0 and i <= first_trainable]
        dead += [output[-2], output[-1]] + ([] if want_pooled else [output[0]])
        ctx.mark_non_differentiable(*dead)


def _encoder_backward(ctx, grads):
    if not ctx.train:
        raise RuntimeError("siglip_hip::encoder_fwd was run with train=False: nothing was saved for backward")
    saved_t = ctx.saved_tensors
    nt = ctx.ntaps
    taps, saved, hs_rest, params = list(saved_t[:nt]), saved_t[nt], saved_t[nt + 1], list(saved_t[nt + 2:])
    needs = [bool(n) for n in ctx.needs_input_grad[1]] if isinstance(ctx.needs_input_grad[1], (list, tuple)) \
        else [p.requires_grad for p in params]
    mod = _module_of(ctx.handle)
    if mod._grad_reducer is not None and mod._grad_reducer.syncing() and mod._accumulating(needs):
        # the exchange of the accumulated gradients runs when the whole autograd pass (every AccumulateGrad) is over
        layout, _ = mod._bucket_layout(needs)
        plist = mod._flat_params()
        work = [(total, [(plist[i], off, n) for i, off, n in entries]) for total, _, entries in layout]
        reducer = mod._grad_reducer
        torch.autograd.Variable._execution_engine.queue_callback(lambda: reducer.reduce_accumulated(work))
    flats = torch.ops.siglip_hip.encoder_bwd(list(grads[:2 + nt]), taps, saved, hs_rest, params, ctx.handle, ctx.image_hw,
                                             ctx.interp, ctx.want_pooled, ctx.tap_ids, needs)
    chunks, _ = _module_of(ctx.handle)._bucket_layout(needs)
    pgrads: List[Optional[torch.Tensor]] = [None] * len(params)
    for flat, (_, _, entries) in zip(flats, chunks):
        for i, off, n in entries:
            pgrads[i] = flat[off:off + n].view(params[i].shape)
    # pytree structure of the inputs: an EMPTY int list is a list node, a non-empty one a leaf (torch/_library/autograd.py)
    return None, pgrads, None, None, None, None, ([] if len(ctx.tap_ids) == 0 else None), None, None, None, None


encoder_fwd.register_autograd(_encoder_backward, setup_context=_encoder_setup_context)


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


def vision_tower_only(state_dict: dict) -> dict:
    """Keep the vision tower of a full SiglipModel checkpoint: keys under ``vision_model.`` when that level exists (the
    published google/siglip* files also carry ``text_model.*``, ``logit_scale``, ``logit_bias``), everything otherwise."""
    if any(k.startswith("vision_model.") for k in state_dict):
        return {k: v for k, v in state_dict.items() if k.startswith("vision_model.")}
    return {k: v for k, v in state_dict.items()
            if not (k.startswith("text_model.") or k in ("logit_scale", "logit_bias"))}


class SiglipVisionModelHIP(nn.Module):
    """Drop-in for ``transformers.SiglipVisionModel`` on the reference's path (see module docstring)."""

    def __init__(self, config, compute_dtype: str = "bf16"):
        super().__init__()
        self.config = get_config(config)
        # "bf16": the benchmarked mode (bf16 MFMA operands, fp32 accumulate / residual stream / statistics);
        # "fp32": strict reference arithmetic (plain fp32 FMAs, no matrix cores): tightest parity, slow;
        # "bf16x3": strict mode on the matrix cores (every GEMM as one bf16 MFMA GEMM over hi/lo-split operands, fp32
        #           accumulate; ~2^-17 relative per product): the north-star "logits within 1e-3" at MFMA speed
        if compute_dtype not in ("bf16", "fp32", "bf16x3"):
            raise ValueError("compute_dtype must be 'bf16', 'fp32' or 'bf16x3'")
        self.compute_dtype = compute_dtype
        cfg = self.config
        self.embeddings = _EmbeddingParams(cfg)
        self.encoder = _EncoderParams(cfg)
        self.post_layernorm = _Affine(cfg.hidden_size)
        self.use_head = bool(cfg.vision_use_head)
        if self.use_head:
            self.head = _HeadParams(cfg)
        self._grad_reducer = None
        self._gradient_checkpointing = False
        self._flat_names = self._build_names()
        self._reset_runtime_state()
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
    def from_pretrained(cls, name_or_path: str, compute_dtype: str = "bf16", seed: int = 0,
                        allow_random_init: bool = False):
        """Local directory (``config.json`` + ``model.safetensors``) or a known config name.  A published
        google/siglip(2) checkpoint is the FULL SiglipModel (``vision_model.*``, ``text_model.*``, ``logit_scale``,
        ``logit_bias``): like ``SiglipVisionModel.from_pretrained`` (Siglip2sidafrozen.py:753) only the vision tower is
        kept.  There is no network: a bare name has no weights to load, which is an error unless
        ``allow_random_init=True`` asks for the closed-form seeded initialisation (``weights.seeded_state_dict``)."""
        if os.path.isdir(name_or_path):
            with open(os.path.join(name_or_path, "config.json")) as f:
                raw = json.load(f)
            raw = raw.get("vision_config", raw)
            fields = SiglipVisionConfig.__dataclass_fields__
            cfg = SiglipVisionConfig(**{k: v for k, v in raw.items() if k in fields})
            model = cls(cfg, compute_dtype)
            from safetensors.torch import load_file
            model.load_state_dict(vision_tower_only(load_file(os.path.join(name_or_path, "model.safetensors"))))
            return model
        if name_or_path in NAMED_CONFIGS:
            if not allow_random_init:
                raise OSError(
                    f"'{name_or_path}' names an architecture, not a local checkpoint directory, and there is no network "
                    "to fetch pretrained weights from.  Pass a directory holding config.json + model.safetensors, or "
                    "allow_random_init=True to get SEEDED RANDOM weights of that architecture (benchmarks / tests).")
            warnings.warn(f"SiglipVisionModelHIP.from_pretrained('{name_or_path}'): no checkpoint — using seeded RANDOM "
                          "weights (allow_random_init=True); outputs are not those of the pretrained model",
                          stacklevel=2)
            model = cls(get_config(name_or_path), compute_dtype)
            model.load_state_dict(seeded_state_dict(model.config, seed))
            return model
        raise OSError(f"{name_or_path} is neither a local checkpoint directory nor a known config name "
                      f"(no network access); known: {sorted(NAMED_CONFIGS)}")

    def gradient_checkpointing_enable(self, **_):
        """Accepted for interface parity (``Siglip2sidafrozen.py:1195-1196``) but does nothing, and says so once:
        activations of a 128-image so400m batch (87 GB) fit the 288 GB of HBM3E, so nothing is recomputed."""
        if not self._gradient_checkpointing:
            warnings.warn("SiglipVisionModelHIP.gradient_checkpointing_enable(): no-op — the HIP encoder keeps all "
                          "activations (0.68 GB per so400m@384 image; 288 GB HBM3E) and never recomputes",
                          stacklevel=2)
        self._gradient_checkpointing = True

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """Accepts HF names with or without the ``vision_model.`` prefix, or an open_clip/timm vision tower
        (``…visual.trunk.blocks.N.attn.qkv.weight``, converted by ``weights_io.timm_to_hf``)."""
        from . import weights_io
        if weights_io.detect_format(state_dict.keys()) == "timm":
            state_dict = weights_io.encoder_state_from_checkpoint(state_dict, self.config)
        return super().load_state_dict(dict(state_dict), strict=strict, **kw)

    def forward(self, pixel_values=None, output_hidden_states: bool = False, interpolate_pos_encoding: bool = False,
                hidden_state_ids=None, patches=None, **_):
        """Traceable by Dynamo: everything device-side happens inside ``torch.ops.siglip_hip.encoder_fwd``.
        ``patches`` (a ``preprocess.PatchOperand``) replaces ``pixel_values``: the resized + normalised images already
        in the patch GEMM's operand layout, so neither an fp32 (B,3,S,S) tensor nor the im2col pass exists."""
        layout, img_h, img_w = 0, 0, 0
        if patches is not None:
            if pixel_values is not None:
                raise ValueError("pass either pixel_values or patches")
            pixel_values, layout, img_h, img_w = patches.data, 2, int(patches.height), int(patches.width)
        if pixel_values.device.type != "cuda":
            raise RuntimeError("SiglipVisionModelHIP runs only on an AMD GPU through libsiglip_hip.so "
                               "(no CPU fallback); move the model and pixel_values to 'cuda'")
        if pixel_values.requires_grad:
            raise RuntimeError("SiglipVisionModelHIP does not differentiate with respect to pixel_values (the reference "
                               "never asks for it); detach the input")
        L = self.config.num_hidden_layers
        if hidden_state_ids is not None:
            tap_ids = tuple(int(i) % (L + 1) for i in hidden_state_ids)
        elif output_hidden_states:
            tap_ids = tuple(range(L + 1))
        else:
            tap_ids = ()
        uniq = sorted(set(tap_ids))
        params = self._flat_params()
        train = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        # first block that can receive a gradient (frozen prefix, Siglip2sidafrozen.py:757-768); 0 when the embeddings train
        first = 0
        if train and not any(p.requires_grad for p in params[:3]):
            first = L
            for (grp, _), p in zip(self._flat_names, params):
                if p.requires_grad and grp.startswith("layer"):
                    first = int(grp[5:])
                    break
        outs = torch.ops.siglip_hip.encoder_fwd(pixel_values, params, self._handle, train,
                                                bool(interpolate_pos_encoding), self.use_head, uniq, first, layout, img_h,
                                                img_w)
        pooled = outs[0] if self.use_head else None
        hs = tuple(outs[2 + uniq.index(i)] for i in tap_ids) if tap_ids else None
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

    def _reset_runtime_state(self):
        """Everything that belongs to THIS Python object (not to its parameters): C context, op handle, caches."""
        self._ctx = None
        self._shadow = None
        self._shadow_key = None
        self._shadow_serial = 0
        self._weights_struct = None
        self._weights_keep = None
        self._weights_key = None
        self._size_cache = {}
        self._ws_cache = None
        self._bucket_cache = {}
        self._pending_fwd = {}
        self._fwd_count = 0
        self._owner = id(self)
        self._handle = _NEXT_HANDLE[0]
        _NEXT_HANDLE[0] += 1
        _MODULES[self._handle] = self

    def __deepcopy__(self, memo):
        """copy.deepcopy (EMA / SWA wrappers): a fresh module with copied parameters and its own C context."""
        new = type(self)(self.config, self.compute_dtype)
        new.load_state_dict(self.state_dict())
        new.to(next(self.parameters()).device)
        for a, b in zip(new.parameters(), self.parameters()):
            a.requires_grad = b.requires_grad
        new.train(self.training)
        memo[id(self)] = new
        return new

    def _workspace(self, nbytes, dev):
        """Scratch reused across calls (stream-ordered; the C side only needs it intact within one forward / backward)."""
        ws = self._ws_cache
        if ws is None or ws.device != dev or ws.numel() < nbytes:
            self._ws_cache = ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return ws

    def _note_forward(self, saved):
        """Remember which weight-shadow generation a training forward used (checked by its backward)."""
        if len(self._pending_fwd) > 64:
            self._pending_fwd.clear()
        self._pending_fwd[saved.data_ptr()] = self._shadow_serial
        self._fwd_count += 1

    def _check_backward(self, saved):
        serial = self._pending_fwd.get(saved.data_ptr())
        if serial is not None and serial != self._shadow_serial:
            raise RuntimeError(
                "SiglipVisionModelHIP: parameters changed between this forward and its backward (optimizer step, EMA "
                "swap or load_state_dict in between re-cast the bf16 weight shadows); run backward before touching them")

    def _bucket_layout(self, needs):
        """Pure function of (which parameters need gradients): the gradient memory plan.

        Groups (embeddings, each block, post-LN + head) are listed in the order the backward completes them (head, block
        L-1 ... first trainable block, embeddings) and cut into at most ``max_buckets`` chunks of consecutive groups; every
        chunk is ONE flat fp32 tensor (= one DDP collective, ddp.py) holding its groups' per-parameter gradients, each
        16-byte aligned.  Returns (chunks, groups): chunks = [(total_elems, [group names], [(param index, offset, numel)])],
        groups = {group name: [param indices]}."""
        names = self._flat_names
        params = self._flat_params()
        groups: dict[str, list[int]] = {}
        for idx, (grp, _) in enumerate(names):
            if needs[idx]:
                groups.setdefault(grp, []).append(idx)
        L = self.config.num_hidden_layers
        order = [g_ for g_ in (["head"] + [f"layer{l}" for l in range(L - 1, -1, -1)] + ["emb"]) if g_ in groups]
        max_buckets = getattr(self._grad_reducer, "max_buckets", 8) if self._grad_reducer is not None else 8
        # q/k/v weight (and bias) gradients back to back: the C side then runs them as one dW GEMM / one column sum
        rank = {"q_w": 0, "k_w": 1, "v_w": 2, "q_b": 3, "k_b": 4, "v_b": 5}
        chunks = []
        for members in _taper(order, max(1, max_buckets)):
            off, entries = 0, []
            for grp in members:
                idxs = sorted(groups[grp], key=lambda i: (rank.get(names[i][1], 6), i))
                for i in idxs:
                    n = params[i].numel()
                    entries.append((i, off, n))
                    off += (n + 3) // 4 * 4
            chunks.append((off, members, entries))
        return chunks, groups

    def _accumulating(self, needs) -> bool:
        """Does a parameter this backward differentiates already carry a gradient (accumulation micro-steps)?"""
        return any(n and p.grad is not None for n, p in zip(needs, self._flat_params()))

    def _alloc_buckets(self, chunks, dev):
        """The flat tensors of ``_bucket_layout``.  The C side overwrites every element, so they are reused from step to
        step (no memset, stable pointers for FusedAdamW's device table) — but only when that is provably safe: no
        parameter's .grad still aliases the cached tensor (gradient accumulation, zero_grad(set_to_none=False)) AND a
        training forward has run since the backward that last filled it (two backward invocations of this module inside
        ONE autograd pass — siamese use, a loss summed over two forward calls — must not share memory: the first one's
        gradients may not have been accumulated yet).  Otherwise this backward gets fresh memory."""
        params = self._flat_params()
        flats = []
        for ci, (total, members, entries) in enumerate(chunks):
            key = (ci, tuple(members), tuple(e[0] for e in entries))
            hit = self._bucket_cache.get(key)
            flat = None
            if hit is not None and hit[0].device == dev and hit[0].numel() == total and hit[1] != self._fwd_count:
                flat = hit[0]
                base, end = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
                if any(params[i].grad is not None and base <= params[i].grad.data_ptr() < end for i, _, _ in entries):
                    flat = None
            if flat is None:
                flat = torch.empty(total, dtype=torch.float32, device=dev)
            if hit is None or hit[1] != self._fwd_count:
                self._bucket_cache[key] = (flat, self._fwd_count)
            flats.append(flat)
        return flats

    def _ensure_ctx(self):
        if self._owner != id(self):      # object was copied field by field (copy.copy): do not share the original's state
            self._reset_runtime_state()
        if self._ctx is None:
            lib = _lib.load()
            cfg = self.config
            c = _lib.SglConfig(cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.num_attention_heads,
                               cfg.patch_size, cfg.native_grid, cfg.layer_norm_eps,
                               {"bf16": _lib.SGL_DTYPE_BF16, "fp32": _lib.SGL_DTYPE_F32,
                                "bf16x3": _lib.SGL_DTYPE_BF16X3}[self.compute_dtype],
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
            self._shadow_serial += 1
        return self._shadow, self._weights_struct

    _SHADOWED = {"q_w", "k_w", "v_w", "q_b", "k_b", "v_b", "o_w", "fc1_w", "fc1_b", "fc2_w",     # per block
                 "patch_w", "in_proj_w", "out_proj_w", "head_fc1_w", "head_fc1_b", "head_fc2_w"}    # globals

    def _unit_keys(self):
        L = self.config.num_hidden_layers
        keys = [[] for _ in range(L + 1)]
        for (grp, _), p in zip(self._flat_names, self._flat_params()):
            keys[int(grp[5:]) if grp.startswith("layer") else L].append((p.data_ptr(), p._version))
        return [tuple(k) for k in keys]

    def _units_in_sync(self):
        """Per block (and, last entry, the globals): is the shadow arena current for the parameters as they are now?"""
        if self._shadow is None or self._shadow_key is None:
            return None
        return [a == b for a, b in zip(self._unit_keys(), self._shadow_key)]

    def _adopt_written_shadows(self, was_in_sync, written_ptrs):
        """Called by ``FusedAdamW`` after a step that wrote the shadows of the parameters in ``written_ptrs`` in its own
        pass: a unit that was in sync before the step, and whose shadowed parameters were all either written or left
        untouched, is in sync again — adopt the new versions so the next forward does not re-cast it."""
        if was_in_sync is None or self._shadow_key is None:
            return
        L = self.config.num_hidden_layers
        now = self._unit_keys()
        changed_ok = [True] * (L + 1)
        pos = [0] * (L + 1)
        for (grp, field), p in zip(self._flat_names, self._flat_params()):
            u = int(grp[5:]) if grp.startswith("layer") else L
            old = self._shadow_key[u][pos[u]]
            pos[u] += 1
            if (p.data_ptr(), p._version) != old and field in self._SHADOWED and p.data_ptr() not in written_ptrs:
                changed_ok[u] = False
        adopted = False
        for u in range(L + 1):
            if was_in_sync[u] and changed_ok[u] and now[u] != self._shadow_key[u]:
                self._shadow_key[u] = now[u]
                adopted = True
        if adopted:
            self._shadow_serial += 1

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._shadow = None
        self._shadow_key = None
        self._weights_struct = None
        self._ws_cache = None
        self._bucket_cache = {}
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

    def encode_image(self, x=None, normalize: bool = False, patches=None):
        out = self.visual(pixel_values=x, interpolate_pos_encoding=False, patches=patches)
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
        model.visual.load_state_dict(vision_tower_only(load_file(os.path.join(pretrained, "model.safetensors"))))
    else:
        if pretrained:   # 'webli' & co.: a hub tag, not a local directory
            warnings.warn(f"create_model_and_transforms('{model_name}', pretrained='{pretrained}'): no network and no "
                          "local checkpoint directory — the encoder gets seeded RANDOM weights, not the pretrained "
                          "ones; pass pretrained=<dir with model.safetensors> for real weights", stacklevel=2)
        model.visual.load_state_dict(seeded_state_dict(cfg, seed))
    model = model.to(device)
    pre = _preprocess_factory(cfg.image_size)
    return model, pre, pre

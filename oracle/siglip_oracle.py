"""CPU ORACLE for the SigLIP-2 vision-encoder hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain, explicit-math CPU restatement (PyTorch CPU tensors, fp32 or fp64) of the algorithm the
reference obtains from third-party ``transformers.SiglipVisionModel`` (reference call site
``Siglip2sidafrozen.py:753,787``; arithmetic in ``TF:models/siglip/modeling_siglip.py`` of transformers
5.15.0, the copy installed in the build container — the reference does not pin a version).  Each function
cites the lines it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it; the product package never does, and fails loudly when the HIP library is missing.

Parity pin: ``oracle/gen_golden.py`` runs the real HF model in the build container on seeded weights and
writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this restatement against those vectors
(the reference itself ships no test or golden vector for this path — SURVEY.md §4).
"""
from __future__ import annotations

import math

import torch


# ----------------------------------------------------------------------------------------------------
# elementary ops, written out
# ----------------------------------------------------------------------------------------------------
def linear(x, w, b=None):
    """nn.Linear: y = x·Wᵀ + b  (W is [out, in])."""
    y = x @ w.transpose(-1, -2)
    return y if b is None else y + b


def layer_norm(x, weight, bias, eps):
    """nn.LayerNorm over the last dim, biased variance (TF:modeling_siglip.py:329,331,567,630)."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * weight + bias


def gelu_tanh(x):
    """F.gelu(approximate='tanh') = 'gelu_pytorch_tanh' (TF:activations.py:31-51, TF:modeling_siglip.py:319)."""
    c = math.sqrt(2.0 / math.pi)
    return 0.5 * x * (1.0 + torch.tanh(c * (x + 0.044715 * x ** 3)))


def _cubic_coeffs(t, a=-0.75):
    """Cubic-convolution weights for taps at offsets −1, 0, +1, +2 (PyTorch upsample_bicubic2d, A = −0.75)."""
    def c1(x):  # |x| <= 1
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0

    def c2(x):  # 1 < |x| < 2
        return ((a * x - 5.0 * a) * x + 8.0 * a) * x - 4.0 * a

    return [c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)]


def bicubic_resize_table(table, src_g, dst_h, dst_w):
    """F.interpolate(mode='bicubic', align_corners=False) of the (src_g, src_g, D) position table to
    (dst_h, dst_w, D) (TF:modeling_siglip.py:137-173).  Separable; source index = (i+0.5)·in/out − 0.5,
    taps clamped to the border."""
    d = table.shape[-1]
    t = table.reshape(src_g, src_g, d)

    def axis_weights(n_in, n_out):
        scale = n_in / n_out
        idx = torch.zeros(n_out, 4, dtype=torch.long)
        wts = torch.zeros(n_out, 4, dtype=table.dtype)
        for o in range(n_out):
            src = (o + 0.5) * scale - 0.5
            i0 = math.floor(src)
            frac = src - i0
            cs = _cubic_coeffs(frac)
            for k in range(4):
                idx[o, k] = min(max(i0 - 1 + k, 0), n_in - 1)
                wts[o, k] = cs[k]
        return idx, wts

    iy, wy = axis_weights(src_g, dst_h)
    ix, wx = axis_weights(src_g, dst_w)
    # rows first
    tmp = torch.zeros(dst_h, src_g, d, dtype=table.dtype)
    for k in range(4):
        tmp += wy[:, k, None, None] * t[iy[:, k]]
    out = torch.zeros(dst_h, dst_w, d, dtype=table.dtype)
    for k in range(4):
        out += wx[None, :, k, None] * tmp[:, ix[:, k]]
    return out.reshape(dst_h * dst_w, d)


# ----------------------------------------------------------------------------------------------------
# encoder pieces
# ----------------------------------------------------------------------------------------------------
def patch_embed(pixel_values, sd, cfg, interpolate_pos_encoding):
    """Conv2d(3→D, k=s=p, 'valid') as a patch GEMM + position table
    (TF:modeling_siglip.py:175-185; table interpolation :137-173)."""
    b, c, h, w = pixel_values.shape
    p = cfg["patch_size"]
    gh, gw = h // p, w // p
    x = pixel_values[:, :, : gh * p, : gw * p]
    x = x.reshape(b, c, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(b, gh * gw, c * p * p)
    wmat = sd["embeddings.patch_embedding.weight"].reshape(cfg["hidden_size"], -1)
    emb = linear(x, wmat, sd["embeddings.patch_embedding.bias"])
    table = sd["embeddings.position_embedding.weight"]
    native = int(math.isqrt(table.shape[0]))
    if interpolate_pos_encoding and not (gh * gw == table.shape[0] and h == w):
        pos = bicubic_resize_table(table, native, gh, gw)
    else:
        pos = table
    return emb + pos.unsqueeze(0)


def attention(x, sd, pre, cfg):
    """SiglipAttention.forward (TF:modeling_siglip.py:273-306) with the eager softmax of :227-247."""
    b, n, d = x.shape
    nh = cfg["num_attention_heads"]
    dh = d // nh
    q = linear(x, sd[pre + "q_proj.weight"], sd[pre + "q_proj.bias"]).view(b, n, nh, dh).transpose(1, 2)
    k = linear(x, sd[pre + "k_proj.weight"], sd[pre + "k_proj.bias"]).view(b, n, nh, dh).transpose(1, 2)
    v = linear(x, sd[pre + "v_proj.weight"], sd[pre + "v_proj.bias"]).view(b, n, nh, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (dh ** -0.5)
    s = s - s.max(dim=-1, keepdim=True).values
    e = torch.exp(s)
    pr = e / e.sum(dim=-1, keepdim=True)
    o = (pr @ v).transpose(1, 2).reshape(b, n, d)
    return linear(o, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])


def mlp(x, sd, pre):
    """SiglipMLP.forward (TF:modeling_siglip.py:318-322)."""
    return linear(gelu_tanh(linear(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"])),
                  sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])


def encoder_layer(x, sd, l, cfg):
    """SiglipEncoderLayer.forward (TF:modeling_siglip.py:335-356)."""
    pre = f"encoder.layers.{l}."
    eps = cfg["layer_norm_eps"]
    x = x + attention(layer_norm(x, sd[pre + "layer_norm1.weight"], sd[pre + "layer_norm1.bias"], eps),
                      sd, pre + "self_attn.", cfg)
    x = x + mlp(layer_norm(x, sd[pre + "layer_norm2.weight"], sd[pre + "layer_norm2.bias"], eps),
                sd, pre + "mlp.")
    return x


def pooling_head(x, sd, cfg):
    """SiglipMultiheadAttentionPoolingHead.forward (TF:modeling_siglip.py:633-643): probe ⊗
    nn.MultiheadAttention (packed in_proj [3D,D] in q,k,v order) → LN → +MLP → [:,0]."""
    b, n, d = x.shape
    nh = cfg["num_attention_heads"]
    dh = d // nh
    w, bias = sd["head.attention.in_proj_weight"], sd["head.attention.in_proj_bias"]
    probe = sd["head.probe"].reshape(1, 1, d).expand(b, 1, d)
    q = linear(probe, w[:d], bias[:d]).view(b, 1, nh, dh).transpose(1, 2)
    k = linear(x, w[d:2 * d], bias[d:2 * d]).view(b, n, nh, dh).transpose(1, 2)
    v = linear(x, w[2 * d:], bias[2 * d:]).view(b, n, nh, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (dh ** -0.5)
    s = s - s.max(dim=-1, keepdim=True).values
    e = torch.exp(s)
    pr = e / e.sum(dim=-1, keepdim=True)
    o = (pr @ v).transpose(1, 2).reshape(b, 1, d)
    h = linear(o, sd["head.attention.out_proj.weight"], sd["head.attention.out_proj.bias"])
    h = h + mlp(layer_norm(h, sd["head.layernorm.weight"], sd["head.layernorm.bias"], cfg["layer_norm_eps"]),
                sd, "head.mlp.")
    return h[:, 0]


def vision_forward(pixel_values, sd, cfg, output_hidden_states=True, interpolate_pos_encoding=False):
    """SiglipVisionModel.forward (TF:modeling_siglip.py:576-619).  Returns a dict with
    ``pooler_output`` (B,D), ``last_hidden_state`` (B,N,D, post-LN) and ``hidden_states`` (L+1 tensors;
    [0] = embeddings, [-1] = last block output BEFORE post_layernorm —
    TF:utils/output_capturing.py:105-117,268-279, tie_last_hidden_states=False)."""
    if not isinstance(cfg, dict):
        cfg = cfg.to_dict()
    x = patch_embed(pixel_values, sd, cfg, interpolate_pos_encoding)
    hs = [x]
    for l in range(cfg["num_hidden_layers"]):
        x = encoder_layer(x, sd, l, cfg)
        hs.append(x)
    last = layer_norm(x, sd["post_layernorm.weight"], sd["post_layernorm.bias"], cfg["layer_norm_eps"])
    pooled = pooling_head(last, sd, cfg) if cfg.get("vision_use_head", True) else None
    return {"pooler_output": pooled, "last_hidden_state": last,
            "hidden_states": tuple(hs) if output_hidden_states else None}


def cast_state_dict(sd, dtype):
    return {k: v.to(dtype) for k, v in sd.items()}


# ----------------------------------------------------------------------------------------------------
# training-parity helper: a fixed scalar loss so gradients can be compared
# ----------------------------------------------------------------------------------------------------
def probe_loss(out, tap_ids=()):
    """A deterministic scalar touching pooled, last_hidden_state and the chosen hidden-state taps, used by
    the gradient-parity vectors.  Coefficients come from cosines of the element index so no RNG is involved."""
    def cw(t):
        n = t.numel()
        idx = torch.arange(n, dtype=t.dtype).reshape(t.shape)
        return torch.cos(idx * 0.37 + 0.11)
    loss = (out["pooler_output"] * cw(out["pooler_output"])).sum()
    loss = loss + 0.01 * (out["last_hidden_state"] * cw(out["last_hidden_state"])).sum()
    for i in tap_ids:
        h = out["hidden_states"][i]
        loss = loss + 0.01 * (h * cw(h)).sum()
    return loss

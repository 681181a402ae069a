"""Generate tests/golden/*.npz from the REAL third-party implementation the reference calls.

Run in the build container only (needs ``transformers``; nothing here travels to the GPU box except the
small .npz outputs).  For every case it instantiates ``transformers.SiglipVisionModel`` from a local config
(the constructor path of ``Siglip2sidafrozen.py:753`` without the network fetch), loads the closed-form
seeded weights, runs forward (+ backward of a fixed scalar loss) in fp32 on CPU, and stores inputs' seeds
and expected outputs (full tensors when small, checksums + strided samples when large).

    python oracle/gen_golden.py
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "deepfake-detection-using-clip-based-siglip-2-vision-transformers_amd")


def _light_package():
    """config + weights only (no HIP library needed to make fixtures)."""
    if "siglip_amd" not in sys.modules:
        pkg = types.ModuleType("siglip_amd")
        pkg.__path__ = [PKG_DIR]
        sys.modules["siglip_amd"] = pkg
    return (importlib.import_module("siglip_amd.config"), importlib.import_module("siglip_amd.weights"))


def _oracle():
    spec = importlib.util.spec_from_file_location("siglip_oracle", os.path.join(ROOT, "oracle", "siglip_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


CASES = [
    # name, config, seed, batch, res, interpolate, taps used in the loss, grads to record
    dict(name="tiny_32", config="tiny", seed=3, batch=2, res=32, interp=False, taps=(1, 3)),
    dict(name="tiny_48_interp", config="tiny", seed=4, batch=2, res=48, interp=True, taps=(0, 2)),
    dict(name="hostile_42", config="hostile", seed=5, batch=2, res=42, interp=True, taps=(1,)),
    dict(name="hostile_98_interp", config="hostile", seed=6, batch=3, res=98, interp=True, taps=(0, 2)),
    dict(name="so400m1_384", config="so400m-1layer", seed=7, batch=1, res=384, interp=True, taps=(1,)),
    dict(name="so400m1_224_interp", config="so400m-1layer", seed=8, batch=2, res=224, interp=True, taps=(0,)),
    # BASELINE.json configs 1/2: SigLIP-2-base-patch16-224 (D 768, head_dim 64, N 196), one block at full width
    dict(name="base1_224", config="base-1layer", seed=9, batch=2, res=224, interp=False, taps=(1,)),
]

GRAD_NAMES = [
    "embeddings.patch_embedding.weight", "embeddings.patch_embedding.bias",
    "embeddings.position_embedding.weight",
    "encoder.layers.0.layer_norm1.weight", "encoder.layers.0.layer_norm1.bias",
    "encoder.layers.0.self_attn.q_proj.weight", "encoder.layers.0.self_attn.k_proj.bias",
    "encoder.layers.0.self_attn.v_proj.weight", "encoder.layers.0.self_attn.out_proj.weight",
    "encoder.layers.0.layer_norm2.weight",
    "encoder.layers.0.mlp.fc1.weight", "encoder.layers.0.mlp.fc1.bias",
    "encoder.layers.0.mlp.fc2.weight", "encoder.layers.0.mlp.fc2.bias",
    "post_layernorm.weight", "head.probe", "head.attention.in_proj_weight", "head.attention.in_proj_bias",
    "head.attention.out_proj.weight", "head.layernorm.bias", "head.mlp.fc1.weight", "head.mlp.fc2.bias",
]

FULL_LIMIT = 1 << 16
NSAMP = 256


def pack(prefix: str, t: torch.Tensor, out: dict):
    a = t.detach().to(torch.float32).contiguous().numpy().reshape(-1)
    out[prefix + ".shape"] = np.asarray(t.shape, dtype=np.int64)
    out[prefix + ".sum"] = np.float64(a.astype(np.float64).sum())
    out[prefix + ".abssum"] = np.float64(np.abs(a.astype(np.float64)).sum())
    if a.size <= FULL_LIMIT:
        out[prefix + ".full"] = a
    else:
        idx = np.linspace(0, a.size - 1, NSAMP).astype(np.int64)
        out[prefix + ".idx"] = idx
        out[prefix + ".samples"] = a[idx]


def pack_err(prefix: str, t: torch.Tensor, rec: dict):
    """Error statistics of a reduced-precision run of the REAL HF model against the fp32 values already packed
    under `prefix`, on exactly the elements the fixture keeps (all of them, or the strided sample): the yardstick
    for the HIP bf16 mode (tests assert HIP-bf16 error <= 2x these)."""
    a = t.detach().to(torch.float32).contiguous().numpy().reshape(-1)
    if prefix + ".full" in rec:
        ref, got = rec[prefix + ".full"], a
    else:
        ref, got = rec[prefix + ".samples"], a[rec[prefix + ".idx"]]
    e = got.astype(np.float64) - ref.astype(np.float64)
    rec["bf16ac." + prefix + ".maxerr"] = np.float64(np.abs(e).max())
    rec["bf16ac." + prefix + ".l2rel"] = np.float64(np.sqrt((e * e).sum()) / (np.sqrt((ref.astype(np.float64) ** 2).sum()) + 1e-30))


def main():
    from transformers import SiglipVisionConfig as HFConfig, SiglipVisionModel
    config, weights = _light_package()
    oracle = _oracle()
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    torch.set_num_threads(8)
    for case in CASES:
        cfg = config.get_config(case["config"])
        sd = weights.seeded_state_dict(cfg, seed=case["seed"])
        hf = SiglipVisionModel(HFConfig(
            hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
            num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
            image_size=cfg.image_size, patch_size=cfg.patch_size, attn_implementation="eager"))
        hf.load_state_dict(sd, strict=True)
        hf.train()  # dropout is 0.0; train() so that gradient flow is the training path
        x = weights.seeded_pixels(case["batch"], case["res"], case["res"], seed=case["seed"] + 1000)
        o = hf(pixel_values=x, output_hidden_states=True, interpolate_pos_encoding=case["interp"])
        out = {"pooler_output": o.pooler_output, "last_hidden_state": o.last_hidden_state,
               "hidden_states": o.hidden_states}
        loss = oracle.probe_loss(out, case["taps"])
        loss.backward()
        rec: dict = {}
        rec["meta.config"] = np.asarray(case["config"])
        rec["meta.seed"] = np.int64(case["seed"])
        rec["meta.batch"] = np.int64(case["batch"])
        rec["meta.res"] = np.int64(case["res"])
        rec["meta.interp"] = np.int64(int(case["interp"]))
        rec["meta.taps"] = np.asarray(case["taps"], dtype=np.int64)
        rec["meta.transformers_version"] = np.asarray(__import__("transformers").__version__)
        pack("pooler_output", o.pooler_output, rec)
        pack("last_hidden_state", o.last_hidden_state, rec)
        for i, h in enumerate(o.hidden_states):
            pack(f"hidden_states.{i}", h, rec)
        rec["loss"] = np.float64(loss.item())
        named = dict(hf.named_parameters())
        for n in GRAD_NAMES:
            if n in named and named[n].grad is not None:
                pack("grad." + n, named[n].grad, rec)
        # the reference trains and evaluates under torch.amp.autocast(bf16) (Siglip2sidafrozen.py:1375,
        # hidf_video_classifier.py:389): the same HF model, same weights and input, under CPU bf16 autocast
        hf.zero_grad(set_to_none=True)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            ob = hf(pixel_values=x, output_hidden_states=True, interpolate_pos_encoding=case["interp"])
            outb = {"pooler_output": ob.pooler_output.float(), "last_hidden_state": ob.last_hidden_state.float(),
                    "hidden_states": tuple(h.float() for h in ob.hidden_states)}
            lossb = oracle.probe_loss(outb, case["taps"])
        lossb.backward()
        pack_err("pooler_output", ob.pooler_output, rec)
        pack_err("last_hidden_state", ob.last_hidden_state, rec)
        for i, h in enumerate(ob.hidden_states):
            pack_err(f"hidden_states.{i}", h, rec)
        rec["bf16ac.loss"] = np.float64(lossb.item())
        for n in GRAD_NAMES:
            if n in named and named[n].grad is not None and ("grad." + n + ".shape") in rec:
                pack_err("grad." + n, named[n].grad, rec)
        path = os.path.join(ROOT, "tests", "golden", case["name"] + ".npz")
        np.savez_compressed(path, **rec)
        print(f"wrote {path}: loss={loss.item():.6f} pooled|max|={o.pooler_output.abs().max().item():.4f} "
              f"({os.path.getsize(path) / 1024:.1f} KiB)")


if __name__ == "__main__":
    main()

"""Generate tests/golden/heads_*.npz by EXECUTING THE REFERENCE'S OWN CLASS DEFINITIONS (build container only).

The reference scripts cannot be imported whole (cv2 / torchvision / open_clip are absent and `coral.py` does not
parse), so the pure-torch classes and functions that sit on the encoder outputs are lifted from the reference's
source text with `ast`, compiled and executed here — nothing of the reference is copied into this repository;
only seeded inputs and the outputs it produced are stored.  Weights are the closed-form seeded tensors of
`siglip_amd.weights.seeded_tensor(<param name>)`, so `tests/test_heads.py` can rebuild identical parameters in this
repo's modules (whose parameter names match the reference's).

    python oracle/gen_golden_heads.py
"""
from __future__ import annotations

import ast
import json
import math
import os
import shutil
import sys
import types
from typing import List, Tuple, Sequence, Optional, Dict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle.gen_golden import _light_package  # noqa: E402

config, weights = _light_package()
OUT = os.path.join(ROOT, "tests", "golden")


# ---------------------------------------------------------------------------------------------------------
def lift(path: str, names: Sequence[str], ns: dict, line_range: Optional[Tuple[int, int]] = None):
    """exec the ClassDef / FunctionDef / Assign nodes called `names` (in file order) from `path` into `ns`."""
    src = open(path, encoding="utf-8").read()
    if line_range is not None:  # files that do not parse as a whole (coral.py)
        lines = src.splitlines()[line_range[0] - 1:line_range[1]]
        src = "\n".join(lines)
    tree = ast.parse(src)
    want = set(names)
    picked = []
    for node in ast.walk(tree):
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name in want:
            picked.append(node)
        elif isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id in want for t in node.targets):
            picked.append(node)
    picked.sort(key=lambda n: n.lineno)
    seen = set()
    for node in picked:
        key = getattr(node, "name", None) or node.targets[0].id
        if key in seen:
            continue
        seen.add(key)
        mod = ast.Module(body=[node], type_ignores=[])
        exec(compile(mod, f"<ref:{os.path.basename(path)}:{node.lineno}>", "exec"), ns)
    missing = want - seen
    assert not missing, f"{path}: not found {missing}"


def base_ns() -> dict:
    return dict(torch=torch, nn=nn, F=F, math=math, np=np, List=List, Tuple=Tuple, Sequence=Sequence,
                Optional=Optional, Dict=Dict)


def seed_module(mod: nn.Module, prefix: str, seed: int = 0):
    """Deterministic parameters/buffers keyed by their names (same rule in tests/test_heads.py)."""
    with torch.no_grad():
        for name, p in list(mod.named_parameters()) + list(mod.named_buffers()):
            full = prefix + name
            if p.dim() >= 2:
                fan_in = int(np.prod(p.shape[1:]))
                t = weights.seeded_tensor(full, p.shape, math.sqrt(3.0 / fan_in), seed)
            elif name.endswith("std"):
                t = weights.seeded_tensor(full, p.shape, 0.3, seed, 1.0)
            elif name.endswith("weight") or name.endswith("alpha") or name.endswith("T"):
                t = weights.seeded_tensor(full, p.shape, 0.2, seed, 1.0)
            else:
                t = weights.seeded_tensor(full, p.shape, 0.2, seed)
            p.copy_(t.reshape(p.shape))


def T(name, shape, bound=1.0, seed=0, offset=0.0):
    return weights.seeded_tensor(name, shape, bound, seed, offset)


class _FakeBackbone(nn.Module):
    """Stands in for the open_clip tower so the reference's model classes can be constructed: returns a preset
    feature tensor (the encoder itself is pinned separately by the HF golden vectors)."""

    def __init__(self, dim):
        super().__init__()
        self.embed_dim = dim
        self.feat = None

    def encode_image(self, x):
        return self.feat


def fake_open_clip(dim_holder):
    m = types.ModuleType("open_clip")

    def create_model_and_transforms(name, pretrained=None, device=None):
        bb = _FakeBackbone(dim_holder["dim"])
        dim_holder["last"] = bb
        return bb, None, None
    m.create_model_and_transforms = create_model_and_transforms
    return m


def main():
    os.makedirs(OUT, exist_ok=True)
    rec: dict = {}

    # ---- SID decoder + losses (Siglip2sidafrozen.py) --------------------------------------------------------
    ns = base_ns()
    sid = os.path.join(REF, "Siglip2sidafrozen.py")
    lift(sid, ["LinearProj", "SegFormerStrongDecoder", "focal_loss", "boundary_aware_loss", "morphological_loss",
               "iou_loss", "combined_segmentation_loss", "bce_dice_loss", "dice_iou_from_logits"], ns)
    B, g, D, E, K, S = 2, 5, 48, 16, 3, 70
    dec = ns["SegFormerStrongDecoder"]([D] * K, embed_dim=E).eval()
    seed_module(dec, "decoder.")
    taps = [T(f"tap{i}", (B, g * g, D)) for i in range(K)]
    seg = dec(taps, (g, g), target_size=S)
    rec["decoder.out"] = seg.detach().numpy()
    rec["decoder.meta"] = np.asarray([B, g, D, E, K, S])
    logits = T("seg_logits", (3, 1, 24, 24), 3.0)
    targets = (T("seg_targets", (3, 1, 24, 24)) > 0.3).float()
    for fn in ["focal_loss", "boundary_aware_loss", "morphological_loss", "iou_loss", "combined_segmentation_loss",
               "bce_dice_loss"]:
        rec["loss." + fn] = np.float64(ns[fn](logits, targets).item())
    dice, iou, pbin = ns["dice_iou_from_logits"](logits, targets)
    rec["loss.dice"], rec["loss.iou"], rec["loss.pbin_sum"] = np.asarray(dice), np.asarray(iou), np.float64(pbin.sum())

    # ---- cifake head (cifake_binary_classifier.py) -------------------------------------------------------------
    holder = {"dim": 64}
    ns = base_ns()
    ns.update(open_clip=fake_open_clip(holder), OPENCLIP_AVAILABLE=True, timm=None, print=lambda *a, **k: None)
    cif = os.path.join(REF, "cifake_binary_classifier.py")
    lift(cif, ["MODEL_CONFIGS", "LightweightAttention", "FastBinaryClassifier", "FocalLoss", "label_smoothing_loss"],
         ns)
    feats = T("cifake_features", (4, 64), 2.0)
    for size in ["tiny", "small", "medium", "large"]:
        holder["dim"] = 64 if size != "large" else 128
        model = ns["FastBinaryClassifier"](model_size=size, device="cpu").eval()
        res = model.resolution
        f_in = feats if size != "large" else T("cifake_features_large", (4, 128), 2.0)
        model.backbone.feat = f_in
        seed_module(model, f"cifake.{size}.")
        x = torch.zeros(4, 3, res, res)
        rec[f"cifake.{size}.logits"] = model(x).detach().numpy()
    y = (T("bin_targets", (16,)) > 0).float()
    z = T("bin_logits", (16,), 3.0)
    rec["loss.FocalLoss"] = np.float64(ns["FocalLoss"](alpha=1.0, gamma=2.0)(z, y).item())
    rec["loss.FocalLoss_pw"] = np.float64(ns["FocalLoss"](alpha=0.5, gamma=1.5, pos_weight=torch.tensor(2.0))(z, y).item())
    rec["loss.label_smoothing"] = np.float64(ns["label_smoothing_loss"](z, y, 0.1).item())

    # ---- video head (hidf_video_classifier.py) ------------------------------------------------------------------
    holder["dim"] = 64
    ns = base_ns()
    ns.update(open_clip=fake_open_clip(holder), OPENCLIP_AVAILABLE=True, print=lambda *a, **k: None)
    lift(os.path.join(REF, "hidf_video_classifier.py"), ["BinaryVideoClassifier"], ns)
    vid = ns["BinaryVideoClassifier"](device="cpu", num_frames=4).eval()
    seed_module(vid, "video.")
    vfeat = T("video_features", (3 * 4, 64), 2.0)
    vid.vision_encoder.feat = vfeat
    rec["video.logits"] = vid(torch.zeros(3, 4, 3, 8, 8)).detach().numpy()

    # ---- HiDF image-track head (simple_classifier.py:115-164; BASELINE config 3) -----------------------------------
    for size, dim in (("small", 768), ("large", 1024)):
        holder["dim"] = dim
        ns = base_ns()
        ns.update(open_clip=fake_open_clip(holder), OPENCLIP_AVAILABLE=True, print=lambda *a, **k: None)
        lift(os.path.join(REF, "simple_classifier.py"), ["BinaryClassifier"], ns)
        img = ns["BinaryClassifier"](model_size=size, device="cpu").eval()
        seed_module(img, f"image.{size}.")
        img.backbone.feat = T(f"image_features_{size}", (5, dim), 2.0)
        rec[f"image.{size}.logits"] = img(torch.zeros(5, 3, img.resolution, img.resolution)).detach().numpy()

    # ---- SE head, FreqMLP v5, adaptive fusion (train_fusion_head_only.py) ------------------------------------------
    holder["dim"] = 1024
    ns = base_ns()
    ns.update(open_clip=fake_open_clip(holder))
    tf = os.path.join(REF, "train_fusion_head_only.py")
    lift(tf, ["SIGLIP_DIM", "IMG_SIZE", "BinaryClassifier", "FeatureNormalizer", "ContrastScaler", "TemperatureScaler",
              "BandGating", "ResidualMLPBlock", "FreqMLP", "AdaptiveFusionHead"], ns)
    se = ns["BinaryClassifier"]("cpu").eval()
    seed_module(se, "se.")
    sefeat = T("se_features", (3, ns["SIGLIP_DIM"]), 2.0)
    se.backbone.feat = sefeat
    rec["se.logits"] = se(torch.zeros(3, 3, ns["IMG_SIZE"], ns["IMG_SIZE"])).detach().numpy()
    rec["se.dim"] = np.int64(ns["SIGLIP_DIM"])
    fm = ns["FreqMLP"]().eval()
    seed_module(fm, "freqv5.")
    rec["freqv5.logits"] = fm(T("freq_in", (5, 24), 2.0)).detach().numpy()
    af = ns["AdaptiveFusionHead"]().eval()
    seed_module(af, "afusion.")
    rec["afusion.z"] = af(T("zf", (7,), 3.0), T("zs", (7,), 3.0)).detach().numpy()

    # ---- shipped app artefacts: FusionHead 2->1, app FreqMLP, CORAL (appv3.py + siglip/*) --------------------------
    ns = base_ns()
    ns.update(CORAL_CUTS=json.load(open(os.path.join(REF, "siglip", "coral_cutpoints.json"))))
    app = os.path.join(REF, "appv3.py")
    lift(app, ["SafeLayerNorm", "FreqMLP", "FusionHead", "_logit", "CoralCalibrator"], ns)
    from safetensors.torch import load_file
    dst = os.path.join(OUT, "ref_siglip")
    os.makedirs(dst, exist_ok=True)
    for fn in ["fusion_head.safetensors", "freq_mlp.safetensors", "coral_cutpoints.json", "coral_temp.json"]:
        shutil.copyfile(os.path.join(REF, "siglip", fn), os.path.join(dst, fn))   # data fixtures the reference ships
    fh = ns["FusionHead"]()
    fh.load_state_dict(load_file(os.path.join(REF, "siglip", "fusion_head.safetensors")))
    probs2 = T("fusion_probs", (6, 2), 0.5, 0, 0.5)
    rec["shipped.fusion_out"] = fh(probs2).detach().numpy()
    fa = ns["FreqMLP"]()
    fa.load_state_dict(load_file(os.path.join(REF, "siglip", "freq_mlp.safetensors")))
    fa.train()  # eval mode adds random jitter (appv3.py:1508-1509); train mode is the deterministic function
    rec["shipped.freq_out"] = fa(T("freq_in", (5, 24), 2.0)).detach().numpy()
    cc = ns["CoralCalibrator"]()
    zs = [-3.0, -0.7, 0.0, 0.4, 2.5, 6.0]
    rec["shipped.coral_probs"] = np.stack([cc.probs(torch.tensor(zv)).numpy() for zv in zs])
    rec["shipped.coral_idx"] = np.asarray([cc.predict(torch.tensor(zv))[0] for zv in zs])
    rec["shipped.coral_z"] = np.asarray(zs)
    rec["shipped.coral_c"] = cc.c.numpy()
    temp = json.load(open(os.path.join(REF, "siglip", "coral_temp.json")))["temperature"]
    rec["shipped.coral_temp"] = np.float64(temp)

    # ---- coral.py (does not parse as a whole: lift the one function by line range) ---------------------------------
    ns = base_ns()
    lift(os.path.join(REF, "coral.py"), ["fit_coral_cutpoints"], ns, line_range=(296, 325))
    fl = T("coral_fit_logits", (257,), 4.0)
    rec["coral.fit_cuts"] = np.asarray(ns["fit_coral_cutpoints"](fl, torch.zeros(257)))

    path = os.path.join(OUT, "heads.npz")
    np.savez_compressed(path, **rec)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB), {len(rec)} entries")


if __name__ == "__main__":
    main()

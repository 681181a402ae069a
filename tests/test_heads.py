"""This repo's heads / decoder / losses / calibration against golden vectors produced by EXECUTING the reference's own
class definitions (oracle/gen_golden_heads.py) and against the data files the reference ships
(siglip/fusion_head.safetensors, freq_mlp.safetensors, coral_*.json — copied as fixtures to tests/golden/ref_siglip).
Every case runs twice: on the CPU (``-m "not gpu"``) and, marked ``gpu``, on the MI355X — the code that actually runs
under these heads in training (PyTorch-ROCm kernels, plus the HIP depthwise stencil of csrc/decoder.hip in the decoder)."""
import json
import math
import os

import numpy as np
import pytest
import torch

import golden_util as gu

REC = dict(np.load(os.path.join(gu.GOLDEN_DIR, "heads.npz")))


@pytest.fixture(params=["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def dev(request):
    return request.param
REF_SIGLIP = os.path.join(gu.GOLDEN_DIR, "ref_siglip")


def seed_module(pkg, mod, prefix, seed=0):
    """Same naming rule as oracle/gen_golden_heads.py::seed_module."""
    W = pkg.weights
    with torch.no_grad():
        for name, p in list(mod.named_parameters()) + list(mod.named_buffers()):
            full = prefix + name
            if p.dim() >= 2:
                t = W.seeded_tensor(full, p.shape, math.sqrt(3.0 / int(np.prod(p.shape[1:]))), seed)
            elif name.endswith("std"):
                t = W.seeded_tensor(full, p.shape, 0.3, seed, 1.0)
            elif name.endswith("weight") or name.endswith("alpha") or name.endswith("T"):
                t = W.seeded_tensor(full, p.shape, 0.2, seed, 1.0)
            else:
                t = W.seeded_tensor(full, p.shape, 0.2, seed)
            p.copy_(t.reshape(p.shape))


def T(pkg, name, shape, bound=1.0, seed=0, offset=0.0):
    return pkg.weights.seeded_tensor(name, shape, bound, seed, offset)


def close(got, ref, tol=2e-5):
    got = np.asarray(got.detach().cpu().numpy() if torch.is_tensor(got) else got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = np.abs(got - ref).max()
    assert err <= tol * max(1.0, np.abs(ref).max()), f"max|err| {err:.3e}"


def test_mask_decoder_matches_reference(pkg, dev):
    B, g, D, E, K, S = (int(v) for v in REC["decoder.meta"])
    H = pkg.heads
    taps = [T(pkg, f"tap{i}", (B, g * g, D)).to(dev) for i in range(K)]
    for early_head in (False, True):   # reference order, and the HBM-saving head-before-upsample order
        dec = H.SegFormerMaskDecoder([D] * K, embed_dim=E, head_before_upsample=early_head).eval()
        seed_module(pkg, dec, "decoder.")
        dec = dec.to(dev)
        close(dec(taps, (g, g), target_size=S), REC["decoder.out"], 5e-6 if not early_head else 2e-5)


def test_segmentation_losses_match_reference(pkg, dev):
    H = pkg.heads
    logits = T(pkg, "seg_logits", (3, 1, 24, 24), 3.0).to(dev)
    targets = (T(pkg, "seg_targets", (3, 1, 24, 24)) > 0.3).float().to(dev)
    for fn in ["focal_loss", "boundary_aware_loss", "morphological_loss", "iou_loss", "combined_segmentation_loss",
               "bce_dice_loss"]:
        close(getattr(H, fn)(logits, targets), REC["loss." + fn], 1e-6)
    dice, iou, pbin = H.dice_iou_from_logits(logits, targets)
    close(torch.as_tensor(dice), REC["loss.dice"], 1e-6)
    close(torch.as_tensor(iou), REC["loss.iou"], 1e-6)
    assert float(pbin.sum()) == float(REC["loss.pbin_sum"])
    y = (T(pkg, "bin_targets", (16,)) > 0).float().to(dev)
    z = T(pkg, "bin_logits", (16,), 3.0).to(dev)
    close(H.FocalLoss(1.0, 2.0)(z, y), REC["loss.FocalLoss"], 1e-6)
    close(H.FocalLoss(0.5, 1.5, pos_weight=torch.tensor(2.0, device=dev))(z, y), REC["loss.FocalLoss_pw"], 1e-6)
    close(H.label_smoothing_loss(z, y, 0.1), REC["loss.label_smoothing"], 1e-6)


@pytest.mark.parametrize("size", ["tiny", "small", "medium", "large"])
def test_cifake_head_matches_reference(pkg, size, dev):
    dim = 128 if size == "large" else 64
    head = pkg.heads.CifakeBinaryHead(dim, model_size=size).eval()
    seed_module(pkg, head, f"cifake.{size}.")
    head = head.to(dev)
    feats = T(pkg, "cifake_features_large" if size == "large" else "cifake_features", (4, dim), 2.0).to(dev)
    close(head(feats), REC[f"cifake.{size}.logits"])


def test_video_and_se_heads_match_reference(pkg, dev):
    vid = pkg.heads.VideoBinaryHead(64, num_frames=4).eval()
    seed_module(pkg, vid, "video.")
    close(vid.to(dev)(T(pkg, "video_features", (12, 64), 2.0).to(dev), batch_size=3), REC["video.logits"])
    dim = int(REC["se.dim"])
    se = pkg.heads.SEBinaryHead(dim).eval()
    seed_module(pkg, se, "se.")
    close(se.to(dev)(T(pkg, "se_features", (3, dim), 2.0).to(dev)), REC["se.logits"])


@pytest.mark.parametrize("size,dim", [("small", 768), ("large", 1024)])
def test_image_binary_head_matches_reference(pkg, size, dim, dev):
    """HiDF image-track head (simple_classifier.py:141-148,159-164; BASELINE config 3) against the reference class."""
    head = pkg.heads.ImageBinaryHead(dim).eval()
    seed_module(pkg, head, f"image.{size}.")
    assert [n for n, _ in head.named_parameters()] == ["classifier.0.weight", "classifier.0.bias", "classifier.2.weight",
                                                       "classifier.2.bias", "classifier.5.weight", "classifier.5.bias"]
    close(head.to(dev)(T(pkg, f"image_features_{size}", (5, dim), 2.0).to(dev)), REC[f"image.{size}.logits"])


def test_fusion_and_freq_match_reference(pkg, dev):
    H = pkg.heads
    fm = H.FreqMLPv5().eval()
    seed_module(pkg, fm, "freqv5.")
    close(fm.to(dev)(T(pkg, "freq_in", (5, 24), 2.0).to(dev)), REC["freqv5.logits"])
    af = H.AdaptiveFusionHead().eval()
    seed_module(pkg, af, "afusion.")
    close(af.to(dev)(T(pkg, "zf", (7,), 3.0).to(dev), T(pkg, "zs", (7,), 3.0).to(dev)), REC["afusion.z"])
    close(np.asarray(H.fit_coral_cutpoints(T(pkg, "coral_fit_logits", (257,), 4.0).to(dev))), REC["coral.fit_cuts"], 1e-7)


def test_shipped_artifacts_known_answers(pkg, dev):
    """The weights / calibration files the reference ships, run through this repo's modules."""
    from safetensors.torch import load_file
    H = pkg.heads
    fh = H.LinearFusionHead()
    fh.load_state_dict(load_file(os.path.join(REF_SIGLIP, "fusion_head.safetensors")))
    close(fh.to(dev)(T(pkg, "fusion_probs", (6, 2), 0.5, 0, 0.5).to(dev)), REC["shipped.fusion_out"], 1e-6)
    fa = H.FreqMLPApp()
    fa.load_state_dict(load_file(os.path.join(REF_SIGLIP, "freq_mlp.safetensors")))
    close(fa.to(dev)(T(pkg, "freq_in", (5, 24), 2.0).to(dev)), REC["shipped.freq_out"], 1e-5)
    cuts = json.load(open(os.path.join(REF_SIGLIP, "coral_cutpoints.json")))
    cc = H.CoralCalibrator(cuts)
    close(cc.c, REC["shipped.coral_c"], 1e-6)
    zs = REC["shipped.coral_z"]
    got = np.stack([cc.probs(float(z)).numpy() for z in zs])
    close(got, REC["shipped.coral_probs"], 1e-6)
    assert [cc.predict(float(z))[0] for z in zs] == [int(i) for i in REC["shipped.coral_idx"]]
    close(cc.probs_batch(torch.tensor(zs, dtype=torch.float32, device=dev)), REC["shipped.coral_probs"], 1e-6)
    temp = json.load(open(os.path.join(REF_SIGLIP, "coral_temp.json")))["temperature"]
    assert abs(temp - float(REC["shipped.coral_temp"])) < 1e-12 and len(H.RISK_NAMES) == 5


def test_mtl_loss_and_shapes(pkg, dev):
    H = pkg.heads
    cls = T(pkg, "cls", (4, 3), 2.0).to(dev)
    seg = T(pkg, "seg", (4, 1, 16, 16), 2.0).to(dev)
    y = torch.tensor([0, 1, 2, 1], device=dev)
    m = (T(pkg, "m", (4, 1, 16, 16)) > 0).float().to(dev)
    hm = torch.tensor([True, False, True, True], device=dev)
    base = torch.nn.functional.cross_entropy(cls, y)
    assert torch.allclose(H.mtl_loss(cls, seg, y, m, torch.zeros(4, dtype=torch.bool, device=dev)), base)
    full = H.mtl_loss(cls, seg, y, m, hm, lam_seg=0.7)
    assert torch.allclose(full, base + 0.7 * H.bce_dice_loss(seg[hm], m[hm]))


def test_batched_inference_signals_equal_the_apps_per_image_loop(pkg, dev):
    """heads.core_signals_batched against a scalar restatement of appv3.py:3221-3300 (`detect_core` after the encoder):
    the app's own arithmetic, one image at a time with Python floats, on the shipped fusion head and CORAL cut-points."""
    from safetensors.torch import load_file
    H = pkg.heads
    fh = H.LinearFusionHead()
    fh.load_state_dict(load_file(os.path.join(REF_SIGLIP, "fusion_head.safetensors")))
    cc = H.CoralCalibrator(json.load(open(os.path.join(REF_SIGLIP, "coral_cutpoints.json"))))
    temp = json.load(open(os.path.join(REF_SIGLIP, "coral_temp.json")))["temperature"]
    B, C = 7, 9
    z_sigs, z_freqs = T(pkg, "app_zs", (B, C), 4.0), T(pkg, "app_zf", (B, C), 3.0)
    z_rot = T(pkg, "app_zr", (B,), 4.0)
    w = torch.tensor([0.2] + [0.1] * 8)
    got = H.core_signals_batched(z_sigs.to(dev), w.to(dev), z_freqs.to(dev), z_rot.to(dev), fh.to(dev), cc, 1.25, temp)
    sig = lambda v: 1.0 / (1.0 + math.exp(-v))
    for b in range(B):
        z_sig = float((z_sigs[b] * w).sum())
        z_freq = float((z_freqs[b] * w).sum())
        visual = 0.6 * sig(z_sig) + 0.4 * sig(float(z_rot[b]))
        p_freq = sig(z_freq / 1.25)
        z = float(fh.cpu()(torch.tensor([[visual, p_freq]], dtype=torch.float32)).item())
        z_scaled = z / max(temp, 1e-3)
        idx, probs = cc.predict(torch.tensor(z_scaled))
        mu = float((torch.arange(5.0) * probs).sum())
        var = float((probs * (torch.arange(5.0) - mu) ** 2).sum())
        want = {"z_sig": H._logit(visual), "z_freq": z_freq, "visual_prob": visual, "p_freq": p_freq, "z": z,
                "z_scaled": z_scaled, "p_fake_raw": sig(z_scaled), "p_fake_coral": max(0.0, min(1.0, mu / 4 + 0.5 * var)),
                "coral_entropy": float(-(probs * torch.log(probs + 1e-8)).sum()),
                "p_blend": max(0.0, min(1.0, 0.7 * sig(z_scaled) + 0.3 * max(0.0, min(1.0, mu / 4 + 0.5 * var))))}
        for k, v in want.items():
            assert abs(float(got[k][b]) - v) <= 2e-5 * max(1.0, abs(v)), (k, b, float(got[k][b]), v)
        assert int(got["risk_idx"][b]) == idx
        close(got["risk_probs"][b], probs.numpy(), 1e-5)
    fh.to(dev)

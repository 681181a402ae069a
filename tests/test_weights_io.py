"""Weight / checkpoint formats (weights_io.py).  CPU only.

What is pinned: the tensor algebra of the open_clip/timm <-> HF conversion (fused qkv / kv split and merge, position
table shape) through exact round trips; that the reference's own loading code path (train_fusion_head_only.py:110-122:
filter by ``k in model.state_dict()`` and equal shape, drop ``backbone.text.*``, ``strict=False``) fills every encoder
tensor of our composed model; and the reference's ``.pt`` checkpoint dictionary.  The timm key NAMES themselves are
"parity unpinned" (timm / open_clip are not installed here and the reference ships no backbone checkpoint)."""
import os

import pytest
import torch

import __graft_entry__ as entry


@pytest.fixture(scope="module")
def pkg():
    return entry.load_package()


def test_hf_timm_round_trip_is_exact(pkg):
    cfg = pkg.get_config("hostile")
    hf = pkg.weights.seeded_state_dict(cfg, 3)
    timm = pkg.weights_io.hf_to_timm(hf, cfg, "visual.trunk.")
    d, L = cfg.hidden_size, cfg.num_hidden_layers
    assert timm["visual.trunk.pos_embed"].shape == (1, cfg.num_positions, d)
    assert timm["visual.trunk.blocks.0.attn.qkv.weight"].shape == (3 * d, d)
    assert timm["visual.trunk.attn_pool.kv.weight"].shape == (2 * d, d)
    assert timm["visual.trunk.attn_pool.latent"].shape == (1, 1, d)
    # q, k, v order inside the fused tensor
    assert torch.equal(timm["visual.trunk.blocks.1.attn.qkv.weight"][d:2 * d], hf["encoder.layers.1.self_attn.k_proj.weight"])
    assert torch.equal(timm["visual.trunk.attn_pool.kv.bias"][d:], hf["head.attention.in_proj_bias"][2 * d:])
    assert len(timm) == 3 + 12 * L + 2 + 13
    back = pkg.weights_io.timm_to_hf(timm, cfg, "visual.trunk.")
    assert set(back) == set(hf)
    for k in hf:
        assert torch.equal(back[k], hf[k]), k
    assert pkg.weights_io.detect_format(timm) == "timm" and pkg.weights_io.detect_format(hf) == "hf"
    assert pkg.weights_io.detect_format({"fc.weight": 0}) == "unknown"
    with pytest.raises(KeyError):
        bad = dict(timm)
        bad.pop("visual.trunk.norm.bias")
        pkg.weights_io.timm_to_hf(bad, cfg, "visual.trunk.")


def test_open_clip_surface_speaks_timm_names_and_loads_reference_checkpoints(pkg, tmp_path):
    from safetensors.torch import save_file, load_file
    cfg = pkg.get_config("hostile")
    hf = pkg.weights.seeded_state_dict(cfg, 5)
    # a checkpoint as the reference's CiFake trainer writes it: backbone.visual.trunk.* + text tower + head
    backbone = pkg.OpenClipStyleEncoder(cfg, "fp32")
    model = pkg.heads.SEBinaryClassifierHIP(backbone)
    ref_ckpt = {"backbone." + k: v.clone() for k, v in pkg.weights_io.hf_to_timm(hf, cfg, "visual.trunk.").items()}
    ref_ckpt["backbone.text.transformer.weight"] = torch.zeros(4, 4)
    ref_ckpt["backbone.logit_scale"] = torch.zeros(())
    for k, v in model.state_dict().items():
        if not k.startswith("backbone."):
            ref_ckpt[k] = torch.full_like(v, 0.25)
    path = str(tmp_path / "best_model.safetensors")
    save_file({k: v.contiguous() for k, v in ref_ckpt.items()}, path)
    # the reference's loader, verbatim in behaviour (train_fusion_head_only.py:110-122)
    state = load_file(path)
    msd = model.state_dict()
    assert all(k.startswith(("backbone.visual.trunk.", "se.", "classifier.")) for k in msd), list(msd)[:5]
    filt = {k: v for k, v in state.items() if (not k.startswith("backbone.text.")) and (k in msd) and (v.shape == msd[k].shape)}
    assert len(filt) == len(msd)                      # every tensor of our model is matched by name AND shape
    res = model.load_state_dict(filt, strict=False)
    assert not res.missing_keys and not res.unexpected_keys
    got = {k[len("vision_model."):]: v for k, v in backbone.visual.state_dict().items()}
    for k in hf:
        assert torch.equal(got[k], hf[k]), k
    # strict load of the unfiltered checkpoint also works: the text tower and logit scale are dropped by the hook
    model2 = pkg.heads.SEBinaryClassifierHIP(pkg.OpenClipStyleEncoder(cfg, "fp32"))
    model2.load_state_dict(state)
    assert torch.equal(model2.backbone.visual.head.probe, hf["head.probe"])
    # an incomplete tower is reported as a load error, not a KeyError from inside the hook
    broken = {k: v for k, v in state.items() if not k.endswith("trunk.norm.bias")}
    with pytest.raises(RuntimeError, match="cannot be converted"):
        pkg.heads.SEBinaryClassifierHIP(pkg.OpenClipStyleEncoder(cfg, "fp32")).load_state_dict(broken)
    # bare encoder accepts the same file too
    enc = pkg.SiglipVisionModelHIP(cfg, "fp32")
    enc.load_state_dict(state)
    assert torch.equal(enc.encoder.layers[1].self_attn.v_proj.bias, hf["encoder.layers.1.self_attn.v_proj.bias"])


def test_mtl_checkpoint_layout_and_pt_round_trip(pkg, tmp_path):
    cfg = pkg.get_config("hostile")
    model = pkg.heads.SigLIP2MTL(pkg.SiglipVisionModelHIP(cfg, "fp32"), seg_layers=(-1, -2), embed_dim=32)
    keys = list(model.state_dict())
    assert "encoder.vision_model.encoder.layers.0.self_attn.q_proj.weight" in keys     # Siglip2sidafrozen.py:753,1639
    assert "encoder.vision_model.head.probe" in keys
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, 3)
    path = str(tmp_path / "best.pt")
    pkg.weights_io.save_checkpoint(path, model, opt, sched, epoch=4, metrics={"f1": 0.5}, seg_layers=[-1, -2])
    raw = torch.load(path, weights_only=True)
    assert set(raw) == {"model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "epoch", "metrics",
                        "seg_layers"}
    model2 = pkg.heads.SigLIP2MTL(pkg.SiglipVisionModelHIP(cfg, "fp32"), seg_layers=(-1, -2), embed_dim=32)
    opt2 = torch.optim.AdamW([p for p in model2.parameters() if p.requires_grad], lr=1e-3)
    extra = pkg.weights_io.load_checkpoint(path, model2, opt2)
    assert extra["epoch"] == 4 and extra["metrics"] == {"f1": 0.5}
    for (k1, v1), (k2, v2) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    sd = pkg.weights_io.load_state_file(path)
    enc = pkg.weights_io.encoder_state_from_checkpoint(sd, cfg)
    assert set(enc) == set(pkg.weights.param_shapes(cfg))

"""Parity of the BENCHMARKED compute mode (bf16 MFMA operands, fp32 accumulate / residual stream / statistics).

The reference runs its models under ``torch.amp.autocast(bf16)`` (Siglip2sidafrozen.py:1375, hidf_video_classifier.py:389),
so the yardstick for "is the bf16 HIP path right" is the error the REAL HF model makes under bf16 autocast against its own
fp32 run.  oracle/gen_golden.py stores that error (max |err| and relative L2, per tensor, on the fixture's elements) under
``bf16ac.*`` in every tests/golden/*.npz.  The bars here, all stated relative to those vectors or to the fp32 CPU oracle:

  * every output, hidden-state tap and the 22 recorded gradients: HIP-bf16 relative-L2 error <= 2x the HF-autocast error
    (measured: 0.3-0.9x, printed per case);
  * full 27-layer so400m@384, forward AND backward, both compute modes, against the fp32 oracle (tolerances next to the
    asserts: 2x the values measured on MI355X, which are printed);
  * BASELINE.json configs 1, 2 and 5 end to end.
"""
import math

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def build(pkg, cfg_name, seed, mode):
    cfg = pkg.get_config(cfg_name)
    model = pkg.SiglipVisionModelHIP(cfg, compute_dtype=mode)
    model.load_state_dict(pkg.weights.seeded_state_dict(cfg, seed=seed))
    return model.to("cuda")


def _probe_loss(out, tap_ids):
    def cw(t):
        idx = torch.arange(t.numel(), dtype=torch.float32, device=t.device).reshape(t.shape)
        return torch.cos(idx * 0.37 + 0.11)
    loss = (out["pooler_output"] * cw(out["pooler_output"])).sum()
    loss = loss + 0.01 * (out["last_hidden_state"] * cw(out["last_hidden_state"])).sum()
    for i in tap_ids:
        h = out["hidden_states"][i]
        loss = loss + 0.01 * (h * cw(h)).sum()
    return loss


def rel_l2(got, ref):
    got, ref = got.detach().double().cpu().reshape(-1), ref.detach().double().cpu().reshape(-1)
    return ((got - ref).norm() / (ref.norm() + 1e-30)).item()


@pytest.mark.parametrize("case", gu.CASES)
def test_bf16_error_within_2x_of_hf_bf16_autocast(case, pkg, hiplib):
    """|HIP_bf16 - fp32 golden| <= 2 x |HF_bf16_autocast - fp32 golden| for pooled, last, every tap and 22 gradients."""
    rec = gu.load(case)
    m = gu.meta(rec)
    model = build(pkg, m["config"], m["seed"], "bf16")
    x = pkg.weights.seeded_pixels(m["batch"], m["res"], m["res"], seed=m["seed"] + 1000).cuda()
    out = model(pixel_values=x, output_hidden_states=True, interpolate_pos_encoding=m["interp"])
    tensors = {"pooler_output": out.pooler_output, "last_hidden_state": out.last_hidden_state}
    for i, h in enumerate(out.hidden_states):
        tensors[f"hidden_states.{i}"] = h
    loss = _probe_loss({"pooler_output": out.pooler_output, "last_hidden_state": out.last_hidden_state,
                        "hidden_states": out.hidden_states}, m["taps"])
    loss.backward()
    named = dict(model.named_parameters())
    for k in rec:
        if k.startswith("grad.") and k.endswith(".shape"):
            name = k[len("grad."):-len(".shape")]
            tensors["grad." + name] = named[name].grad
    ratios = {}
    for prefix, t in tensors.items():
        if prefix.endswith("k_proj.bias"):
            continue   # exactly zero in real arithmetic (softmax shift invariance): both sides are rounding noise
        mx, l2 = gu.err_stats(rec, prefix, t.detach().float().cpu())
        hf_l2 = float(rec["bf16ac." + prefix + ".l2rel"])
        hf_mx = float(rec["bf16ac." + prefix + ".maxerr"])
        ratios[prefix] = (l2 / (hf_l2 + 1e-12), mx / (hf_mx + 1e-12), l2, hf_l2)
        assert l2 <= 2.0 * hf_l2 + 1e-5, f"{prefix}: rel-L2 err {l2:.3e} vs HF bf16 autocast {hf_l2:.3e}"
        assert mx <= 3.0 * hf_mx + 1e-6, f"{prefix}: max err {mx:.3e} vs HF bf16 autocast {hf_mx:.3e}"
    hf_loss_err = abs(float(rec["bf16ac.loss"]) - float(rec["loss"]))
    loss_err = abs(loss.item() - float(rec["loss"]))
    # the probe loss is a random-sign weighted SUM, so one run's error is itself random (HF's own ranges from 1e-3 to
    # 0.3 over the cases): allow 2x HF's, or 0.2 % of the sum's L1 mass, whichever is larger
    mass = out.pooler_output.abs().sum().item() + 0.01 * out.last_hidden_state.abs().sum().item() + \
        0.01 * sum(out.hidden_states[i].abs().sum().item() for i in m["taps"])
    assert loss_err <= max(2.0 * hf_loss_err, 2e-3 * mass), (loss_err, hf_loss_err, mass)
    worst = max(ratios.items(), key=lambda kv: kv[1][0])
    acts = [v[0] for k, v in ratios.items() if not k.startswith("grad.")]
    grads = [v[0] for k, v in ratios.items() if k.startswith("grad.")]
    print(f"[{case}] HIP-bf16 / HF-bf16-autocast rel-L2 error ratio: activations max {max(acts):.2f}, gradients max "
          f"{max(grads):.2f} (worst {worst[0]}: {worst[1][2]:.2e} vs {worst[1][3]:.2e}); loss err {loss_err:.2e} "
          f"vs HF {hf_loss_err:.2e}")


# -----------------------------------------------------------------------------------------------------------------
# full depth, forward AND backward, both modes, against the fp32 oracle
# -----------------------------------------------------------------------------------------------------------------
FULL_GRADS = ["embeddings.patch_embedding.weight", "embeddings.position_embedding.weight",
              "encoder.layers.0.self_attn.q_proj.weight", "encoder.layers.0.mlp.fc1.weight",
              "encoder.layers.0.layer_norm1.weight", "encoder.layers.13.self_attn.v_proj.weight",
              "encoder.layers.13.mlp.fc2.weight", "encoder.layers.13.mlp.fc1.bias",
              "encoder.layers.26.self_attn.out_proj.weight", "encoder.layers.26.mlp.fc1.weight",
              "encoder.layers.26.layer_norm2.bias", "post_layernorm.weight", "head.probe",
              "head.attention.in_proj_weight", "head.mlp.fc2.weight"]


@pytest.fixture(scope="module")
def full_depth_reference(pkg, oracle):
    cfg = pkg.get_config("so400m-patch14-384")
    sd = pkg.weights.seeded_state_dict(cfg, seed=21)
    x = pkg.weights.seeded_pixels(2, 384, 384, seed=22)
    sdr = {k: (v.clone().requires_grad_(True) if k in FULL_GRADS else v) for k, v in sd.items()}
    ref = oracle.vision_forward(x, sdr, cfg, True, True)
    loss = _probe_loss(ref, (1, 14, 27))
    loss.backward()
    keep = {"pooler_output": ref["pooler_output"].detach(), "last_hidden_state": ref["last_hidden_state"].detach(),
            "hs1": ref["hidden_states"][1].detach(), "hs14": ref["hidden_states"][14].detach(),
            "hs27": ref["hidden_states"][27].detach(), "loss": loss.item()}
    for k in FULL_GRADS:
        keep["grad." + k] = sdr[k].grad.detach().clone()
    return cfg, sd, x, keep


# tolerances: relative L2 per tensor; fp32 mode is the north-star "logits within 1e-3" path, bf16 the benchmarked one.
# Values are 2x what the MI355X run measured (printed by the test).
# measured (r2): fp32 act 1.4e-6, grads <= 6.8e-6, pooled 7.0e-6 abs; bf16 act 5.4e-3, grads <= 1.03e-2, pooled 2.2e-2 abs
# bf16x3 = strict mode on the matrix cores (split-bf16 GEMMs; attention in fp32): its bound is the north-star one, pooled
# output within 1e-3 abs of the fp32 reference, plus 2x what round 3 measured for the other tensors.
FULL_TOL = {"fp32": dict(act=3e-6, grad=1.4e-5, pooled_abs=2e-5, loss=2e-6),
            "bf16x3": dict(act=2e-5, grad=3.6e-5, pooled_abs=8e-5, loss=2e-5),   # measured 9.7e-6 / 1.8e-5 / 4.0e-5
            "bf16": dict(act=1.1e-2, grad=2.1e-2, pooled_abs=4.4e-2, loss=8e-3)}


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "bf16"])
def test_full_depth_forward_backward_vs_oracle(mode, pkg, hiplib, full_depth_reference):
    """so400m-patch14-384, all 27 blocks, B=2: outputs, taps 1/14/27 and gradients of the patch embedding, blocks 0, 13,
    26 and the pooling head against the fp32 CPU oracle (bf16 drift through 27 layers of backward is pinned here)."""
    cfg, sd, x, ref = full_depth_reference
    model = pkg.SiglipVisionModelHIP(cfg, compute_dtype=mode)
    model.load_state_dict(sd)
    model = model.to("cuda")
    out = model(pixel_values=x.cuda(), hidden_state_ids=[1, 14, 27], interpolate_pos_encoding=True)
    hs = {1: out.hidden_states[0], 14: out.hidden_states[1], 27: out.hidden_states[2]}
    loss = _probe_loss({"pooler_output": out.pooler_output, "last_hidden_state": out.last_hidden_state,
                        "hidden_states": hs}, (1, 14, 27))
    loss.backward()
    tol = FULL_TOL[mode]
    errs = {"pooler_output": rel_l2(out.pooler_output, ref["pooler_output"]),
            "last_hidden_state": rel_l2(out.last_hidden_state, ref["last_hidden_state"]),
            "hs1": rel_l2(hs[1], ref["hs1"]), "hs14": rel_l2(hs[14], ref["hs14"]), "hs27": rel_l2(hs[27], ref["hs27"])}
    pooled_abs = (out.pooler_output.detach().cpu() - ref["pooler_output"]).abs().max().item()
    named = dict(model.named_parameters())
    gerrs = {k: rel_l2(named[k].grad, ref["grad." + k]) for k in FULL_GRADS}
    print(f"[full depth {mode}] pooled max|err| {pooled_abs:.2e} (scale {ref['pooler_output'].abs().max():.2f}); rel-L2: "
          + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()) + " | grads: "
          + ", ".join(f"{k.replace('encoder.layers.', 'L')} {v:.2e}" for k, v in gerrs.items())
          + f" | loss {loss.item():.5f} vs {ref['loss']:.5f}")
    assert pooled_abs <= tol["pooled_abs"]
    for k, v in errs.items():
        assert v <= tol["act"], (k, v)
    for k, v in gerrs.items():
        assert v <= tol["grad"], (k, v)
    assert abs(loss.item() - ref["loss"]) <= tol["loss"] * max(1.0, abs(ref["loss"]))
    if mode in ("fp32", "bf16x3"):
        assert pooled_abs < 1e-3     # the north-star bound ("logits within 1e-3 of the HF reference"), 140x margin


# -----------------------------------------------------------------------------------------------------------------
# BASELINE.json config 2: SigLIP-2-base-patch16-224 full fine-tune, bf16, batch 256
# -----------------------------------------------------------------------------------------------------------------
def test_config2_base_patch16_224_full_finetune(pkg, oracle, hiplib):
    """All 12 blocks of base-patch16-224 (D 768, head_dim 64, N 196): (a) B=8 forward+backward in bf16 against the fp32
    oracle, (b) the configuration's real batch, 256 images: finite, and two runs are bitwise identical."""
    cfg = pkg.get_config("base-patch16-224")
    sd = pkg.weights.seeded_state_dict(cfg, seed=31)
    model = pkg.SiglipVisionModelHIP(cfg, compute_dtype="bf16")
    model.load_state_dict(sd)
    model = model.to("cuda")
    x = pkg.weights.seeded_pixels(8, 224, 224, seed=32)
    watch = ["embeddings.patch_embedding.weight", "encoder.layers.0.self_attn.q_proj.weight",
             "encoder.layers.6.mlp.fc1.weight", "encoder.layers.11.mlp.fc2.weight", "head.mlp.fc1.weight"]
    sdr = {k: (v.clone().requires_grad_(True) if k in watch else v) for k, v in sd.items()}
    ref = oracle.vision_forward(x, sdr, cfg, False, False)
    (ref["pooler_output"].square().mean() + ref["last_hidden_state"].mean()).backward()
    out = model(pixel_values=x.cuda())
    (out.pooler_output.square().mean() + out.last_hidden_state.mean()).backward()
    e_p, e_l = rel_l2(out.pooler_output, ref["pooler_output"]), rel_l2(out.last_hidden_state, ref["last_hidden_state"])
    named = dict(model.named_parameters())
    ge = {k: rel_l2(named[k].grad, sdr[k].grad) for k in watch}
    print(f"[config 2, B=8 bf16] rel-L2 pooled {e_p:.2e} last {e_l:.2e} grads " +
          ", ".join(f"{k.replace('encoder.layers.', 'L')} {v:.2e}" for k, v in ge.items()))
    assert e_p <= 8e-3 and e_l <= 8e-3
    assert max(ge.values()) <= 3e-2
    # the configuration's batch
    xb = pkg.weights.seeded_pixels(256, 224, 224, seed=33).cuda()
    runs = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        o = model(pixel_values=xb)
        (o.pooler_output.square().mean() + o.last_hidden_state.mean()).backward()
        runs.append((o.pooler_output.detach().clone(), named["encoder.layers.5.mlp.fc1.weight"].grad.clone(),
                     named["embeddings.patch_embedding.weight"].grad.clone()))
    assert all(torch.isfinite(t).all() for t in runs[0])
    assert all(torch.equal(a, b) for a, b in zip(*runs))
    # image 3 of the 256-batch equals the same image run in a batch of 8 (images are independent: LayerNorm only)
    with torch.no_grad():
        small = model(pixel_values=xb[:8]).pooler_output
    assert torch.allclose(runs[0][0][:8], small, atol=1e-5)


# -----------------------------------------------------------------------------------------------------------------
# BASELINE.json config 1: frozen base-patch16-224 backbone + linear head, batch 4 (GPU counterpart of the CPU plumbing run)
# -----------------------------------------------------------------------------------------------------------------
def test_config1_frozen_base_linear_head_batch4(pkg, oracle, hiplib):
    """cifake_binary_classifier.py 'tiny' head (Dropout -> Linear(D,1), :657-662) on a frozen base-patch16-224 encoder,
    B=4: logits against oracle o CPU head, gradients reach ONLY the head, and match the CPU composition."""
    import copy
    H = pkg.heads
    cfg = pkg.get_config("base-patch16-224")
    sd = pkg.weights.seeded_state_dict(cfg, seed=41)
    enc = pkg.OpenClipStyleEncoder(cfg, "bf16")
    enc.visual.load_state_dict(sd)
    for p in enc.parameters():
        p.requires_grad = False                     # --freeze_backbone (hidf_video_classifier.py:2914-2915 idiom)
    torch.manual_seed(3)
    clf = H.FastBinaryClassifierHIP(enc, model_size="tiny").eval()
    head_cpu = copy.deepcopy(clf.head)
    clf = clf.cuda()
    x = pkg.weights.seeded_pixels(4, 224, 224, seed=42)
    y = torch.tensor([1.0, 0.0, 0.0, 1.0])
    logits = clf(x.cuda())
    F = torch.nn.functional
    F.binary_cross_entropy_with_logits(logits, y.cuda()).backward()
    assert all(p.grad is None for p in enc.parameters())
    ref = oracle.vision_forward(x, sd, cfg, False, False)
    ref_logits = head_cpu(ref["pooler_output"])
    F.binary_cross_entropy_with_logits(ref_logits, y).backward()
    err = (logits.detach().cpu() - ref_logits.detach()).abs().max().item()
    print(f"[config 1] logits {logits.detach().cpu().tolist()} max|err| vs CPU composition {err:.2e}")
    assert err <= 2.5e-3        # measured 1.24e-3: bf16 encoder (12 blocks) -> L2-norm -> LN -> attention -> Linear
    for (n, p), (_, q) in zip(clf.head.named_parameters(), head_cpu.named_parameters()):
        assert p.grad is not None, n
        assert rel_l2(p.grad, q.grad) <= 5e-2 or q.grad.abs().max() < 1e-6, n
    # strict mode reproduces the CPU logits to the north-star bound
    enc32 = pkg.OpenClipStyleEncoder(cfg, "fp32")
    enc32.visual.load_state_dict(sd)
    clf32 = H.FastBinaryClassifierHIP(enc32, model_size="tiny").eval()
    clf32.head.load_state_dict(head_cpu.state_dict())
    with torch.no_grad():
        l32 = clf32.cuda()(x.cuda())
    assert (l32.cpu() - ref_logits.detach()).abs().max().item() < 1e-3


# -----------------------------------------------------------------------------------------------------------------
# BASELINE.json config 5: video track, 32-frame clips, per-frame so400m encoder + temporal mean-pool
# -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_config5_video_32_frames(mode, pkg, oracle, hiplib):
    """BinaryVideoClassifierHIP on (2, 32, 3, 384, 384) clips: encoder batch = 64 frames of the full-width so400m block
    (so400m-1layer keeps the CPU oracle affordable), L2-norm, temporal mean, MLP -> (2,) logits, forward and backward,
    against oracle o CPU head (hidf_video_classifier.py:299-320).

    Conditioning note (measured, gpurun_out/cfg5.log): with random weights the 64 frame embeddings are nearly parallel
    (cosine 0.97), so with OPPOSITE clip labels the two clips' contributions to every weight gradient cancel about 6:1 and
    any forward rounding is amplified by that factor (bf16: 1.4 % on d loss/d embedding becomes 4-8 % on weight
    gradients, while strict fp32 stays at 1e-5 on the very same problem).  The strict mode is therefore checked on the
    hard opposite-label problem, the bf16 mode on equal labels, where its error is the error of the kernels."""
    import copy
    H = pkg.heads
    cfg = pkg.get_config("so400m-1layer")
    sd = pkg.weights.seeded_state_dict(cfg, seed=51)
    enc = pkg.OpenClipStyleEncoder(cfg, mode)
    enc.visual.load_state_dict(sd)
    torch.manual_seed(5)
    vid = H.BinaryVideoClassifierHIP(enc, num_frames=32).eval()
    head_cpu = copy.deepcopy(vid.head)
    vid = vid.cuda()
    clips = pkg.weights.seeded_pixels(64, 384, 384, seed=52).view(2, 32, 3, 384, 384)
    y = torch.tensor([1.0, 0.0] if mode == "fp32" else [1.0, 1.0])
    F = torch.nn.functional
    logits = vid(clips.cuda())
    assert logits.shape == (2,)
    F.binary_cross_entropy_with_logits(logits, y.cuda()).backward()
    watch = ["encoder.layers.0.mlp.fc1.weight", "encoder.layers.0.self_attn.q_proj.weight",
             "head.attention.in_proj_weight", "embeddings.patch_embedding.weight"]
    sdr = {k: (v.clone().requires_grad_(True) if k in watch else v) for k, v in sd.items()}
    ref = oracle.vision_forward(clips.view(64, 3, 384, 384), sdr, cfg, False, False)
    ref_logits = head_cpu(ref["pooler_output"], batch_size=2)
    F.binary_cross_entropy_with_logits(ref_logits, y).backward()
    err = (logits.detach().cpu() - ref_logits.detach()).abs().max().item()
    named = dict(enc.visual.named_parameters())
    ge = {k: rel_l2(named[k].grad, sdr[k].grad) for k in watch}
    hg = max(rel_l2(p.grad, q.grad) for (n, p), (_, q) in zip(vid.head.named_parameters(), head_cpu.named_parameters())
             if q.grad.abs().max() > 1e-7)
    print(f"[config 5 {mode}] clip logits {logits.detach().cpu().tolist()} max|err| {err:.2e}; encoder grads "
          + ", ".join(f"{k.split('.')[-3]}.{k.split('.')[-2]} {v:.2e}" for k, v in ge.items()) + f"; head grads {hg:.2e}")
    if mode == "fp32":      # measured 1.2e-5 worst
        assert err <= 1e-5 and max(ge.values()) <= 3e-5 and hg <= 3e-5
    else:
        assert err <= 5e-4          # measured 2.2e-4
        assert max(ge.values()) <= 2.4e-2 and hg <= 2.2e-2      # measured 1.18e-2 / 1.10e-2


# -----------------------------------------------------------------------------------------------------------------
# BASELINE.json config 4: SID multi-task model, bf16 autocast, the script's default 11-tap E=512 decoder
# -----------------------------------------------------------------------------------------------------------------
def test_config4_sid_default_decoder_bf16_autocast(pkg, oracle, hiplib):
    """SigLIP2MTL with the reference's DEFAULT decoder (taps 1..10 and -1, embed_dim 512, Siglip2sidafrozen.py:1139-1140)
    under torch.autocast(bf16) exactly as its train step runs (:1375) — the path tests/bench_mtl.py times, including the
    HIP GEMM dispatch of the decoder's wide 1x1 convolutions — against the fp32 CPU composition oracle o heads.
    Encoder: 12 full-width so400m blocks at 224 px (N 256) so that all 11 taps exist and the CPU side stays affordable."""
    import copy
    H = pkg.heads
    cfg = pkg.get_config(dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=12, num_attention_heads=16,
                              image_size=224, patch_size=14))
    sd = pkg.weights.seeded_state_dict(cfg, seed=61)
    enc = pkg.SiglipVisionModelHIP(cfg, compute_dtype="bf16")
    enc.load_state_dict(sd)
    torch.manual_seed(7)
    seg_layers = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, -1)
    model = H.SigLIP2MTL(enc, seg_layers=seg_layers, embed_dim=512, freeze_below=9)
    heads_cpu = copy.deepcopy({"cls": model.cls_head, "dec": model.decoder})
    model = model.cuda()
    B = 8    # M = 8 * 256 = 2048 tokens: the decoder's Linear layers take the HIP MFMA path (heads._linear_tokens)
    x = pkg.weights.seeded_pixels(B, 224, 224, seed=62)
    y = torch.tensor([0, 1, 2, 1, 0, 2, 2, 1])
    masks = (pkg.weights.seeded_tensor("masks", (B, 1, 224, 224), 1.0) > 0.2).float()
    has = torch.tensor([True] * B)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        cls_logit, seg_logits = model(x.cuda())
        loss = H.mtl_loss(cls_logit.float(), seg_logits.float(), y.cuda(), masks.cuda(), has.cuda())
    loss.backward()
    watch = ["encoder.layers.9.mlp.fc1.weight", "encoder.layers.11.self_attn.q_proj.weight", "head.mlp.fc2.weight"]
    sdr = {k: (v.clone().requires_grad_(True) if k in watch else v) for k, v in sd.items()}
    ref = oracle.vision_forward(x, sdr, cfg, True, True)
    cls_ref = heads_cpu["cls"](ref["pooler_output"])
    feats = [ref["hidden_states"][i + 1 if i >= 0 else cfg.num_hidden_layers] for i in seg_layers]
    seg_ref = heads_cpu["dec"](feats, (16, 16), target_size=224)
    loss_ref = H.mtl_loss(cls_ref, seg_ref, y, masks, has)
    loss_ref.backward()
    e_cls = (cls_logit.float().cpu() - cls_ref).abs().max().item()
    e_seg = (seg_logits.float().cpu() - seg_ref).abs().max().item()
    seg_l2 = rel_l2(seg_logits.float(), seg_ref)
    named = dict(model.encoder.named_parameters())
    ge = {k: rel_l2(named[k].grad, sdr[k].grad) for k in watch}
    gd = rel_l2(model.decoder.projs[10].proj.weight.grad, heads_cpu["dec"].projs[10].proj.weight.grad)
    gf = rel_l2(model.decoder.fuse[0].weight.grad, heads_cpu["dec"].fuse[0].weight.grad)
    print(f"[config 4, bf16 autocast, 11 taps E=512] cls max|err| {e_cls:.2e} (scale {cls_ref.abs().max():.2f}); seg "
          f"max|err| {e_seg:.2e} rel-L2 {seg_l2:.2e} (scale {seg_ref.abs().max():.2f}); loss {loss.item():.4f} vs "
          f"{loss_ref.item():.4f}; grads {ge} decoder proj {gd:.2e} fuse {gf:.2e}")
    # stated tolerance of the benchmarked path (2x the values measured on MI355X: 9.7e-3, 8.9e-3, 1.5e-3, 1.6e-2,
    # 6.4e-3): class logits 2e-2 abs, mask logits 1.8e-2 rel-L2, loss 0.3 %, gradients 3.3e-2 / 1.3e-2 rel-L2
    assert e_cls <= 2e-2
    assert seg_l2 <= 1.8e-2
    assert abs(loss.item() - loss_ref.item()) <= 3e-3 * abs(loss_ref.item())
    assert named["encoder.layers.8.mlp.fc1.weight"].grad is None
    assert max(ge.values()) <= 3.3e-2 and gd <= 1.3e-2 and gf <= 1.3e-2

"""BASELINE.json configs 3, 4 and 5 at FULL depth (all 27 so400m blocks at 384 px) against oracle o CPU heads.

Round 2 pinned these configurations at reduced depth only (12 blocks at 224 px for the SID model, one block for the video
model, no model at all for the HiDF image head).  One CPU oracle forward of the B=2 batch is shared by the image-head and
SID tests (module-scoped fixture, graph kept; each test differentiates its own loss through it with
``torch.autograd.grad(..., retain_graph=True)``), so the CPU cost of the 27-block graph is paid once.
"""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
F = torch.nn.functional

WATCH = ["encoder.layers.26.mlp.fc1.weight", "encoder.layers.25.self_attn.q_proj.weight",
         "encoder.layers.21.mlp.fc1.weight", "encoder.layers.26.self_attn.q_proj.weight",
         "encoder.layers.3.layer_norm1.weight", "encoder.layers.26.layer_norm2.bias", "post_layernorm.weight",
         "head.layernorm.weight", "head.mlp.fc2.weight"]


def rel_l2(got, ref):
    got, ref = got.detach().double().cpu().reshape(-1), ref.detach().double().cpu().reshape(-1)
    return ((got - ref).norm() / (ref.norm() + 1e-30)).item()


@pytest.fixture(scope="module")
def so400m_graph(pkg, oracle):
    """fp32 CPU oracle forward of so400m-patch14-384 on a seeded B=2 batch, every hidden state, autograd graph kept."""
    cfg = pkg.get_config("so400m-patch14-384")
    sd = pkg.weights.seeded_state_dict(cfg, seed=31)
    x = pkg.weights.seeded_pixels(2, 384, 384, seed=32)
    sdr = {k: (v.clone().requires_grad_(True) if k in WATCH else v) for k, v in sd.items()}
    ref = oracle.vision_forward(x, sdr, cfg, True, True)
    return cfg, sd, x, sdr, ref


# measured on MI355X (round 3, printed by the test); bounds are 2x.  bf16: logits 7.7e-4 abs; gradients 1.3e-2 .. 1.7e-2
# rel-L2 except the query projections of the last two blocks (5.3e-2 / 7.8e-2): with a loss that reaches the encoder through
# two pooled logits only, d loss / d W_q of the last blocks is a small difference of large per-token terms (the softmax
# Jacobian removes the common part), so bf16 rounding of P and dP shows up amplified there; the strict mode has 1e-5 on the
# very same tensors, i.e. it is conditioning, not a kernel defect (same effect as documented for config 5 in round 2).
CFG3_TOL = {"fp32": dict(logit=2e-5, grad=4e-5, grad_q=4e-5), "bf16": dict(logit=2e-3, grad=3.5e-2, grad_q=1.6e-1)}


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_config3_hidf_image_binary_head_on_full_so400m(mode, pkg, hiplib, so400m_graph):
    """BASELINE config 3: so400m-patch14-384 + the HiDF image head (simple_classifier.py:141-148,159-164), fine-tuned with
    the script's name-based partial unfreezing (:483-496, generalised from blocks.23/22 to the last two blocks): logits,
    and gradients of the last two blocks, of a LayerNorm deep in the frozen part ('norm' matches norm1/norm2 of every
    block), of the final norm and of the pooling head's norm; everything else must stay without a gradient."""
    cfg, sd, x, sdr, ref = so400m_graph
    enc = pkg.OpenClipStyleEncoder(cfg, mode)
    enc.visual.load_state_dict(sd)
    torch.manual_seed(3)
    model = pkg.heads.ImageBinaryClassifierHIP(enc).eval()
    head_cpu = copy.deepcopy(model.head)
    model = model.cuda()
    n_unfrozen = model.partially_unfreeze_backbone()
    names = {n for n, p in model.backbone.named_parameters() if p.requires_grad}
    assert "visual.trunk.blocks.26.mlp.fc1.weight" in names and "visual.trunk.blocks.25.attn.q_proj.bias" in names
    assert "visual.trunk.blocks.3.norm1.weight" in names and "visual.trunk.norm.bias" in names
    assert "visual.trunk.attn_pool.norm.weight" in names
    assert "visual.trunk.blocks.24.mlp.fc1.weight" not in names and "visual.trunk.patch_embed.proj.weight" not in names
    assert "visual.trunk.attn_pool.latent" not in names
    per_block = sum(p.numel() for p in enc.visual.encoder.layers[0].parameters())
    assert n_unfrozen == 2 * per_block + 25 * 4 * cfg.hidden_size + 2 * cfg.hidden_size + 2 * cfg.hidden_size

    y = torch.tensor([1.0, 0.0])
    logits = model(x.cuda())
    assert logits.shape == (2,)
    F.binary_cross_entropy_with_logits(logits, y.cuda()).backward()
    ref_logits = head_cpu(ref["pooler_output"])
    watch = [k for k in WATCH if k not in ("encoder.layers.21.mlp.fc1.weight", "head.mlp.fc2.weight")]
    gref = torch.autograd.grad(F.binary_cross_entropy_with_logits(ref_logits, y), [sdr[k] for k in watch],
                               retain_graph=True)
    named = dict(enc.visual.named_parameters())
    err = (logits.detach().cpu() - ref_logits.detach()).abs().max().item()
    ge = {k: rel_l2(named[k].grad, g) for k, g in zip(watch, gref)}
    hg = max(rel_l2(p.grad, q) for p, q in zip(model.head.parameters(), torch.autograd.grad(
        F.binary_cross_entropy_with_logits(head_cpu(ref["pooler_output"].detach()), y), list(head_cpu.parameters()))))
    print(f"[config 3 {mode}] logits {logits.detach().cpu().tolist()} max|err| {err:.2e}; grads "
          + ", ".join(f"{k.replace('encoder.layers.', 'L')} {v:.2e}" for k, v in ge.items()) + f"; head {hg:.2e}")
    assert named["encoder.layers.21.mlp.fc1.weight"].grad is None and named["head.mlp.fc2.weight"].grad is None
    assert named["embeddings.patch_embedding.weight"].grad is None and named["head.probe"].grad is None
    tol = CFG3_TOL[mode]
    assert err <= tol["logit"]
    for k, v in ge.items():
        assert v <= (tol["grad_q"] if "q_proj" in k else tol["grad"]), (k, v)
    assert hg <= tol["grad"]


def test_config4_sid_multitask_full_27_blocks_384(pkg, hiplib, so400m_graph):
    """BASELINE config 4 at full size: SigLIP2MTL on all 27 so400m blocks at 384 px with the script's DEFAULT decoder (taps
    1..10 and -1, embed_dim 512, Siglip2sidafrozen.py:1139-1140), embeddings and blocks < 21 frozen (:757-768), under
    torch.autocast(bf16) as the train step runs it (:1375), against oracle o CPU heads in fp32."""
    cfg, sd, x, sdr, ref = so400m_graph
    H = pkg.heads
    enc = pkg.SiglipVisionModelHIP(cfg, compute_dtype="bf16")
    enc.load_state_dict(sd)
    torch.manual_seed(7)
    seg_layers = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, -1)
    model = H.SigLIP2MTL(enc, seg_layers=seg_layers, embed_dim=512, freeze_below=21)
    heads_cpu = copy.deepcopy({"cls": model.cls_head, "dec": model.decoder})
    model = model.cuda()
    y = torch.tensor([2, 1])
    masks = (pkg.weights.seeded_tensor("masks_full", (2, 1, 384, 384), 1.0) > 0.2).float()
    has = torch.tensor([True, True])
    with torch.autocast("cuda", dtype=torch.bfloat16):
        cls_logit, seg_logits = model(x.cuda())
        loss = H.mtl_loss(cls_logit.float(), seg_logits.float(), y.cuda(), masks.cuda(), has.cuda())
    assert cls_logit.shape == (2, 3) and seg_logits.shape == (2, 1, 384, 384)
    loss.backward()
    cls_ref = heads_cpu["cls"](ref["pooler_output"])
    feats = [ref["hidden_states"][i + 1 if i >= 0 else cfg.num_hidden_layers] for i in seg_layers]
    seg_ref = heads_cpu["dec"](feats, (27, 27), target_size=384)
    loss_ref = H.mtl_loss(cls_ref, seg_ref, y, masks, has)
    watch = ["encoder.layers.21.mlp.fc1.weight", "encoder.layers.26.self_attn.q_proj.weight", "head.mlp.fc2.weight",
             "post_layernorm.weight"]
    dec_params = [heads_cpu["dec"].projs[10].proj.weight, heads_cpu["dec"].fuse[0].weight, heads_cpu["cls"].weight]
    gref = torch.autograd.grad(loss_ref, [sdr[k] for k in watch] + dec_params, retain_graph=True)
    named = dict(model.encoder.named_parameters())
    e_cls = (cls_logit.float().cpu() - cls_ref).abs().max().item()
    seg_l2 = rel_l2(seg_logits.float(), seg_ref)
    ge = {k: rel_l2(named[k].grad, g) for k, g in zip(watch, gref)}
    gd = rel_l2(model.decoder.projs[10].proj.weight.grad, gref[len(watch)])
    gf = rel_l2(model.decoder.fuse[0].weight.grad, gref[len(watch) + 1])
    gc = rel_l2(model.cls_head.weight.grad, gref[len(watch) + 2])
    print(f"[config 4 full depth, bf16 autocast] cls max|err| {e_cls:.2e} (scale {cls_ref.abs().max():.2f}); seg rel-L2 "
          f"{seg_l2:.2e} (scale {seg_ref.abs().max():.2f}); loss {loss.item():.4f} vs {loss_ref.item():.4f}; grads "
          + ", ".join(f"{k.replace('encoder.layers.', 'L')} {v:.2e}" for k, v in ge.items())
          + f"; decoder proj {gd:.2e} fuse {gf:.2e} cls {gc:.2e}")
    assert named["encoder.layers.20.mlp.fc1.weight"].grad is None
    assert named["embeddings.patch_embedding.weight"].grad is None
    # bounds = 2x the values measured on MI355X in round 3 (printed above): cls 5.9e-3 abs, seg 3.2e-3 rel-L2, loss 4e-4
    # relative, gradients 8.6e-3 .. 9.3e-3 (decoder 5.1e-3 .. 8.4e-3) and 5.6e-2 for the last block's query projection (the
    # ill-conditioned tensor explained at CFG3_TOL)
    assert e_cls <= 1.2e-2
    assert seg_l2 <= 6.4e-3
    assert abs(loss.item() - loss_ref.item()) <= 1e-3 * abs(loss_ref.item())
    for k, v in ge.items():
        assert v <= (1.2e-1 if "q_proj" in k else 1.9e-2), (k, v)
    assert gd <= 1.6e-2 and gf <= 1.1e-2 and gc <= 1.7e-2


@pytest.fixture(scope="module")
def clip_reference(pkg, oracle):
    """fp32 CPU oracle embeddings of one seeded 32-frame clip (computed once for both compute modes)."""
    cfg = pkg.get_config("so400m-patch14-384")
    sd = pkg.weights.seeded_state_dict(cfg, seed=41)
    clip = pkg.weights.seeded_pixels(32, 384, 384, seed=42).view(1, 32, 3, 384, 384)
    with torch.no_grad():
        ref = oracle.vision_forward(clip.view(32, 3, 384, 384), sd, cfg, False, False)["pooler_output"]
    return cfg, sd, clip, ref


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
def test_config5_video_clip_of_32_frames_through_all_27_blocks(mode, pkg, hiplib, clip_reference):
    """BASELINE config 5 at full depth, the script's frozen-backbone default (hidf_video_classifier.py:2913-2916): one clip
    of 32 frames -> 32 images through all 27 so400m blocks -> L2-norm -> temporal mean -> MLP -> one logit
    (:299-320), forward only, against oracle o CPU head; strict mode = the one that runs on the matrix cores."""
    H = pkg.heads
    cfg, sd, clip, ref_pooled = clip_reference
    enc = pkg.OpenClipStyleEncoder(cfg, mode)
    enc.visual.load_state_dict(sd)
    torch.manual_seed(5)
    vid = H.BinaryVideoClassifierHIP(enc, num_frames=32).eval()
    head_cpu = copy.deepcopy(vid.head)
    vid = vid.cuda()
    for p in vid.vision_encoder.parameters():
        p.requires_grad = False
    with torch.no_grad():
        logit = vid(clip.cuda())
        feats = enc.encode_image(clip.view(32, 3, 384, 384).cuda())
        ref_logit = head_cpu(ref_pooled, batch_size=1)
    assert logit.shape == (1,)
    err = (logit.cpu() - ref_logit).abs().max().item()
    ferr = rel_l2(feats, ref_pooled)
    print(f"[config 5 full depth {mode}] clip logit {logit.item():.6f} vs {ref_logit.item():.6f} (|err| {err:.2e}); "
          f"frame embeddings rel-L2 {ferr:.2e}")
    # bounds = 2x measured (round 3)
    if mode == "bf16x3":
        assert err <= 1e-4 and ferr <= 4e-5
    else:
        assert err <= 1e-2 and ferr <= 2e-2

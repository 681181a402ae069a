"""SID decoder tail on the GPU (second half; csrc/decoder_tail.hip): sigmoid(gate)*x in one pass and
`bce_dice_loss` evaluated straight from the low-resolution logit map, against the reference formulation — the PyTorch
composition that tests/test_heads.py pins to the reference's own functions (Siglip2sidafrozen.py:174-181,741-745)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
F = torch.nn.functional


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize("shape", [(729 * 3, 5632), (4096,), (37, 8, 16)])
def test_gate_mul_forward_backward(pkg, hiplib, dtype, tol, shape):
    H = pkg.heads
    torch.manual_seed(1)
    g = (torch.randn(shape, device="cuda") * 2).to(dtype).requires_grad_(True)
    x = torch.randn(shape, device="cuda").to(dtype).requires_grad_(True)
    dy = torch.randn(shape, device="cuda").to(dtype)
    y = H._gate_mul(g, x)
    y.backward(dy)
    gr, xr = g.detach().float().requires_grad_(True), x.detach().float().requires_grad_(True)
    ref = torch.sigmoid(gr) * xr
    ref.backward(dy.float())

    def rel(a, b):
        return ((a.float() - b).norm() / (b.norm() + 1e-30)).item()
    assert y.dtype == dtype and rel(y, ref) < tol
    assert rel(g.grad, gr.grad) < tol and rel(x.grad, xr.grad) < tol


@pytest.mark.parametrize("B,g,S", [(3, 27, 384), (2, 16, 224), (4, 3, 42), (1, 14, 224), (2, 27, 320)])
def test_seg_loss_from_lowres_matches_upsample_then_bce_dice(pkg, hiplib, B, g, S):
    """Forward value and d loss / d low-res logits against F.interpolate(bilinear, align_corners=False) -> bce_dice_loss
    on the images that carry a mask (the reference's `seg_logits[has_mask]`), including the empty selection."""
    H = pkg.heads
    torch.manual_seed(B * 100 + g)
    lr0 = torch.randn(B, 1, g, g, device="cuda") * 2.5
    masks = (torch.rand(B, 1, S, S, device="cuda") < 0.3).float()
    for sel in ([True] * B, [i % 2 == 0 for i in range(B)], [False] * B):
        has = torch.tensor(sel, device="cuda")
        lr = lr0.clone().requires_grad_(True)
        loss = 1.7 * H.bce_dice_loss_from_lowres(lr, masks, has)
        loss.backward()
        ref_lr = lr0.clone().requires_grad_(True)
        if any(sel):
            up = F.interpolate(ref_lr, size=(S, S), mode="bilinear", align_corners=False)
            ref = 1.7 * H.bce_dice_loss(up[has], masks[has])
            ref.backward()
            assert abs(loss.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
            err = (lr.grad - ref_lr.grad).abs().max().item() / (ref_lr.grad.abs().max().item() + 1e-30)
            assert err < 2e-5, err
        else:
            assert loss.item() == 0.0 and lr.grad.abs().max().item() == 0.0
    # bitwise reproducible (no atomics anywhere)
    a = lr0.clone().requires_grad_(True)
    b = lr0.clone().requires_grad_(True)
    H.bce_dice_loss_from_lowres(a, masks).backward()
    H.bce_dice_loss_from_lowres(b, masks).backward()
    assert torch.equal(a.grad, b.grad)


def test_training_loss_equals_forward_then_mtl_loss(pkg, hiplib):
    """SigLIP2MTL.training_loss (decoder tail fused: never forms the (B,1,S,S) logits) against forward() + mtl_loss, values
    and gradients, on the frozen-prefix model in strict mode."""
    H = pkg.heads
    cfg = pkg.get_config("hostile")
    enc = pkg.SiglipVisionModelHIP(cfg, compute_dtype="fp32")
    enc.load_state_dict(pkg.weights.seeded_state_dict(cfg, seed=13))
    torch.manual_seed(0)
    model = H.SigLIP2MTL(enc, seg_layers=(0, 1, -1), embed_dim=32, freeze_below=1).cuda()
    x = pkg.weights.seeded_pixels(3, 56, 56, seed=17).cuda()
    y = torch.tensor([2, 0, 1]).cuda()
    masks = (pkg.weights.seeded_tensor("masks", (3, 1, 56, 56), 1.0) > 0.2).float().cuda()
    has = torch.tensor([True, False, True]).cuda()
    cls, seg = model(x)
    ref = H.mtl_loss(cls, seg, y, masks, has, lam_seg=0.8)
    ref.backward()
    want = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    loss, cls2, seg_lr = model.training_loss(x, y, masks, has, lam_seg=0.8)
    loss.backward()
    assert seg_lr.shape == (3, 1, 4, 4) and torch.equal(cls, cls2)
    assert abs(loss.item() - ref.item()) <= 2e-6 * abs(ref.item())
    for n, p in model.named_parameters():
        if n in want and not (n.endswith("k_proj.bias") or n.endswith("in_proj_bias")):   # exactly-zero gradients: noise
            err = (p.grad - want[n]).abs().max().item() / (want[n].abs().max().item() + 1e-30)
            assert err < 5e-5, (n, err)


@pytest.mark.parametrize("B,T,D", [(2, 32, 1152), (3, 4, 64), (1, 1, 768), (5, 7, 100)])
def test_video_l2norm_temporal_mean_forward_backward(pkg, hiplib, B, T, D):
    """heads.l2norm_temporal_mean (one HIP launch each way) against the reference's composition
    f / f.norm(dim=-1, keepdim=True) -> view(B, T, D) -> mean(1)  (hidf_video_classifier.py:308-316)."""
    H = pkg.heads
    torch.manual_seed(B * 10 + T)
    f = (torch.randn(B * T, D, device="cuda") * 3).requires_grad_(True)
    g = torch.randn(B, D, device="cuda")
    out = H.l2norm_temporal_mean(f, B)
    out.backward(g)
    fr = f.detach().clone().requires_grad_(True)
    ref = (fr / fr.norm(dim=-1, keepdim=True)).view(B, T, D).mean(1)
    ref.backward(g)
    assert (out - ref).abs().max().item() < 2e-6
    assert (f.grad - fr.grad).abs().max().item() <= 2e-6 * max(1.0, fr.grad.abs().max().item())

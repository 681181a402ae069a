"""GPU input pipeline (csrc/preprocess.hip): Resize(antialias) + MixUp + Normalize, optionally
written straight into the patch GEMM's operand, against oracle/preprocess_oracle.py (torch's own antialiased bilinear
resize on the CPU = the arithmetic of the reference's torchvision CPU transform; kornia's K.Resize is not installed:
parity with it is unpinned)."""
import importlib.util
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def po():
    spec = importlib.util.spec_from_file_location("preprocess_oracle", os.path.join(ROOT, "oracle", "preprocess_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_oracle_explicit_filter_equals_torch_antialias(po):
    """The explicit triangle-filter restatement against torch's implementation (down- and up-scaling, odd sizes)."""
    g = torch.Generator().manual_seed(0)
    for (h, w, s) in [(64, 80, 48), (33, 47, 42), (30, 30, 42), (100, 64, 56)]:
        x = torch.rand(2, 3, h, w, generator=g)
        ref = torch.nn.functional.interpolate(x, size=(s, s), mode="bilinear", antialias=True, align_corners=False)
        assert (po.triangle_resize(x, s) - ref).abs().max().item() < 2e-6
    x = torch.rand(1, 3, 28, 28, generator=g)
    patches = po.patch_operand(x, 14)
    assert patches.shape == (4, 640) and torch.equal(patches[3, 196 + 14:196 + 28], x[0, 1, 15, 14:28])
    assert patches[:, 588:].abs().max().item() == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("src", ["u8", "f32"])
@pytest.mark.parametrize("hs,ws,size", [(512, 512, 384), (300, 420, 224), (384, 384, 384), (160, 200, 224), (97, 131, 42)])
def test_resize_normalize_matches_oracle(pkg, hiplib, po, src, hs, ws, size):
    g = torch.Generator().manual_seed(hs * 7 + ws)
    if src == "u8":
        img = torch.randint(0, 256, (3, hs, ws, 3), generator=g, dtype=torch.uint8)
    else:
        img = torch.rand(3, 3, hs, ws, generator=g)
    idx = torch.tensor([2, 0, 1])
    for mix in (None, idx):
        ref = po.gpu_transform(img, size, 0.5, 0.5, mix, 0.3)
        got = pkg.preprocess.resize_normalize(img.cuda(), size, 0.5, 0.5, None if mix is None else mix.cuda(), 0.3)
        assert got.shape == (3, 3, size, size)
        assert (got.cpu() - ref).abs().max().item() < 5e-6, (src, hs, ws, size, mix is not None)
    mod = pkg.preprocess.GpuTransform(size).cuda()
    assert torch.equal(mod(img.cuda()), pkg.preprocess.resize_normalize(img.cuda(), size))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name,hs,ws", [("hostile", 97, 131), ("tiny", 50, 64), ("so400m-1layer", 512, 512)])
def test_patch_operand_feeds_the_encoder(pkg, hiplib, po, cfg_name, hs, ws):
    """to_patch_operand == patch gather of the oracle transform (bit-exact after the same bf16 rounding), and the encoder
    run from it (no fp32 pixel tensor, no im2col pass) equals the encoder run on the materialised pixels, forward and
    backward (the saved operand feeds the patch-embedding weight gradient)."""
    cfg = pkg.get_config(cfg_name)
    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, 256, (2, hs, ws, 3), generator=g, dtype=torch.uint8)
    S, P = cfg.image_size, cfg.patch_size
    ref_px = po.gpu_transform(img, S)
    ref_op = po.patch_operand(ref_px, P)
    for mode, dt, tol in (("fp32", torch.float32, 5e-6), ("bf16", torch.bfloat16, 8e-3)):
        op = pkg.preprocess.to_patch_operand(img.cuda(), cfg, compute_dtype=mode)
        assert op.data.dtype == dt and op.data.shape == ref_op.shape and (op.batch, op.height, op.width) == (2, S, S)
        assert (op.data.float().cpu() - ref_op).abs().max().item() < tol
        assert op.data[:, 3 * P * P:].float().abs().sum().item() == 0.0     # K padding (none when 3*p*p % 64 == 0)
        model = pkg.SiglipVisionModelHIP(cfg, compute_dtype=mode)
        model.load_state_dict(pkg.weights.seeded_state_dict(cfg, seed=2))
        model = model.cuda()
        px = pkg.preprocess.resize_normalize(img.cuda(), S)
        outs = []
        for kw in (dict(pixel_values=px), dict(patches=op)):
            model.zero_grad(set_to_none=True)
            o = model(**kw)
            (o.pooler_output.square().mean() + o.last_hidden_state.mean()).backward()
            outs.append((o.pooler_output.detach().clone(), model.embeddings.patch_embedding.weight.grad.clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def _aug_params():
    import math
    a1, a2 = math.radians(4.0), math.radians(-5.0)
    return [
        dict(flip=False, cos=1.0, sin=0.0, brightness=1.0, contrast=1.0, saturation=1.0, hue=0.0, order=None),  # identity
        dict(flip=True, cos=math.cos(a1), sin=math.sin(a1), brightness=1.07, contrast=0.93, saturation=1.09, hue=0.04,
             order=[2, 0, 3, 1]),
        dict(flip=False, cos=math.cos(a2), sin=math.sin(a2), brightness=0.91, contrast=1.1, saturation=0.9, hue=-0.05,
             order=[1, 3, 0, 2]),
        dict(flip=True, cos=1.0, sin=0.0, brightness=1.1, contrast=1.05, saturation=1.0, hue=0.0, order=[0, 1, 2, 3]),
    ]


def test_oracle_colour_operators_are_identities_at_neutral_parameters(po):
    """The restated operators at their neutral parameters (factor 1, hue 0, angle 0) reproduce the plain transform; a pure
    90-degree-free sanity check of the HSV round trip and of the rotation sampler."""
    g = torch.Generator().manual_seed(5)
    img = torch.rand(2, 3, 40, 40, generator=g)
    neutral = [dict(flip=False, cos=1.0, sin=0.0, brightness=1.0, contrast=1.0, saturation=1.0, hue=0.0, order=[3, 1, 0, 2])] * 2
    assert (po.augment_transform(img, 40, neutral) - po.gpu_transform(img, 40)).abs().max().item() < 2e-6
    x = torch.rand(3, 17, 17, generator=g)
    assert torch.equal(po.rotate_bilinear_zeros(x, 1.0, 0.0), x)
    # a quarter turn maps pixels exactly (no interpolation): out(y, x) = in(x, S-1-y) for the counter-clockwise convention
    r = po.rotate_bilinear_zeros(x, 0.0, 1.0)
    assert (r - x.transpose(-1, -2).flip(-2)).abs().max().item() < 1e-5 or \
        (r - x.transpose(-1, -2).flip(-1)).abs().max().item() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("src", ["u8", "f32"])
@pytest.mark.parametrize("hs,ws,size", [(96, 120, 56), (56, 56, 56), (40, 70, 42)])
def test_augmentation_branch_matches_oracle(pkg, hiplib, po, src, hs, ws, size):
    """Resize -> flip -> rotation(+-5 deg) -> colour jitter (all four operators, three different orders) -> normalize, fused
    (csrc/preprocess.hip), against the explicit restatement in oracle/preprocess_oracle.py; kornia itself: parity unpinned."""
    g = torch.Generator().manual_seed(hs + ws)
    if src == "u8":
        img = torch.randint(0, 256, (4, hs, ws, 3), generator=g, dtype=torch.uint8)
    else:
        img = torch.rand(4, 3, hs, ws, generator=g)
    params = _aug_params()
    ref = po.augment_transform(img, size, params)
    got = pkg.preprocess.augment_resize_normalize(img.cuda(), size, params)
    err = (got.cpu() - ref).abs()
    # hue: a pixel whose two largest channels tie within rounding may pick the other branch of the piecewise HSV formula on
    # either side; both are correct to rounding but differ by O(1e-3) for that pixel — allow a handful
    assert (err > 2e-5).float().mean().item() < 1e-4, err.max().item()
    assert err.max().item() < 5e-3
    assert torch.equal(got[0], pkg.preprocess.resize_normalize(img.cuda(), size)[0])     # identity record == plain transform


@pytest.mark.gpu
def test_augmented_patch_operand_and_module(pkg, hiplib, po):
    cfg = pkg.get_config("hostile")
    g = torch.Generator().manual_seed(11)
    img = torch.randint(0, 256, (4, 60, 75, 3), generator=g, dtype=torch.uint8)
    params = _aug_params()
    S, P = cfg.image_size, cfg.patch_size
    px = pkg.preprocess.augment_resize_normalize(img.cuda(), S, params)
    op = pkg.preprocess.augment_to_patch_operand(img.cuda(), cfg, params, compute_dtype="fp32")
    assert torch.equal(op.data.cpu(), po.patch_operand(px.cpu(), P))                       # same pixels, operand layout
    opb = pkg.preprocess.augment_to_patch_operand(img.cuda(), cfg, params, compute_dtype="bf16")
    assert torch.equal(opb.data.float().cpu(), po.patch_operand(px.cpu(), P).bfloat16().float())
    gen = torch.Generator().manual_seed(3)
    mod = pkg.preprocess.GpuTransform(S, data_augmentation=True, generator=gen).cuda().train()
    want = pkg.preprocess.sample_augmentation(4, torch.Generator().manual_seed(3))
    assert torch.equal(mod(img.cuda()), pkg.preprocess.augment_resize_normalize(img.cuda(), S, want))
    assert torch.equal(mod.eval()(img.cuda()), pkg.preprocess.resize_normalize(img.cuda(), S))   # eval: no augmentation

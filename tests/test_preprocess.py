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

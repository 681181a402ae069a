"""Fused AdamW + global-norm clip (csrc/optimizer.hip) against the reference's own implementation of the step tail:
torch.nn.utils.clip_grad_norm_ followed by torch.optim.AdamW.step() (Siglip2sidafrozen.py:1396-1398), run on the CPU in
fp32 on the same seeded tensors.  Floating point: the kernel follows torch's operation order; the only differences are
fused multiply-adds and 1/x pre-computation of the two bias-correction factors -> tolerance 2e-6 relative."""
import pytest
import torch

import __graft_entry__ as entry

pytestmark = pytest.mark.gpu

SHAPES = [(1,), (7,), (4096,), (4097,), (1152, 1152), (3, 5, 7), (538, 144), (12289,)]


def _make(seed, misalign):
    g = torch.Generator().manual_seed(seed)
    params = []
    for i, shp in enumerate(SHAPES):
        n = 1
        for s in shp:
            n *= s
        if misalign and i % 3 == 1:   # a view starting 4 bytes into its buffer: exercises the scalar path
            buf = torch.randn(n + 1, generator=g)
            t = buf[1:].view(shp)
        else:
            t = torch.randn(shp, generator=g)
        params.append(t)
    return params


@pytest.mark.parametrize("max_norm", [None, 1.0, 1e6])
@pytest.mark.parametrize("misalign", [False, True])
def test_fused_adamw_matches_torch(max_norm, misalign):
    pkg = entry.load_package()
    ref_p = [torch.nn.Parameter(t.clone()) for t in _make(0, False)]
    src = _make(0, misalign)
    if misalign:  # same values, misaligned storage on the GPU side
        dev_p = []
        for t, r in zip(src, ref_p):
            if t.storage_offset() != 0:
                buf = torch.empty(t.numel() + 1, device="cuda")
                v = buf[1:].view(t.shape)
                v.copy_(r.data)
                dev_p.append(torch.nn.Parameter(v))
            else:
                dev_p.append(torch.nn.Parameter(r.data.clone().cuda()))
    else:
        dev_p = [torch.nn.Parameter(r.data.clone().cuda()) for r in ref_p]
    groups = lambda ps: [{"params": ps[:4], "lr": 3e-3, "weight_decay": 0.05}, {"params": ps[4:], "lr": 1e-3}]
    ref = torch.optim.AdamW(groups(ref_p), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    opt = pkg.FusedAdamW(groups(dev_p), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_grad_norm=max_norm)
    for step in range(4):
        grads = _make(100 + step, False)
        for i, (r, d, g) in enumerate(zip(ref_p, dev_p, grads)):
            if step == 2 and i == 3:      # a parameter without a gradient this step: skipped by both
                r.grad = None
                d.grad = None
                continue
            r.grad = g.clone() * (10.0 if step == 1 else 0.01)
            d.grad = r.grad.clone().cuda()
        if max_norm is not None:
            # torch/nn/utils/clip_grad.py formula.  The norm is taken in float64: torch's fp32 CPU reduction is off by
            # 1.5e-5 relative on these 1.4 M elements (checked below), more than this test's tolerance.
            live = [r for r in ref_p if r.grad is not None]
            total = torch.sqrt(sum((r.grad.double() ** 2).sum() for r in live)).float()
            torch_total = torch.nn.utils.clip_grad_norm_(live, 1e30)   # torch's own value, no clipping applied
            assert abs(torch_total.item() - total.item()) <= 5e-5 * total.item()
            coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
            for r in live:
                r.grad.mul_(coef)
        ref.step()
        opt.step()
        if max_norm is not None:
            assert abs(opt.last_grad_norm.item() - total.item()) <= 2e-6 * total.item()
        for r, d in zip(ref_p, dev_p):
            torch.testing.assert_close(d.detach().cpu(), r.detach(), rtol=2e-6, atol=1e-7)
            torch.testing.assert_close(opt.state[d]["exp_avg"].cpu(), ref.state[r]["exp_avg"], rtol=2e-6, atol=1e-9)
            torch.testing.assert_close(opt.state[d]["exp_avg_sq"].cpu(), ref.state[r]["exp_avg_sq"], rtol=2e-6,
                                       atol=1e-12)
    # the gradients themselves are not modified by the fused clip
    torch.testing.assert_close(dev_p[0].grad.cpu(), _make(103, False)[0] * 0.01)
    assert ref.state[ref_p[3]]["step"].item() == 3 and opt.state[dev_p[3]]["step"].item() == 3  # skipped once


def test_state_dict_round_trips_with_torch_adamw():
    pkg = entry.load_package()
    p = torch.nn.Parameter(torch.randn(300, 40, device="cuda"))
    opt = pkg.FusedAdamW([p], lr=1e-2, weight_decay=0.1)
    p.grad = torch.randn_like(p)
    opt.step()
    sd = opt.state_dict()
    q = torch.nn.Parameter(p.detach().clone())
    t = torch.optim.AdamW([q], lr=1e-2, weight_decay=0.1)
    t.load_state_dict(sd)                       # torch accepts our state
    q.grad = p.grad.clone()
    t.step()
    opt.step()
    torch.testing.assert_close(p.detach(), q.detach(), rtol=2e-6, atol=1e-7)
    opt2 = pkg.FusedAdamW([p], lr=1e-2, weight_decay=0.1)
    opt2.load_state_dict(t.state_dict())        # and we accept torch's
    opt2.step()
    t.step()
    torch.testing.assert_close(p.detach(), q.detach(), rtol=2e-6, atol=1e-7)


def test_global_grad_norm_and_errors():
    pkg = entry.load_package()
    ps = [torch.nn.Parameter(torch.zeros(n, device="cuda")) for n in (5, 70000)]
    for p in ps:
        p.grad = torch.randn_like(p)
    want = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in ps)).item()
    got = pkg.global_grad_norm(ps).item()
    assert abs(got - want) <= 2e-6 * want
    cpu = torch.nn.Parameter(torch.zeros(4))
    cpu.grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        pkg.FusedAdamW([cpu]).step()


def test_ema_matches_reference_formula_and_swaps_weights():
    """ExponentialMovingAverage.update/apply_shadow/restore against the reference's per-tensor loop
    (cifake_binary_classifier.py:211-236), restated on the CPU."""
    pkg = entry.load_package()
    torch.manual_seed(3)
    model = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.LayerNorm(53), torch.nn.Linear(53, 4101)).cuda()
    model[1].bias.requires_grad = False                      # frozen tensors are not tracked (reference: requires_grad)
    ema = pkg.ExponentialMovingAverage(model, decay=0.99)
    assert "1.bias" not in ema.shadow and len(ema.shadow) == 5
    ref = {n: p.detach().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
    for step in range(3):
        with torch.no_grad():
            for p in model.parameters():
                p.add_(torch.randn_like(p) * 0.1)
        ema.update()
        for n, p in model.named_parameters():
            if p.requires_grad:
                ref[n] = ref[n] * 0.99 + p.detach().cpu() * (1 - 0.99)
    for n in ref:
        torch.testing.assert_close(ema.shadow[n].cpu(), ref[n], rtol=2e-6, atol=1e-7)
    live = {n: p.detach().clone() for n, p in model.named_parameters()}
    ema.apply_shadow()
    assert torch.equal(model[0].weight, ema.shadow["0.weight"]) and torch.equal(model[1].bias, live["1.bias"])
    ema.restore()
    for n, p in model.named_parameters():
        assert torch.equal(p, live[n])


def test_step_writes_encoder_shadows_ema_and_folds_grad_scale():
    """Kernel work-list k11: FusedAdamW with an attached encoder writes the bf16 weight shadows (and their transposes) in
    the AdamW pass, so the next forward re-casts nothing; attach_ema folds the EMA in; grad_scale folds 1/world in; a
    changed learning rate needs no table upload.  Checked against torch.optim.AdamW + explicit formulas."""
    pkg = entry.load_package()
    for cfg_name, mode in (("hostile", "bf16"), ("tiny", "fp32")):
        cfg = pkg.get_config(cfg_name)
        sd = pkg.weights.seeded_state_dict(cfg, seed=3)
        res = cfg.image_size
        x = pkg.weights.seeded_pixels(2, res, res, seed=4).cuda()

        def make():
            m = pkg.SiglipVisionModelHIP(cfg, compute_dtype=mode)
            m.load_state_dict(sd)
            return m.cuda()
        fused, plain = make(), make()
        opt_f = pkg.FusedAdamW(fused.parameters(), lr=1e-3, weight_decay=0.05, max_grad_norm=0.7, grad_scale=0.5)
        opt_f.attach_encoder(fused)
        ema = pkg.ExponentialMovingAverage(fused, decay=0.9)
        opt_f.attach_ema(ema)
        opt_p = pkg.FusedAdamW(plain.parameters(), lr=1e-3, weight_decay=0.05, max_grad_norm=0.7)
        ema_ref = {n: p.detach().clone() for n, p in fused.named_parameters()}
        for step in range(3):
            for m, opt in ((fused, opt_f), (plain, opt_p)):
                out = m(pixel_values=x)
                loss = out.pooler_output.square().mean() + out.last_hidden_state.mean()
                opt.zero_grad(set_to_none=True)
                loss.backward()
            with torch.no_grad():            # fused holds rank SUMS of a 2-rank job: twice the mean gradient
                for p in fused.parameters():
                    p.grad.mul_(2.0)
            if step == 2:                    # an LR scheduler step: hyper-parameters are launch arguments
                for opt in (opt_f, opt_p):
                    opt.param_groups[0]["lr"] = 5e-4
                key_before = opt_f._table_key
            serial = fused._shadow_serial
            opt_f.step()
            opt_p.step()
            if step == 2:
                assert opt_f._table_key is key_before
            assert all(fused._units_in_sync()), "every block's shadows were written by the optimizer"
            assert fused._shadow_serial == serial + 1
            assert not any(plain._units_in_sync()[:-1]), "without attach_encoder the next forward must re-cast"
            for (n, a), (_, b) in zip(fused.named_parameters(), plain.named_parameters()):
                assert torch.equal(a.detach(), b.detach()), n     # tiled (shadow-writing) and linear paths: same bits
                ema_ref[n] = ema_ref[n] * 0.9 + a.detach() * 0.1
                torch.testing.assert_close(ema.shadow[n], ema_ref[n], rtol=3e-6, atol=1e-7)
            # the shadows the optimizer wrote give bit for bit the forward of a from-scratch re-cast of the same parameters
            with torch.no_grad():
                o_f = fused(pixel_values=x).pooler_output.clone()
                assert fused._shadow_serial == serial + 1, "the forward after the step must not have re-cast anything"
                fused._shadow_key = None                      # force sgl_prepare_weights on the next forward
                o_r = fused(pixel_values=x).pooler_output
                assert fused._shadow_serial == serial + 2
            assert torch.equal(o_f, o_r), (o_f - o_r).abs().max()
        # a parameter changed behind the optimizer's back is not adopted: the encoder re-casts that block itself
        with torch.no_grad():
            fused.encoder.layers[0].mlp.fc1.weight.mul_(1.01)
        assert fused._units_in_sync()[0] is False

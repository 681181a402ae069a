"""CPU (no GPU): the C-ABI library builds, loads and exports every symbol include/siglip_hip.h declares; the ctypes
binding matches the header's parameter counts; host-only entry points (create / query_sizes / status strings)
behave; the Python host surface mirrors the reference's interface (state-dict keys, freezing, error behaviour).
No kernel is launched here."""
import ctypes as C
import os
import re

import pytest
import torch


def _prototypes(header_path):
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(sgl_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        protos[m.group(1)] = n
    return protos


def test_library_exports_every_declared_symbol(pkg, hiplib):
    protos = _prototypes(pkg.lib.HEADER_PATH)
    assert len(protos) >= 20
    for name, nargs in protos.items():
        fn = getattr(hiplib, name)  # raises AttributeError when the symbol is missing
        assert fn.argtypes is not None, f"{name} has no ctypes signature"
        assert len(fn.argtypes) == nargs, f"{name}: header has {nargs} parameters, binding {len(fn.argtypes)}"
    assert set(pkg.lib.declared_symbols()) == set(protos)
    assert hiplib.sgl_abi_version() == 3
    assert hiplib.sgl_status_string(0) == b"ok" and hiplib.sgl_status_string(-3) == b"buffer too small"


def test_create_and_query_sizes_host_logic(pkg, hiplib):
    L = pkg.lib
    good = L.SglConfig(1152, 4304, 27, 16, 14, 27, 1e-6, L.SGL_DTYPE_BF16, 1)
    ctx = hiplib.sgl_create(C.byref(good))
    assert ctx
    a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
    assert hiplib.sgl_query_sizes(ctx, 64, 384, 384, 1, C.byref(a), C.byref(b), C.byref(c)) == 0
    shadow64, saved64, ws64 = a.value, b.value, c.value
    assert hiplib.sgl_query_sizes(ctx, 32, 384, 384, 1, C.byref(a), C.byref(b), C.byref(c)) == 0
    assert a.value == shadow64 and b.value < saved64 and c.value < ws64
    # bf16 shadows: 2 copies (plain + transposed) of every matrix, a bit above 2 * 2 bytes * matrix params
    mats = 27 * (4 * 1152 * 1152 + 2 * 1152 * 4304)
    assert 4 * mats < shadow64 < 4.4 * mats + (1 << 24)
    # saved activations fit the 288 GB HBM3E budget with room to spare at the benchmark batch
    assert saved64 < 60 * (1 << 30)
    assert hiplib.sgl_query_sizes(ctx, 64, 384, 384, 0, C.byref(a), C.byref(b), C.byref(c)) == 0
    assert b.value == 0 and c.value > 0
    assert hiplib.sgl_query_sizes(ctx, 0, 384, 384, 1, None, None, None) == -1      # bad batch
    assert hiplib.sgl_query_sizes(ctx, 2, 10, 384, 1, None, None, None) == -1       # smaller than a patch
    assert hiplib.sgl_query_sizes(None, 2, 384, 384, 1, None, None, None) == -5
    hiplib.sgl_destroy(ctx)
    for bad in [L.SglConfig(1150, 4304, 27, 16, 14, 27, 1e-6, 1, 1),    # D not divisible by heads
                L.SglConfig(1152, 4304, 27, 8, 14, 27, 1e-6, 1, 1),     # head_dim 144 > 96
                L.SglConfig(1152, 4304, 27, 16, 14, 27, 1e-6, 7, 1),    # unknown dtype
                L.SglConfig(1152, 0, 27, 16, 14, 27, 1e-6, 1, 1)]:
        assert not hiplib.sgl_create(C.byref(bad))


def test_state_dict_matches_hf_names_and_freezing_surface(pkg):
    cfg = pkg.get_config("hostile")
    m = pkg.SiglipVisionModelHIP(cfg, compute_dtype="fp32")
    ref_shapes = pkg.weights.param_shapes(cfg)
    sd = m.state_dict()
    assert list(sd.keys()) == ["vision_model." + k for k in ref_shapes]   # transformers' SiglipVisionModel keys, in order
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(ref_shapes[k[len("vision_model."):]]), k
    assert sum(p.numel() for p in m.parameters()) == cfg.num_params()
    seeded = pkg.weights.seeded_state_dict(cfg, 1)
    m.load_state_dict(seeded)
    m.load_state_dict({"vision_model." + k: v for k, v in seeded.items()})   # transformers-4.x prefixed keys
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in list(seeded.items())[:-1]})        # strict by default
    # the reference's freezing code (Siglip2sidafrozen.py:757-768) runs unchanged
    for p in m.vision_model.embeddings.parameters():
        p.requires_grad = False
    for i, layer in enumerate(m.vision_model.encoder.layers):
        for p in layer.parameters():
            p.requires_grad = i >= 1
    assert m.config.hidden_size == 144
    assert hasattr(m, "gradient_checkpointing_enable")
    m.gradient_checkpointing_enable()
    assert not m.embeddings.patch_embedding.weight.requires_grad
    assert m.encoder.layers[1].mlp.fc1.weight.requires_grad
    # no CPU fallback: calling the model on CPU tensors is an error, not a silent slow path
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(pixel_values=torch.zeros(1, 3, 42, 42))


def test_named_configs_and_flop_model(pkg):
    so = pkg.get_config("so400m-patch14-384")
    assert (so.hidden_size, so.intermediate_size, so.num_hidden_layers, so.head_dim, so.native_grid) == \
        (1152, 4304, 27, 72, 27)
    # BASELINE.md §2 figures
    assert abs(so.fwd_flops_per_image() / 1e9 - 670.35) < 0.01
    assert abs(so.train_flops_per_image() / 1e12 - 2.0110) < 1e-3
    base = pkg.get_config("base-patch16-224")
    assert abs(base.fwd_flops_per_image() / 1e9 - 35.42) < 0.01
    assert abs(so.num_params() / 1e6 - 428.2) < 0.1 and abs(base.num_params() / 1e6 - 92.9) < 0.1
    with pytest.raises(KeyError):
        pkg.get_config("no-such-model")
    with pytest.raises(ValueError):
        pkg.SiglipVisionConfig(hidden_size=100, num_attention_heads=3)
    for name in ("ViT-B-16-SigLIP", "ViT-L-16-SigLIP-384", "ViT-SO400M-16-SigLIP2-512",
                 "google/siglip2-large-patch16-384"):
        assert pkg.get_config(name).hidden_size in (768, 1024, 1152)


def test_from_pretrained_local_dir_and_open_clip_factory_errors(pkg, tmp_path):
    import json
    from safetensors.torch import save_file
    cfg = pkg.get_config("tiny")
    sd = pkg.weights.seeded_state_dict(cfg, 2)
    d = tmp_path / "ckpt"
    d.mkdir()
    (d / "config.json").write_text(json.dumps({"vision_config": cfg.to_dict()}))
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "model.safetensors"))
    m = pkg.SiglipVisionModelHIP.from_pretrained(str(d), compute_dtype="fp32")
    assert torch.equal(m.state_dict()["vision_model.head.probe"], sd["head.probe"])
    # a published google/siglip* file is the FULL SiglipModel: only the vision tower is kept (Siglip2sidafrozen.py:753)
    full = {("vision_model." + k): v.contiguous() for k, v in sd.items()}
    full.update({"text_model.embeddings.token_embedding.weight": torch.zeros(8, 4), "text_model.head.bias": torch.zeros(4),
                 "logit_scale": torch.zeros(1), "logit_bias": torch.zeros(1)})
    d2 = tmp_path / "full"
    d2.mkdir()
    (d2 / "config.json").write_text(json.dumps({"vision_config": cfg.to_dict(), "text_config": {"hidden_size": 4}}))
    save_file(full, str(d2 / "model.safetensors"))
    mf = pkg.SiglipVisionModelHIP.from_pretrained(str(d2), compute_dtype="fp32")
    assert torch.equal(mf.state_dict()["vision_model.encoder.layers.1.mlp.fc1.weight"], sd["encoder.layers.1.mlp.fc1.weight"])
    # a bare architecture name has no weights to load: an error, unless seeded random init is asked for (and then it warns)
    with pytest.raises(OSError, match="allow_random_init"):
        pkg.SiglipVisionModelHIP.from_pretrained("tiny")
    with pytest.warns(UserWarning, match="RANDOM"):
        m2 = pkg.SiglipVisionModelHIP.from_pretrained("tiny", allow_random_init=True)
    assert m2.config.num_hidden_layers == 3
    with pytest.raises(OSError):
        pkg.SiglipVisionModelHIP.from_pretrained("google/not-a-model")
    with pytest.warns(UserWarning, match="RANDOM"):
        pkg.create_model_and_transforms("tiny", pretrained="webli", device="cpu")
    with pytest.warns(UserWarning, match="no-op"):
        m2.gradient_checkpointing_enable()


def test_custom_ops_are_registered_with_fake_impls(pkg):
    """torch.ops.siglip_hip.encoder_fwd / encoder_bwd exist and their fake (meta) implementations give the right shapes
    without touching a GPU (what torch.compile traces through; SURVEY.md 8b 'Who calls it')."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    assert hasattr(torch.ops.siglip_hip, "encoder_fwd") and hasattr(torch.ops.siglip_hip, "encoder_bwd")
    cfg = pkg.get_config("tiny")
    m = pkg.SiglipVisionModelHIP(cfg, "bf16")
    params = m._flat_params()
    with FakeTensorMode(allow_non_fake_inputs=True):
        x = torch.empty(2, 3, 32, 32)
        outs = torch.ops.siglip_hip.encoder_fwd(x, params, m._handle, True, False, True, [1, 3], 0, 0, 0, 0)
    assert [tuple(o.shape) for o in outs[:4]] == [(2, 64), (2, 4, 64), (2, 4, 64), (2, 4, 64)]
    assert outs[4].dtype == torch.uint8 and outs[4].numel() == m._sizes(2, 32, 32, True)[1]
    assert tuple(outs[5].shape) == (2, 8, 64)          # 4 slots, 2 of them handed out as taps


def test_adamw_plan_host_helper(hiplib):
    """sgl_adamw_plan is host-only: (tensor, chunk) pairs of 4096 elements, in table order."""
    lib = hiplib
    numel = (C.c_uint64 * 4)(1, 4096, 4097, 0)
    nb = lib.sgl_adamw_plan(numel, 4, None, 0)
    assert nb == 4
    bm = (C.c_int32 * (2 * nb))()
    assert lib.sgl_adamw_plan(numel, 4, bm, nb) == nb
    assert list(bm) == [0, 0, 1, 0, 2, 0, 2, 1]
    assert lib.sgl_adamw_plan(None, 1, None, 0) < 0


def test_open_clip_surface_parameter_names_support_the_reference_unfreezing(pkg):
    """simple_classifier.py:484-493: freeze the backbone, then unfreeze by name substring ('blocks.<last>', 'norm', ...)."""
    cfg = pkg.get_config("tiny")                      # 3 blocks
    backbone = pkg.OpenClipStyleEncoder(cfg, "fp32")
    names = [n for n, _ in backbone.named_parameters()]
    assert "visual.trunk.blocks.2.mlp.fc1.weight" in names and "visual.trunk.norm.weight" in names
    assert "visual.trunk.patch_embed.proj.weight" in names and "visual.trunk.pos_embed" in names
    assert "visual.trunk.attn_pool.latent" in names and "visual.trunk.blocks.0.attn.q_proj.bias" in names
    assert len(names) == len(list(backbone.parameters())) == len(set(names))
    for p in backbone.parameters():
        p.requires_grad = False
    for n, p in backbone.named_parameters():
        if any(x in n for x in ["blocks.2", "ln_final", "norm"]):
            p.requires_grad = True
    on = {n for n, p in backbone.named_parameters() if p.requires_grad}
    assert all(("blocks.2." in n) or ("norm" in n) for n in on)
    assert "visual.trunk.blocks.2.attn.proj.weight" in on and "visual.trunk.blocks.0.norm1.weight" in on
    assert "visual.trunk.blocks.1.mlp.fc1.weight" not in on

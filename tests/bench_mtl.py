"""Developer tool: the SID multi-task training step (frozen-prefix encoder + SegFormer decoder + CE/BCE/Dice loss,
Siglip2sidafrozen.py:750-803,1375-1398) against the encoder alone, to see what the PyTorch heads cost around the HIP path.
   python tests/bench_mtl.py [B] [freeze_below]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g

pkg = g.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 21
ULTRA = len(sys.argv) > 3 and sys.argv[3] == "ultra"   # the script's default decoder (Siglip2sidafrozen.py:1139-1140)
SEG_LAYERS = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, -1) if ULTRA else (2, 6, 10, -1)
EMBED = 512 if ULTRA else 256
cfg = pkg.get_config("so400m-patch14-384")
enc = pkg.SiglipVisionModelHIP(cfg, "bf16")
enc.load_state_dict(pkg.weights.seeded_state_dict(cfg, 0))
model = pkg.heads.SigLIP2MTL(enc, seg_layers=SEG_LAYERS, embed_dim=EMBED, freeze_below=K).cuda()
x = pkg.weights.seeded_pixels(B, 384, 384, seed=1).cuda()
y = torch.randint(0, 3, (B,), device="cuda")
masks = (torch.rand(B, 1, 384, 384, device="cuda") < 0.1).float()
has = torch.ones(B, dtype=torch.bool, device="cuda")
opt = pkg.FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.01, max_grad_norm=1.0)

def step_full():
    with torch.autocast("cuda", dtype=torch.bfloat16):     # as the reference's train step (Siglip2sidafrozen.py:1375)
        cls, seg = model(x)
        loss = pkg.heads.mtl_loss(cls.float(), seg.float(), y, masks, has)
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()

def step_enc():
    out = enc(pixel_values=x, interpolate_pos_encoding=True)
    loss = out.pooler_output.square().mean()
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()

for name, fn in (("encoder only (+AdamW)", step_enc), ("SID multi-task step (+AdamW)", step_full)):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {dt*1e3:8.1f} ms/step  {B/dt:8.1f} img/s  (B={B}, blocks<{K} frozen, decoder taps={len(SEG_LAYERS)} E={EMBED})")

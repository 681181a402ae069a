"""Developer tool: one training step (fwd+bwd) of so400m-patch14-384 in each compute mode.  python tests/bench_modes.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = pkg.get_config("so400m-patch14-384")
x = pkg.weights.seeded_pixels(B, 384, 384, seed=3).cuda()
for mode in (sys.argv[2:] or ["bf16", "bf16x3", "fp32"]):
    m = pkg.SiglipVisionModelHIP(cfg, compute_dtype=mode)
    m.load_state_dict(pkg.weights.seeded_state_dict(cfg, seed=0))
    m = m.cuda()
    def step():
        out = m(pixel_values=x, interpolate_pos_encoding=True)
        out.pooler_output.square().mean().backward()
        for p in m.parameters(): p.grad = None
        return out
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 3
    for _ in range(n): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{mode:7s} B={B}: {dt*1e3:9.1f} ms/step  {B/dt:8.2f} img/s", flush=True)
    del m; torch.cuda.empty_cache()

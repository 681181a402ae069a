"""Developer tool: inference (no_grad forward) latency / throughput of the so400m encoder at small batches — the shape of
the reference app's 9-crop call pattern (appv3.py:3221-3247).   python tests/bench_infer.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g

pkg = g.load_package()
cfg = pkg.get_config("so400m-patch14-384")
model = pkg.SiglipVisionModelHIP(cfg, "bf16")
model.load_state_dict(pkg.weights.seeded_state_dict(cfg, 0))
model = model.cuda().eval()
for B in (1, 2, 4, 9, 18, 64):
    x = pkg.weights.seeded_pixels(B, 384, 384, seed=1).cuda()
    with torch.no_grad():
        for _ in range(3):
            model(pixel_values=x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            model(pixel_values=x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"B={B:3d}: {dt*1e3:7.2f} ms/forward  {B/dt:8.1f} img/s  ({B*0.67035/dt:6.1f} TF/s)")

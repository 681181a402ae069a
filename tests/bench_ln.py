"""Developer micro-benchmark of the LayerNorm kernels through the C ABI (not part of the test-suite).
   python tests/bench_ln.py [B]   -> microseconds and effective TB/s at M = B*729 rows, D = 1152."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g

pkg = g.load_package(); lib = pkg.lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M, D = B * 729, 1152
st = torch.cuda.current_stream()
x = torch.randn(M, D, device="cuda"); gam = torch.randn(D, device="cuda"); bet = torch.randn(D, device="cuda")
y = torch.empty(M, D, device="cuda", dtype=torch.bfloat16); mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
dy = torch.randn(M, D, device="cuda").bfloat16(); dres = torch.randn(M, D, device="cuda"); dx = torch.empty(M, D, device="cuda")
dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
scratch = torch.empty(16 << 20, device="cuda", dtype=torch.uint8)

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps

def fwd():
    assert lib.sgl_op_layernorm_fwd(x.data_ptr(), gam.data_ptr(), bet.data_ptr(), y.data_ptr(), 1, mean.data_ptr(), rstd.data_ptr(), M, D, 1e-6, st.cuda_stream) == 0
def bwd():
    assert lib.sgl_op_layernorm_bwd(dy.data_ptr(), 1, x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gam.data_ptr(), dres.data_ptr(), dx.data_ptr(), None, 0,
                                    dg.data_ptr(), db.data_ptr(), scratch.data_ptr(), scratch.numel(), M, D, st.cuda_stream) == 0
t = timeit(fwd); print(f"ln_fwd  M={M} D={D}: {t*1e6:7.1f} us  {M*D*6/t/1e12:5.2f} TB/s")
t = timeit(bwd); print(f"ln_bwd  M={M} D={D}: {t*1e6:7.1f} us  {M*D*14/t/1e12:5.2f} TB/s (x f32 + dy bf16 + dres f32 in, dx f32 out)")

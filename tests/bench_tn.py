"""Developer tool: time the four dW (TN) GEMM shapes of one so400m block.  python tests/bench_tn.py [B] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package(); lib = pkg.lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
D, I, N = 1152, 4304, 729
Ip = 4352
M = B * N
st = torch.cuda.current_stream()
scratch = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
tot_t = tot_f = 0.0
for name, N1, N2, l1, l2 in (("fc2 dW", D, I, D, Ip), ("fc1 dW", I, D, Ip, D), ("out dW", D, D, D, D), ("qkv dW", 3 * D, D, 3 * D, D)):
    A = torch.randn(M, l1, device="cuda").bfloat16(); Bm = torch.randn(M, l2, device="cuda").bfloat16()
    out = torch.empty(N1, N2, device="cuda")
    f = lambda: lib.sgl_op_gemm_tn_ws(1, A.data_ptr(), l1, Bm.data_ptr(), l2, M, N1, N2, 0, out.data_ptr(), N2, 0, scratch.data_ptr(), scratch.numel(), st.cuda_stream)
    assert f() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): f()
    e1.record(st); e1.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    fl = 2.0 * M * N1 * N2
    tot_t += t; tot_f += fl
    print(f"{name}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")
    del A, Bm, out
print(f"total: {tot_t*1e6:8.1f} us  {tot_f/tot_t/1e12:7.1f} TF/s")

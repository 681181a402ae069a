"""Developer micro-benchmark of the GEMM / attention kernels through the C ABI (not part of the test-suite).
   python tests/bench_kernels.py [B]      -> TFLOP/s per shape at batch B (tokens = B*729), random bf16 data."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g

pkg = g.load_package(); lib = pkg.lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = B * 729
D, I, Ip = 1152, 4304, 4352
st = torch.cuda.current_stream()

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps

def nt(Mm, N, K, epi=0):
    A = torch.randn(Mm, K, device="cuda").bfloat16(); Bw = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    out = torch.empty(Mm, N, device="cuda", dtype=torch.bfloat16)
    def f():
        assert lib.sgl_op_gemm_nt(1, A.data_ptr(), K, Bw.data_ptr(), K, Mm, N, K, epi, out.data_ptr(), N, None, 0, None, None, 0, None, 0, None, 1, 1, 1, 8, 8, 1, st.cuda_stream) == 0
    t = timeit(f); return t, 2.0 * Mm * N * K / t / 1e12

def tn(Mr, N1, N2):
    A = torch.randn(Mr, N1, device="cuda").bfloat16(); Bm = torch.randn(Mr, N2, device="cuda").bfloat16()
    out = torch.empty(N1, N2, device="cuda")
    def f():
        assert lib.sgl_op_gemm_tn(1, A.data_ptr(), N1, Bm.data_ptr(), N2, Mr, N1, N2, 0, out.data_ptr(), N2, 0, st.cuda_stream) == 0
    # splits=0 -> library default is chosen by the encoder host code; here emulate it
    t = timeit(f); return t, 2.0 * Mr * N1 * N2 / t / 1e12

def nt_epi(Mm, N, K, epi):
    A = torch.randn(Mm, K, device="cuda").bfloat16(); Bw = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device="cuda")
    f32 = epi == 2
    out = torch.empty(Mm, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    out2 = torch.empty(Mm, N, device="cuda", dtype=torch.bfloat16) if epi == 1 else None
    res = torch.randn(Mm, N, device="cuda") if epi == 2 else None
    aux = torch.randn(Mm, N, device="cuda").bfloat16() if epi == 4 else None
    def f():
        assert lib.sgl_op_gemm_nt(1, A.data_ptr(), K, Bw.data_ptr(), K, Mm, N, K, epi, out.data_ptr(), N,
                                  None if out2 is None else out2.data_ptr(), N, bias.data_ptr(),
                                  None if res is None else res.data_ptr(), N, None if aux is None else aux.data_ptr(), N,
                                  None, 1, 1, 1, 8, 8, 1, st.cuda_stream) == 0
    t = timeit(f); return t, 2.0 * Mm * N * K / t / 1e12

print(f"gen={os.environ.get('SGL_GEMM_GEN','2')} B={B} M={M}")
for name, (Mm, N, K, epi) in {"fc1+gelu": (M, Ip, D, 1), "gelu_bwd": (M, Ip, D, 4), "out+res": (M, D, D, 2), "fc2+res": (M, D, Ip, 2)}.items():
    t, tf = nt_epi(Mm, N, K, epi); print(f"NT {name:8s} M={Mm:6d} N={N:5d} K={K:5d}  {t*1e3:8.3f} ms  {tf:7.1f} TF/s")
for name, (Mm, N, K) in {"qkv": (M, 3*D, D), "out": (M, D, D), "fc1": (M, Ip, D), "fc2": (M, D, Ip), "dX_qkv": (M, D, 3*D),
                         "sq4096": (4096, 4096, 4096), "sq8192": (8192, 8192, 8192)}.items():
    t, tf = nt(Mm, N, K); print(f"NT {name:8s} M={Mm:6d} N={N:5d} K={K:5d}  {t*1e3:8.3f} ms  {tf:7.1f} TF/s")
for name, (Mr, N1, N2) in {"dWqkv": (M, 3*D, D), "dWo": (M, D, D), "dW1": (M, Ip, D), "dW2": (M, D, Ip), "sq4096": (4096, 4096, 4096)}.items():
    t, tf = tn(Mr, N1, N2); print(f"TN {name:8s} Mred={Mr:6d} N1={N1:5d} N2={N2:5d}  {t*1e3:8.3f} ms  {tf:7.1f} TF/s")

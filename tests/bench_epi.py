"""Developer tool: per-tile epilogue cost of the NT GEMM without HBM contention (a grid smaller than the chip).
   python tests/bench_epi.py"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package(); lib = pkg.lib.load()
st = torch.cuda.current_stream()
def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps
def run(Mm, N, K, epi):
    A = torch.randn(Mm, K, device="cuda").bfloat16(); Bw = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device="cuda")
    f32 = epi in (2, 6)
    out = torch.empty(Mm, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    out2 = torch.empty(Mm, N, device="cuda", dtype=torch.bfloat16) if epi == 1 else None
    res = torch.randn(Mm, N, device="cuda") if epi == 2 else None
    aux = torch.randn(Mm, N, device="cuda").bfloat16() if epi == 4 else None
    def f():
        assert lib.sgl_op_gemm_nt(1, A.data_ptr(), K, Bw.data_ptr(), K, Mm, N, K, epi, out.data_ptr(), N,
                                  None if out2 is None else out2.data_ptr(), N, bias.data_ptr(),
                                  None if res is None else res.data_ptr(), N, None if aux is None else aux.data_ptr(), N,
                                  None, 1, 1, 1, 8, 8, 1, st.cuda_stream) == 0
    return timeit(f)
names = {0: "STORE", 1: "BIAS_GELU", 2: "RES_F32", 4: "GELU_BWD", 6: "F32"}
for (Mm, N, K) in [(2048, 4352, 1152), (2048, 4352, 128), (2048 * 23, 4352, 1152), (2048 * 23, 4352, 128)]:
    tiles = ((Mm + 255) // 256) * ((N + 255) // 256)
    print(f"M={Mm} N={N} K={K} tiles={tiles}: " + "  ".join(f"{names[e]} {run(Mm, N, K, e)*1e6:7.1f}us" for e in (0, 1, 2, 4, 6)))

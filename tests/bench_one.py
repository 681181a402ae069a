"""Developer tool: launch one GEMM shape a few times (for rocprofv3 --pmc runs).  python tests/bench_one.py nt M N K"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package(); lib = pkg.lib.load()
kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
st = torch.cuda.current_stream()
if kind == "nt":
    A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    f = lambda: lib.sgl_op_gemm_nt(1, A.data_ptr(), K, B.data_ptr(), K, M, N, K, 0, out.data_ptr(), N, None, 0, None, None, 0, None, 0, None, 1, 1, 1, 8, 8, 1, st.cuda_stream)
else:
    A = torch.randn(M, N, device="cuda").bfloat16(); B = torch.randn(M, K, device="cuda").bfloat16()
    out = torch.empty(N, K, device="cuda")
    f = lambda: lib.sgl_op_gemm_tn(1, A.data_ptr(), N, B.data_ptr(), K, M, N, K, 0, out.data_ptr(), K, 0, st.cuda_stream)
for _ in range(reps):
    assert f() == 0
torch.cuda.synchronize()

"""Developer tool: time / profile the attention kernels at the encoder shape.  python tests/bench_attn.py [B] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package(); lib = pkg.lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H, N, dh, DP = (16, 729, 72, 80) if os.environ.get("ATT_SHAPE", "so400m") == "so400m" else (12, 196, 64, 64)
D = H * dh
st = torch.cuda.current_stream()
LAYOUT = os.environ.get("ATT_LAYOUT", "token")   # token: [B*N, 3D] as the encoder's QKV GEMM writes it; head: [3,B,H,N,DP]
if LAYOUT == "token":
    tok = torch.randn(B * N, 3 * D, device="cuda").bfloat16()
    qkv = [tok[:, j * D:] for j in range(3)]
    LD = 3 * D
else:
    qkv = torch.zeros(3, B, H, N, DP, device="cuda", dtype=torch.bfloat16)
    qkv[..., :dh] = torch.randn(3, B, H, N, dh, device="cuda").bfloat16()
    LD = 0
out = torch.empty(B * N, D, device="cuda", dtype=torch.bfloat16)
dout = torch.randn(B * N, D, device="cuda").bfloat16()
lse = torch.empty(B, H, N, device="cuda"); delta = torch.empty(2, B, H, N, device="cuda")
dqkv = torch.empty(B * N, 3 * D, device="cuda", dtype=torch.bfloat16)
def fwd(): assert lib.sgl_op_attn_fwd(1, qkv[0].data_ptr(), qkv[1].data_ptr(), qkv[2].data_ptr(), out.data_ptr(), lse.data_ptr(), B, H, N, dh, DP, LD, st.cuda_stream) == 0
def bwd(): assert lib.sgl_op_attn_bwd(1, qkv[0].data_ptr(), qkv[1].data_ptr(), qkv[2].data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), delta.data_ptr(), B, H, N, dh, DP, LD, st.cuda_stream) == 0
for name, fn, fl in (("fwd", fwd, 4.0), ("bwd", bwd, 10.0)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); e1.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    print(f"attn {name}: {t*1e6:8.1f} us   {fl*B*H*N*N*dh/t/1e12:7.1f} TF/s (algorithmic)")

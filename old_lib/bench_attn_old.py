"""A/B helper (scratch): time the attention kernels of a library built from an earlier commit (head-major ABI 2)."""
import ctypes as C, os, sys, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsiglip_hip_old.so"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H, N, dh, DP = 16, 729, 72, 80
D = H * dh
st = torch.cuda.current_stream()
qkv = torch.zeros(3, B, H, N, DP, device="cuda", dtype=torch.bfloat16)
qkv[..., :dh] = torch.randn(3, B, H, N, dh, device="cuda").bfloat16()
out = torch.empty(B * N, D, device="cuda", dtype=torch.bfloat16)
dout = torch.randn(B * N, D, device="cuda").bfloat16()
lse = torch.empty(B, H, N, device="cuda"); delta = torch.empty(2, B, H, N, device="cuda")
dqkv = torch.empty(B * N, 3 * D, device="cuda", dtype=torch.bfloat16)
P = lambda t: C.c_void_p(t.data_ptr())
S = C.c_void_p(st.cuda_stream)
def fwd(): assert lib.sgl_op_attn_fwd(1, P(qkv[0]), P(qkv[1]), P(qkv[2]), P(out), P(lse), B, H, N, dh, DP, S) == 0
def bwd(): assert lib.sgl_op_attn_bwd(1, P(qkv[0]), P(qkv[1]), P(qkv[2]), P(out), P(dout), P(lse), P(dqkv), P(delta), B, H, N, dh, DP, S) == 0
for name, fn, fl in (("fwd", fwd, 4.0), ("bwd", bwd, 10.0)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); e1.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    print(f"OLD attn {name}: {t*1e6:8.1f} us   {fl*B*H*N*N*dh/t/1e12:7.1f} TF/s (algorithmic)")

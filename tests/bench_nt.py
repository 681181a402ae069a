"""Developer tool: the encoder's eight NT GEMM launches per block (4 forward, 4 dX) at one batch size, each checked
against torch and timed.  A/B the generations in separate processes:
   python tests/bench_nt.py [B]            SGL_GEMM_GEN=6 python tests/bench_nt.py [B]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package(); lib = pkg.lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
check = len(sys.argv) <= 2 or sys.argv[2] != "nocheck"
N_tok, H, hd, hdp = 729, 16, 72, 80
M, D, I, Ip = B * N_tok, 1152, 4304, 4352
st = torch.cuda.current_stream()
def gelu(x): return torch.nn.functional.gelu(x, approximate="tanh")
def relerr(a, b): return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps
shapes = [("qkv", 3 * D, D, 3, 3 * D, D), ("out_proj+res", D, D, 2, D, D), ("fc1+gelu", Ip, D, 1, I, D),
          ("fc2+res", D, Ip, 2, D, I), ("d(fc2)*gelu'", Ip, D, 4, I, D), ("d(fc1) store", D, Ip, 0, D, I),
          ("d(out) store", D, D, 0, D, D), ("d(qkv) store", D, 3 * D, 0, D, 3 * D)]
tot_t = tot_f = 0.0
for name, N, K, epi, Na, Ka in shapes:
    torch.manual_seed(N + K + epi)
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda") if epi == 2 else None
    aux = (torch.randn(M, N, device="cuda") * 1.5).bfloat16() if epi == 4 else None
    if epi == 3: out = torch.full((3 * B * H * N_tok * hdp,), float("nan"), device="cuda", dtype=torch.bfloat16)
    elif epi == 2: out = torch.full((M, N), float("nan"), device="cuda")
    else: out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    out2 = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16) if epi == 1 else None
    csum = torch.zeros((M + 127) // 128, N, device="cuda") if epi == 4 else None
    P = lambda t: None if t is None else t.data_ptr()
    def f():
        rc = lib.sgl_op_gemm_nt(1, A.data_ptr(), K, W.data_ptr(), K, M, N, K, epi, out.data_ptr(), N, P(out2), N,
                                bias.data_ptr() if epi in (1, 2, 3) else None, P(res), N, P(aux), N, None, 1, N_tok, H,
                                hd, hdp, B, st.cuda_stream)
        assert rc == 0, rc
    t = timeit(f)
    err = ""
    if check:
        acc = A[:4096].float() @ W.float().t()
        sl = slice(0, 4096)
        tail = A[-300:].float() @ W.float().t()
        if epi == 0: e = max(relerr(out[sl], acc), relerr(out[-300:], tail))
        elif epi == 1: e = max(relerr(out[sl], acc + bias), relerr(out2[sl], gelu(acc + bias)), relerr(out2[-300:], gelu(tail + bias)))
        elif epi == 2: e = max(relerr(out[sl], res[sl] + acc + bias), relerr(out[-300:], res[-300:] + tail + bias))
        elif epi == 4:
            u = aux[sl].float().requires_grad_(True); gelu(u).backward(acc); e = relerr(out[sl], u.grad)
        else:
            q = out.view(3, B, H, N_tok, hdp)
            full = (acc + bias).view(-1, 3, H, hd)      # tokens of the first images
            nimg = 4096 // N_tok
            ref = full[:nimg * N_tok].view(nimg, N_tok, 3, H, hd).permute(2, 0, 3, 1, 4)
            e = max(relerr(q[:, :nimg, :, :, :hd], ref), float(q[..., hd:].float().abs().max()))
        assert not torch.isnan(out.float()).any(), f"{name}: NaN left in the output"
        err = f"  relerr {e:.1e}"
        assert e < 1e-2, (name, e)
    fl = 2.0 * M * Na * Ka
    tot_t += t; tot_f += fl
    print(f"{name:16s} M={M} N={N} K={K}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s{err}", flush=True)
print(f"sum: {tot_t*1e3:.3f} ms  {tot_f/tot_t/1e12:.1f} TF/s  (gen {os.environ.get('SGL_GEMM_GEN', '6')})")

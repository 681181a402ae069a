"""Developer tool: sample the shader clock (rocm-smi) while the NT GEMM runs back to back, to relate achieved TFLOP/s to the
clock the chip actually sustains under MFMA load.   python tests/clock_probe.py"""
import os, sys, subprocess, threading, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g

pkg = g.load_package(); lib = pkg.lib.load()
st = torch.cuda.current_stream()
M = N = K = 8192
A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
samples = []
stop = False
def probe():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            samples.append([l.strip() for l in r.splitlines() if ("sclk" in l or "Power" in l or "power" in l)])
        except Exception as e:
            samples.append([repr(e)])
        time.sleep(0.5)
th = threading.Thread(target=probe); th.start()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
while time.time() - t0 < 6.0:
    for _ in range(50):
        assert lib.sgl_op_gemm_nt(1, A.data_ptr(), K, B.data_ptr(), K, M, N, K, 0, out.data_ptr(), N, None, 0, None, None, 0, None, 0, None, 1, 1, 1, 8, 8, 1, st.cuda_stream) == 0
    n += 50
    torch.cuda.synchronize()
e1.record(st); e1.synchronize()
stop = True; th.join()
t = e0.elapsed_time(e1) * 1e-3
print(f"{n} launches of 8192^3 in {t:.2f} s -> {2.0*M*N*K*n/t/1e12:.1f} TF/s sustained")
for s in samples[:3] + samples[-3:]:
    print(s)

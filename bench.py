"""bench.py — headline benchmark of the hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one training forward + backward of the SigLIP-2 so400m-patch14-384 vision encoder (+ pooling head)
on one synthetic batch per GPU that is already resident in HBM: bf16 MFMA operands, fp32 accumulate / residual
stream / master weights, the per-step refresh of the bf16 weight shadows (what autocast re-does every step in the
reference, Siglip2sidafrozen.py:1375) and, for N > 1, the bucketed gradient all-reduce over RCCL overlapped with
backward.  No optimizer step (metric is train fwd+bwd).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

PEAK_BF16_DENSE = 2.5e15   # MI355X dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM = 8.0e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="so400m-patch14-384")
    ap.add_argument("--batch", type=int, default=128, help="images per GPU per step (SURVEY.md 8d sweep: 16..128)")
    ap.add_argument("--res", type=int, default=0, help="image side (default: the config's native size)")
    ap.add_argument("--mode", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--freeze-below", type=int, default=0,
                    help="secondary metric (SURVEY.md 8d): freeze embeddings and blocks < K as Siglip2sidafrozen.py:757-768")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--kernel-reps", type=int, default=20,
                    help="launches per kernel for the roofline / breakdown legs; 0 skips them (clean rocprofv3 traces)")
    ap.add_argument("--no-optimizer", action="store_true",
                    help="skip the optimizer-step and full-train-step legs (clean rocprofv3 traces of the timed step)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary metrics of SURVEY.md 8d (batch sweep, base-patch16-224 B=256, frozen prefix)")
    ap.add_argument("--wire", default="fp32", choices=["fp32", "bf16"],
                    help="N > 1: gradient exchange format (ddp.GradBucketReducer)")
    ap.add_argument("--max-buckets", type=int, default=8, help="N > 1: gradient chunks (= collectives) per step")
    ap.add_argument("--rccl-channels", type=int, default=0,
                    help="N > 1: cap RCCL's channel count (NCCL_MAX_NCHANNELS), i.e. the CUs its kernels take from the GEMMs")
    ap.add_argument("--train-steps", type=int, default=10, help="timed steps of the train_step_with_optimizer leg")
    return ap.parse_args()


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment (how the driver calls it): start the N
    ranks as CHILD processes through torch.distributed.run.  This parent has made no GPU call (importing torch does not
    initialise HIP) and never execs; it relays rank 0's single JSON line and returns non-zero if any rank failed."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    lines = []
    for ln in proc.stdout:                      # stderr is inherited: progress lines stream through as they come
        if ln.startswith("{"):
            lines.append(ln.rstrip("\n"))
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc == 0 and len(lines) != 1:
        print(f"[bench] expected exactly one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        rc = 1
    for ln in lines[-1:]:
        print(ln, flush=True)
    return rc


def gemm_kernel_roofline(pkg, cfg, batch, res, reps):
    """Time the dominant kernel (the bf16 MFMA NT GEMM at the encoder's own shapes) with HIP events on the
    stream it is launched on, and convert to achieved TFLOP/s.  One launch per shape per rep; the figure is
    algorithmic FLOPs (2*M*N*K of the un-padded problem) / mean launch duration."""
    import ctypes as C
    lib = pkg.lib.load()
    gh = res // cfg.patch_size
    M = batch * gh * gh
    D, I = cfg.hidden_size, cfg.intermediate_size
    Ip = (I + 127) // 128 * 128
    dev = "cuda"
    stream = torch.cuda.current_stream()
    shapes = [  # (name, N, K, epi, N_alg, K_alg)
        ("qkv", 3 * D, D, 3, 3 * D, D), ("out_proj", D, D, 2, D, D), ("fc1", Ip, D, 1, I, D), ("fc2", D, Ip, 2, D, I),
    ]
    per = {}
    tot_t, tot_f = 0.0, 0.0
    for name, N, K, epi, Na, Ka in shapes:
        A = torch.randn(M, K, device=dev).bfloat16()
        Bw = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        bias = torch.randn(N, device=dev)
        res_t = torch.randn(M, N, device=dev) if epi == 2 else None
        hd, H = cfg.head_dim, cfg.num_attention_heads
        hdp = (hd + 15) // 16 * 16
        if epi == 3:
            out = torch.empty(3 * batch * H * gh * gh * hdp, device=dev, dtype=torch.bfloat16)
        elif epi == 2:
            out = torch.empty(M, N, device=dev)
        else:
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        out2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == 1 else None

        def launch():
            st = lib.sgl_op_gemm_nt(1, A.data_ptr(), K, Bw.data_ptr(), K, M, N, K, epi, out.data_ptr(), N,
                                    None if out2 is None else out2.data_ptr(), N, bias.data_ptr(),
                                    None if res_t is None else res_t.data_ptr(), N, None, 0, None, 1, gh * gh, H, hd,
                                    hdp, batch, stream.cuda_stream)
            assert st == 0, st
        launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            launch()
        e1.record(stream)
        e1.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / reps
        fl = 2.0 * M * Na * Ka
        per[name] = {"ms": round(t * 1e3, 4), "tflops": round(fl / t / 1e12, 1)}
        tot_t += t
        tot_f += fl
        del A, Bw, out, out2, res_t
    achieved = tot_f / tot_t / 1e12
    sclk = None
    try:  # shader clock the chip holds while this kernel runs back to back (power-capped well below the 2.4 GHz boost)
        A = torch.randn(M, D, device=dev).bfloat16()
        Bw = (torch.randn(Ip, D, device=dev) / D ** 0.5).bfloat16()
        out = torch.empty(M, Ip, device=dev, dtype=torch.bfloat16)

        def launch_fc1():
            lib.sgl_op_gemm_nt(1, A.data_ptr(), D, Bw.data_ptr(), D, M, Ip, D, 0, out.data_ptr(), Ip, None, 0, None, None,
                               0, None, 0, None, 1, 1, 1, 8, 8, 1, stream.cuda_stream)
        sclk = sustained_sclk_mhz(launch_fc1)
        del A, Bw, out
    except Exception:
        sclk = None
    traffic, alg_bytes = None, None
    tpath = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
    if not os.path.exists(tpath):
        tpath = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
    if not os.path.exists(tpath):
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(tpath):  # PMC FETCH_SIZE/WRITE_SIZE of the fc1-shape launch at this batch, measured offline
        ent = json.load(open(tpath)).get("by_batch", {}).get(str(batch))
        if ent and res == cfg.image_size and cfg.hidden_size == 1152:
            traffic, alg_bytes = ent["traffic_bytes_per_launch"], ent["algorithmic_bytes_per_launch"]
    return {"bound": "mfma", "kernel": "sgl::gemm_nt6_kernel (bf16 MFMA NT GEMM, the 4 forward shapes of one block)",
            "achieved": round(achieved, 1), "peak": PEAK_BF16_DENSE / 1e12, "unit": "TFLOP/s",
            "frac": round(achieved * 1e12 / PEAK_BF16_DENSE, 4), "traffic": traffic,
            "sustained_sclk_mhz": sclk,
            "sclk_note": "rocm-smi shader clock sampled while the kernel runs back to back for ~2 s; peak assumes 2400 MHz",
            "traffic_note": "bytes leaving the XCD L2s (HBM + Infinity Cache) for ONE fc1-shape launch: rocprofv3 "
                            f"FETCH_SIZE x2 + WRITE_SIZE, collected OFFLINE in separate --pmc passes ({os.path.basename(tpath)}; "
                            f"a process cannot read PMC counters for itself); algorithmic bytes of that launch: {alg_bytes}",
            "per_shape": per}


def step_breakdown(pkg, cfg, batch, res, reps, nt_per_shape):
    """Where one step's time goes, by kernel family, measured live: every family's kernel is launched `reps` times at the
    step's own shapes between two HIP events on the launch stream, and its mean duration is multiplied by the number of
    launches one step makes (L blocks: 4 forward NT GEMMs, 4 dX NT GEMMs, 4 dW TN GEMMs, 1 attention forward, 1 attention
    backward, 2 LayerNorm forward, 2 LayerNorm backward).  `accounted_ms` against `ms_per_step` shows what is left for the
    patch embedding, the pooling head, the weight-shadow casts, reductions and launch gaps."""
    import math
    lib = pkg.lib.load()
    gh = res // cfg.patch_size
    Ntok = gh * gh
    M = batch * Ntok
    D, I, L, Hh, dh = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.head_dim
    Ip, DP = (I + 127) // 128 * 128, (dh + 15) // 16 * 16
    st = torch.cuda.current_stream()
    dev = "cuda"

    def timeit(fn):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            fn()
        e1.record(st)
        e1.synchronize()
        return e0.elapsed_time(e1) / reps      # ms

    fam = {"nt_fwd": sum(v["ms"] for v in nt_per_shape.values())}
    # dX NT GEMMs: d(fc2)*gelu' [M,Ip]<-[M,D], d(fc1) [M,D]<-[M,Ip], d(out_proj) [M,D]<-[M,D], d(qkv) [M,D]<-[M,3D]
    t_dx = 0.0
    for N, K, epi in ((Ip, D, 4), (D, Ip, 0), (D, D, 0), (D, 3 * D, 0)):
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        aux = torch.randn(M, N, device=dev).bfloat16() if epi == 4 else None
        t_dx += timeit(lambda: lib.sgl_op_gemm_nt(1, A.data_ptr(), K, W.data_ptr(), K, M, N, K, epi, out.data_ptr(), N,
                                                  None, 0, None, None, 0, None if aux is None else aux.data_ptr(), N,
                                                  None, 1, 1, 1, 8, 8, 1, st.cuda_stream))
        del A, W, out, aux
    fam["nt_dx"] = t_dx
    # dW TN GEMMs (deterministic split-K with scratch): fc2 [D,I], fc1 [I,D], out_proj [D,D], qkv [3D,D]
    scratch = torch.empty(64 << 20, device=dev, dtype=torch.uint8)
    t_dw = 0.0
    for N1, N2, l1, l2 in ((D, I, D, Ip), (I, D, Ip, D), (D, D, D, D), (3 * D, D, 3 * D, D)):
        A = torch.randn(M, l1, device=dev).bfloat16()
        Bm = torch.randn(M, l2, device=dev).bfloat16()
        out = torch.empty(N1, N2, device=dev)
        t_dw += timeit(lambda: lib.sgl_op_gemm_tn_ws(1, A.data_ptr(), l1, Bm.data_ptr(), l2, M, N1, N2, 0, out.data_ptr(),
                                                     N2, 0, scratch.data_ptr(), scratch.numel(), st.cuda_stream))
        del A, Bm, out
    fam["tn_dw"] = t_dw
    qkv = torch.zeros(3, batch, Hh, Ntok, DP, device=dev, dtype=torch.bfloat16)   # head-major, as EPI_QKV writes it
    qkv[..., :dh] = torch.randn(3, batch, Hh, Ntok, dh, device=dev).bfloat16()
    o = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    do = torch.randn(M, D, device=dev).bfloat16()
    lse = torch.empty(batch, Hh, Ntok, device=dev)
    delta = torch.empty(2, batch, Hh, Ntok, device=dev)
    dqkv = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
    fam["attn_fwd"] = timeit(lambda: lib.sgl_op_attn_fwd(1, qkv[0].data_ptr(), qkv[1].data_ptr(), qkv[2].data_ptr(),
                                                         o.data_ptr(), lse.data_ptr(), batch, Hh, Ntok, dh, DP,
                                                         0, st.cuda_stream))
    fam["attn_bwd"] = timeit(lambda: lib.sgl_op_attn_bwd(1, qkv[0].data_ptr(), qkv[1].data_ptr(), qkv[2].data_ptr(),
                                                         o.data_ptr(), do.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                                         delta.data_ptr(), batch, Hh, Ntok, dh, DP, 0,
                                                         st.cuda_stream))
    del qkv, o, do, dqkv
    x = torch.randn(M, D, device=dev)
    gam, bet = torch.randn(D, device=dev), torch.randn(D, device=dev)
    y = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    dy = torch.randn(M, D, device=dev).bfloat16()
    dres, dx = torch.randn(M, D, device=dev), torch.empty(M, D, device=dev)
    dxlp = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    sc = torch.empty(32 << 20, device=dev, dtype=torch.uint8)
    fam["ln_fwd"] = 2 * timeit(lambda: lib.sgl_op_layernorm_fwd(x.data_ptr(), gam.data_ptr(), bet.data_ptr(), y.data_ptr(),
                                                                1, mean.data_ptr(), rstd.data_ptr(), M, D, 1e-6,
                                                                st.cuda_stream))
    fam["ln_bwd"] = 2 * timeit(lambda: lib.sgl_op_layernorm_bwd(dy.data_ptr(), 1, x.data_ptr(), mean.data_ptr(),
                                                                rstd.data_ptr(), gam.data_ptr(), dres.data_ptr(),
                                                                dx.data_ptr(), dxlp.data_ptr(), 1, dg.data_ptr(),
                                                                db.data_ptr(), sc.data_ptr(), sc.numel(), M, D,
                                                                st.cuda_stream))
    per_step = {k: round(v * L, 2) for k, v in fam.items()}
    flops_layer = {"nt_fwd": 2.0 * M * (4 * D * D + 2 * D * I), "nt_dx": 2.0 * M * (4 * D * D + 2 * D * I),
                   "tn_dw": 2.0 * M * (4 * D * D + 2 * D * I), "attn_fwd": 4.0 * batch * Hh * Ntok * Ntok * dh,
                   "attn_bwd": 10.0 * batch * Hh * Ntok * Ntok * dh}
    tf = {k: round(flops_layer[k] / (fam[k] * 1e-3) / 1e12, 1) for k in flops_layer}
    return {"ms_per_step": per_step, "accounted_ms": round(sum(per_step.values()), 2), "tflops": tf,
            "note": f"per-family kernel time = mean launch duration (HIP events, {reps} launches) x launches per step "
                    f"({L} blocks); LayerNorm rows are x2 (two per block)"}


def _median(v):
    v = sorted(v)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def full_train_step(pkg, model, x, steps=10, warmup=2):
    """fwd + bwd + FusedAdamW (clip + AdamW + bf16 weight-shadow writes in ONE pass: no separate re-cast) per step.

    Every step is bracketed by HIP events on the launch stream and reported individually (min / median / max), with the
    phases inside it (forward+backward, optimizer, zero_grad) and what could make a step slow for reasons that are not
    kernels: caching-allocator traffic (`num_device_alloc` = hipMalloc calls, `num_alloc_retries`) and whether FusedAdamW
    rebuilt and re-uploaded its device table inside the timed region.  Round 2's driver run showed 503.7 ms here against
    304 ms of fwd+bwd with 1 warm-up and 3 wall-clock-timed steps: the first optimizer step allocates 3.4 GB of AdamW state
    out of the cached 87 GB activation block the previous backward had just freed, so the NEXT step's activation arena
    needs a fresh hipMalloc of 87 GB - a one-off that a 3-step mean spreads over the figure.  Two warm-ups absorb it."""
    params = [p for p in model.parameters() if p.requires_grad]
    opt = pkg.FusedAdamW(params, lr=1e-5, weight_decay=0.01, max_grad_norm=1.0).attach_encoder(model)
    st = torch.cuda.current_stream()
    ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
    marks = []
    uploads = [0]
    orig_tables = opt._device_tables

    def counted_tables(lib, ents, dev):
        key0 = (opt._table_key, opt._aux_key)
        out = orig_tables(lib, ents, dev)
        uploads[0] += int((opt._table_key, opt._aux_key) != key0)
        return out
    opt._device_tables = counted_tables

    def step(record):
        e = [ev() for _ in range(4)]
        e[0].record(st)
        out = model(pixel_values=x, interpolate_pos_encoding=True)
        out.pooler_output.square().mean().backward()
        e[1].record(st)
        opt.step()
        e[2].record(st)
        opt.zero_grad(set_to_none=True)
        e[3].record(st)
        if record:
            marks.append(e)

    def mem():
        s_ = torch.cuda.memory_stats()
        return {k: int(s_.get(k, 0)) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries")}
    m_before_warm = mem()
    for _ in range(warmup):
        step(False)
    torch.cuda.synchronize()
    up_warm, uploads[0] = uploads[0], 0
    m0 = mem()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    m1 = mem()
    per = [e[0].elapsed_time(e[3]) for e in marks]
    fb = [e[0].elapsed_time(e[1]) for e in marks]
    op = [e[1].elapsed_time(e[2]) for e in marks]
    med = _median(per)
    return {"images_per_sec": round(x.shape[0] / (med * 1e-3), 2), "ms_per_step": round(med, 2),
            "ms_per_step_min": round(min(per), 2), "ms_per_step_max": round(max(per), 2),
            "ms_per_step_wall_mean": round(wall * 1e3, 2), "per_step_ms": [round(v, 1) for v in per],
            "fwd_bwd_ms_median": round(_median(fb), 2), "optimizer_ms_median": round(_median(op), 3),
            "timed_steps": steps, "warmup_steps": warmup,
            "allocator_delta_timed": {k: m1[k] - m0[k] for k in m0},
            "allocator_delta_warmup": {k: m0[k] - m_before_warm[k] for k in m0},
            "optimizer_table_uploads": {"warmup": up_warm, "timed": uploads[0]},
            "what": "forward + backward + FusedAdamW(max_grad_norm) with the weight shadows written by the optimizer; "
                    "HIP events per step on the launch stream, median reported"}


def optimizer_step_roofline(pkg, model, x, reps=10):
    """The step tail, reported beside the headline ('optimizer step reported separately'): fused
    global-norm clip + AdamW over every encoder parameter (csrc/optimizer.hip), HIP-event timed on its stream.
    Algorithmic bytes: 32 per parameter (norm pass reads g; update reads p,g,m,v and writes p,m,v)."""
    out = model(pixel_values=x[:2], interpolate_pos_encoding=True)
    out.pooler_output.square().mean().backward()
    params = [p for p in model.parameters() if p.grad is not None]
    opt = pkg.FusedAdamW(params, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0)
    stream = torch.cuda.current_stream()
    for _ in range(2):
        opt.step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        opt.step()
    e1.record(stream)
    e1.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    n = sum(p.numel() for p in params)
    return {"kernel": "sgl::grad_sqnorm_kernel + sgl::adamw_ex_kernel (clip_grad_norm_ + AdamW.step, fp32; no shadows attached)",
            "params": n, "ms": round(t * 1e3, 3), "bound": "hbm", "achieved": round(32.0 * n / t / 1e9, 1),
            "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": round(32.0 * n / t / PEAK_HBM, 4)}


def sustained_sclk_mhz(launch, max_seconds=8.0):
    """Median rocm-smi sclk (MHz) over three samples, every one of them taken (call start to call end) while `launch` is
    being re-issued back to back; None when rocm-smi is unavailable or too slow.  Informational: explains the distance
    between `peak` (datasheet, 2.4 GHz) and what the pool sustains under MFMA load."""
    import re
    import subprocess
    import threading
    samples, done = [], threading.Event()

    def probe():
        try:
            for i in range(4):
                r = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
                m = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", r)
                if m and i > 0:          # the first call may have started before the load did
                    samples.append(int(m.group(1)))
        except Exception:
            pass
        done.set()
    for _ in range(20):
        launch()
    th = threading.Thread(target=probe, daemon=True)
    t0 = time.perf_counter()
    th.start()
    while not done.is_set() and time.perf_counter() - t0 < max_seconds:
        for _ in range(20):
            launch()
        torch.cuda.synchronize()
    ok = done.is_set()       # samples are valid only if the load outlasted the probe
    th.join(timeout=15)
    samples.sort()
    return samples[len(samples) // 2] if (ok and samples) else None


def host_cores() -> int:
    """Cores this process may really use: cgroup quota (cpu.max) if set, else the affinity mask."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(pkg, cfg, res, steps):
    """The CPU oracle (port of the HF path, validated against HF in tests/) timed on this box's host cores, on a
    bounded sample of the same workload (batch 2, a few steps): fp32 (`value`) and, as SURVEY.md 8d asks, under
    torch.autocast("cpu", bfloat16) - the precision the reference trains in (Siglip2sidafrozen.py:1375)."""
    oracle = entry.load_oracle()
    cores = host_cores()
    torch.set_num_threads(cores)
    B = 2
    sd = {k: v.clone().requires_grad_(True) for k, v in pkg.weights.seeded_state_dict(cfg, seed=0).items()}
    x = pkg.weights.seeded_pixels(B, res, res, seed=1234)

    keep = {}

    def step(xb):
        out = oracle.vision_forward(xb, sd, cfg, False, True)
        keep.setdefault("pooled", out["pooler_output"].detach().float().clone())
        out["pooler_output"].float().square().mean().backward()
        for v in sd.values():
            v.grad = None
    print(f"[bench] cpu_baseline: warm-up on {cores} cores ...", file=sys.stderr, flush=True)
    step(x[:1])
    t0 = time.perf_counter()
    for i in range(steps):
        step(x)
        print(f"[bench] cpu_baseline: step {i + 1}/{steps} at {time.perf_counter() - t0:.1f} s", file=sys.stderr,
              flush=True)
    dt = (time.perf_counter() - t0) / steps
    out = {"value": round(B / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
           "sample": f"{cfg_name_of(cfg)} fp32 oracle, batch {B}, {steps} timed fwd+bwd steps after a 1-image warm-up",
           "_pooled_ref": keep["pooled"], "_x_ref": x[:1]}
    # bf16 autocast leg, bounded: a CPU without native bf16 dot products runs oneDNN's reference path (10-100x slower
    # than fp32), so the cost is estimated from one GEMM of the step's shape first
    a = torch.randn(B * (res // cfg.patch_size) ** 2, cfg.hidden_size)
    w = torch.randn(cfg.intermediate_size, cfg.hidden_size)

    def t_mm(a_, w_):
        a_ @ w_.t()
        t = time.perf_counter()
        for _ in range(3):
            a_ @ w_.t()
        return (time.perf_counter() - t) / 3
    ratio = t_mm(a.bfloat16(), w.bfloat16()) / t_mm(a, w)
    est = dt * ratio
    if est > 45.0:
        out["bf16_autocast"] = {"value": None, "note": f"skipped: a bf16 GEMM of the step's shape is {ratio:.1f}x slower "
                                f"than fp32 on this host (no native bf16 dot product), one step would take ~{est:.0f} s"}
        return out
    with torch.autocast("cpu", dtype=torch.bfloat16):
        step(x[:1])
        t0 = time.perf_counter()
        for i in range(steps):
            step(x)
            print(f"[bench] cpu_baseline (bf16 autocast): step {i + 1}/{steps} at {time.perf_counter() - t0:.1f} s",
                  file=sys.stderr, flush=True)
        dtb = (time.perf_counter() - t0) / steps
    out["bf16_autocast"] = {"value": round(B / dtb, 4), "unit": "images/sec",
                            "sample": f"same oracle under torch.autocast('cpu', bfloat16), batch {B}, {steps} timed steps"}
    return out


def tolerance_and_strict_mode(pkg, cfg, res, dev, ref_pooled, x_ref, strict_batch=32):
    """north_star states a tolerance ("logits within 1e-3 of the HF reference") next to the throughput target: print both.
    The fp32 CPU oracle's pooled output for one image (computed by the cpu_baseline leg's warm-up) against the benchmarked
    bf16 mode and against the strict mode on the matrix cores (compute_dtype "bf16x3": split-bf16 GEMMs, fp32-MFMA
    attention), and that strict mode's own training throughput."""
    out = {}
    for mode in ("bf16", "bf16x3"):
        model = build_model(pkg, cfg, mode, dev)
        with torch.no_grad():
            got = model(pixel_values=x_ref.to(dev), interpolate_pos_encoding=True).pooler_output.float().cpu()
        err = (got - ref_pooled).abs().max().item()
        if mode == "bf16":
            out["bf16_pooled_abs_err"] = round(err, 6)
        else:
            xb = pkg.weights.seeded_pixels(strict_batch, res, res, seed=77).to(dev)
            _, per = timed_steps(model, xb, 1, 3)
            gh = res // cfg.patch_size
            ips = strict_batch / (_median(per) * 1e-3)
            out["strict_mode"] = {"compute_dtype": "bf16x3", "images_per_sec": round(ips, 2), "batch": strict_batch,
                                  "pooled_abs_err": round(err, 7), "within_1e-3": bool(err < 1e-3),
                                  "step_mfma_frac_algorithmic": round(ips * cfg.train_flops_per_image(
                                      gh * cfg.patch_size, gh * cfg.patch_size) / PEAK_BF16_DENSE, 4),
                                  "what": "train fwd+bwd with every GEMM as ONE bf16 MFMA GEMM over hi/lo-split operands "
                                          "(3x the reduction length, fp32 accumulate) and fp32-MFMA attention; the error "
                                          "is max |pooled - fp32 CPU oracle| on one seeded image (output scale ~4.6)"}
            del xb
        del model
        torch.cuda.empty_cache()
    out["pooled_scale"] = round(ref_pooled.abs().max().item(), 3)
    return out


def cfg_name_of(cfg):
    return f"D{cfg.hidden_size}-I{cfg.intermediate_size}-L{cfg.num_hidden_layers}-p{cfg.patch_size}@{cfg.image_size}"


def build_model(pkg, cfg, mode, dev, freeze_below=0):
    model = pkg.SiglipVisionModelHIP(cfg, compute_dtype=mode)
    model.load_state_dict(pkg.weights.seeded_state_dict(cfg, seed=0))
    model = model.to(dev)
    set_frozen_prefix(model, freeze_below)
    return model


def set_frozen_prefix(model, k):
    """Siglip2sidafrozen.py:757-768: embeddings and blocks < k frozen (k = 0: everything trains)."""
    for p in model.vision_model.embeddings.parameters():
        p.requires_grad = k == 0
    for i, layer in enumerate(model.vision_model.encoder.layers):
        for p in layer.parameters():
            p.requires_grad = i >= k


def timed_steps(model, x, warmup, steps, world=1):
    """W untimed steps, then K steps bracketed by barrier + synchronize on both sides (the contract's wall clock), with a
    HIP event between steps on the launch stream for the per-step distribution.  Returns (seconds, [ms per step])."""
    params = [p for p in model.parameters()]
    trainable = [p for p in params if p.requires_grad]
    st = torch.cuda.current_stream()

    def step():
        # an optimizer step happened: every trainable parameter changed, so its bf16 shadows are re-cast in this step
        # (what autocast re-does per step in the reference, Siglip2sidafrozen.py:1375); frozen tensors keep theirs
        torch.autograd.graph.increment_version(trainable)
        out = model(pixel_values=x, interpolate_pos_encoding=True)
        loss = out.pooler_output.square().mean()
        loss.backward()
        for p in params:
            p.grad = None

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    evs[0].record(st)
    for i in range(steps):
        step()
        evs[i + 1].record(st)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt, [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]


def frozen_prefix_flops(cfg, gh, k):
    """fwd + 2*[(L-k)*layer + head]."""
    Nn, D, I, L = gh * gh, cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    layer = 8 * Nn * D * D + 4 * Nn * Nn * D + 4 * Nn * D * I
    head = 4 * Nn * D * D + 4 * D * D + 4 * Nn * D + 4 * D * I
    fwd = cfg.fwd_flops_per_image(gh * cfg.patch_size, gh * cfg.patch_size)
    return fwd + 2 * ((L - min(k, L)) * layer + head)


def secondary_metrics(pkg, cfg, model, res, dev, batch):
    """SURVEY.md 8d's secondary figures, measured in the same run (N = 1), HIP-event median over a few steps each:
    per-GPU batch sweep, the frozen-prefix SID configuration (blocks >= 21 trainable), the video configuration (frames/s =
    images/s at B = 32*b; clips/s = /32) and BASELINE config 2 (base-patch16-224, B = 256)."""
    out = {}
    gh = res // cfg.patch_size
    tf = cfg.train_flops_per_image(gh * cfg.patch_size, gh * cfg.patch_size)
    sweep = {}
    for b in (16, 32, 64):
        if b >= batch:
            continue
        xb = pkg.weights.seeded_pixels(b, res, res, seed=99).to(dev)
        _, per = timed_steps(model, xb, 2, 5)
        ips = b / (_median(per) * 1e-3)
        sweep[str(b)] = {"images_per_sec": round(ips, 1), "step_mfma_frac": round(ips * tf / PEAK_BF16_DENSE, 4)}
        del xb
    out["batch_sweep"] = sweep
    k = cfg.num_hidden_layers - 6
    if k > 0:
        set_frozen_prefix(model, k)
        xb = pkg.weights.seeded_pixels(batch, res, res, seed=1234).to(dev)
        _, per = timed_steps(model, xb, 2, 5)
        ips = batch / (_median(per) * 1e-3)
        out["frozen_prefix"] = {"what": f"embeddings + blocks < {k} frozen (Siglip2sidafrozen.py:757-768), B={batch}",
                                "images_per_sec": round(ips, 1),
                                "step_mfma_frac": round(ips * frozen_prefix_flops(cfg, gh, k) / PEAK_BF16_DENSE, 4)}
        set_frozen_prefix(model, 0)
        del xb
    return out


def base224_metric(pkg, dev):
    cfg = pkg.get_config("base-patch16-224")
    model = build_model(pkg, cfg, "bf16", dev)
    xb = pkg.weights.seeded_pixels(256, 224, 224, seed=1234).to(dev)
    _, per = timed_steps(model, xb, 3, 8)
    ips = 256 / (_median(per) * 1e-3)
    tf = cfg.train_flops_per_image(224, 224)
    return {"what": "BASELINE config 2: base-patch16-224 full fine-tune fwd+bwd, bf16, B=256",
            "images_per_sec": round(ips, 1), "ms_per_step": round(_median(per), 2),
            "step_mfma_frac": round(ips * tf / PEAK_BF16_DENSE, 4)}


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # rehearsal hooks (tests/test_bench_multiproc_gpu.py): every rank on one device over gloo; the driver never sets them
    backend = os.environ.get("SGL_BENCH_BACKEND", "nccl")
    if os.environ.get("SGL_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rccl_channels > 0:
            os.environ["NCCL_MAX_NCHANNELS"] = str(args.rccl_channels)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    pkg = entry.load_package()
    pkg.lib.load()
    cfg = pkg.get_config(args.config)
    res = args.res or cfg.image_size
    dev = torch.device("cuda", local)

    model = build_model(pkg, cfg, args.mode, dev, args.freeze_below)
    reducer = None
    if world > 1:
        pkg.ddp.broadcast_parameters(model, src=0)
        reducer = pkg.GradBucketReducer(wire=args.wire, max_buckets=args.max_buckets).attach(model)
        reducer.time_exposed = True
    x = pkg.weights.seeded_pixels(args.batch, res, res, seed=1234 + rank).to(dev)

    if reducer is not None:     # warm-up first, then measure only the timed region's exposed waits
        dt_w, _ = timed_steps(model, x, args.warmup, 0, world)
        reducer.exposed_ms()
        dt, per = timed_steps(model, x, 0, args.steps, world)
    else:
        dt, per = timed_steps(model, x, args.warmup, args.steps, world)
    my_rate = args.batch * args.steps / dt
    rates = [my_rate]
    exposed = reducer.exposed_ms() if reducer is not None else []
    exposed_per_step = sum(exposed) / max(1, args.steps)
    if world > 1:
        t = torch.tensor([dt, my_rate, exposed_per_step], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        dt = max(a[0].item() for a in allt)
        rates = [a[1].item() for a in allt]
        exposed_per_step = max(a[2].item() for a in allt)
    ms_per_step = dt / args.steps * 1e3
    images = args.batch * world * args.steps
    value = images / dt
    gh = res // cfg.patch_size
    train_flops = cfg.train_flops_per_image(gh * cfg.patch_size, gh * cfg.patch_size)
    if args.freeze_below > 0:
        train_flops = frozen_prefix_flops(cfg, gh, args.freeze_below)

    line = {
        "metric": ("images/sec (train fwd+bwd) SigLIP-2-so400m@384 bf16"
                   if args.config == "so400m-patch14-384" and args.freeze_below == 0
                   else f"images/sec (train fwd+bwd) {args.config}@{res} {args.mode}"
                        + (f" blocks<{args.freeze_below} frozen" if args.freeze_below else "")),
        "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.mode, "data": "synthetic",
        "config": {"workload": f"{args.config} encoder+pool-head train fwd+bwd, {res}x{res}, per-GPU batch "
                               f"{args.batch}, seeded random weights", "global_batch": args.batch * world,
                   "tokens_per_image": gh * gh, "parallelism": f"dp{world}"},
        "step_mfma_frac": round(value / world * train_flops / PEAK_BF16_DENSE, 4),
        "train_tflop_per_image": round(train_flops / 1e12, 4),
        "ms_per_step_events": {"median": round(_median(per), 3), "min": round(min(per), 3), "max": round(max(per), 3),
                               "note": "rank 0, HIP events between steps on the launch stream"},
        "video_equivalent": {"frames_per_sec": round(value, 2), "clips_per_sec_32_frames": round(value / 32, 3),
                             "note": "BASELINE config 5: a clip is 32 frames through the same encoder (hidf_video_"
                                     "classifier.py:299-320); per-GPU batch = 32*b frames"},
    }
    if world > 1:
        line["ranks_seen"] = dist.get_world_size()
        line["per_rank_images_per_sec"] = [round(r, 2) for r in rates]
        line["exposed_comm_ms"] = round(exposed_per_step, 3)
        line["wire"] = args.wire
        line["collectives_per_step"] = reducer.collectives_issued // max(1, args.steps + args.warmup)
        line["backend"] = backend
        line["rccl_max_nchannels"] = os.environ.get("NCCL_MAX_NCHANNELS")
    if rank == 0:
        print(f"[bench] {value:.1f} images/s, {ms_per_step:.1f} ms/step; measuring kernel roofline + CPU baseline ...",
              file=sys.stderr, flush=True)
        if world == 1:
            def leg(key, fn, into=None):
                """Optional legs never cost the headline: a failure is reported in the line instead of killing it."""
                try:
                    val = fn()
                    if val is not None:
                        (line if into is None else into)[key] = val
                except Exception as e:   # noqa: BLE001
                    line.setdefault("leg_errors", {})[key] = f"{type(e).__name__}: {e}"[:300]
                    torch.cuda.empty_cache()

            if args.kernel_reps > 0:
                leg("roofline", lambda: gemm_kernel_roofline(pkg, cfg, args.batch, res, args.kernel_reps))
                if args.mode == "bf16" and "roofline" in line:
                    leg("step_breakdown", lambda: step_breakdown(pkg, cfg, args.batch, res, max(3, args.kernel_reps // 4),
                                                                 line["roofline"]["per_shape"]))
            if not args.no_secondary and args.mode == "bf16" and args.freeze_below == 0:
                leg("secondary", lambda: secondary_metrics(pkg, cfg, model, res, dev, args.batch))
            if not args.no_optimizer:
                leg("optimizer_step", lambda: optimizer_step_roofline(pkg, model, x))
                if args.freeze_below == 0:
                    leg("train_step_with_optimizer", lambda: full_train_step(pkg, model, x, steps=args.train_steps))
            del model
            torch.cuda.empty_cache()
            if not args.no_secondary and args.mode == "bf16" and args.config == "so400m-patch14-384" \
                    and args.freeze_below == 0:
                leg("base_patch16_224_B256", lambda: base224_metric(pkg, dev), into=line.setdefault("secondary", {}))
                torch.cuda.empty_cache()
            if not args.no_cpu_baseline:
                def cpu_leg():
                    cb = cpu_baseline(pkg, cfg, res, args.cpu_steps)
                    ref_pooled, x_ref = cb.pop("_pooled_ref"), cb.pop("_x_ref")
                    line["cpu_baseline"] = cb
                    if args.mode == "bf16" and not args.no_secondary:
                        leg("tolerance", lambda: tolerance_and_strict_mode(pkg, cfg, res, dev, ref_pooled, x_ref))
                leg("cpu_baseline_leg", cpu_leg)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

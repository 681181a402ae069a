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
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--kernel-reps", type=int, default=20,
                    help="launches per kernel for the roofline / breakdown legs; 0 skips them (clean rocprofv3 traces)")
    ap.add_argument("--no-optimizer", action="store_true",
                    help="skip the optimizer-step and full-train-step legs (clean rocprofv3 traces of the timed step)")
    return ap.parse_args()


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
    qkv = torch.zeros(3, batch, Hh, Ntok, DP, device=dev, dtype=torch.bfloat16)
    qkv[..., :dh] = torch.randn(3, batch, Hh, Ntok, dh, device=dev).bfloat16()
    o = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    do = torch.randn(M, D, device=dev).bfloat16()
    lse = torch.empty(batch, Hh, Ntok, device=dev)
    delta = torch.empty(2, batch, Hh, Ntok, device=dev)
    dqkv = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
    fam["attn_fwd"] = timeit(lambda: lib.sgl_op_attn_fwd(1, qkv[0].data_ptr(), qkv[1].data_ptr(), qkv[2].data_ptr(),
                                                         o.data_ptr(), lse.data_ptr(), batch, Hh, Ntok, dh, DP,
                                                         st.cuda_stream))
    fam["attn_bwd"] = timeit(lambda: lib.sgl_op_attn_bwd(1, qkv[0].data_ptr(), qkv[1].data_ptr(), qkv[2].data_ptr(),
                                                         o.data_ptr(), do.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                                         delta.data_ptr(), batch, Hh, Ntok, dh, DP, st.cuda_stream))
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


def full_train_step(pkg, model, x, steps=3):
    """fwd + bwd + FusedAdamW (clip + AdamW + bf16 weight-shadow writes in ONE pass: no separate re-cast) per step."""
    params = [p for p in model.parameters() if p.requires_grad]
    opt = pkg.FusedAdamW(params, lr=1e-5, weight_decay=0.01, max_grad_norm=1.0).attach_encoder(model)

    def step():
        out = model(pixel_values=x, interpolate_pos_encoding=True)
        out.pooler_output.square().mean().backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"images_per_sec": round(x.shape[0] / dt, 2), "ms_per_step": round(dt * 1e3, 2),
            "what": "forward + backward + FusedAdamW(max_grad_norm) with the weight shadows written by the optimizer"}


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
    bounded sample of the same workload (batch 2, a few steps)."""
    oracle = entry.load_oracle()
    cores = host_cores()
    torch.set_num_threads(cores)
    B = 2
    sd = {k: v.clone().requires_grad_(True) for k, v in pkg.weights.seeded_state_dict(cfg, seed=0).items()}
    x = pkg.weights.seeded_pixels(B, res, res, seed=1234)

    def step(xb):
        out = oracle.vision_forward(xb, sd, cfg, False, True)
        out["pooler_output"].square().mean().backward()
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
    return {"value": round(B / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{cfg_name_of(cfg)} fp32 oracle, batch {B}, {steps} timed fwd+bwd steps after a 1-image warm-up"}


def cfg_name_of(cfg):
    return f"D{cfg.hidden_size}-I{cfg.intermediate_size}-L{cfg.num_hidden_layers}-p{cfg.patch_size}@{cfg.image_size}"


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    # rehearsal hooks (tests/test_bench_multiproc_gpu.py): every rank on one device over gloo; the driver never sets them
    backend = os.environ.get("SGL_BENCH_BACKEND", "nccl")
    if os.environ.get("SGL_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    pkg = entry.load_package()
    pkg.lib.load()
    cfg = pkg.get_config(args.config)
    res = args.res or cfg.image_size
    dev = torch.device("cuda", local)

    model = pkg.SiglipVisionModelHIP(cfg, compute_dtype=args.mode)
    model.load_state_dict(pkg.weights.seeded_state_dict(cfg, seed=0))
    model = model.to(dev)
    if args.freeze_below > 0:
        for p in model.vision_model.embeddings.parameters():
            p.requires_grad = False
        for i, layer in enumerate(model.vision_model.encoder.layers):
            for p in layer.parameters():
                p.requires_grad = i >= args.freeze_below
    if world > 1:
        pkg.ddp.broadcast_parameters(model, src=0)
        pkg.GradBucketReducer().attach(model)
    x = pkg.weights.seeded_pixels(args.batch, res, res, seed=1234 + rank).to(dev)
    params = [p for p in model.parameters()]
    trainable = [p for p in params if p.requires_grad]

    def step():
        # an optimizer step happened: every trainable parameter changed, so its bf16 shadows are re-cast in this step
        # (what autocast re-does per step in the reference, Siglip2sidafrozen.py:1375); frozen tensors keep theirs
        torch.autograd.graph.increment_version(trainable)
        out = model(pixel_values=x, interpolate_pos_encoding=True)
        loss = out.pooler_output.square().mean()
        loss.backward()
        for p in params:
            p.grad = None

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms_per_step = dt / args.steps * 1e3
    images = args.batch * world * args.steps
    value = images / dt
    gh = res // cfg.patch_size
    train_flops = cfg.train_flops_per_image(gh * cfg.patch_size, gh * cfg.patch_size)
    if args.freeze_below > 0:  # fwd + 2*[(L-k)*layer + head]
        Nn, D, I, L = gh * gh, cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
        layer = 8 * Nn * D * D + 4 * Nn * Nn * D + 4 * Nn * D * I
        head = 4 * Nn * D * D + 4 * D * D + 4 * Nn * D + 4 * D * I
        fwd = cfg.fwd_flops_per_image(gh * cfg.patch_size, gh * cfg.patch_size)
        train_flops = fwd + 2 * ((L - min(args.freeze_below, L)) * layer + head)

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
    }
    if rank == 0:
        print(f"[bench] {value:.1f} images/s, {ms_per_step:.1f} ms/step; measuring kernel roofline + CPU baseline ...",
              file=sys.stderr, flush=True)
        if world == 1:
            if args.kernel_reps > 0:
                line["roofline"] = gemm_kernel_roofline(pkg, cfg, args.batch, res, args.kernel_reps)
                if args.mode == "bf16":
                    line["step_breakdown"] = step_breakdown(pkg, cfg, args.batch, res, max(3, args.kernel_reps // 4),
                                                            line["roofline"]["per_shape"])
            if not args.no_optimizer:
                line["optimizer_step"] = optimizer_step_roofline(pkg, model, x)
                if args.freeze_below == 0:
                    line["train_step_with_optimizer"] = full_train_step(pkg, model, x)
            del model
            torch.cuda.empty_cache()
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(pkg, cfg, res, args.cpu_steps)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

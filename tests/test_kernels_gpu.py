"""GPU parity tests of the individual HIP kernels, called through the C ABI (``sgl_op_*``).

Reference for each kernel = the same op in plain PyTorch fp32 on the same inputs (bf16 inputs are rounded first,
so the only difference is accumulation order / output rounding).  Tolerances are stated per test:
bf16 outputs carry 2^-9 relative rounding; fp32-strict kernels must agree to ~1e-5.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = 1, 0


@pytest.fixture(scope="module")
def lib(hiplib):
    assert torch.cuda.is_available()
    return hiplib


def stream():
    return torch.cuda.current_stream().cuda_stream


def P(t):
    return None if t is None else t.data_ptr()


def relerr(got, ref):
    got, ref = got.float(), ref.float()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-12)).item()


def ok(st):
    assert st == 0, f"C ABI status {st}"


def gelu_tanh(x):
    return torch.nn.functional.gelu(x, approximate="tanh")


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,D", [(5, 64), (77, 144), (1458, 1152), (300, 768), (33, 2048)])
@pytest.mark.parametrize("odt", [F32, BF16])
def test_layernorm_fwd_bwd(lib, M, D, odt):
    torch.manual_seed(M + D)
    dev = "cuda"
    x = torch.randn(M, D, device=dev) * 2 + 0.3
    g = torch.randn(D, device=dev) * 0.2 + 1
    b = torch.randn(D, device=dev) * 0.1
    tdt = torch.bfloat16 if odt == BF16 else torch.float32
    y = torch.empty(M, D, device=dev, dtype=tdt)
    mean = torch.empty(M, device=dev)
    rstd = torch.empty(M, device=dev)
    ok(lib.sgl_op_layernorm_fwd(P(x), P(g), P(b), P(y), odt, P(mean), P(rstd), M, D, 1e-6, stream()))
    xr = x.clone().requires_grad_(True)
    gr = g.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
    tol = 1e-5 if odt == F32 else 6e-3
    assert relerr(y, ref) < tol
    assert relerr(mean, x.mean(-1)) < 1e-5
    # backward
    dy = torch.randn(M, D, device=dev).to(tdt)
    dres = torch.randn(M, D, device=dev)
    dx = torch.empty(M, D, device=dev)
    dxlp = torch.empty(M, D, device=dev, dtype=tdt)
    dg = torch.empty(D, device=dev)
    db = torch.empty(D, device=dev)
    scratch = torch.empty(1024 * 3 * D, device=dev)
    ok(lib.sgl_op_layernorm_bwd(P(dy), odt, P(x), P(mean), P(rstd), P(g), P(dres), P(dx), P(dxlp), odt, P(dg), P(db),
                                P(scratch), scratch.numel() * 4, M, D, stream()))
    ref.backward(dy.float())
    assert relerr(dx, xr.grad + dres) < 2e-5
    assert relerr(dxlp, xr.grad + dres) < tol
    assert relerr(dg, gr.grad) < 2e-5
    assert relerr(db, br.grad) < 2e-5


# ---------------------------------------------------------------------------------------------------------
def call_gemm_nt(lib, dtype, A, B, M, N, K, epi, out, ldo, out2=None, ldo2=0, bias=None, res=None, ldr=0, aux=None,
                 ldaux=0, pos=None, pos_rows=1, tokens=1, heads=1, hd=8, hdp=8, batch=1):
    return lib.sgl_op_gemm_nt(dtype, P(A), A.stride(0), P(B), B.stride(0), M, N, K, epi, P(out), ldo, P(out2), ldo2,
                              P(bias), P(res), ldr, P(aux), ldaux, P(pos), pos_rows, tokens, heads, hd, hdp, batch,
                              stream())


GEMM_SHAPES = [(128, 128, 64), (1, 8, 8), (200, 136, 72), (729, 1152, 1152), (1458, 3456, 1152), (300, 1152, 4352),
               (64, 4352, 1152), (257, 144, 640),
               # large enough for the 256x256-tile LDS-DMA generation (M >= 2048, N >= 256), with M/N/K tails
               (4096, 1152, 1152), (2300, 4352, 1152), (2125, 1152, 4352), (2916, 3456, 1152), (2049, 264, 72),
               (2100, 304, 40), (2560, 512, 64), (2051, 1160, 200)]   # one K-step / exactly one / odd tails


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("dtype", [BF16, F32])
def test_gemm_nt_store_bias(lib, M, N, K, dtype):
    torch.manual_seed(M * 7 + N + K)
    tdt = torch.bfloat16 if dtype == BF16 else torch.float32
    A = torch.randn(M, K, device="cuda").to(tdt)
    B = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(tdt)
    bias = torch.randn(N, device="cuda")
    out = torch.full((M, N), float("nan"), device="cuda", dtype=tdt)
    ok(call_gemm_nt(lib, dtype, A, B, M, N, K, 0, out, N, bias=bias))
    ref = A.float() @ B.float().t() + bias
    assert relerr(out, ref) < (6e-3 if dtype == BF16 else 2e-5)


@pytest.mark.parametrize("dtype,M", [(BF16, 333), (BF16, 2333), (F32, 333)])
def test_gemm_nt_epilogues(lib, dtype, M):
    torch.manual_seed(5)
    tdt = torch.bfloat16 if dtype == BF16 else torch.float32
    tol = 8e-3 if dtype == BF16 else 3e-5
    N, K = 384, 192
    A = torch.randn(M, K, device="cuda").to(tdt)
    B = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(tdt)
    bias = torch.randn(N, device="cuda")
    acc = A.float() @ B.float().t()
    # bias + gelu (pre-activation and activation)
    u = torch.empty(M, N, device="cuda", dtype=tdt)
    a = torch.empty(M, N, device="cuda", dtype=tdt)
    ok(call_gemm_nt(lib, dtype, A, B, M, N, K, 1, u, N, out2=a, ldo2=N, bias=bias))
    assert relerr(u, acc + bias) < tol
    assert relerr(a, gelu_tanh(acc + bias)) < tol
    # residual fp32
    res = torch.randn(M, N, device="cuda")
    o = torch.empty(M, N, device="cuda")
    ok(call_gemm_nt(lib, dtype, A, B, M, N, K, 2, o, N, bias=bias, res=res, ldr=N))
    assert relerr(o, res + acc + bias) < 2e-5
    # gelu backward
    uu = (torch.randn(M, N, device="cuda") * 1.5).to(tdt)
    o2 = torch.empty(M, N, device="cuda", dtype=tdt)
    ok(call_gemm_nt(lib, dtype, A, B, M, N, K, 4, o2, N, aux=uu, ldaux=N))
    ur = uu.float().requires_grad_(True)
    gelu_tanh(ur).backward(acc)
    assert relerr(o2, ur.grad) < tol
    # position add fp32
    pos = torch.randn(37, N, device="cuda")
    o3 = torch.empty(M, N, device="cuda")
    ok(call_gemm_nt(lib, dtype, A, B, M, N, K, 5, o3, N, bias=bias, pos=pos, pos_rows=37))
    ref = acc + bias + pos[torch.arange(M, device="cuda") % 37]
    assert relerr(o3, ref) < 2e-5
    # fp32 plain
    o4 = torch.empty(M, N, device="cuda")
    ok(call_gemm_nt(lib, dtype, A, B, M, N, K, 6, o4, N, bias=bias))
    assert relerr(o4, acc + bias) < 2e-5


@pytest.mark.parametrize("dtype", [BF16, F32])
@pytest.mark.parametrize("batch,tokens,heads,hd", [(2, 9, 2, 72), (3, 49, 4, 16), (1, 729, 16, 72), (2, 196, 12, 64),
                                                   (3, 729, 16, 72), (11, 196, 12, 64)])
def test_gemm_nt_qkv_scatter(lib, dtype, batch, tokens, heads, hd):
    torch.manual_seed(11)
    tdt = torch.bfloat16 if dtype == BF16 else torch.float32
    D = heads * hd
    hdp = (hd + 15) // 16 * 16
    M, N, K = batch * tokens, 3 * D, D
    A = torch.randn(M, K, device="cuda").to(tdt)
    B = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(tdt)
    bias = torch.randn(N, device="cuda")
    out = torch.full((3, batch, heads, tokens, hdp), float("nan"), device="cuda", dtype=tdt)
    ok(call_gemm_nt(lib, dtype, A, B, M, N, K, 3, out, 0, bias=bias, tokens=tokens, heads=heads, hd=hd, hdp=hdp,
                    batch=batch))
    ref = (A.float() @ B.float().t() + bias).view(batch, tokens, 3, heads, hd).permute(2, 0, 3, 1, 4)
    assert relerr(out[..., :hd], ref) < (8e-3 if dtype == BF16 else 2e-5)
    if hdp > hd:
        assert (out[..., hd:] == 0).all(), "pad columns must be written as zero"


@pytest.mark.parametrize("Mred,N1,N2,splits", [(64, 128, 128, 1), (729, 144, 144, 1), (1458, 1152, 1152, 4),
                                                 (2187, 538, 144, 3), (300, 256, 588, 1), (5000, 4304, 1152, 2),
                                                 (130, 8, 16, 1), (4100, 3456, 1152, 1), (2048, 1152, 1152, 1),
                                                 (2916, 1152, 4304, 1), (2500, 538, 640, 1), (9000, 1152, 588, 1),
                                                 (2048, 512, 512, 16), (2090, 520, 1152, 8), (2048, 1152, 512, 1)])
@pytest.mark.parametrize("dtype", [BF16, F32])
def test_gemm_tn(lib, Mred, N1, N2, splits, dtype):
    torch.manual_seed(Mred + N1)
    tdt = torch.bfloat16 if dtype == BF16 else torch.float32
    lda, ldb = (N1 + 7) // 8 * 8 + 8, (N2 + 7) // 8 * 8
    A = torch.zeros(Mred, lda, device="cuda", dtype=tdt)
    B = torch.zeros(Mred, ldb, device="cuda", dtype=tdt)
    A[:, :N1] = torch.randn(Mred, N1, device="cuda").to(tdt)
    B[:, :N2] = torch.randn(Mred, N2, device="cuda").to(tdt)
    ref = A[:, :N1].float().t() @ B[:, :N2].float()
    out = torch.full((N1, N2), float("nan"), device="cuda")
    ok(lib.sgl_op_gemm_tn(dtype, P(A), lda, P(B), ldb, Mred, N1, N2, splits, P(out), N2, 0, stream()))
    assert relerr(out, ref) < 3e-5
    # accumulate
    ok(lib.sgl_op_gemm_tn(dtype, P(A), lda, P(B), ldb, Mred, N1, N2, splits, P(out), N2, 1, stream()))
    assert relerr(out, 2 * ref) < 3e-5


# ---------------------------------------------------------------------------------------------------------
def attn_reference(q, k, v, dout):
    """fp32 eager attention + autograd on (B,H,N,dh) tensors."""
    q, k, v = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    dh = q.shape[-1]
    s = (q @ k.transpose(-1, -2)) * dh ** -0.5
    p = torch.softmax(s, dim=-1)
    o = p @ v
    o.backward(dout.float())
    lse = torch.logsumexp(s, dim=-1)
    return o.detach(), lse.detach(), q.grad, k.grad, v.grad


@pytest.mark.parametrize("B,H,N,dh", [(1, 2, 4, 16), (2, 2, 9, 72), (2, 4, 49, 16), (1, 3, 196, 64), (2, 16, 729, 72),
                                      (1, 2, 1024, 72), (1, 2, 130, 32)])
@pytest.mark.parametrize("dtype", [BF16, F32, 2])   # 2 = SGL_DTYPE_BF16X3: fp32 operands on the fp32 MFMA (attention_f32.hip)
@pytest.mark.parametrize("layout", ["token", "head"])
def test_attention_fwd_bwd(lib, B, H, N, dh, dtype, layout):
    """layout "token": q | k | v are the column blocks of one token-major [B*N, 3D] matrix (what the encoder's QKV GEMM
    writes; each head's dh-element row segments are gathered by the kernels, nothing padded in memory), with NaN in a
    4th column block behind v to catch any read past a head's segment; "head": the legacy [B,H,N,DP] matrices."""
    torch.manual_seed(N + dh)
    tdt = torch.bfloat16 if dtype == BF16 else torch.float32
    DP = (dh + 15) // 16 * 16
    D = H * dh
    qkv = torch.zeros(3, B, H, N, DP, device="cuda", dtype=tdt)
    qkv[..., :dh] = (torch.randn(3, B, H, N, dh, device="cuda") * 1.2).to(tdt)
    if layout == "token":
        ld = 3 * D + 8
        tok = torch.full((B * N, ld), float("nan"), device="cuda", dtype=tdt)
        tok[:, :3 * D] = qkv[..., :dh].permute(1, 3, 0, 2, 4).reshape(B * N, 3 * D)
        ptrs = [P(tok) + j * D * tok.element_size() for j in range(3)]
    else:
        ld, ptrs = 0, [P(qkv[0]), P(qkv[1]), P(qkv[2])]
    dout_tok = torch.randn(B * N, D, device="cuda").to(tdt)
    out = torch.full((B * N, D), float("nan"), device="cuda", dtype=tdt)
    lse = torch.empty(B, H, N, device="cuda")
    ok(lib.sgl_op_attn_fwd(dtype, ptrs[0], ptrs[1], ptrs[2], P(out), P(lse), B, H, N, dh, DP, ld, stream()))
    dout_h = dout_tok.view(B, N, H, dh).permute(0, 2, 1, 3)
    o_ref, lse_ref, dq, dk, dv = attn_reference(qkv[0][..., :dh], qkv[1][..., :dh], qkv[2][..., :dh], dout_h)
    o_tok = o_ref.permute(0, 2, 1, 3).reshape(B * N, D)
    tol = 1.5e-2 if dtype == BF16 else 2e-5
    assert relerr(out, o_tok) < tol, "forward output"
    assert (lse - lse_ref).abs().max().item() < (2e-2 if dtype == BF16 else 1e-4), "log-sum-exp"
    dqkv = torch.full((B * N, 3 * D), float("nan"), device="cuda", dtype=tdt)
    delta = torch.empty(2, B, H, N, device="cuda")   # scratch: 2*B*H*N floats (include/siglip_hip.h)
    # the backward consumes the forward's own output (as the encoder does)
    ok(lib.sgl_op_attn_bwd(dtype, ptrs[0], ptrs[1], ptrs[2], P(out), P(dout_tok), P(lse), P(dqkv), P(delta), B, H,
                           N, dh, DP, ld, stream()))
    got = dqkv.view(B, N, 3, H, dh).permute(2, 0, 3, 1, 4).float()
    btol = 3e-2 if dtype == BF16 else 5e-5
    assert relerr(got[0], dq) < btol, "dQ"
    assert relerr(got[1], dk) < btol, "dK"
    assert relerr(got[2], dv) < btol, "dV"


def test_attention_softmax_rescale_branch(lib):
    """Force the online-softmax running max to jump at a late KV tile (guide §5.4 rule 26): one key far above the
    rest for one query, placed in the last tile."""
    torch.manual_seed(0)
    B, H, N, dh, DP = 1, 1, 300, 64, 64
    qkv = (torch.randn(3, B, H, N, DP, device="cuda") * 0.5)
    qkv[0, 0, 0, 7] = 4.0
    qkv[1, 0, 0, 290] = 4.0  # score 4*4*64/8 = 128 >> others, inside the last 64-key tile
    qkv = qkv.to(torch.bfloat16)
    out = torch.empty(N, dh, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, N, device="cuda")
    ok(lib.sgl_op_attn_fwd(BF16, P(qkv[0]), P(qkv[1]), P(qkv[2]), P(out), P(lse), B, H, N, dh, DP, 0, stream()))
    q, k, v = (t.float() for t in qkv)
    ref = torch.softmax((q @ k.transpose(-1, -2)) / 8.0, -1) @ v
    assert relerr(out, ref[0, 0]) < 1.5e-2
    assert torch.isfinite(out.float()).all()


# ---------------------------------------------------------------------------------------------------------
def test_colsum_im2col_posresize(lib, oracle):
    torch.manual_seed(2)
    x = torch.randn(1000, 136, device="cuda").to(torch.bfloat16)
    out = torch.empty(136, device="cuda")
    scratch = torch.empty(256 * 136, device="cuda")
    ok(lib.sgl_op_colsum(BF16, P(x), 136, 1000, 136, P(out), 0, P(scratch), scratch.numel() * 4, stream()))
    assert relerr(out, x.float().sum(0)) < 1e-5
    # im2col == unfold of the 'valid' strided conv (incl. an image size that is not a multiple of the patch)
    for (Bn, Hh, Ww, Pp, cl) in [(2, 42, 42, 14, 0), (1, 45, 45, 14, 0), (2, 32, 32, 16, 1)]:
        pix = torch.randn(Bn, 3, Hh, Ww, device="cuda")
        K = 3 * Pp * Pp
        Kp = (K + 63) // 64 * 64
        gh, gw = Hh // Pp, Ww // Pp
        o = torch.empty(Bn * gh * gw, Kp, device="cuda")
        src = pix.contiguous(memory_format=torch.channels_last) if cl else pix
        ok(lib.sgl_op_im2col(P(src), cl, P(o), F32, Bn, Hh, Ww, Pp, Kp, stream()))
        ref = torch.nn.functional.unfold(pix[:, :, :gh * Pp, :gw * Pp], Pp, stride=Pp).transpose(1, 2).reshape(-1, K)
        assert torch.equal(o[:, :K], ref)
        assert (o[:, K:] == 0).all()
    for g0, gh in [(2, 3), (3, 7), (27, 16), (14, 20)]:
        t = torch.randn(g0 * g0, 40, device="cuda")
        o = torch.empty(gh * gh, 40, device="cuda")
        ok(lib.sgl_op_pos_resize(P(t), g0, P(o), gh, gh, 40, stream()))
        ref = torch.nn.functional.interpolate(t.view(1, g0, g0, 40).permute(0, 3, 1, 2), size=(gh, gh), mode="bicubic",
                                              align_corners=False).permute(0, 2, 3, 1).reshape(gh * gh, 40)
        assert (o - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,gh,gw,E", [(3, 27, 27, 256), (2, 27, 27, 512), (2, 5, 7, 64), (1, 1, 1, 128), (2, 3, 9, 8)])
@pytest.mark.parametrize("tdt", [torch.float32, torch.bfloat16])
def test_depthwise_conv3x3_fwd_and_grads(B, gh, gw, E, tdt):
    """csrc/decoder.hip against the operator it replaces: nn.Conv2d(E, E, 3, padding=1, groups=E)
    (Siglip2sidafrozen.py:713-718) in fp32 on the same (bf16-rounded) inputs."""
    import __graft_entry__ as entry
    pkg = entry.load_package()
    torch.manual_seed(B * 100 + E)
    conv = torch.nn.Conv2d(E, E, 3, padding=1, groups=E).cuda()
    x = torch.randn(B, gh, gw, E, device="cuda").to(tdt).requires_grad_(True)
    dy = torch.randn(B, gh, gw, E, device="cuda").to(tdt)
    y = pkg.heads._DepthwiseConv3x3Fn.apply(x, conv.weight, conv.bias)
    assert y.dtype == tdt
    gx, gw_, gb = torch.autograd.grad(y, [x, conv.weight, conv.bias], dy)
    xr = x.detach().float().requires_grad_(True)
    yr = conv(xr.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    rx, rw, rb = torch.autograd.grad(yr, [xr, conv.weight, conv.bias], dy.float())
    tol = 2e-5 if tdt == torch.float32 else 1.2e-2
    assert relerr(y, yr) < tol and relerr(gx, rx) < tol
    assert relerr(gw_, rw) < (1e-4 if tdt == torch.float32 else 1.2e-2) and relerr(gb, rb) < (1e-4 if tdt == torch.float32 else 1.2e-2)


@pytest.mark.parametrize("M,K,N", [(2916, 1152, 512), (729, 512, 512), (1458, 2048, 512), (300, 136, 72)])
def test_hip_linear_function_matches_autocast_linear(M, K, N):
    """heads._HipLinearFn (the decoder's GEMMs on the encoder's MFMA kernels) against what it replaces under autocast:
    F.linear with bf16 operands, checked in fp32 on the same bf16-rounded inputs (forward, dX, dW, db)."""
    import __graft_entry__ as entry
    pkg = entry.load_package()
    torch.manual_seed(M + N)
    lin = torch.nn.Linear(K, N).cuda()
    x = torch.randn(M, K, device="cuda").bfloat16().requires_grad_(True)
    dy = torch.randn(M, N, device="cuda").bfloat16()
    y = pkg.heads._HipLinearFn.apply(x, lin.weight, lin.bias)
    gx, gw, gb = torch.autograd.grad(y, [x, lin.weight, lin.bias], dy)
    xr = x.detach().float().requires_grad_(True)
    wr = lin.weight.detach().bfloat16().float().requires_grad_(True)
    yr = xr @ wr.t() + lin.bias
    rx, rw, rb = torch.autograd.grad(yr, [xr, wr, lin.bias], dy.float())
    assert y.dtype == torch.bfloat16 and relerr(y, yr) < 8e-3
    assert relerr(gx, rx) < 8e-3 and relerr(gw, rw) < 8e-3 and relerr(gb, rb) < 8e-3
    with torch.autocast("cuda", dtype=torch.bfloat16):       # the dispatcher picks the HIP path only under autocast
        y2 = pkg.heads._linear_tokens(x.detach().reshape(1, M, K), lin.weight, lin.bias)
    assert y2.shape == (1, M, N) and relerr(y2[0], yr) < 8e-3
    if M >= 2048 and N >= 512 and K >= 512:
        assert torch.equal(y2[0], y)                         # ... and only for shapes where it pays


def test_gemm_tn_with_scratch_is_reproducible_and_correct(lib):
    """sgl_op_gemm_tn_ws: split-K partial tiles in caller scratch, summed in a fixed order -> same bits every run."""
    torch.manual_seed(1)
    Mred, N1, N2 = 6000, 1152, 512
    A = torch.randn(Mred, N1, device="cuda").bfloat16()
    B = torch.randn(Mred, N2, device="cuda").bfloat16()
    scratch = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
    outs = []
    for _ in range(2):
        out = torch.full((N1, N2), float("nan"), device="cuda")
        ok(lib.sgl_op_gemm_tn_ws(BF16, P(A), N1, P(B), N2, Mred, N1, N2, 0, P(out), N2, 0, P(scratch), scratch.numel(),
                                 torch.cuda.current_stream().cuda_stream))
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert relerr(outs[0], A.float().t() @ B.float()) < 2e-3
    acc = outs[0].clone()
    ok(lib.sgl_op_gemm_tn_ws(BF16, P(A), N1, P(B), N2, Mred, N1, N2, 0, P(acc), N2, 1, P(scratch), scratch.numel(),
                             torch.cuda.current_stream().cuda_stream))
    assert relerr(acc, 2 * outs[0]) < 1e-6


@pytest.mark.parametrize("gen", ["2", "7", "8"])
def test_opt_in_nt_gemm_generations_are_correct(gen):
    """The A/B NT GEMM generations (SGL_GEMM_GEN=2 one barrier per K-step, =7 persistent, =8 four-wave; DESIGN.md negative
    results) live only in the developer library libsiglip_hip_ab.so (make AB=1; __graft_entry__.build() builds it) and stay
    correct on the encoder's eight shapes: tests/bench_nt.py checks every launch against torch (head, tail rows, pad
    columns) and asserts.  The generation is latched per process, hence the subprocess."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ab = os.path.join(root, "deepfake-detection-using-clip-based-siglip-2-vision-transformers_amd", "libsiglip_hip_ab.so")
    if not os.path.exists(ab):
        pytest.skip("developer A/B library not built (make -C csrc AB=1)")
    env = dict(os.environ, SGL_GEMM_GEN=gen, SGL_LIB_PATH=ab)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "bench_nt.py"), "8"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("relerr") == 8, r.stdout

// Fused GEMM epilogues, applied to NV consecutive output columns of one row (row-contiguous so that bias,
// residual and output traffic is 16-byte vectorised).  Shared by the MFMA kernels (NV = 8, after the
// accumulator tile has been transposed through LDS) and the strict-fp32 generic kernel (NV = 4).
//
// Reference ops being fused (TF:models/siglip/modeling_siglip.py):
//   EPI_QKV       q/k/v_proj bias + view(B,N,H,dh).transpose(1,2)            :284-286
//   EPI_RES_F32   out_proj / fc2 bias + residual add                        :303-304,349,354
//   EPI_BIAS_GELU fc1 bias + gelu_pytorch_tanh                              :319-320
//   EPI_POS_F32   patch conv bias + position embedding add                  :178-184
#pragma once
#include "common.hip.h"
#include "kernels.h"

namespace sgl {

template <int EPI, typename TOut, int NV>
__device__ __forceinline__ void epi_apply(const EpiParams& p, int row, int col, int N, float* v) {
  if constexpr (EPI == EPI_STORE) {
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] *= p.alpha;
    if (p.bias) {
      float b[NV];
      Vec<float, NV>::ld(p.bias + col, b);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j] += b[j];
    }
    Vec<TOut, NV>::st_nt(reinterpret_cast<TOut*>(p.out) + (size_t)row * p.ldo + col, v);
  } else if constexpr (EPI == EPI_BIAS_GELU) {
    float b[NV], a[NV];
    Vec<float, NV>::ld(p.bias + col, b);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      v[j] += b[j];
      if (p.out && p.gelu_grad_form) gelu_tanh_both(v[j], a[j], v[j]);   // v := gelu'(u)
      else a[j] = gelu_tanh(v[j]);
    }
    // the pre-activation u (or gelu'(u), gelu_grad_form) is only read by the backward GELU': inference and frozen blocks
    // pass out == nullptr
    if (p.out) Vec<TOut, NV>::st_nt(reinterpret_cast<TOut*>(p.out) + (size_t)row * p.ldo + col, v);
    Vec<TOut, NV>::st_nt(reinterpret_cast<TOut*>(p.out2) + (size_t)row * p.ldo2 + col, a);
  } else if constexpr (EPI == EPI_RES_F32) {
    float b[NV], r[NV];
    Vec<float, NV>::ld(p.bias + col, b);
    Vec<float, NV>::ld(p.res + (size_t)row * p.ldr + col, r);
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = r[j] + (v[j] + b[j]);
    Vec<float, NV>::st_nt(reinterpret_cast<float*>(p.out) + (size_t)row * p.ldo + col, v);
  } else if constexpr (EPI == EPI_QKV) {
    const int dm = p.heads * p.head_dim;
    const int which = col / dm;
    const int hc = col - which * dm;
    const int h = hc / p.head_dim;
    const int d = hc - h * p.head_dim;
    // row / tokens without the ~35-instruction integer division (16 of them per thread and tile otherwise): float
    // reciprocal estimate, then one exact correction step (rows < 2^22)
    int b = (int)((float)row * __builtin_amdgcn_rcpf((float)p.tokens));
    int n = row - b * p.tokens;
    if (n < 0) { b -= 1; n += p.tokens; }
    else if (n >= p.tokens) { b += 1; n -= p.tokens; }
    float bb[NV];
    Vec<float, NV>::ld(p.bias + col, bb);
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] += bb[j];
    TOut* dst = reinterpret_cast<TOut*>(p.out) +
                ((((size_t)which * p.batch + b) * p.heads + h) * p.tokens + n) * p.head_dim_pad + d;
    Vec<TOut, NV>::st_nt(dst, v);
    if (d + NV == p.head_dim) {
      // zero the pad columns [head_dim, head_dim_pad): head_dim % 8 == 0 and head_dim_pad = round_up(head_dim, 16), so the
      // pad is 0 or 8 elements, i.e. whole NV-chunks: vector stores (scalar 2-byte stores here cost the QKV GEMM ~15 %)
      float z[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) z[j] = 0.f;
      for (int j = p.head_dim; j + NV <= p.head_dim_pad; j += NV) Vec<TOut, NV>::st_nt(dst + (j - d), z);
    }
  } else if constexpr (EPI == EPI_GELU_BWD) {
    float u[NV];
    Vec<TOut, NV>::ld(reinterpret_cast<const TOut*>(p.aux) + (size_t)row * p.ldaux + col, u);
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] *= p.gelu_grad_form ? u[j] : gelu_tanh_grad(u[j]);
    Vec<TOut, NV>::st_nt(reinterpret_cast<TOut*>(p.out) + (size_t)row * p.ldo + col, v);
  } else if constexpr (EPI == EPI_POS_F32) {
    float b[NV], e[NV];
    Vec<float, NV>::ld(p.bias + col, b);
    Vec<float, NV>::ld(p.pos + (size_t)(row % p.pos_rows) * N + col, e);
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = (v[j] + b[j]) + e[j];
    Vec<float, NV>::st(reinterpret_cast<float*>(p.out) + (size_t)row * p.ldo + col, v);
  } else {  // EPI_F32
    float* dst = reinterpret_cast<float*>(p.out) + (size_t)row * p.ldo + col;
    const int nv = (col + NV <= N) ? NV : (N - col);
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] *= p.alpha;
    if (p.bias) {
      for (int j = 0; j < nv; ++j) v[j] += p.bias[col + j];
    }
    if (p.atomic) {
      for (int j = 0; j < nv; ++j) atomicAdd(dst + j, v[j]);
    } else if (nv == NV && (p.ldo % NV) == 0) {
      if (p.accumulate) {
        float o[NV];
        Vec<float, NV>::ld(dst, o);
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] += o[j];
      }
      Vec<float, NV>::st(dst, v);
    } else {
      for (int j = 0; j < nv; ++j) dst[j] = p.accumulate ? dst[j] + v[j] : v[j];
    }
  }
}

}  // namespace sgl

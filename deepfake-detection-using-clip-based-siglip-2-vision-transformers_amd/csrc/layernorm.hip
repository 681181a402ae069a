// LayerNorm forward / backward for the fp32 residual stream (gfx950).
// Reference math: nn.LayerNorm(eps=1e-6), TF:models/siglip/modeling_siglip.py:329,331,567,630.
// HBM-bound: one wave (64 lanes) per row, the row lives in registers (float4 per lane), fp32 statistics by
// wave-64 shuffles, output written in the GEMM operand dtype (bf16 or fp32).
// Algorithmic bytes per row: fwd  D*4 (x) + D*sizeof(T) (y);  bwd  D*(4 + sizeof(T) + 4 [+4 dres] + sizeof(T)).
#include "common.hip.h"
#include "kernels.h"

namespace sgl {

constexpr int LN_MAXV_MAX = 8;  // float4 per lane -> D <= 2048 (instantiated for 2, 5, 8 to keep occupancy)

// Forward: each wave walks rows with stride gridDim*4 and keeps the NEXT row's loads in flight while it reduces and
// writes the current one (one wave per row with a fresh workgroup per 4 rows was dispatch/latency bound: 3.6 TB/s;
// this form measures 4.7 TB/s at M = 46656, D = 1152).  gamma/beta are re-read per row (cache hits) so that the
// kernel stays under 80 VGPRs.
template <typename TOut, int LN_MAXV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TOut* __restrict__ y, int ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M, int D,
                                                     float eps) {
  const int lane = lane_id();
  const int nv = D >> 2;
  const int stride = gridDim.x * 4;
  int row = blockIdx.x * 4 + wave_id();
  if (row >= M) return;
  f32x4 v[LN_MAXV], nx[LN_MAXV];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    v[i] = (c < nv) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + (size_t)row * D) + c) : z;
    nx[i] = z;
  }
  const float invD = 1.0f / (float)D;
  for (; row < M; row += stride) {
    const int nrow = row + stride;
    if (nrow < M) {
#pragma unroll
      for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) nx[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + (size_t)nrow * D) + c);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);  // lanes past nv hold zeros
    const float mu = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = v[i][j] - mu;
          q += d * d;
        }
      }
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) * invD + eps);
    TOut* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[c];
        const f32x4 b = reinterpret_cast<const f32x4*>(beta)[c];
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mu) * rs * g[j] + b[j];
        Vec<TOut, 4>::st_nt(yr + c * 4, o);
      }
    }
    if (lane == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) v[i] = nx[i];
  }
}

template <typename TOut>
static hipError_t ln_fwd_launch(const float* x, const float* gamma, const float* beta, TOut* y, int ldy, float* mean,
                                float* rstd, int M, int D, float eps, hipStream_t s) {
  int blocks = (M + 3) / 4;
  // each wave streams its rows (stride = 4 * gridDim) with a one-row prefetch.  Grid size, measured at M = 93 312 (round 3):
  // 2 048 workgroups (8 per CU, but only 6 are resident at < 80 VGPRs: 1.33 rounds) 135 us = 4.8 TB/s; 1 536 (exactly one
  // resident round) 115 us; >= 12 288 (a few rows per wave, the dispatcher balances the tail) 111 us = 5.8 TB/s
  if (blocks > 16384) blocks = 16384;
  dim3 grid(blocks), block(256);
  if (D <= 512)
    hipLaunchKernelGGL((ln_fwd_kernel<TOut, 2>), grid, block, 0, s, x, gamma, beta, y, ldy, mean, rstd, M, D, eps);
  else if (D <= 1280)
    hipLaunchKernelGGL((ln_fwd_kernel<TOut, 5>), grid, block, 0, s, x, gamma, beta, y, ldy, mean, rstd, M, D, eps);
  else
    hipLaunchKernelGGL((ln_fwd_kernel<TOut, 8>), grid, block, 0, s, x, gamma, beta, y, ldy, mean, rstd, M, D, eps);
  return hipGetLastError();
}

hipError_t layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_dtype, int ldy,
                         float* mean, float* rstd, int M, int D, float eps, hipStream_t s) {
  if (D % 4 || D > 64 * 4 * LN_MAXV_MAX || ldy % 4) return hipErrorInvalidValue;
  if (M == 0) return hipSuccess;
  if (y_dtype == DT_BF16) return ln_fwd_launch<bf16>(x, gamma, beta, (bf16*)y, ldy, mean, rstd, M, D, eps, s);
  return ln_fwd_launch<float>(x, gamma, beta, (float*)y, ldy, mean, rstd, M, D, eps, s);
}

// Backward.  dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma;  optional "+ dres" fuses the
// residual-branch gradient; optional low-precision copy of dx feeds the next backward GEMM's A operand.
// dgamma/dbeta/colsum(dx): per-lane register partials over the block's rows -> LDS -> partial[block][3D].
template <typename TDy, typename TLp, int LN_MAXV>
__global__ __launch_bounds__(256, (LN_MAXV <= 5 ? 3 : 1)) void ln_bwd_kernel(const TDy* __restrict__ dy, int lddy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres,
                                                     float* __restrict__ dx, TLp* __restrict__ dx_lp,
                                                     float* __restrict__ partial, int M, int D) {
  extern __shared__ __attribute__((aligned(16))) float ln_smem[];  // [4][D] fold buffer, then gamma[D]
  const int lane = lane_id(), w = wave_id();
  const int nv = D >> 2;
  // gamma lives in LDS, not in 20 registers, and a row's x / dy stay RAW in registers between the two passes (dy packed as
  // loaded): that frees the registers to request the row's residual-gradient chunk together with x and dy, so a row
  // exposes ONE memory round trip instead of two (the second pass used to start with the dres load) at the same 3 waves/SIMD.
  float* gsm = ln_smem + (size_t)4 * D;
  for (int j = threadIdx.x; j < D; j += 256) gsm[j] = gamma[j];
  __syncthreads();
  using DyVec = __attribute__((ext_vector_type(4))) TDy;
  f32x4 dg[LN_MAXV], db[LN_MAXV], ds[LN_MAXV];  // ds: column sums of the dx written
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    dg[i] = z;
    db[i] = z;
    ds[i] = z;
  }
  const float invD = 1.0f / (float)D;
  for (int row = blockIdx.x * 4 + w; row < M; row += gridDim.x * 4) {
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * D);
    const TDy* dyr = dy + (size_t)row * lddy;
    const f32x4* rr = dres ? reinterpret_cast<const f32x4*>(dres + (size_t)row * D) : nullptr;
    f32x4 xv[LN_MAXV], rv[LN_MAXV];
    DyVec dv[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        xv[i] = __builtin_nontemporal_load(xr + c);
        dv[i] = __builtin_nontemporal_load(reinterpret_cast<const DyVec*>(dyr + c * 4));
        if (rr) rv[i] = __builtin_nontemporal_load(rr + c);
      }
    }
    const float mu = mean[row], rs = rstd[row];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        const f32x4 gm = reinterpret_cast<const f32x4*>(gsm)[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = (float)dv[i][j];
          const float h = (xv[i][j] - mu) * rs;
          const float g = d * gm[j];
          c1 += g;
          c2 += g * h;
          dg[i][j] += d * h;
          db[i][j] += d;
        }
      }
    }
    c1 = wave_sum(c1) * invD;
    c2 = wave_sum(c2) * invD;
    float* dxr = dx + (size_t)row * D;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        const f32x4 gm = reinterpret_cast<const f32x4*>(gsm)[c];
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float h = (xv[i][j] - mu) * rs;           // same operations as the first pass: bit-identical h, g
          const float g = (float)dv[i][j] * gm[j];
          o[j] = rs * (g - c1 - h * c2);
        }
        if (rr) {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] += rv[i][j];
        }
        Vec<float, 4>::st_nt(dxr + c * 4, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) ds[i][j] += o[j];
        if (dx_lp) Vec<TLp, 4>::st_nt(dx_lp + (size_t)row * D + c * 4, o);
      }
    }
  }
  if (!partial) return;
  // cross-wave fold through one [4][D] LDS buffer, reused for the three vectors (keeps LDS at 4*D*4 bytes so that
  // occupancy is set by registers, not LDS)
  float* mine = ln_smem + (size_t)w * D;
  float* out = partial + (size_t)blockIdx.x * 3 * D;
#pragma unroll
  for (int ph = 0; ph < 3; ++ph) {
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) reinterpret_cast<f32x4*>(mine)[c] = (ph == 0) ? dg[i] : ((ph == 1) ? db[i] : ds[i]);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < D; j += 256)
      out[ph * D + j] = (ln_smem[j] + ln_smem[D + j]) + (ln_smem[2 * D + j] + ln_smem[3 * D + j]);
    __syncthreads();
  }
}

int layernorm_bwd_blocks(int M) {
  int b = (M + 3) / 4;
  return b < 1 ? 1 : (b > 768 ? 768 : b);
}

template <typename TDy, typename TLp>
static hipError_t ln_bwd_launch2(const void* dy, int lddy, const float* x, const float* mean, const float* rstd,
                                 const float* gamma, const float* dres, float* dx, void* dx_lp, float* partial,
                                 int nblk, int M, int D, hipStream_t s) {
  dim3 grid(nblk), block(256);
  const size_t smem = (size_t)5 * D * sizeof(float);   // [4][D] fold buffer + gamma[D]
#define SGL_LNB(V)                                                                                                  \
  hipLaunchKernelGGL((ln_bwd_kernel<TDy, TLp, V>), grid, block, smem, s, (const TDy*)dy, lddy, x, mean, rstd, gamma, \
                     dres, dx, (TLp*)dx_lp, partial, M, D)
  if (D <= 512)
    SGL_LNB(2);
  else if (D <= 1280)
    SGL_LNB(5);
  else
    SGL_LNB(8);
#undef SGL_LNB
  return hipGetLastError();
}

template <typename TDy>
static hipError_t ln_bwd_launch(const void* dy, int lddy, const float* x, const float* mean, const float* rstd,
                                const float* gamma, const float* dres, float* dx, void* dx_lp, int lp_dtype,
                                float* partial, int nblk, int M, int D, hipStream_t s) {
  if (lp_dtype == DT_BF16)
    return ln_bwd_launch2<TDy, bf16>(dy, lddy, x, mean, rstd, gamma, dres, dx, dx_lp, partial, nblk, M, D, s);
  return ln_bwd_launch2<TDy, float>(dy, lddy, x, mean, rstd, gamma, dres, dx, dx_lp, partial, nblk, M, D, s);
}

hipError_t layernorm_bwd(const void* dy, int dy_dtype, int lddy, const float* x, const float* mean,
                         const float* rstd, const float* gamma, const float* dres, float* dx, void* dx_lp,
                         int lp_dtype, float* partial, int nblk, int M, int D, hipStream_t s) {
  if (D % 4 || D > 64 * 4 * LN_MAXV_MAX || lddy % 4) return hipErrorInvalidValue;
  if (M == 0) return hipSuccess;
  if (dy_dtype == DT_BF16)
    return ln_bwd_launch<bf16>(dy, lddy, x, mean, rstd, gamma, dres, dx, dx_lp, lp_dtype, partial, nblk, M, D, s);
  return ln_bwd_launch<float>(dy, lddy, x, mean, rstd, gamma, dres, dx, dx_lp, lp_dtype, partial, nblk, M, D, s);
}

// out[j] (+)= sum_b partial[b*stride + j]: 32 columns x 8 row-groups per block, LDS tree across the groups
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int nblk, int stride,
                                                              float* __restrict__ out, int n, int accumulate) {
  __shared__ float red[8][33];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + c;
  float s0 = 0.f, s1 = 0.f;
  if (j < n) {
    int b = g;
    for (; b + 8 < nblk; b += 16) {
      s0 += partial[(size_t)b * stride + j];
      s1 += partial[(size_t)(b + 8) * stride + j];
    }
    if (b < nblk) s0 += partial[(size_t)b * stride + j];
  }
  red[g][c] = s0 + s1;
  __syncthreads();
  if (g == 0 && j < n) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) r += red[k][c];
    out[j] = accumulate ? out[j] + r : r;
  }
}

// three outputs in one launch: out_k[j] (+)= sum_b partial[b*stride + k*n + j]  (null outputs are skipped)
__global__ __launch_bounds__(256) void reduce_partials3_kernel(const float* __restrict__ partial, int nblk, int stride,
                                                               float* __restrict__ o0, float* __restrict__ o1,
                                                               float* __restrict__ o2, int n, int a0, int a1, int a2) {
  __shared__ float red[8][33];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + c;
  const int which = j / n, jj = j - which * n;
  float* o = which == 0 ? o0 : (which == 1 ? o1 : o2);
  const int acc = which == 0 ? a0 : (which == 1 ? a1 : a2);
  const bool live = (j < 3 * n) && o;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (live) {
    int b = g;
    for (; b + 24 < nblk; b += 32) {
      s0 += partial[(size_t)b * stride + j];
      s1 += partial[(size_t)(b + 8) * stride + j];
      s2 += partial[(size_t)(b + 16) * stride + j];
      s3 += partial[(size_t)(b + 24) * stride + j];
    }
    for (; b < nblk; b += 8) s0 += partial[(size_t)b * stride + j];
  }
  red[g][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && live) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) r += red[k][c];
    o[jj] = acc ? o[jj] + r : r;
  }
}

hipError_t reduce_partials3(const float* partial, int nblk, int stride, float* o0, float* o1, float* o2, int n, int a0,
                            int a1, int a2, hipStream_t s) {
  if (n == 0 || (!o0 && !o1 && !o2)) return hipSuccess;
  hipLaunchKernelGGL(reduce_partials3_kernel, dim3((3 * n + 31) / 32), dim3(256), 0, s, partial, nblk, stride, o0, o1,
                     o2, n, a0, a1, a2);
  return hipGetLastError();
}

hipError_t reduce_partials(const float* partial, int nblk, int stride, float* out, int n, int accumulate,
                           hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 31) / 32), dim3(256), 0, s, partial, nblk, stride, out, n,
                     accumulate);
  return hipGetLastError();
}

}  // namespace sgl

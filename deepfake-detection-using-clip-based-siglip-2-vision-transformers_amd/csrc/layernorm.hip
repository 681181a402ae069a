// LayerNorm forward / backward for the fp32 residual stream (gfx950).
// Reference math: nn.LayerNorm(eps=1e-6), TF:models/siglip/modeling_siglip.py:329,331,567,630.
// HBM-bound: one wave (64 lanes) per row, the row lives in registers (float4 per lane), fp32 statistics by
// wave-64 shuffles, output written in the GEMM operand dtype (bf16 or fp32).
// Algorithmic bytes per row: fwd  D*4 (x) + D*sizeof(T) (y);  bwd  D*(4 + sizeof(T) + 4 [+4 dres] + sizeof(T)).
#include "common.cuh"
#include "kernels.h"

namespace sgl {

constexpr int LN_MAXV = 8;  // float4 per lane -> D <= 2048

template <typename TOut>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TOut* __restrict__ y, int ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M, int D,
                                                     float eps) {
  const int row = blockIdx.x * 4 + wave_id();
  if (row >= M) return;
  const int lane = lane_id();
  const int nv = D >> 2;
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * D);
  f32x4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      v[i] = xr[c];
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
  }
  const float mu = wave_sum(s) / (float)D;
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
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
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
      Vec<TOut, 4>::st(yr + c * 4, o);
    }
  }
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

hipError_t layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_dtype, int ldy,
                         float* mean, float* rstd, int M, int D, float eps, hipStream_t s) {
  if (D % 4 || D > 64 * 4 * LN_MAXV || ldy % 4) return hipErrorInvalidValue;
  if (M == 0) return hipSuccess;
  dim3 grid((M + 3) / 4), block(256);
  if (y_dtype == DT_BF16)
    hipLaunchKernelGGL(ln_fwd_kernel<bf16>, grid, block, 0, s, x, gamma, beta, (bf16*)y, ldy, mean, rstd, M, D, eps);
  else
    hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, block, 0, s, x, gamma, beta, (float*)y, ldy, mean, rstd, M, D, eps);
  return hipGetLastError();
}

// Backward.  dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma;  optional "+ dres" fuses the
// residual-branch gradient; optional low-precision copy of dx feeds the next backward GEMM's A operand.
// dgamma/dbeta: per-lane register partials over the block's rows -> LDS -> partial[block][2D].
template <typename TDy, typename TLp>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDy* __restrict__ dy, int lddy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres,
                                                     float* __restrict__ dx, TLp* __restrict__ dx_lp,
                                                     float* __restrict__ partial, int M, int D) {
  extern __shared__ __attribute__((aligned(16))) float ln_smem[];  // [4][2][D]
  const int lane = lane_id(), w = wave_id();
  const int nv = D >> 2;
  f32x4 gam[LN_MAXV], dg[LN_MAXV], db[LN_MAXV];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    dg[i] = z;
    db[i] = z;
    gam[i] = (c < nv) ? reinterpret_cast<const f32x4*>(gamma)[c] : z;
  }
  const float invD = 1.0f / (float)D;
  for (int row = blockIdx.x * 4 + w; row < M; row += gridDim.x * 4) {
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * D);
    const TDy* dyr = dy + (size_t)row * lddy;
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[LN_MAXV], gy[LN_MAXV];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        const f32x4 xv = xr[c];
        float d[4];
        Vec<TDy, 4>::ld(dyr + c * 4, d);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float h = (xv[j] - mu) * rs;
          const float g = d[j] * gam[i][j];
          xh[i][j] = h;
          gy[i][j] = g;
          c1 += g;
          c2 += g * h;
          dg[i][j] += d[j] * h;
          db[i][j] += d[j];
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
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rs * (gy[i][j] - c1 - xh[i][j] * c2);
        if (dres) {
          const f32x4 r = reinterpret_cast<const f32x4*>(dres + (size_t)row * D)[c];
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] += r[j];
        }
        Vec<float, 4>::st(dxr + c * 4, o);
        if (dx_lp) Vec<TLp, 4>::st(dx_lp + (size_t)row * D + c * 4, o);
      }
    }
  }
  if (!partial) return;
  float* mine = ln_smem + (size_t)w * 2 * D;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      reinterpret_cast<f32x4*>(mine)[c] = dg[i];
      reinterpret_cast<f32x4*>(mine + D)[c] = db[i];
    }
  }
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * 2 * D;
  for (int j = threadIdx.x; j < 2 * D; j += 256)
    out[j] = (ln_smem[j] + ln_smem[2 * D + j]) + (ln_smem[4 * D + j] + ln_smem[6 * D + j]);
}

int layernorm_bwd_blocks(int M) {
  int b = (M + 3) / 4;
  return b < 1 ? 1 : (b > 1024 ? 1024 : b);
}

template <typename TDy>
static hipError_t ln_bwd_launch(const void* dy, int lddy, const float* x, const float* mean, const float* rstd,
                                const float* gamma, const float* dres, float* dx, void* dx_lp, int lp_dtype,
                                float* partial, int nblk, int M, int D, hipStream_t s) {
  dim3 grid(nblk), block(256);
  size_t smem = (size_t)4 * 2 * D * sizeof(float);
  if (lp_dtype == DT_BF16)
    hipLaunchKernelGGL((ln_bwd_kernel<TDy, bf16>), grid, block, smem, s, (const TDy*)dy, lddy, x, mean, rstd, gamma,
                       dres, dx, (bf16*)dx_lp, partial, M, D);
  else
    hipLaunchKernelGGL((ln_bwd_kernel<TDy, float>), grid, block, smem, s, (const TDy*)dy, lddy, x, mean, rstd, gamma,
                       dres, dx, (float*)dx_lp, partial, M, D);
  return hipGetLastError();
}

hipError_t layernorm_bwd(const void* dy, int dy_dtype, int lddy, const float* x, const float* mean,
                         const float* rstd, const float* gamma, const float* dres, float* dx, void* dx_lp,
                         int lp_dtype, float* partial, int nblk, int M, int D, hipStream_t s) {
  if (D % 4 || D > 64 * 4 * LN_MAXV || lddy % 4) return hipErrorInvalidValue;
  if (M == 0) return hipSuccess;
  if (dy_dtype == DT_BF16)
    return ln_bwd_launch<bf16>(dy, lddy, x, mean, rstd, gamma, dres, dx, dx_lp, lp_dtype, partial, nblk, M, D, s);
  return ln_bwd_launch<float>(dy, lddy, x, mean, rstd, gamma, dres, dx, dx_lp, lp_dtype, partial, nblk, M, D, s);
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int nblk, int stride,
                                                              float* __restrict__ out, int n, int accumulate) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  float s0 = 0.f, s1 = 0.f;
  int b = 0;
  for (; b + 1 < nblk; b += 2) {
    s0 += partial[(size_t)b * stride + j];
    s1 += partial[(size_t)(b + 1) * stride + j];
  }
  if (b < nblk) s0 += partial[(size_t)b * stride + j];
  const float r = s0 + s1;
  out[j] = accumulate ? out[j] + r : r;
}

hipError_t reduce_partials(const float* partial, int nblk, int stride, float* out, int n, int accumulate,
                           hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 255) / 256), dim3(256), 0, s, partial, nblk, stride, out, n,
                     accumulate);
  return hipGetLastError();
}

}  // namespace sgl

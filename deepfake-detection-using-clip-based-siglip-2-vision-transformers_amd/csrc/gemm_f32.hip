// Strict-fp32 generic strided GEMM:  C[m,n] = sum_k A(m,k) * B(n,k),  A(m,k) = A[m*sam + k*sak],
// B(n,k) = B[n*sbn + k*sbk].  One kernel covers the NT (forward), NN (dX) and TN (dW) forms by strides.
// It exists for the fp32-strict compute mode (parity against the fp32 CPU oracle through the same host
// orchestration and the same fused epilogues as the bf16 MFMA path); it is not a performance path.
// 64x64 tile, BK = 16, 256 threads, 4x4 outputs per thread, fp32 FMA accumulation in k order.
#include "common.hip.h"
#include "epilogue.hip.h"
#include "kernels.h"

namespace sgl {

template <int EPI, typename TOut>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, long sam, long sak,
                                                       const float* __restrict__ B, long sbn, long sbk, int M, int N,
                                                       int K, EpiParams p) {
  __shared__ float As[16][68];
  __shared__ float Bs[16][68];
  const int t = threadIdx.x;
  const int ty = t >> 4, tx = t & 15;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = t + q * 256;
      int mm, kk;
      if (sak == 1) { kk = idx & 15; mm = idx >> 4; } else { mm = idx & 63; kk = idx >> 6; }
      const int gm = m0 + mm, gk = k0 + kk;
      As[kk][mm] = (gm < M && gk < K) ? A[(long)gm * sam + (long)gk * sak] : 0.f;
      int nn, kb;
      if (sbk == 1) { kb = idx & 15; nn = idx >> 4; } else { nn = idx & 63; kb = idx >> 6; }
      const int gn = n0 + nn, gkb = k0 + kb;
      Bs[kb][nn] = (gn < N && gkb < K) ? B[(long)gn * sbn + (long)gkb * sbk] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&As[kk][ty * 4]);
      const f32x4 b = *reinterpret_cast<const f32x4*>(&Bs[kk][tx * 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
  const int col = n0 + tx * 4;
  if (col >= N) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + ty * 4 + i;
    if (row < M) epi_apply<EPI, TOut, 4>(p, row, col, N, acc[i]);
  }
}

template <int EPI, typename TOut>
static hipError_t launch_f32(const float* A, long sam, long sak, const float* B, long sbn, long sbk, int M, int N,
                             int K, const EpiParams& p, hipStream_t s) {
  dim3 grid((N + 63) / 64, (M + 63) / 64), block(256);
  hipLaunchKernelGGL((gemm_f32_kernel<EPI, TOut>), grid, block, 0, s, A, sam, sak, B, sbn, sbk, M, N, K, p);
  return hipGetLastError();
}

hipError_t gemm_f32_generic(const float* A, long sam, long sak, const float* B, long sbn, long sbk, int M, int N,
                            int K, int epi, int out_dtype, const EpiParams& p, hipStream_t s) {
  if (M == 0 || N == 0) return hipSuccess;
  if (epi != EPI_F32 && (N % 4)) return hipErrorInvalidValue;
#define SGL_CASE(E)                                                                               \
  case E:                                                                                         \
    return out_dtype == DT_BF16 ? launch_f32<E, bf16>(A, sam, sak, B, sbn, sbk, M, N, K, p, s)    \
                                : launch_f32<E, float>(A, sam, sak, B, sbn, sbk, M, N, K, p, s);
  switch (epi) {
    SGL_CASE(EPI_STORE)
    SGL_CASE(EPI_BIAS_GELU)
    SGL_CASE(EPI_QKV)
    SGL_CASE(EPI_GELU_BWD)
    case EPI_RES_F32: return launch_f32<EPI_RES_F32, float>(A, sam, sak, B, sbn, sbk, M, N, K, p, s);
    case EPI_POS_F32: return launch_f32<EPI_POS_F32, float>(A, sam, sak, B, sbn, sbk, M, N, K, p, s);
    case EPI_F32: return launch_f32<EPI_F32, float>(A, sam, sak, B, sbn, sbk, M, N, K, p, s);
  }
#undef SGL_CASE
  return hipErrorInvalidValue;
}

}  // namespace sgl

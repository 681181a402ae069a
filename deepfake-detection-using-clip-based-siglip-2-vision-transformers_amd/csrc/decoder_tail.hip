// SID mask-decoder tail, second half of SURVEY.md §8f row 1 (Siglip2sidafrozen.py:731-745 decoder tail, :174-181 loss):
//
//   gate_mul       y = sigmoid(g) * x            the SE-style channel gate applied to the concatenated taps; one pass
//                  (dg, dx) from dy              instead of sigmoid + mul (and, backward, four elementwise kernels).
//                                                (B*N, E*K) bf16 at the default decoder: 525 MB per tensor at B = 64.
//   seg_loss_fwd   per-image partial sums of the BCE-with-logits and Dice terms of `bce_dice_loss` taken DIRECTLY from the
//                  low-resolution (B,1,g,g) logit map: every output pixel's logit is the bilinear (align_corners=False)
//                  interpolation of four low-res logits, evaluated in registers, so the (B,1,S,S) up-sampled logits are
//                  never written or read (147 k pixels per image; in the reference also the (B,512,S,S) features).
//   seg_loss_bwd   d loss / d low-res logits: the transposed bilinear interpolation of (p - t)/n + dice', gathered per
//                  low-res pixel in a fixed order (no atomics: bitwise reproducible).
//
// Loss definition being reproduced (heads.bce_dice_loss == Siglip2sidafrozen.py:174-181, on the images that carry a mask):
//   bce  = mean over all pixels of the selected images of  max(z,0) - z t + log(1 + exp(-|z|))
//   dice = 1 - mean_b( 2 sum(p t) / (sum p + sum t + eps) ),  p = sigmoid(z)
//   loss = bce_w * bce + dice_w * dice
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hip.h"
#include "kernels.h"
#include "siglip_hip.h"

namespace sgl {

__device__ __forceinline__ float sigmoidf_fast(float z) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-z * SGL_LOG2E));
}

template <typename T, int NV>
__global__ __launch_bounds__(256) void gate_mul_kernel(const T* __restrict__ g, const T* __restrict__ x,
                                                       T* __restrict__ y, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    float a[NV], b[NV];
    Vec<T, NV>::ld(g + i * NV, a);
    Vec<T, NV>::ld(x + i * NV, b);
#pragma unroll
    for (int j = 0; j < NV; ++j) a[j] = sigmoidf_fast(a[j]) * b[j];
    Vec<T, NV>::st(y + i * NV, a);
  }
}

template <typename T, int NV>
__global__ __launch_bounds__(256) void gate_mul_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ g,
                                                           const T* __restrict__ x, T* __restrict__ dg,
                                                           T* __restrict__ dx, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    float d[NV], a[NV], b[NV], og[NV], ox[NV];
    Vec<T, NV>::ld(dy + i * NV, d);
    Vec<T, NV>::ld(g + i * NV, a);
    Vec<T, NV>::ld(x + i * NV, b);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const float s = sigmoidf_fast(a[j]);
      ox[j] = d[j] * s;
      og[j] = d[j] * b[j] * (s * (1.0f - s));
    }
    if (dg) Vec<T, NV>::st(dg + i * NV, og);
    if (dx) Vec<T, NV>::st(dx + i * NV, ox);
  }
}

// bilinear source position of output index i (PyTorch upsample_bilinear2d, align_corners = False)
__device__ __forceinline__ void bil_src(int i, float scale, int g, int& i0, int& i1, float& w1) {
  float s = scale * ((float)i + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > g - 1) i0 = g - 1;
  i1 = i0 + 1 < g ? i0 + 1 : g - 1;
  w1 = s - (float)i0;
}

constexpr int SEG_ROWS = 8;   // output rows per block in the forward kernel

// partial[b][chunk][4] = { sum bce, sum p*t, sum p, sum t } over SEG_ROWS output rows of image b
__global__ __launch_bounds__(256) void seg_loss_fwd_kernel(const float* __restrict__ lr /*[B][g][g]*/,
                                                           const float* __restrict__ tgt /*[B][S][S]*/,
                                                           float* __restrict__ partial, int g, int S, float scale,
                                                           int chunks) {
  __shared__ float red[4][4];
  const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
  const float* L = lr + (size_t)b * g * g;
  const float* T = tgt + (size_t)b * S * S;
  float s_bce = 0.f, s_pt = 0.f, s_p = 0.f, s_t = 0.f;
  const int r_begin = ch * SEG_ROWS, r_end = (r_begin + SEG_ROWS < S) ? r_begin + SEG_ROWS : S;
  for (int i = r_begin; i < r_end; ++i) {
    int y0, y1;
    float wy;
    bil_src(i, scale, g, y0, y1, wy);
    for (int j = threadIdx.x; j < S; j += 256) {
      int x0, x1;
      float wx;
      bil_src(j, scale, g, x0, x1, wx);
      const float top = L[y0 * g + x0] * (1.f - wx) + L[y0 * g + x1] * wx;
      const float bot = L[y1 * g + x0] * (1.f - wx) + L[y1 * g + x1] * wx;
      const float z = top * (1.f - wy) + bot * wy;
      const float t = T[(size_t)i * S + j];
      const float e = __expf(-fabsf(z));
      s_bce += fmaxf(z, 0.f) - z * t + log1pf(e);
      const float p = z >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
      s_pt += p * t;
      s_p += p;
      s_t += t;
    }
  }
  s_bce = wave_sum(s_bce); s_pt = wave_sum(s_pt); s_p = wave_sum(s_p); s_t = wave_sum(s_t);
  if (lane_id() == 0) {
    red[wave_id()][0] = s_bce; red[wave_id()][1] = s_pt; red[wave_id()][2] = s_p; red[wave_id()][3] = s_t;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    partial[((size_t)b * chunks + ch) * 4 + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
  }
}

// One block per (image b, low-res row gy).  sums[b] = {sum bce, sum p t, sum p, sum t} of image b (forward output, folded);
// coef[b] = {c_bce, c_dice}: d loss / d z(pixel) = c_bce * (p - t) + c_dice * p (1-p) * ((2 t D - 2 I) / D^2) with
// I = sum p t, D = sum p + sum t + eps.  (c_bce = upstream * bce_w / n_pixels_total, c_dice = -upstream * dice_w / n_imgs;
// both 0 for images without a mask.)  dlr[b][gy][gx] = sum over output pixels of dz * bilinear weight: each thread owns
// output columns, sums them over the rows of the band in order, then one thread per gx folds its columns in order.
__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const float* __restrict__ lr, const float* __restrict__ tgt,
                                                           const float* __restrict__ sums,
                                                           const float* __restrict__ coef, float* __restrict__ dlr,
                                                           int g, int S, float scale, float eps) {
  extern __shared__ float colacc[];   // [S][2]: contribution of output column j to (gy, x0(j)) and (gy, x1(j))
  const int b = blockIdx.x / g, gy = blockIdx.x - b * g;
  const float* L = lr + (size_t)b * g * g;
  const float* T = tgt + (size_t)b * S * S;
  const float c_bce = coef[2 * b], c_dice = coef[2 * b + 1];
  const float I = sums[4 * b + 1], D = sums[4 * b + 2] + sums[4 * b + 3] + eps;
  const float invD2 = 1.0f / (D * D);
  // output rows whose bilinear support can include gy (conservative range; exact test inside)
  const float inv = 1.0f / scale;
  int r_lo = (int)floorf(((float)gy - 1.0f + 0.5f) * inv - 0.5f) - 1;
  int r_hi = (int)ceilf(((float)gy + 1.0f + 0.5f) * inv - 0.5f) + 1;
  if (r_lo < 0) r_lo = 0;
  if (r_hi > S - 1) r_hi = S - 1;
  for (int j = threadIdx.x; j < S; j += 256) {
    int x0, x1;
    float wx;
    bil_src(j, scale, g, x0, x1, wx);
    float a0 = 0.f, a1 = 0.f;
    for (int i = r_lo; i <= r_hi; ++i) {
      int y0, y1;
      float wy;
      bil_src(i, scale, g, y0, y1, wy);
      const float wrow = (y0 == gy ? (1.f - wy) : 0.f) + (y1 == gy ? wy : 0.f);
      if (wrow == 0.f) continue;
      const float top = L[y0 * g + x0] * (1.f - wx) + L[y0 * g + x1] * wx;
      const float bot = L[y1 * g + x0] * (1.f - wx) + L[y1 * g + x1] * wx;
      const float z = top * (1.f - wy) + bot * wy;
      const float t = T[(size_t)i * S + j];
      const float e = __expf(-fabsf(z));
      const float p = z >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
      const float dz = c_bce * (p - t) + c_dice * (p * (1.f - p)) * ((2.f * t * D - 2.f * I) * invD2);
      a0 += dz * wrow * (1.f - wx);
      a1 += dz * wrow * wx;
    }
    colacc[2 * j] = a0;
    colacc[2 * j + 1] = a1;
  }
  __syncthreads();
  for (int gx = threadIdx.x; gx < g; gx += 256) {
    int c_lo = (int)floorf(((float)gx - 1.0f + 0.5f) * inv - 0.5f) - 1;
    int c_hi = (int)ceilf(((float)gx + 1.0f + 0.5f) * inv - 0.5f) + 1;
    if (c_lo < 0) c_lo = 0;
    if (c_hi > S - 1) c_hi = S - 1;
    float acc = 0.f;
    for (int j = c_lo; j <= c_hi; ++j) {
      int x0, x1;
      float wx;
      bil_src(j, scale, g, x0, x1, wx);
      // x0 == x1 at the right border: both halves belong to the same low-res pixel
      if (x0 == gx) acc += colacc[2 * j];
      if (x1 == gx) acc += colacc[2 * j + 1];
    }
    dlr[((size_t)b * g + gy) * g + gx] = acc;
  }
}

}  // namespace sgl

extern "C" {

int sgl_op_gate_mul(const void* g, const void* x, void* y, size_t n, int dtype, sgl_stream stream) {
  if (!g || !x || !y) return SGL_ERR_NULL;
  if (dtype != SGL_DTYPE_BF16 && dtype != SGL_DTYPE_F32) return SGL_ERR_UNSUPPORTED;
  const int nv = dtype == SGL_DTYPE_BF16 ? 8 : 4;
  if (n % nv || ((((uintptr_t)g) | ((uintptr_t)x) | ((uintptr_t)y)) & 15)) return SGL_ERR_BAD_SHAPE;
  const size_t nvec = n / nv;
  if (nvec == 0) return SGL_OK;
  const int blocks = (int)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SGL_DTYPE_BF16)
    hipLaunchKernelGGL((sgl::gate_mul_kernel<sgl::bf16, 8>), dim3(blocks), dim3(256), 0, s, (const sgl::bf16*)g,
                       (const sgl::bf16*)x, (sgl::bf16*)y, nvec);
  else
    hipLaunchKernelGGL((sgl::gate_mul_kernel<float, 4>), dim3(blocks), dim3(256), 0, s, (const float*)g, (const float*)x,
                       (float*)y, nvec);
  return hipGetLastError() == hipSuccess ? SGL_OK : SGL_ERR_HIP;
}

int sgl_op_gate_mul_bwd(const void* dy, const void* g, const void* x, void* dg, void* dx, size_t n, int dtype,
                        sgl_stream stream) {
  if (!dy || !g || !x) return SGL_ERR_NULL;
  if (dtype != SGL_DTYPE_BF16 && dtype != SGL_DTYPE_F32) return SGL_ERR_UNSUPPORTED;
  const int nv = dtype == SGL_DTYPE_BF16 ? 8 : 4;
  if (n % nv || ((((uintptr_t)dy) | ((uintptr_t)g) | ((uintptr_t)x) | ((uintptr_t)dg) | ((uintptr_t)dx)) & 15))
    return SGL_ERR_BAD_SHAPE;
  const size_t nvec = n / nv;
  if (nvec == 0) return SGL_OK;
  const int blocks = (int)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SGL_DTYPE_BF16)
    hipLaunchKernelGGL((sgl::gate_mul_bwd_kernel<sgl::bf16, 8>), dim3(blocks), dim3(256), 0, s, (const sgl::bf16*)dy,
                       (const sgl::bf16*)g, (const sgl::bf16*)x, (sgl::bf16*)dg, (sgl::bf16*)dx, nvec);
  else
    hipLaunchKernelGGL((sgl::gate_mul_bwd_kernel<float, 4>), dim3(blocks), dim3(256), 0, s, (const float*)dy,
                       (const float*)g, (const float*)x, (float*)dg, (float*)dx, nvec);
  return hipGetLastError() == hipSuccess ? SGL_OK : SGL_ERR_HIP;
}

int sgl_op_seg_loss_chunks(int S) { return S > 0 ? (S + sgl::SEG_ROWS - 1) / sgl::SEG_ROWS : 0; }

int sgl_op_seg_loss_fwd(const float* logits_lr, const float* targets, float* partial, int B, int g, int S,
                        sgl_stream stream) {
  if (!logits_lr || !targets || !partial) return SGL_ERR_NULL;
  if (B <= 0 || g <= 0 || S <= 0 || g > 4096 || S > 16384) return SGL_ERR_BAD_SHAPE;
  const int chunks = sgl_op_seg_loss_chunks(S);
  hipLaunchKernelGGL(sgl::seg_loss_fwd_kernel, dim3((unsigned)(B * chunks)), dim3(256), 0, (hipStream_t)stream, logits_lr,
                     targets, partial, g, S, (float)g / (float)S, chunks);
  return hipGetLastError() == hipSuccess ? SGL_OK : SGL_ERR_HIP;
}

int sgl_op_seg_loss_bwd(const float* logits_lr, const float* targets, const float* sums, const float* coef,
                        float* dlogits_lr, int B, int g, int S, float eps, sgl_stream stream) {
  if (!logits_lr || !targets || !sums || !coef || !dlogits_lr) return SGL_ERR_NULL;
  if (B <= 0 || g <= 0 || S <= 0 || g > 4096 || S > 8192) return SGL_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(sgl::seg_loss_bwd_kernel, dim3((unsigned)(B * g)), dim3(256), (size_t)S * 2 * sizeof(float),
                     (hipStream_t)stream, logits_lr, targets, sums, coef, dlogits_lr, g, S, (float)g / (float)S, eps);
  return hipGetLastError() == hipSuccess ? SGL_OK : SGL_ERR_HIP;
}

}  // extern "C"

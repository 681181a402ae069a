// SID mask-decoder tail: the depthwise 3x3 convolution of `SegFormerStrongDecoder`'s per-tap
// "smooth" block (Siglip2sidafrozen.py:713-718: nn.Conv2d(E, E, 3, padding=1, groups=E)) on the CHANNELS-LAST
// (B, gh, gw, E) layout the token-major decoder uses.  MIOpen only offers a naive fp32 NHWC solver for this shape
// (12-24 ms per call measured); as an HBM-bound stencil it is one read and one write of the tensor:
//   forward / data-gradient : y[b,y,x,e] = bias[e] + sum_{dy,dx} w[e][dy][dx] * x[b, y+dy-1, x+dx-1, e]   (zero padding)
//                             (the data gradient is the same stencil with the 3x3 taps flipped and no bias)
//   weight / bias gradient  : dw[e][dy][dx] = sum_{b,y,x} x[b, y+dy-1, x+dx-1, e] * dy_[b,y,x,e];  db[e] = sum dy_
//                             two deterministic stages: per-chunk partial sums, then a fixed-order fold.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.cuh"
#include "kernels.h"
#include "siglip_hip.h"

namespace sgl {

// one thread = one pixel x 4 consecutive channels
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, T* __restrict__ y, int B,
                                                        int gh, int gw, int E, int flip) {
  const int cv = E >> 2;
  const size_t total = (size_t)B * gh * gw * cv;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int c = (int)(idx % cv) * 4;
    const size_t pix = idx / cv;
    const int px = (int)(pix % gw);
    const int py = (int)((pix / gw) % gh);
    const size_t b = pix / ((size_t)gw * gh);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) Vec<float, 4>::ld(bias + c, acc);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = py + dy - 1;
      if (yy < 0 || yy >= gh) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = px + dx - 1;
        if (xx < 0 || xx >= gw) continue;
        float v[4];
        Vec<T, 4>::ld(x + (((b * gh + yy) * gw + xx) * (size_t)E + c), v);
        const int k = flip ? (8 - (dy * 3 + dx)) : (dy * 3 + dx);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(w[(c + j) * 9 + k], v[j], acc[j]);
      }
    }
    Vec<T, 4>::st(y + (pix * (size_t)E + c), acc);
  }
}

// stage 1 of the weight gradient: block `blockIdx.x` owns pixels [p0, p1); thread = channel quad x pixel lane;
// partial[blockIdx.x][10][E]  (rows 0..8: the nine taps, row 9: bias)
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy_,
                                                              float* __restrict__ partial, int B, int gh, int gw,
                                                              int E, int pix_per_block) {
  extern __shared__ float red[];  // [lanes][10][E]
  const int cv = E >> 2;
  const int lanes = 256 / cv;     // pixel lanes per block (E <= 1024, E % 4 == 0, cv divides 256: checked by the host)
  const int cq = threadIdx.x % cv, pl = threadIdx.x / cv;
  const int c = cq * 4;
  const size_t npix = (size_t)B * gh * gw;
  const size_t p0 = (size_t)blockIdx.x * pix_per_block;
  const size_t p1 = (p0 + pix_per_block < npix) ? p0 + pix_per_block : npix;
  float acc[10][4];
#pragma unroll
  for (int k = 0; k < 10; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[k][j] = 0.f;
  if (pl < lanes) {
    for (size_t pix = p0 + pl; pix < p1; pix += lanes) {
      const int px = (int)(pix % gw);
      const int py = (int)((pix / gw) % gh);
      const size_t b = pix / ((size_t)gw * gh);
      float g[4];
      Vec<T, 4>::ld(dy_ + (pix * (size_t)E + c), g);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[9][j] += g[j];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int yy = py + dy - 1;
        if (yy < 0 || yy >= gh) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int xx = px + dx - 1;
          if (xx < 0 || xx >= gw) continue;
          float v[4];
          Vec<T, 4>::ld(x + (((b * gh + yy) * gw + xx) * (size_t)E + c), v);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[dy * 3 + dx][j] = fmaf(v[j], g[j], acc[dy * 3 + dx][j]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[((size_t)pl * 10 + k) * E + c + j] = acc[k][j];
  }
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * 10 * E;
  for (int i = threadIdx.x; i < 10 * E; i += 256) {
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += red[(size_t)l * 10 * E + i];
    out[i] = s;
  }
}

// stage 2: dw[e][k] (+)= sum_blk partial[blk][k][e]; db[e] (+)= sum_blk partial[blk][9][e]
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_fold_kernel(const float* __restrict__ partial, int nblk, int E,
                                                                   float* __restrict__ dw, float* __restrict__ db,
                                                                   int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // over 10*E
  if (i >= 10 * E) return;
  const int k = i / E, e = i - k * E;
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += partial[(size_t)b * 10 * E + i];
  if (k < 9) {
    float* o = dw + (size_t)e * 9 + k;
    *o = accumulate ? *o + s : s;
  } else if (db) {
    db[e] = accumulate ? db[e] + s : s;
  }
}

static bool dw_shape_ok(int B, int gh, int gw, int E) {
  return B > 0 && gh > 0 && gw > 0 && E >= 4 && E <= 1024 && (E % 4) == 0 && (256 % (E / 4)) == 0;
}

}  // namespace sgl

extern "C" {

int sgl_op_dwconv3x3(const void* x, int dtype, const float* w, const float* bias, void* y, int B, int gh, int gw, int E,
                     int flip, sgl_stream stream) {
  if (!x || !w || !y) return SGL_ERR_NULL;
  if (!sgl::dw_shape_ok(B, gh, gw, E)) return SGL_ERR_BAD_SHAPE;
  const size_t total = (size_t)B * gh * gw * (E / 4);
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SGL_DTYPE_BF16)
    hipLaunchKernelGGL(sgl::dwconv3x3_kernel<sgl::bf16>, dim3(blocks), dim3(256), 0, s, (const sgl::bf16*)x, w, bias,
                       (sgl::bf16*)y, B, gh, gw, E, flip);
  else if (dtype == SGL_DTYPE_F32)
    hipLaunchKernelGGL(sgl::dwconv3x3_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, w, bias, (float*)y,
                       B, gh, gw, E, flip);
  else
    return SGL_ERR_UNSUPPORTED;
  return hipGetLastError() == hipSuccess ? SGL_OK : SGL_ERR_HIP;
}

size_t sgl_op_dwconv3x3_wgrad_scratch_bytes(int B, int gh, int gw, int E) {
  (void)B; (void)gh; (void)gw;
  return (size_t)512 * 10 * E * sizeof(float);
}

int sgl_op_dwconv3x3_wgrad(const void* x, const void* dy, int dtype, float* dw, float* dbias, int accumulate,
                           float* scratch, size_t scratch_bytes, int B, int gh, int gw, int E, sgl_stream stream) {
  if (!x || !dy || !dw || !scratch) return SGL_ERR_NULL;
  if (!sgl::dw_shape_ok(B, gh, gw, E)) return SGL_ERR_BAD_SHAPE;
  const size_t npix = (size_t)B * gh * gw;
  int nblk = (int)(npix < 512 ? npix : 512);
  const int ppb = (int)((npix + nblk - 1) / nblk);
  nblk = (int)((npix + ppb - 1) / ppb);
  if (scratch_bytes < (size_t)nblk * 10 * E * sizeof(float)) return SGL_ERR_WORKSPACE;
  const int lanes = 256 / (E / 4);
  const size_t smem = (size_t)lanes * 10 * E * sizeof(float);  // <= 40 KiB
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SGL_DTYPE_BF16)
    hipLaunchKernelGGL(sgl::dwconv3x3_wgrad_kernel<sgl::bf16>, dim3(nblk), dim3(256), smem, s, (const sgl::bf16*)x,
                       (const sgl::bf16*)dy, scratch, B, gh, gw, E, ppb);
  else if (dtype == SGL_DTYPE_F32)
    hipLaunchKernelGGL(sgl::dwconv3x3_wgrad_kernel<float>, dim3(nblk), dim3(256), smem, s, (const float*)x,
                       (const float*)dy, scratch, B, gh, gw, E, ppb);
  else
    return SGL_ERR_UNSUPPORTED;
  if (hipGetLastError() != hipSuccess) return SGL_ERR_HIP;
  hipLaunchKernelGGL(sgl::dwconv3x3_wgrad_fold_kernel, dim3((10 * E + 255) / 256), dim3(256), 0, s, scratch, nblk, E, dw,
                     dbias, accumulate);
  return hipGetLastError() == hipSuccess ? SGL_OK : SGL_ERR_HIP;
}

}  // extern "C"

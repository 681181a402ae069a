// SID mask-decoder tail: the depthwise 3x3 convolution of `SegFormerStrongDecoder`'s per-tap
// "smooth" block (Siglip2sidafrozen.py:713-718: nn.Conv2d(E, E, 3, padding=1, groups=E)) on the CHANNELS-LAST
// (B, gh, gw, E) layout the token-major decoder uses.  MIOpen only offers a naive fp32 NHWC solver for this shape
// (12-24 ms per call measured); as an HBM-bound stencil it is one read and one write of the tensor:
//   forward / data-gradient : y[b,y,x,e] = bias[e] + sum_{dy,dx} w[e][dy][dx] * x[b, y+dy-1, x+dx-1, e]   (zero padding)
//                             (the data gradient is the same stencil with the 3x3 taps flipped and no bias)
//   weight / bias gradient  : dw[e][dy][dx] = sum_{b,y,x} x[b, y+dy-1, x+dx-1, e] * dy_[b,y,x,e];  db[e] = sum dy_
//                             two deterministic stages: per-chunk partial sums, then a fixed-order fold.
// Weights cross the ABI TAP-MAJOR ([9][E], i.e. Conv2d.weight.view(E, 9).t()), gradients as [10][E] (nine tap rows and
// the bias row): the channels a thread owns are then one 16/32-byte vector per tap.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hip.h"
#include "kernels.h"
#include "siglip_hip.h"

namespace sgl {

// Weights arrive TAP-MAJOR, w9[k][e] (k = dy*3 + dx), so the NV channels a thread owns are one vector load per tap.
// one thread = one pixel x NV consecutive channels (NV = 8 for bf16, 4 for fp32: 16-byte accesses)
template <typename T, int NV>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const T* __restrict__ x, const float* __restrict__ w9,
                                                        const float* __restrict__ bias, T* __restrict__ y, int B,
                                                        int gh, int gw, int E, int flip) {
  const int cv = E / NV;
  const size_t total = (size_t)B * gh * gw * cv;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int c = (int)(idx % cv) * NV;
    const size_t pix = idx / cv;
    const int px = (int)(pix % gw);
    const int py = (int)((pix / gw) % gh);
    const size_t b = pix / ((size_t)gw * gh);
    float acc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] = 0.f;
    if (bias) Vec<float, NV>::ld(bias + c, acc);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = py + dy - 1;
      if (yy < 0 || yy >= gh) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = px + dx - 1;
        if (xx < 0 || xx >= gw) continue;
        float v[NV], wk[NV];
        Vec<T, NV>::ld(x + (((b * gh + yy) * gw + xx) * (size_t)E + c), v);
        const int k = flip ? (8 - (dy * 3 + dx)) : (dy * 3 + dx);
        Vec<float, NV>::ld(w9 + (size_t)k * E + c, wk);
#pragma unroll
        for (int j = 0; j < NV; ++j) acc[j] = fmaf(wk[j], v[j], acc[j]);
      }
    }
    Vec<T, NV>::st(y + (pix * (size_t)E + c), acc);
  }
}

// stage 1 of the weight gradient: block `blockIdx.x` owns pixels [p0, p1); thread = channel group x pixel lane;
// partial[blockIdx.x][10][E]  (rows 0..8: the nine taps, row 9: bias) -- the same [10][E] layout as the result
template <typename T, int NV>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy_,
                                                              float* __restrict__ partial, int B, int gh, int gw,
                                                              int E, int pix_per_block) {
  extern __shared__ float red[];  // [lanes][10][E]
  const int cv = E / NV;
  const int lanes = 256 / cv;     // pixel lanes per block (cv divides 256: checked by the host)
  const int cq = threadIdx.x % cv, pl = threadIdx.x / cv;
  const int c = cq * NV;
  const size_t npix = (size_t)B * gh * gw;
  const size_t p0 = (size_t)blockIdx.x * pix_per_block;
  const size_t p1 = (p0 + pix_per_block < npix) ? p0 + pix_per_block : npix;
  float acc[10][NV];
#pragma unroll
  for (int k = 0; k < 10; ++k)
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[k][j] = 0.f;
  for (size_t pix = p0 + pl; pix < p1; pix += lanes) {
    const int px = (int)(pix % gw);
    const int py = (int)((pix / gw) % gh);
    const size_t b = pix / ((size_t)gw * gh);
    float g[NV];
    Vec<T, NV>::ld(dy_ + (pix * (size_t)E + c), g);
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[9][j] += g[j];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = py + dy - 1;
      if (yy < 0 || yy >= gh) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = px + dx - 1;
        if (xx < 0 || xx >= gw) continue;
        float v[NV];
        Vec<T, NV>::ld(x + (((b * gh + yy) * gw + xx) * (size_t)E + c), v);
#pragma unroll
        for (int j = 0; j < NV; ++j) acc[dy * 3 + dx][j] = fmaf(v[j], g[j], acc[dy * 3 + dx][j]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 10; ++k)
#pragma unroll
    for (int j = 0; j < NV; ++j) red[((size_t)pl * 10 + k) * E + c + j] = acc[k][j];
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * 10 * E;
  for (int i = threadIdx.x; i < 10 * E; i += 256) {
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += red[(size_t)l * 10 * E + i];
    out[i] = s;
  }
}

static bool dw_shape_ok(int B, int gh, int gw, int E, int nv) {
  return B > 0 && gh > 0 && gw > 0 && E >= nv && E <= 1024 && (E % nv) == 0 && (256 % (E / nv)) == 0;
}

}  // namespace sgl

extern "C" {

int sgl_op_dwconv3x3(const void* x, int dtype, const float* w, const float* bias, void* y, int B, int gh, int gw, int E,
                     int flip, sgl_stream stream) {
  if (!x || !w || !y) return SGL_ERR_NULL;
  if (dtype != SGL_DTYPE_BF16 && dtype != SGL_DTYPE_F32) return SGL_ERR_UNSUPPORTED;
  const int nv = dtype == SGL_DTYPE_BF16 ? 8 : 4;
  if (!sgl::dw_shape_ok(B, gh, gw, E, nv)) return SGL_ERR_BAD_SHAPE;
  const size_t total = (size_t)B * gh * gw * (E / nv);
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SGL_DTYPE_BF16)
    hipLaunchKernelGGL((sgl::dwconv3x3_kernel<sgl::bf16, 8>), dim3(blocks), dim3(256), 0, s, (const sgl::bf16*)x, w, bias,
                       (sgl::bf16*)y, B, gh, gw, E, flip);
  else
    hipLaunchKernelGGL((sgl::dwconv3x3_kernel<float, 4>), dim3(blocks), dim3(256), 0, s, (const float*)x, w, bias,
                       (float*)y, B, gh, gw, E, flip);
  return hipGetLastError() == hipSuccess ? SGL_OK : SGL_ERR_HIP;
}

size_t sgl_op_dwconv3x3_wgrad_scratch_bytes(int B, int gh, int gw, int E) {
  (void)B; (void)gh; (void)gw;
  return (size_t)512 * 10 * E * sizeof(float);
}

int sgl_op_dwconv3x3_wgrad(const void* x, const void* dy, int dtype, float* dw10, int accumulate, float* scratch,
                           size_t scratch_bytes, int B, int gh, int gw, int E, sgl_stream stream) {
  if (!x || !dy || !dw10 || !scratch) return SGL_ERR_NULL;
  if (dtype != SGL_DTYPE_BF16 && dtype != SGL_DTYPE_F32) return SGL_ERR_UNSUPPORTED;
  const int nv = dtype == SGL_DTYPE_BF16 ? 8 : 4;
  if (!sgl::dw_shape_ok(B, gh, gw, E, nv)) return SGL_ERR_BAD_SHAPE;
  const size_t npix = (size_t)B * gh * gw;
  int nblk = (int)(npix < 512 ? npix : 512);
  const int ppb = (int)((npix + nblk - 1) / nblk);
  nblk = (int)((npix + ppb - 1) / ppb);
  if (scratch_bytes < (size_t)nblk * 10 * E * sizeof(float)) return SGL_ERR_WORKSPACE;
  const int lanes = 256 / (E / nv);
  const size_t smem = (size_t)lanes * 10 * E * sizeof(float);  // 40 KiB (fp32) or 80 KiB (bf16)
  hipStream_t s = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&sgl::dwconv3x3_wgrad_kernel<sgl::bf16, 8>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
      return SGL_ERR_HIP;
    attr = true;
  }
  if (dtype == SGL_DTYPE_BF16)
    hipLaunchKernelGGL((sgl::dwconv3x3_wgrad_kernel<sgl::bf16, 8>), dim3(nblk), dim3(256), smem, s, (const sgl::bf16*)x,
                       (const sgl::bf16*)dy, scratch, B, gh, gw, E, ppb);
  else
    hipLaunchKernelGGL((sgl::dwconv3x3_wgrad_kernel<float, 4>), dim3(nblk), dim3(256), smem, s, (const float*)x,
                       (const float*)dy, scratch, B, gh, gw, E, ppb);
  if (hipGetLastError() != hipSuccess) return SGL_ERR_HIP;
  // stage 2: the result has the partials' own [10][E] layout (nine tap-major weight rows, then the bias row)
  return sgl::reduce_partials(scratch, nblk, 10 * E, dw10, 10 * E, accumulate, s) == hipSuccess ? SGL_OK : SGL_ERR_HIP;
}

}  // extern "C"

// GPU input pipeline, first slice of SURVEY.md §8f row 2: the reference's per-batch GPU transform
//     K.Resize(target_resolution, antialias=True) -> K.Normalize(mean=0.5, std=0.5)      cifake_binary_classifier.py:1791-1794
//     (+ optionally MixUp: lam * images + (1 - lam) * images[index]                       cifake_binary_classifier.py:812-817)
// fused with the patch gather of the patch-embedding convolution (TF:modeling_siglip.py:175-185): source images
// (decoded uint8 NHWC bytes, or the float [0,1] NCHW tensors the reference's CPU transform produces) are resampled,
// normalised and written straight into the bf16 patch-major A operand [B*gh*gw][Kp] of the patch GEMM — the fp32
// (B,3,S,S) pixel tensor of the reference is never materialised and the im2col pass (encoder.hip) disappears.
//
// Resampling = separable triangle filter with the support stretched by the down-scale factor: the arithmetic of
// torch's upsample_bilinear2d(antialias=True) (aten/native/cpu/UpSampleKernel.cpp, _compute_indices_weights_aa), which
// is what torchvision Resize(antialias=True) runs in the reference's CPU transform (cifake...:1795-1797).  kornia (the
// GPU transform) is not installed here: its result is "parity unpinned" (it blurs with a Gaussian before sampling).
// HBM-bound; one thread = one operand element (coalesced bf16 stores along k), source taps come from L1/L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hip.h"
#include "kernels.h"
#include "siglip_hip.h"

namespace sgl {

struct AaAxis {
  int lo, n;
  float center, invscale, inv_total;
};

// taps of output index i along one axis (in -> out), weights w_j = tri((j + lo - center + 0.5) * invscale) / total
__device__ __forceinline__ AaAxis aa_axis(int i, int in, float scale) {
  AaAxis a;
  const float support = scale >= 1.0f ? scale : 1.0f;
  a.invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
  a.center = scale * ((float)i + 0.5f);
  int lo = (int)(a.center - support + 0.5f);
  if (lo < 0) lo = 0;
  int hi = (int)(a.center + support + 0.5f);
  if (hi > in) hi = in;
  a.lo = lo;
  a.n = hi - lo;
  float total = 0.f;
  for (int j = 0; j < a.n; ++j) {
    float x = ((float)(j + lo) - a.center + 0.5f) * a.invscale;
    x = x < 0.f ? -x : x;
    total += x < 1.0f ? 1.0f - x : 0.f;
  }
  a.inv_total = total != 0.f ? 1.0f / total : 0.f;
  return a;
}
__device__ __forceinline__ float aa_w(const AaAxis& a, int j) {
  float x = ((float)(j + a.lo) - a.center + 0.5f) * a.invscale;
  x = x < 0.f ? -x : x;
  return (x < 1.0f ? 1.0f - x : 0.f) * a.inv_total;
}

template <bool SRC_U8>
__device__ __forceinline__ float src_px(const void* src, int b, int c, int y, int x, int Hs, int Ws) {
  if constexpr (SRC_U8)   // NHWC bytes
    return (float)reinterpret_cast<const uint8_t*>(src)[(((size_t)b * Hs + y) * Ws + x) * 3 + c] * (1.0f / 255.0f);
  else                    // NCHW float in [0, 1]
    return reinterpret_cast<const float*>(src)[(((size_t)b * 3 + c) * Hs + y) * Ws + x];
}

template <bool SRC_U8>
__device__ __forceinline__ float resample(const void* src, int b, int c, int oy, int ox, int Hs, int Ws, float sy,
                                          float sx) {
  if (sy == 1.0f && sx == 1.0f) return src_px<SRC_U8>(src, b, c, oy, ox, Hs, Ws);
  const AaAxis ay = aa_axis(oy, Hs, sy), ax = aa_axis(ox, Ws, sx);
  float acc = 0.f;
  for (int jy = 0; jy < ay.n; ++jy) {
    float row = 0.f;
    for (int jx = 0; jx < ax.n; ++jx) row += aa_w(ax, jx) * src_px<SRC_U8>(src, b, c, ay.lo + jy, ax.lo + jx, Hs, Ws);
    acc += aa_w(ay, jy) * row;
  }
  return acc;
}

// patch_major: out[(b*gh + gy)*gw + gx][k], k = c*P*P + ky*P + kx (k >= 3*P*P zero)   | else out[b][c][y][x] (NCHW)
template <bool SRC_U8, typename TOut>
__global__ __launch_bounds__(256) void preprocess_kernel(const void* __restrict__ src, TOut* __restrict__ out, int B,
                                                         int Hs, int Ws, int S, int P, int Kp, int patch_major,
                                                         float mean, float inv_std, const int* __restrict__ mix_index,
                                                         float lam) {
  const int g = S / P, K0 = 3 * P * P;
  const size_t total = patch_major ? (size_t)B * g * g * Kp : (size_t)B * 3 * S * S;
  const float sy = (float)Hs / (float)S, sx = (float)Ws / (float)S;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    int b, c, oy, ox;
    bool live = true;
    if (patch_major) {
      const int k = (int)(idx % Kp);
      const size_t m = idx / Kp;
      const int gx = (int)(m % g), gy = (int)((m / g) % g);
      b = (int)(m / ((size_t)g * g));
      live = k < K0;
      c = k / (P * P);
      const int r = k - c * P * P;
      oy = gy * P + r / P;
      ox = gx * P + r % P;
    } else {
      ox = (int)(idx % S);
      oy = (int)((idx / S) % S);
      c = (int)((idx / ((size_t)S * S)) % 3);
      b = (int)(idx / ((size_t)3 * S * S));
    }
    float v = 0.f;
    if (live) {
      v = resample<SRC_U8>(src, b, c, oy, ox, Hs, Ws, sy, sx);
      if (mix_index) v = lam * v + (1.0f - lam) * resample<SRC_U8>(src, mix_index[b], c, oy, ox, Hs, Ws, sy, sx);
      v = (v - mean) * inv_std;
    }
    Elem<TOut>::st(out + idx, v);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Augmentation branch of the video trainer's GPU transform (hidf_video_classifier.py:2868-2874), fused into the same pass:
//     K.Resize(S, antialias=True) -> K.RandomHorizontalFlip(p=0.5) -> K.RandomRotation(degrees=5, p=0.3)
//     -> K.ColorJitter(brightness=0.1, contrast=0.1, saturation=0.1, hue=0.05, p=0.3) -> K.Normalize(0.5, 0.5)
// Randomness stays with the caller: per-sample parameters arrive in a device table (sgl_aug_sample).  Definitions used
// (kornia is not installed: "parity unpinned" against it; oracle/preprocess_oracle.py restates exactly these):
//   flip      out(y, x) = in(y, S-1-x)
//   rotation  about the image centre ((S-1)/2, (S-1)/2), positive angle counter-clockwise (OpenCV / kornia
//             get_rotation_matrix2d), out(p) = in(M^-1 p) sampled bilinearly, zeros outside
//   colour    torchvision-style operators on [0,1] RGB, each clamped to [0,1], applied in the order given per sample:
//             0 brightness x*f | 1 contrast (x - m)*f + m, m = mean of the image's grey level at that point of the chain |
//             2 saturation (x - grey)*f + grey | 3 hue: RGB -> HSV, h = frac(h + shift), HSV -> RGB; grey = .299R+.587G+.114B
// The contrast operator needs a per-image mean of the partly transformed image: a first launch (aug_mean_kernel, one
// workgroup per image, fixed summation order) evaluates the chain up to the contrast step and reduces it.
struct AugSample {   // == sgl_aug_sample
  float flip, cos_a, sin_a, brightness, contrast, saturation, hue;
  int order[4];
  int reserved;
};

__device__ __forceinline__ float aug_clamp01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }
__device__ __forceinline__ float aug_grey(const float* c) { return 0.299f * c[0] + 0.587f * c[1] + 0.114f * c[2]; }

__device__ __forceinline__ void aug_hue(float* c, float shift) {
  const float r = c[0], g = c[1], b = c[2];
  const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
  const bool eq = maxc == minc;
  const float cr = maxc - minc;
  const float sat = cr / (eq ? 1.0f : maxc);
  const float crd = eq ? 1.0f : cr;
  const float rc = (maxc - r) / crd, gc = (maxc - g) / crd, bc = (maxc - b) / crd;
  const float hr = (maxc == r) ? (bc - gc) : 0.f;
  const float hg = (maxc == g && maxc != r) ? (2.0f + rc - bc) : 0.f;
  const float hb = (maxc != g && maxc != r) ? (4.0f + gc - rc) : 0.f;
  float h = (hr + hg + hb) / 6.0f + 1.0f;
  h = h - floorf(h);                 // fmod(., 1)
  h = h + shift;
  h = h - floorf(h);                 // (h + shift) % 1
  const float v = maxc;
  const float h6 = h * 6.0f;
  const float fi = floorf(h6);
  const float f = h6 - fi;
  int i = (int)fi % 6;
  const float p = aug_clamp01(v * (1.0f - sat));
  const float q = aug_clamp01(v * (1.0f - f * sat));
  const float t = aug_clamp01(v * (1.0f - (1.0f - f) * sat));
  switch (i) {
    case 0: c[0] = v; c[1] = t; c[2] = p; break;
    case 1: c[0] = q; c[1] = v; c[2] = p; break;
    case 2: c[0] = p; c[1] = v; c[2] = t; break;
    case 3: c[0] = p; c[1] = q; c[2] = v; break;
    case 4: c[0] = t; c[1] = p; c[2] = v; break;
    default: c[0] = v; c[1] = p; c[2] = q; break;
  }
}

// resized + flipped + rotated RGB of output pixel (oy, ox)
template <bool SRC_U8>
__device__ __forceinline__ void aug_geo(const void* src, int b, int oy, int ox, int Hs, int Ws, int S, float sy, float sx,
                                        const AugSample& a, float* rgb) {
  const bool flip = a.flip != 0.f;
  auto px = [&](int c, int y, int x) {
    return resample<SRC_U8>(src, b, c, y, flip ? S - 1 - x : x, Hs, Ws, sy, sx);
  };
  if (a.sin_a == 0.f && a.cos_a == 1.f) {
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[c] = px(c, oy, ox);
    return;
  }
  const float ctr = 0.5f * (float)(S - 1);
  const float dx = (float)ox - ctr, dy = (float)oy - ctr;
  const float xs = a.cos_a * dx - a.sin_a * dy + ctr;
  const float ys = a.sin_a * dx + a.cos_a * dy + ctr;
  const float x0f = floorf(xs), y0f = floorf(ys);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const float fx = xs - x0f, fy = ys - y0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = 0.f;
#pragma unroll
    for (int jy = 0; jy < 2; ++jy)
#pragma unroll
      for (int jx = 0; jx < 2; ++jx) {
        const int yy = y0 + jy, xx = x0 + jx;
        const float w = (jy ? fy : 1.0f - fy) * (jx ? fx : 1.0f - fx);
        if (yy >= 0 && yy < S && xx >= 0 && xx < S && w != 0.f) v += w * px(c, yy, xx);
      }
    rgb[c] = v;
  }
}

// colour chain on rgb; stop_before_contrast: evaluate only the operators in front of the contrast step (mean pre-pass)
__device__ __forceinline__ void aug_colour(float* rgb, const AugSample& a, float grey_mean, bool stop_before_contrast) {
  if (a.order[0] < 0) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int op = a.order[k];
    if (op == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb[c] = aug_clamp01(rgb[c] * a.brightness);
    } else if (op == 1) {
      if (stop_before_contrast) return;
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb[c] = aug_clamp01((rgb[c] - grey_mean) * a.contrast + grey_mean);
    } else if (op == 2) {
      const float g = aug_grey(rgb);
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb[c] = aug_clamp01((rgb[c] - g) * a.saturation + g);
    } else {
      aug_hue(rgb, a.hue);
    }
  }
}

template <bool SRC_U8>
__global__ __launch_bounds__(256) void aug_mean_kernel(const void* __restrict__ src, int Hs, int Ws, int S,
                                                       const AugSample* __restrict__ aug, float* __restrict__ grey_mean) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const AugSample a = aug[b];
  bool need = false;
  if (a.order[0] >= 0)
    for (int k = 0; k < 4; ++k) need = need || a.order[k] == 1;
  if (!need) {
    if (threadIdx.x == 0) grey_mean[b] = 0.f;
    return;
  }
  const float sy = (float)Hs / (float)S, sx = (float)Ws / (float)S;
  float acc = 0.f;
  for (int i = threadIdx.x; i < S * S; i += 256) {
    float rgb[3];
    aug_geo<SRC_U8>(src, b, i / S, i % S, Hs, Ws, S, sy, sx, a, rgb);
    aug_colour(rgb, a, 0.f, true);
    acc += aug_grey(rgb);
  }
  acc = wave_sum(acc);
  if (lane_id() == 0) red[wave_id()] = acc;
  __syncthreads();
  if (threadIdx.x == 0) grey_mean[b] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)(S * S);
}

template <bool SRC_U8, typename TOut>
__global__ __launch_bounds__(256) void preprocess_aug_kernel(const void* __restrict__ src, TOut* __restrict__ out, int B,
                                                             int Hs, int Ws, int S, int P, int Kp, int patch_major,
                                                             float mean, float inv_std,
                                                             const AugSample* __restrict__ aug,
                                                             const float* __restrict__ grey_mean) {
  // one thread = one output PIXEL (all three channels: the colour operators mix them)
  const int g = patch_major ? S / P : 0;
  const int side = patch_major ? g * P : S;             // pixels past the last whole patch are dropped ('valid' conv)
  const size_t total = (size_t)B * side * side;
  const float sy = (float)Hs / (float)S, sx = (float)Ws / (float)S;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int ox = (int)(idx % side), oy = (int)((idx / side) % side), b = (int)(idx / ((size_t)side * side));
    const AugSample a = aug[b];
    float rgb[3];
    aug_geo<SRC_U8>(src, b, oy, ox, Hs, Ws, S, sy, sx, a, rgb);
    aug_colour(rgb, a, grey_mean[b], false);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = (rgb[c] - mean) * inv_std;
      size_t o;
      if (patch_major)
        o = (((size_t)b * g + oy / P) * g + ox / P) * Kp + (size_t)c * P * P + (oy % P) * P + (ox % P);
      else
        o = (((size_t)b * 3 + c) * S + oy) * S + ox;
      Elem<TOut>::st(out + o, v);
    }
  }
}

// zero the K padding columns [3*P*P, Kp) of a patch-major operand (the augmentation kernel writes pixels only)
template <typename TOut>
__global__ __launch_bounds__(256) void patch_pad_zero_kernel(TOut* __restrict__ out, size_t rows, int K0, int Kp) {
  const int padw = Kp - K0;
  const size_t total = rows * (size_t)padw;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256)
    Elem<TOut>::st(out + (idx / padw) * Kp + K0 + idx % padw, 0.f);
}

}  // namespace sgl

extern "C" {

int sgl_op_preprocess(const void* src, int src_is_u8_nhwc, int B, int Hs, int Ws, void* out, int out_dtype, int S, int P,
                      int Kp, int patch_major, float mean, float std, const int* mix_index, float lam,
                      sgl_stream stream) {
  if (!src || !out) return SGL_ERR_NULL;
  if (B <= 0 || Hs <= 0 || Ws <= 0 || S <= 0 || std == 0.f) return SGL_ERR_BAD_SHAPE;
  if (patch_major && (P <= 0 || S < P || Kp < 3 * P * P)) return SGL_ERR_BAD_SHAPE;
  if (out_dtype != SGL_DTYPE_BF16 && out_dtype != SGL_DTYPE_F32) return SGL_ERR_UNSUPPORTED;
  if ((float)Hs / (float)S > 16.f || (float)Ws / (float)S > 16.f) return SGL_ERR_UNSUPPORTED;  // tap loops stay short
  const int g = patch_major ? S / P : 0;
  const size_t total = patch_major ? (size_t)B * g * g * Kp : (size_t)B * 3 * S * S;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipStream_t s = (hipStream_t)stream;
  const float inv_std = 1.0f / std;
#define SGL_PP(U8, T)                                                                                               \
  hipLaunchKernelGGL((sgl::preprocess_kernel<U8, T>), dim3(blocks), dim3(256), 0, s, src, (T*)out, B, Hs, Ws, S, P, Kp, \
                     patch_major, mean, inv_std, mix_index, lam)
  if (src_is_u8_nhwc) {
    if (out_dtype == SGL_DTYPE_BF16) SGL_PP(true, sgl::bf16); else SGL_PP(true, float);
  } else {
    if (out_dtype == SGL_DTYPE_BF16) SGL_PP(false, sgl::bf16); else SGL_PP(false, float);
  }
#undef SGL_PP
  return hipGetLastError() == hipSuccess ? SGL_OK : SGL_ERR_HIP;
}

int sgl_op_preprocess_aug(const void* src, int src_is_u8_nhwc, int B, int Hs, int Ws, void* out, int out_dtype, int S,
                          int P, int Kp, int patch_major, float mean, float std, const sgl_aug_sample* aug,
                          float* grey_mean, sgl_stream stream) {
  static_assert(sizeof(sgl_aug_sample) == sizeof(sgl::AugSample) && sizeof(sgl_aug_sample) == 48, "table layout");
  if (!src || !out || !aug || !grey_mean) return SGL_ERR_NULL;
  if (B <= 0 || Hs <= 0 || Ws <= 0 || S <= 0 || std == 0.f) return SGL_ERR_BAD_SHAPE;
  if (patch_major && (P <= 0 || S < P || Kp < 3 * P * P)) return SGL_ERR_BAD_SHAPE;
  if (out_dtype != SGL_DTYPE_BF16 && out_dtype != SGL_DTYPE_F32) return SGL_ERR_UNSUPPORTED;
  if ((float)Hs / (float)S > 16.f || (float)Ws / (float)S > 16.f) return SGL_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const sgl::AugSample* tab = reinterpret_cast<const sgl::AugSample*>(aug);
  if (src_is_u8_nhwc)
    hipLaunchKernelGGL((sgl::aug_mean_kernel<true>), dim3((unsigned)B), dim3(256), 0, s, src, Hs, Ws, S, tab, grey_mean);
  else
    hipLaunchKernelGGL((sgl::aug_mean_kernel<false>), dim3((unsigned)B), dim3(256), 0, s, src, Hs, Ws, S, tab, grey_mean);
  const int g = patch_major ? S / P : 0;
  const int side = patch_major ? g * P : S;
  const size_t total = (size_t)B * side * side;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  const float inv_std = 1.0f / std;
#define SGL_PA(U8, T)                                                                                                   \
  do {                                                                                                                  \
    if (patch_major && Kp > 3 * P * P)                                                                                  \
      hipLaunchKernelGGL((sgl::patch_pad_zero_kernel<T>), dim3(1024), dim3(256), 0, s, (T*)out, (size_t)B * g * g,      \
                         3 * P * P, Kp);                                                                                \
    hipLaunchKernelGGL((sgl::preprocess_aug_kernel<U8, T>), dim3(blocks), dim3(256), 0, s, src, (T*)out, B, Hs, Ws, S, P, \
                       Kp, patch_major, mean, inv_std, tab, grey_mean);                                                 \
  } while (0)
  if (src_is_u8_nhwc) {
    if (out_dtype == SGL_DTYPE_BF16) SGL_PA(true, sgl::bf16); else SGL_PA(true, float);
  } else {
    if (out_dtype == SGL_DTYPE_BF16) SGL_PA(false, sgl::bf16); else SGL_PA(false, float);
  }
#undef SGL_PA
  return hipGetLastError() == hipSuccess ? SGL_OK : SGL_ERR_HIP;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Video tail (hidf_video_classifier.py:304-316): per-frame embeddings (B*T, D) -> L2-normalise each frame -> mean over the T
// frames of a clip -> (B, D).  One workgroup per clip; forward keeps 1/|f| per frame for the backward:
//   out[b] = (1/T) sum_t f_t / |f_t|          d f_t = (g - fhat_t (fhat_t . g)) / (T |f_t|),  g = d out[b]
// ---------------------------------------------------------------------------------------------------------------
namespace sgl {

__global__ __launch_bounds__(256) void l2norm_tmean_fwd_kernel(const float* __restrict__ f, float* __restrict__ out,
                                                               float* __restrict__ inv_norm, int T, int D) {
  extern __shared__ float acc[];   // [D]
  __shared__ float red[4];
  const int b = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += 256) acc[d] = 0.f;
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const float* row = f + ((size_t)b * T + t) * D;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) s += row[d] * row[d];
    s = wave_sum(s);
    if (lane_id() == 0) red[wave_id()] = s;
    __syncthreads();
    const float inv = 1.0f / sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    if (threadIdx.x == 0) inv_norm[(size_t)b * T + t] = inv;
    for (int d = threadIdx.x; d < D; d += 256) acc[d] += row[d] * inv;
    __syncthreads();
  }
  const float it = 1.0f / (float)T;
  for (int d = threadIdx.x; d < D; d += 256) out[(size_t)b * D + d] = acc[d] * it;
}

__global__ __launch_bounds__(256) void l2norm_tmean_bwd_kernel(const float* __restrict__ f,
                                                               const float* __restrict__ inv_norm,
                                                               const float* __restrict__ dout, float* __restrict__ df,
                                                               int T, int D) {
  __shared__ float red[4];
  const int bt = blockIdx.x, b = bt / T;
  const float* row = f + (size_t)bt * D;
  const float* g = dout + (size_t)b * D;
  const float inv = inv_norm[bt];
  float s = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) s += row[d] * g[d];
  s = wave_sum(s);
  if (lane_id() == 0) red[wave_id()] = s;
  __syncthreads();
  const float dot = ((red[0] + red[1]) + (red[2] + red[3])) * inv;   // fhat . g
  const float k = inv / (float)T;
  for (int d = threadIdx.x; d < D; d += 256) df[(size_t)bt * D + d] = (g[d] - row[d] * inv * dot) * k;
}

}  // namespace sgl

extern "C" {

int sgl_op_l2norm_tmean_fwd(const float* f, float* out, float* inv_norm, int B, int T, int D, sgl_stream stream) {
  if (!f || !out || !inv_norm) return SGL_ERR_NULL;
  if (B <= 0 || T <= 0 || D <= 0 || D > 16384) return SGL_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(sgl::l2norm_tmean_fwd_kernel, dim3((unsigned)B), dim3(256), (size_t)D * sizeof(float),
                     (hipStream_t)stream, f, out, inv_norm, T, D);
  return hipGetLastError() == hipSuccess ? SGL_OK : SGL_ERR_HIP;
}

int sgl_op_l2norm_tmean_bwd(const float* f, const float* inv_norm, const float* dout, float* df, int B, int T, int D,
                            sgl_stream stream) {
  if (!f || !inv_norm || !dout || !df) return SGL_ERR_NULL;
  if (B <= 0 || T <= 0 || D <= 0) return SGL_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(sgl::l2norm_tmean_bwd_kernel, dim3((unsigned)(B * T)), dim3(256), 0, (hipStream_t)stream, f, inv_norm,
                     dout, df, T, D);
  return hipGetLastError() == hipSuccess ? SGL_OK : SGL_ERR_HIP;
}

}  // extern "C"

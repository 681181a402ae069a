// Optimizer step tail of the training loop: multi-tensor AdamW with the global-norm gradient
// clip folded in, no host synchronisation anywhere.
//
// What it replaces in the reference (every trainer does the same three calls per step):
//     torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)      Siglip2sidafrozen.py:1396,
//                                                                        cifake_binary_classifier.py:836-858,
//                                                                        hidf_video_classifier.py:396-412
//     optimizer.step()   with  torch.optim.AdamW(model.parameters(), lr=..., weight_decay=...)
//                                                                        Siglip2sidafrozen.py:1241-1244,
//                                                                        cifake_binary_classifier.py:1916-1920,
//                                                                        hidf_video_classifier.py:2941
// torch's clip returns the norm to the host (`.item()` in the logging path, Siglip2sidafrozen.py:1391) and its
// AdamW is a dozen elementwise passes per tensor.  Here: one launch computes per-block partial sums of squares, one
// tiny launch folds them (fixed order -> bitwise reproducible) into {norm, clip coefficient} in device memory, one
// launch applies AdamW to every tensor reading the coefficient from device memory.
//
// Update rule (torch/optim/adamw.py single-tensor path, same operation order, fp32):
//     p <- p * (1 - lr*wd);  m <- m + (g - m)*(1 - b1);  v <- v*b2 + g*g*(1 - b2)
//     p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
//
// HBM-bound: 28 algorithmic bytes per parameter (read p,g,m,v; write p,m,v), +4 for the norm pass.
// Work decomposition: the host plan (sgl_adamw_plan) cuts every tensor into 4096-element chunks and lists
// (tensor, chunk) pairs; a 256-thread block takes one pair, 16 elements per thread as four 16-byte accesses.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "common.hip.h"
#include "siglip_hip.h"

namespace sgl {

constexpr int OPT_CHUNK = 4096;

__device__ __forceinline__ bool aligned16(const void* a, const void* b, const void* c, const void* d) {
  return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) == 0;
}

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const sgl_adamw_tensor* __restrict__ T,
                                                          const int32_t* __restrict__ map,
                                                          float* __restrict__ partial) {
  __shared__ float red[4];
  const int ti = map[2 * blockIdx.x], ch = map[2 * blockIdx.x + 1];
  const float* g = T[ti].g;
  const uint64_t n = T[ti].n;
  const uint64_t base = (uint64_t)ch * OPT_CHUNK;
  float s = 0.f;
  if (g) {
    if ((((uintptr_t)g) & 15) == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint64_t i = base + (uint64_t)(k * 256 + threadIdx.x) * 4;
        if (i + 3 < n) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(g + i);
          s += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
        } else {
          for (uint64_t j = i; j < n && j < i + 4; ++j) s += g[j] * g[j];
        }
      }
    } else {
      for (int k = 0; k < 16; ++k) {
        const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
        if (i < n) s += g[i] * g[i];
      }
    }
  }
  s = wave_sum(s);
  if (lane_id() == 0) red[wave_id()] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = ||g||_2 ; out[1] = min(1, max_norm / (norm + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(1024) void grad_norm_finish_kernel(const float* __restrict__ partial, int n,
                                                                float max_norm, float* __restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) s += (double)partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += red[i];
    const float norm = (float)sqrt(t);
    out[0] = norm;
    float coef = 1.0f;
    if (max_norm > 0.f) {
      coef = max_norm / (norm + 1e-6f);
      if (coef > 1.0f) coef = 1.0f;
    }
    out[1] = coef;
  }
}

struct AdamConst {
  float omb1, beta2, omb2, eps, inv_bc1, inv_sqrt_bc2;  // omb = 1 - beta, rounded from double like torch's scalars
};

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, float lr, float wd,
                                          const AdamConst& c) {
  // no FMA contraction: every product and sum rounds once, as torch's separate elementwise kernels do, and — what matters
  // here — identically in every kernel this is inlined into (the tiled shadow-writing path and the linear path gave
  // parameters one ulp apart before, which bf16 weight rounding then amplifies into different training trajectories)
#pragma clang fp contract(off)
  p = p * (1.0f - lr * wd);
  m = m + (g - m) * c.omb1;
  v = v * c.beta2 + (g * g) * c.omb2;
  const float denom = sqrtf(v) * c.inv_sqrt_bc2 + c.eps;
  p = p - (lr * c.inv_bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(const sgl_adamw_tensor* __restrict__ T,
                                                    const int32_t* __restrict__ map, AdamConst c,
                                                    const float* __restrict__ clip /* [2] or null */) {
  const int ti = map[2 * blockIdx.x], ch = map[2 * blockIdx.x + 1];
  const sgl_adamw_tensor t = T[ti];
  if (!t.g) return;  // parameter without a gradient this step: untouched, as torch skips p.grad is None
  const float gs = clip ? clip[1] : 1.0f;
  const uint64_t base = (uint64_t)ch * OPT_CHUNK;
  if (aligned16(t.p, t.g, t.m, t.v)) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint64_t i = base + (uint64_t)(k * 256 + threadIdx.x) * 4;
      if (i + 3 < t.n) {
        f32x4 p = *reinterpret_cast<const f32x4*>(t.p + i);
        const f32x4 g = *reinterpret_cast<const f32x4*>(t.g + i);
        f32x4 m = *reinterpret_cast<const f32x4*>(t.m + i);
        f32x4 v = *reinterpret_cast<const f32x4*>(t.v + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float pj = p[j], mj = m[j], vj = v[j];
          adamw_one(pj, g[j] * gs, mj, vj, t.lr, t.weight_decay, c);
          p[j] = pj;
          m[j] = mj;
          v[j] = vj;
        }
        *reinterpret_cast<f32x4*>(t.p + i) = p;
        *reinterpret_cast<f32x4*>(t.m + i) = m;
        *reinterpret_cast<f32x4*>(t.v + i) = v;
      } else {
        for (uint64_t j = i; j < t.n && j < i + 4; ++j) adamw_one(t.p[j], t.g[j] * gs, t.m[j], t.v[j], t.lr, t.weight_decay, c);
      }
    }
  } else {
    for (int k = 0; k < 16; ++k) {
      const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
      if (i < t.n) adamw_one(t.p[i], t.g[i] * gs, t.m[i], t.v[i], t.lr, t.weight_decay, c);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// AdamW + everything that must follow a parameter update, in the same pass over the parameter (SURVEY.md §8f row 3,
// kernel work-list k11): the compute-dtype weight shadow the GEMMs read (row-major copy, zero-padded leading dimension),
// its transpose (the dX GEMMs' "B" operand), fp32 bias copies, and the CiFake trainer's weight EMA
// (cifake_binary_classifier.py:222-225).  Replaces the per-step cast_pad / cast_transpose launches of
// sgl_prepare_weights (what autocast re-does every step in the reference, Siglip2sidafrozen.py:1375): +4 B/parameter of
// stores here instead of a 6 B/parameter read-modify pass of their own.
//
// Matrices with a shadow are walked in 64x64 tiles (one tile per 256-thread block: 4 x 16-byte accesses per thread per
// array, 256-byte row segments), the bf16 tile is transposed through LDS so that both copies are written in >= 128-byte
// row segments.  Everything else uses the linear 4096-element chunks of adamw_kernel.
// ---------------------------------------------------------------------------------------------------------------
struct GroupHyper {
  float lr[16], wd[16];
  int n;
};

template <typename T> __device__ __forceinline__ T cvt_out(float x);
template <> __device__ __forceinline__ float cvt_out<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 cvt_out<bf16>(float x) { return (bf16)x; }

template <typename T>
__device__ __forceinline__ void adamw_tile(const sgl_adamw_tensor& t, const sgl_adamw_aux& a, int tile, float lr,
                                           float wd, float gs, const AdamConst& c, float ema_decay, T* lt /*[64][66]*/) {
  const int cols = a.cols, rows = a.rows;
  const int tiles_c = (cols + 63) >> 6;
  const int tr = tile / tiles_c, tc = tile - tr * tiles_c;
  const int r0 = tr * 64, c0 = tc * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const bool vec = ((cols & 3) == 0) && aligned16(t.p, t.g, t.m, t.v) && (!a.ema || ((((uintptr_t)a.ema) & 15) == 0));
  T* dst = reinterpret_cast<T*>(a.dst);
  // interior tiles: every load of the tile in flight before the first dependent instruction
  const bool interior = vec && (r0 + 64 <= rows) && (c0 + 64 <= cols);
  f32x4 P4[4], G4[4], M4[4], V4[4];
  if (interior) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t i = (size_t)(r0 + ty + 16 * k) * cols + c0 + tx * 4;
      P4[k] = *reinterpret_cast<const f32x4*>(t.p + i);
      G4[k] = *reinterpret_cast<const f32x4*>(t.g + i);
      M4[k] = *reinterpret_cast<const f32x4*>(t.m + i);
      V4[k] = *reinterpret_cast<const f32x4*>(t.v + i);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 16 * k, cc = c0 + tx * 4;
    float out[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < rows && cc < cols) {
      const size_t i = (size_t)r * cols + cc;
      if (vec && cc + 3 < cols) {
        f32x4 p = interior ? P4[k] : *reinterpret_cast<const f32x4*>(t.p + i);
        const f32x4 g = interior ? G4[k] : *reinterpret_cast<const f32x4*>(t.g + i);
        f32x4 m = interior ? M4[k] : *reinterpret_cast<const f32x4*>(t.m + i);
        f32x4 v = interior ? V4[k] : *reinterpret_cast<const f32x4*>(t.v + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float pj = p[j], mj = m[j], vj = v[j];
          adamw_one(pj, g[j] * gs, mj, vj, lr, wd, c);
          p[j] = pj; m[j] = mj; v[j] = vj; out[j] = pj;
        }
        *reinterpret_cast<f32x4*>(t.p + i) = p;
        *reinterpret_cast<f32x4*>(t.m + i) = m;
        *reinterpret_cast<f32x4*>(t.v + i) = v;
        if (a.ema) {
          const f32x4 e = *reinterpret_cast<const f32x4*>(a.ema + i);
          *reinterpret_cast<f32x4*>(a.ema + i) = e * ema_decay + p * (1.0f - ema_decay);
        }
      } else {
        for (int j = 0; j < 4 && cc + j < cols; ++j) {
          adamw_one(t.p[i + j], t.g[i + j] * gs, t.m[i + j], t.v[i + j], lr, wd, c);
          out[j] = t.p[i + j];
          if (a.ema) a.ema[i + j] = a.ema[i + j] * ema_decay + out[j] * (1.0f - ema_decay);
        }
      }
      if (dst && r >= a.row0) {
        T* d = dst + (size_t)(r - a.row0) * a.ld + cc;
        if (cc + 3 < cols && ((((uintptr_t)d) & (4 * sizeof(T) - 1)) == 0)) {
          T o4[4] = {cvt_out<T>(out[0]), cvt_out<T>(out[1]), cvt_out<T>(out[2]), cvt_out<T>(out[3])};
          if constexpr (sizeof(T) == 2) {
            u32x2 w;
            __builtin_memcpy(&w, o4, 8);
            *reinterpret_cast<u32x2*>(d) = w;
          } else {
            u32x4 w;
            __builtin_memcpy(&w, o4, 16);
            *reinterpret_cast<u32x4*>(d) = w;
          }
        } else {
          for (int j = 0; j < 4 && cc + j < cols; ++j) d[j] = cvt_out<T>(out[j]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) lt[(ty + 16 * k) * 66 + tx * 4 + j] = cvt_out<T>(out[j]);
  }
  if (!a.dst_t) return;
  __syncthreads();
  // transposed copy: output row = parameter column c0 + oc, 64 consecutive elements = parameter rows r0 .. r0+63
  T* dt = reinterpret_cast<T*>(a.dst_t);
  const int oc = threadIdx.x >> 2, seg = threadIdx.x & 3;   // 64 output rows x 4 segments of 16 elements
  if (c0 + oc < cols) {
    const int rb = r0 + seg * 16;
    T* drow = dt + (size_t)(c0 + oc) * a.ld_t + (rb - a.row0);
    T vals[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) vals[j] = lt[(seg * 16 + j) * 66 + oc];
    if (rb >= a.row0 && rb + 15 < rows && ((((uintptr_t)drow) & 15) == 0)) {
      // 16 consecutive elements of one output row: 16-byte stores
      constexpr int PER = 16 / sizeof(T);
#pragma unroll
      for (int q = 0; q < 16 / PER; ++q) {
        u32x4 w;
        __builtin_memcpy(&w, &vals[q * PER], 16);
        *reinterpret_cast<u32x4*>(drow + q * PER) = w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (rb + j < rows && rb + j >= a.row0) drow[j] = vals[j];
    }
  }
}

__global__ __launch_bounds__(256) void adamw_ex_kernel(const sgl_adamw_tensor* __restrict__ T,
                                                       const sgl_adamw_aux* __restrict__ A,
                                                       const int32_t* __restrict__ map, AdamConst c, GroupHyper gh,
                                                       const float* __restrict__ clip, float ema_decay) {
  __shared__ float lds_tile[64 * 66];
  const int ti = map[2 * blockIdx.x], ch = map[2 * blockIdx.x + 1];
  const sgl_adamw_tensor t = T[ti];
  if (!t.g) return;
  const sgl_adamw_aux a = A[ti];
  const float gs = clip ? clip[1] : 1.0f;
  const float lr = (gh.n > 0 && a.group >= 0 && a.group < gh.n) ? gh.lr[a.group] : t.lr;
  const float wd = (gh.n > 0 && a.group >= 0 && a.group < gh.n) ? gh.wd[a.group] : t.weight_decay;
  if (a.dst || a.dst_t) {  // tiled matrix with shadow copies
    if (a.dtype == SGL_DTYPE_BF16)
      adamw_tile<bf16>(t, a, ch, lr, wd, gs, c, ema_decay, reinterpret_cast<bf16*>(lds_tile));
    else
      adamw_tile<float>(t, a, ch, lr, wd, gs, c, ema_decay, lds_tile);
    return;
  }
  const uint64_t base = (uint64_t)ch * OPT_CHUNK;
  // linear chunk: 1-D tensors and matrices without shadows
  const bool vec = aligned16(t.p, t.g, t.m, t.v) && (!a.ema || ((((uintptr_t)a.ema) & 15) == 0)) &&
                   (!a.dst_f32 || ((((uintptr_t)a.dst_f32) & 15) == 0));
  if (vec && base + OPT_CHUNK <= t.n) {
    // full chunk: all sixteen 16-byte loads (and the EMA's four) are issued before the first dependent instruction —
    // with them inside the per-k loop behind the dst_f32 / ema branches the kernel ran at 45 % of HBM peak instead of 73 %
    f32x4 P[4], G[4], Mm[4], V[4], E[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint64_t i = base + (uint64_t)(k * 256 + threadIdx.x) * 4;
      P[k] = *reinterpret_cast<const f32x4*>(t.p + i);
      G[k] = *reinterpret_cast<const f32x4*>(t.g + i);
      Mm[k] = *reinterpret_cast<const f32x4*>(t.m + i);
      V[k] = *reinterpret_cast<const f32x4*>(t.v + i);
    }
    if (a.ema) {
#pragma unroll
      for (int k = 0; k < 4; ++k) E[k] = *reinterpret_cast<const f32x4*>(a.ema + base + (uint64_t)(k * 256 + threadIdx.x) * 4);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint64_t i = base + (uint64_t)(k * 256 + threadIdx.x) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float pj = P[k][j], mj = Mm[k][j], vj = V[k][j];
        adamw_one(pj, G[k][j] * gs, mj, vj, lr, wd, c);
        P[k][j] = pj; Mm[k][j] = mj; V[k][j] = vj;
      }
      *reinterpret_cast<f32x4*>(t.p + i) = P[k];
      *reinterpret_cast<f32x4*>(t.m + i) = Mm[k];
      *reinterpret_cast<f32x4*>(t.v + i) = V[k];
      if (a.dst_f32) *reinterpret_cast<f32x4*>(a.dst_f32 + i) = P[k];
      if (a.ema) *reinterpret_cast<f32x4*>(a.ema + i) = E[k] * ema_decay + P[k] * (1.0f - ema_decay);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint64_t i = base + (uint64_t)(k * 256 + threadIdx.x) * 4;
    if (vec && i + 3 < t.n) {
      f32x4 p = *reinterpret_cast<const f32x4*>(t.p + i);
      const f32x4 g = *reinterpret_cast<const f32x4*>(t.g + i);
      f32x4 m = *reinterpret_cast<const f32x4*>(t.m + i);
      f32x4 v = *reinterpret_cast<const f32x4*>(t.v + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float pj = p[j], mj = m[j], vj = v[j];
        adamw_one(pj, g[j] * gs, mj, vj, lr, wd, c);
        p[j] = pj; m[j] = mj; v[j] = vj;
      }
      *reinterpret_cast<f32x4*>(t.p + i) = p;
      *reinterpret_cast<f32x4*>(t.m + i) = m;
      *reinterpret_cast<f32x4*>(t.v + i) = v;
      if (a.dst_f32) *reinterpret_cast<f32x4*>(a.dst_f32 + i) = p;
      if (a.ema) {
        const f32x4 e = *reinterpret_cast<const f32x4*>(a.ema + i);
        *reinterpret_cast<f32x4*>(a.ema + i) = e * ema_decay + p * (1.0f - ema_decay);
      }
    } else {
      for (uint64_t j = i; j < t.n && j < i + 4; ++j) {
        adamw_one(t.p[j], t.g[j] * gs, t.m[j], t.v[j], lr, wd, c);
        if (a.dst_f32) a.dst_f32[j] = t.p[j];
        if (a.ema) a.ema[j] = a.ema[j] * ema_decay + t.p[j] * (1.0f - ema_decay);
      }
    }
  }
}

// out[0] = s*||g||_2 ; out[1] = s*min(1, max_norm / (s*norm + 1e-6)): the factor every gradient is multiplied by when the
// stored gradients are rank SUMS and s = 1/world (ddp.GradBucketReducer(average="defer")); s = 1 is grad_norm_finish.
__global__ __launch_bounds__(1024) void grad_norm_finish_scaled_kernel(const float* __restrict__ partial, int n,
                                                                       float max_norm, float gscale,
                                                                       float* __restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) s += (double)partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += red[i];
    const float norm = (float)sqrt(t) * gscale;
    out[0] = norm;
    float coef = 1.0f;
    if (max_norm > 0.f) {
      coef = max_norm / (norm + 1e-6f);
      if (coef > 1.0f) coef = 1.0f;
    }
    out[1] = coef * gscale;
  }
}

// shadow <- shadow*decay + p*(1-decay) for every table entry (p = .p, shadow = .m); the reference's
// ExponentialMovingAverage.update (cifake_binary_classifier.py:222-225) as one launch.  12 B/parameter.
__global__ __launch_bounds__(256) void ema_kernel(const sgl_adamw_tensor* __restrict__ T,
                                                  const int32_t* __restrict__ map, float decay, float omd) {
  const int ti = map[2 * blockIdx.x], ch = map[2 * blockIdx.x + 1];
  const float* p = T[ti].p;
  float* sh = T[ti].m;
  const uint64_t n = T[ti].n;
  const uint64_t base = (uint64_t)ch * OPT_CHUNK;
  if (((((uintptr_t)p) | ((uintptr_t)sh)) & 15) == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint64_t i = base + (uint64_t)(k * 256 + threadIdx.x) * 4;
      if (i + 3 < n) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(p + i);
        const f32x4 y = *reinterpret_cast<const f32x4*>(sh + i);
        *reinterpret_cast<f32x4*>(sh + i) = y * decay + x * omd;
      } else {
        for (uint64_t j = i; j < n && j < i + 4; ++j) sh[j] = sh[j] * decay + p[j] * omd;
      }
    }
  } else {
    for (int k = 0; k < 16; ++k) {
      const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
      if (i < n) sh[i] = sh[i] * decay + p[i] * omd;
    }
  }
}

}  // namespace sgl

extern "C" {

int sgl_op_ema(const sgl_adamw_tensor* table, const int32_t* blockmap, int64_t nblocks, double decay,
               sgl_stream stream) {
  if (!table || !blockmap) return SGL_ERR_NULL;
  if (nblocks < 0 || nblocks > 0x7fffffff) return SGL_ERR_BAD_SHAPE;
  if (nblocks == 0) return SGL_OK;
  hipLaunchKernelGGL(sgl::ema_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, table, blockmap,
                     (float)decay, (float)(1.0 - decay));
  return hipGetLastError() == hipSuccess ? SGL_OK : SGL_ERR_HIP;
}

int64_t sgl_adamw_plan(const uint64_t* numel, int ntensors, int32_t* blockmap, int64_t capacity_pairs) {
  if (!numel || ntensors < 0) return SGL_ERR_NULL;
  int64_t nb = 0;
  for (int t = 0; t < ntensors; ++t) {
    const uint64_t chunks = (numel[t] + sgl::OPT_CHUNK - 1) / sgl::OPT_CHUNK;
    for (uint64_t c = 0; c < chunks; ++c, ++nb) {
      if (blockmap && nb < capacity_pairs) {
        blockmap[2 * nb] = t;
        blockmap[2 * nb + 1] = (int32_t)c;
      }
    }
  }
  return nb;
}

int sgl_op_grad_norm(const sgl_adamw_tensor* table, const int32_t* blockmap, int64_t nblocks, float max_norm,
                     float* partials, float* norm_and_coef, sgl_stream stream) {
  if (!table || !blockmap || !partials || !norm_and_coef) return SGL_ERR_NULL;
  if (nblocks < 0 || nblocks > 0x7fffffff) return SGL_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (nblocks > 0) hipLaunchKernelGGL(sgl::grad_sqnorm_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, table, blockmap, partials);
  hipLaunchKernelGGL(sgl::grad_norm_finish_kernel, dim3(1), dim3(1024), 0, s, partials, (int)nblocks, max_norm,
                     norm_and_coef);
  return hipGetLastError() == hipSuccess ? SGL_OK : SGL_ERR_HIP;
}


int sgl_op_grad_norm_scaled(const sgl_adamw_tensor* table, const int32_t* blockmap, int64_t nblocks, float max_norm,
                            float grad_scale, float* partials, float* norm_and_coef, sgl_stream stream) {
  if (!table || !blockmap || !partials || !norm_and_coef) return SGL_ERR_NULL;
  if (nblocks < 0 || nblocks > 0x7fffffff) return SGL_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (nblocks > 0) hipLaunchKernelGGL(sgl::grad_sqnorm_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, table, blockmap, partials);
  hipLaunchKernelGGL(sgl::grad_norm_finish_scaled_kernel, dim3(1), dim3(1024), 0, s, partials, (int)nblocks, max_norm,
                     grad_scale, norm_and_coef);
  return hipGetLastError() == hipSuccess ? SGL_OK : SGL_ERR_HIP;
}

int sgl_op_adamw_ex(const sgl_adamw_tensor* table, const sgl_adamw_aux* aux, const int32_t* blockmap, int64_t nblocks,
                    double beta1, double beta2, double eps, int step, const float* norm_and_coef,
                    const float* group_lr_wd_host, int ngroups, double ema_decay, sgl_stream stream) {
  if (!table || !aux || !blockmap) return SGL_ERR_NULL;
  if (nblocks < 0 || nblocks > 0x7fffffff || step < 1 || ngroups < 0 || ngroups > 16) return SGL_ERR_BAD_SHAPE;
  if (ngroups > 0 && !group_lr_wd_host) return SGL_ERR_NULL;
  if (nblocks == 0) return SGL_OK;
  sgl::AdamConst c;
  c.omb1 = (float)(1.0 - beta1);
  c.beta2 = (float)beta2;
  c.omb2 = (float)(1.0 - beta2);
  c.eps = (float)eps;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  c.inv_bc1 = (float)(1.0 / bc1);
  c.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  sgl::GroupHyper gh;
  gh.n = ngroups;
  for (int i = 0; i < 16; ++i) {
    gh.lr[i] = i < ngroups ? group_lr_wd_host[2 * i] : 0.f;
    gh.wd[i] = i < ngroups ? group_lr_wd_host[2 * i + 1] : 0.f;
  }
  hipLaunchKernelGGL(sgl::adamw_ex_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, table, aux,
                     blockmap, c, gh, norm_and_coef, (float)ema_decay);
  return hipGetLastError() == hipSuccess ? SGL_OK : SGL_ERR_HIP;
}

int sgl_op_adamw(const sgl_adamw_tensor* table, const int32_t* blockmap, int64_t nblocks, double beta1, double beta2,
                 double eps, int step, const float* norm_and_coef, sgl_stream stream) {
  if (!table || !blockmap) return SGL_ERR_NULL;
  if (nblocks < 0 || nblocks > 0x7fffffff || step < 1) return SGL_ERR_BAD_SHAPE;
  if (nblocks == 0) return SGL_OK;
  sgl::AdamConst c;
  // every scalar is formed in double and rounded once, as torch does with its Python-float hyper-parameters
  // (1 - 0.999f evaluated in fp32 is off by 1.3e-5 relative, which would show up in exp_avg_sq)
  c.omb1 = (float)(1.0 - beta1);
  c.beta2 = (float)beta2;
  c.omb2 = (float)(1.0 - beta2);
  c.eps = (float)eps;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  c.inv_bc1 = (float)(1.0 / bc1);
  c.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  hipLaunchKernelGGL(sgl::adamw_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, table, blockmap, c,
                     norm_and_coef);
  return hipGetLastError() == hipSuccess ? SGL_OK : SGL_ERR_HIP;
}

}  // extern "C"

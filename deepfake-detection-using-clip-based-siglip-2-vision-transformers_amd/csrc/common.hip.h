// Shared device helpers for the gfx950 (MI355X / CDNA4) SigLIP-2 encoder kernels.
// Wavefront = 64 lanes everywhere; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgl {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define SGL_WAVE 64
#define SGL_LDS __attribute__((address_space(3)))

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---- wave-64 reductions (DPP/shuffle; every lane ends with the result) ------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- typed element access --------------------------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16> {
  static __device__ __forceinline__ float ld(const bf16* p) { return (float)*p; }
  static __device__ __forceinline__ void st(bf16* p, float v) { *p = (bf16)v; }
};

// load / store NV (4 or 8) consecutive elements as float; pointers must be NV*sizeof(T)-aligned
template <typename T, int NV> struct Vec;
// st_nt / ld_nt: streaming ("nt") cache policy for tensors that are larger than the L2s and touched once per kernel: an
// epilogue's outputs written with plain stores evict the weight / activation panels the NEXT tiles' main loops are reading
// (fc1+GELU: 1025 -> 916 us at B = 128 with nontemporal stores alone).
template <> struct Vec<float, 4> {
  static __device__ __forceinline__ void ld(const float* p, float* v) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  static __device__ __forceinline__ void ld_nt(const float* p, float* v) {
    f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  static __device__ __forceinline__ void st(float* p, const float* v) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  }
  static __device__ __forceinline__ void st_nt(float* p, const float* v) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
  }
};
template <> struct Vec<float, 8> {
  static __device__ __forceinline__ void ld(const float* p, float* v) {
    Vec<float, 4>::ld(p, v); Vec<float, 4>::ld(p + 4, v + 4);
  }
  static __device__ __forceinline__ void st(float* p, const float* v) {
    Vec<float, 4>::st(p, v); Vec<float, 4>::st(p + 4, v + 4);
  }
  static __device__ __forceinline__ void st_nt(float* p, const float* v) {
    Vec<float, 4>::st_nt(p, v); Vec<float, 4>::st_nt(p + 4, v + 4);
  }
};
template <> struct Vec<bf16, 4> {
  static __device__ __forceinline__ void ld(const bf16* p, float* v) {
    bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)t[i];
  }
  static __device__ __forceinline__ void st(bf16* p, const float* v) {
    bf16x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = (bf16)v[i];
    *reinterpret_cast<bf16x4*>(p) = t;
  }
  static __device__ __forceinline__ void st_nt(bf16* p, const float* v) {
    bf16x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = (bf16)v[i];
    __builtin_nontemporal_store(t, reinterpret_cast<bf16x4*>(p));
  }
};
template <> struct Vec<bf16, 8> {
  static __device__ __forceinline__ void ld(const bf16* p, float* v) {
    bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
  }
  static __device__ __forceinline__ void st(bf16* p, const float* v) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
    *reinterpret_cast<bf16x8*>(p) = t;
  }
  static __device__ __forceinline__ void st_nt(bf16* p, const float* v) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
    __builtin_nontemporal_store(t, reinterpret_cast<bf16x8*>(p));
  }
};

// ---- GELU (tanh form, 'gelu_pytorch_tanh') and its derivative ---------------------------------------
// 0.5 x (1 + tanh(z)) == x * sigmoid(2z),  z = c (x + a x^3).  Written with one v_exp_f32 and one v_rcp_f32 (both
// hardware transcendentals, ~1 ulp) and no IEEE division, because these run in GEMM epilogues where VALU time is
// exposed.  sigmoid(2z) = 1 / (1 + exp2(w)),  w = -2z*log2(e) = x * (k1 + k2 x^2); saturates to 0 / 1 without NaNs.
#define SGL_GELU_C 0.7978845608028654f
#define SGL_GELU_A 0.044715f
#define SGL_LOG2E 1.4426950408889634f
__device__ __forceinline__ float gelu_sigmoid(float x, float x2) {
  const float k1 = -2.0f * SGL_GELU_C * SGL_LOG2E, k2 = -2.0f * SGL_GELU_C * SGL_GELU_A * SGL_LOG2E;
  const float e = __builtin_amdgcn_exp2f(x * fmaf(k2, x2, k1));
  return __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_tanh(float x) { return x * gelu_sigmoid(x, x * x); }
// d/dx [x s(2z)] = s + x s (1 - s) (2z)',   (2z)' = 2c (1 + 3 a x^2)
__device__ __forceinline__ float gelu_tanh_grad(float x) {
  const float x2 = x * x;
  const float s = gelu_sigmoid(x, x2);
  const float zp = fmaf(6.0f * SGL_GELU_C * SGL_GELU_A, x2, 2.0f * SGL_GELU_C);
  return fmaf(x * (s * (1.0f - s)), zp, s);
}

// gelu(x) and d gelu / dx from one sigmoid evaluation (the forward epilogue can save the derivative instead of x)
__device__ __forceinline__ void gelu_tanh_both(float x, float& a, float& g) {
  const float x2 = x * x;
  const float s = gelu_sigmoid(x, x2);
  const float zp = fmaf(6.0f * SGL_GELU_C * SGL_GELU_A, x2, 2.0f * SGL_GELU_C);
  a = x * s;
  g = fmaf(x * (s * (1.0f - s)), zp, s);     // same operations as gelu_tanh_grad
}

// ---- buffer resources (hardware bounds check: out-of-range loads return 0, stores are dropped) -------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
#define SGL_OOB 0x80000000u  // any voffset >= num_records reads as zero

}  // namespace sgl

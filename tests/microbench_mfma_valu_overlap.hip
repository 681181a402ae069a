// micro-benchmark: can VALU work of one wave overlap MFMA work of another wave on the same SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int OP>
__device__ __forceinline__ float vop(float v) {
  if constexpr (OP == 0) return fmaf(v, 1.0001f, 0.5f);
  else if constexpr (OP == 1) return __builtin_amdgcn_exp2f(v);
  else if constexpr (OP == 2) return __int_as_float((__float_as_int(v) ^ 0x5a5a) + 12345);
  else return fmaxf(v * 0.99f, -v);
}
template <int OP>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  const int w = threadIdx.x >> 6;
  const bool do_mfma = (mode == 0) || (mode == 2 && w < 4) || (mode == 3);
  const bool do_valu = (mode == 1) || (mode == 2 && w >= 4) || (mode == 3);
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x & 7); y[i] = (__bf16)1.0f; }
  float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
  if (mode == 3) {  // same wave does both, interleaved by the compiler
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v0 = vop<OP>(v0); v1 = vop<OP>(v1); v2 = vop<OP>(v2); v3 = vop<OP>(v3);
        v4 = vop<OP>(v4); v5 = vop<OP>(v5); v6 = vop<OP>(v6); v7 = vop<OP>(v7);
      }
    }
  } else {
    if (do_mfma)
      for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
      }
    if (do_valu)
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v0 = vop<OP>(v0); v1 = vop<OP>(v1); v2 = vop<OP>(v2); v3 = vop<OP>(v3);
          v4 = vop<OP>(v4); v5 = vop<OP>(v5); v6 = vop<OP>(v6); v7 = vop<OP>(v7);
        }
      }
  }
  float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 512 * 4);
  const int iters = 20000;
  const char* names[] = {"MFMA only, 8 waves/CU (2/SIMD): 4 MFMA32 per iter", "VALU only, 8 waves/CU: 32 v_fma per iter",
                         "waves 0-3 MFMA, waves 4-7 VALU (1+1 per SIMD)", "every wave both (compiler interleave)"};
  for (int op = 0; op < 4; ++op)
  for (int mode = 0; mode < 4; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto L = [&](int it) { if (op == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, it, mode); else if (op == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, it, mode); else if (op == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, it, mode); else hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, it, mode); };
    L(100);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    L(iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("op %d mode %d: %8.3f ms  (%s)\n", op, mode, ms, names[mode]);
  }
  return 0;
}

// micro-benchmark: does a store-bound (and VALU-heavy) epilogue running in one wave overlap MFMA work of the OTHER wave
// on the same SIMD?  (The premise of hiding a GEMM's epilogue behind a co-resident tile's main loop.)
//   hipcc --offload-arch=gfx950 -O3 tests/microbench_mfma_store_overlap.hip -o /tmp/mso && /tmp/mso
// One 512-thread workgroup per CU (160 KiB of LDS requested so that no second workgroup fits), 256 workgroups.
//   role M (waves 0-3, one per SIMD): ITERS x 8 back-to-back v_mfma_f32_16x16x32_bf16 on 8 accumulators
//   role S (waves 4-7, the SIMD partners): streams BYTES_PER_WAVE of 16-byte/lane stores (a GEMM epilogue's store phase),
//          optionally with G tanh-GELU evaluations per stored element group (the fc1 epilogue's VALU work)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float gelu_like(float x) {
  const float e = __builtin_amdgcn_exp2f(x * (-2.3022f - 0.1029f * x * x));
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

__global__ __launch_bounds__(512) void k(float* out, f32x4* sink, int iters, int store_iters, int mode, int gelu) {
  extern __shared__ char lds[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool role_m = (w < 4) && (mode & 1);
  const bool role_s = (w >= 4) && (mode & 2);
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)((threadIdx.x + i) & 7); y[i] = (__bf16)1.0f; }
  if (role_m) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[j], 0, 0, 0);
    }
  }
  float s = 0.f;
  if (role_s) {
    // each storing wave owns a private 1 KiB-per-instruction stream; consecutive instructions write consecutive KiB
    f32x4* base = sink + ((size_t)blockIdx.x * 4 + (w - 4)) * (size_t)store_iters * 64 + lane;
    f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
    for (int i = 0; i < store_iters; ++i) {
      if (gelu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = gelu_like(v[j] + (float)i * 1e-3f);   // 4 gelu per 16 stored bytes... x2 below
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = gelu_like(v[j] * 0.5f + 0.25f);
      }
      base[(size_t)i * 64] = v;
    }
    s = v[0];
  }
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  if (lds[threadIdx.x] == 77) s += 1.f;
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  const int ITERS = 4000;          // 32000 MFMA16 per M-wave  = 512 k cycles of matrix pipe per SIMD
  const int STORE_ITERS = 2048;    // 2 MiB per storing wave, 8 MiB per CU, 2 GiB per launch
  float* out; f32x4* sink;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&sink, (size_t)256 * 4 * STORE_ITERS * 64 * 16);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"", "M only   (waves 0-3 MFMA, partners idle)", "S only   (waves 4-7 store, partners idle)",
                         "M + S    (MFMA in waves 0-3, stores in their SIMD partners)"};
  for (int gelu = 0; gelu < 2; ++gelu)
    for (int mode = 1; mode <= 3; ++mode) {
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 160 * 1024, 0, out, sink, ITERS, STORE_ITERS, mode, gelu);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 160 * 1024, 0, out, sink, ITERS, STORE_ITERS, mode, gelu);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      const double tf = (mode & 1) ? 256.0 * 4 * ITERS * 8 * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12 : 0.0;
      const double tb = (mode & 2) ? 256.0 * 4 * STORE_ITERS * 1024.0 / (ms * 1e-3) / 1e12 : 0.0;
      printf("gelu=%d  %-62s %8.3f ms   %7.1f TFLOP/s   %5.2f TB/s stored\n", gelu, names[mode], ms, tf, tb);
    }
  return 0;
}

// micro-benchmark: issue cost (shader cycles per instruction, one wave per SIMD and 2 per SIMD) of global->LDS DMA
// (buffer_load_dwordx4 ... lds) against plain buffer_load_dwordx4, for in-cache and out-of-range addresses.
//   hipcc --offload-arch=gfx950 -O3 tests/microbench_dma_issue.hip -o build/mdi && build/mdi
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void dma16(u32x4 desc, uint32_t lds_addr, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(desc) : "memory");
}
__device__ __forceinline__ void dma4(u32x4 desc, uint32_t lds_addr, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(desc) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(512) void k(const char* src, long long* cyc, float* out, int iters, int oob, int waves) {
  extern __shared__ char lds[];
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint64_t q = (uint64_t)src;
  const u32x4 d = {(uint32_t)q, (uint32_t)(q >> 32) & 0xffffu, 1u << 20, 0x00020000u};
  const uint32_t lds0 = (uint32_t)(size_t)((__attribute__((address_space(3))) char*)lds) + w * 4096;
  uint32_t voff = oob ? 0x80000000u : (uint32_t)(lane * 16 + w * 1024);
  u32x4 acc = {0, 0, 0, 0};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (w < waves) {
    for (int i = 0; i < iters; ++i) {
      if constexpr (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dma16(d, lds0 + j * 1024, voff);
      } else if constexpr (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dma4(d, lds0 + j * 256, voff);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          u32x4 v;
          asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(d) : "memory");
          asm volatile("" :: "v"(v));
        }
      }
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + threadIdx.x] = (float)acc[0] + lds[threadIdx.x];
  if (lane == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
}
int main() {
  char* src; long long* cyc; float* out;
  hipMalloc(&src, 1 << 21); hipMemset(src, 1, 1 << 21); hipMalloc(&cyc, 64); hipMalloc(&out, 256 * 512 * 4);
  const int iters = 2000;
  const char* names[] = {"DMA dwordx4 -> LDS", "DMA dword -> LDS  ", "buffer_load_dwordx4 -> VGPR"};
  for (int grid : {1, 256}) for (int waves : {4, 8}) for (int oob = 0; oob < 2; ++oob) for (int mode = 0; mode < 3; ++mode) {
    for (int r = 0; r < 2; ++r) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 65536, 0, src, cyc, out, iters, oob, waves);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 65536, 0, src, cyc, out, iters, oob, waves);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 65536, 0, src, cyc, out, iters, oob, waves);
      hipDeviceSynchronize();
    }
    long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("grid %3d  %d waves/CU  %-28s %s : %6.1f cycles per instruction per wave (wave 0), %6.1f (wave %d)\n", grid, waves, names[mode],
           oob ? "out of range" : "L1-resident ", (double)h[0] / iters / 4, (double)h[waves - 1] / iters / 4, waves - 1);
  }
  return 0;
}

// micro-benchmark: LDS read throughput per CU of the two fragment-read instructions the GEMM main loops are made of,
// ds_read_b128 (NT: k-contiguous operands) and ds_read_b64_tr_b16 (TN: transposed reads), with the kernels' own conflict-free
// address patterns, 8 waves per workgroup as in gemm_nt6 / gemm_tn6.  Question (round 3): is gemm_tn6's 68 % matrix-pipe
// utilisation an LDS-bandwidth bound?  A K-step moves 8 waves x 24 KiB of fragments either way.
//   hipcc --offload-arch=gfx950 -O3 tests/microbench_lds_read_rates.hip -o build/mlr && build/mlr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// MODE 0: 24 x ds_read_b128 per iteration, 1: 48 x ds_read_b64_tr_b16, 2: 48 x ds_read_b64 (plain).
// PARTNER: waves 4-7 (the SIMD partners of waves 0-3) run a saturated v_mfma_f32_16x16x32_bf16 stream instead of reading,
// as in the anti-phase GEMM slots (4 reading waves next to 4 MFMA waves): bytes per iteration halve, the reported rate is
// that of the 4 reading waves.
template <int MODE, bool PARTNER, int PRIO_READ = 0, int PRIO_MFMA = 0>
__global__ __launch_bounds__(512) void k(long long* cyc, unsigned* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int i = t; i < 65536 / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  unsigned addr[8];
  if (MODE == 0) {   // NT image: 128-byte rows, chunk ((4h + lane>>4) ^ (row & 7)) << 4
    const int frow = lane & 15, fg = lane >> 4, fsw = frow & 7;
    for (int f = 0; f < 8; ++f)
      addr[f] = lds0 + (unsigned)((((w >> 1) & 1) * 128 + (f & 3) * 16 + frow) * 128 + ((((f >> 2) * 4 + fg) ^ fsw) << 4));
  } else {           // TN image: 512-byte rows, two transposed 8-byte reads per fragment
    const int fg = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
    const unsigned swz = 32 * fq + 128 * (fg & 1);
    const unsigned frow = (8 * fg + fq) * 512;
    for (int f = 0; f < 8; ++f) addr[f] = lds0 + frow + ((((w >> 1) & 1) * 256 + 8 * fp) ^ swz ^ (unsigned)(f * 32));
  }
  u32x4 acc4 = {0, 0, 0, 0};
  u32x2 acc2 = {0, 0};
  __syncthreads();
  if (PARTNER && w >= 4) {
    __builtin_amdgcn_s_setprio(PRIO_MFMA);
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)((lane * 7 + i) & 15) * 0.125f; y[i] = (__bf16)(0.5f + 0.01f * i); }
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters * 8; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float sacc = 0.f;
    for (int i = 0; i < 8; ++i) sacc += c[i][0] + c[i][3];
    out[blockIdx.x * 512 + t] = (unsigned)sacc;
    if (lane == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
    return;
  }
  __builtin_amdgcn_s_setprio(PRIO_READ);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          u32x4 v;
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr[f]), "n"(0));
          asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
          acc4 ^= v;
        }
    } else {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          u32x2 a, b;
          if (MODE == 1) {
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a) : "v"(addr[f]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(b) : "v"(addr[f]));
          } else {
            asm volatile("ds_read_b64 %0, %1" : "=v"(a) : "v"(addr[f]));
            asm volatile("ds_read_b64 %0, %1 offset:2048" : "=v"(b) : "v"(addr[f]));
          }
          asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
          acc2 ^= a;
          acc2 ^= b;
        }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + t] = acc4[0] ^ acc4[1] ^ acc4[2] ^ acc4[3] ^ acc2[0] ^ acc2[1];
  if (lane == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
}

int main() {
  long long* d_cyc; unsigned* d_out;
  hipMalloc(&d_cyc, 64); hipMalloc(&d_out, 512 * 4 * 256);
  const int iters = 20000;
  const char* names[3] = {"24 x ds_read_b128      ", "48 x ds_read_b64_tr_b16", "48 x ds_read_b64       "};
  for (int mode = 0; mode < 3; ++mode) {
    for (int grid = 1; grid <= 256; grid *= 256) {
      if (mode == 0) hipLaunchKernelGGL((k<0, false>), dim3(grid), dim3(512), 65536, 0, d_cyc, d_out, iters);
      if (mode == 1) hipLaunchKernelGGL((k<1, false>), dim3(grid), dim3(512), 65536, 0, d_cyc, d_out, iters);
      if (mode == 2) hipLaunchKernelGGL((k<2, false>), dim3(grid), dim3(512), 65536, 0, d_cyc, d_out, iters);
      hipDeviceSynchronize();
      long long h[8];
      hipMemcpy(h, d_cyc, 64, hipMemcpyDeviceToHost);
      long long mx = 0;
      for (int i = 0; i < 8; ++i) mx = h[i] > mx ? h[i] : mx;
      const double per_iter = (double)mx / iters;
      printf("%s  8 waves, %3d workgroup(s): %8.1f cycles per 24 KiB/wave iteration = %6.1f B/clk/CU  (a GEMM K-step has 2048 "
             "cycles of MFMA work per SIMD)\n", names[mode], grid, per_iter, 8.0 * 24576.0 / per_iter);
    }
  }
  for (int mode = 0; mode < 6; ++mode) {   // 4 reading waves next to 4 MFMA waves (one of each per SIMD); s_setprio variants
    if (mode == 0) hipLaunchKernelGGL((k<0, true>), dim3(1), dim3(512), 65536, 0, d_cyc, d_out, iters);
    if (mode == 1) hipLaunchKernelGGL((k<1, true>), dim3(1), dim3(512), 65536, 0, d_cyc, d_out, iters);
    if (mode == 2) hipLaunchKernelGGL((k<0, true, 0, 1>), dim3(1), dim3(512), 65536, 0, d_cyc, d_out, iters);
    if (mode == 3) hipLaunchKernelGGL((k<1, true, 0, 1>), dim3(1), dim3(512), 65536, 0, d_cyc, d_out, iters);
    if (mode == 4) hipLaunchKernelGGL((k<0, true, 3, 1>), dim3(1), dim3(512), 65536, 0, d_cyc, d_out, iters);
    if (mode == 5) hipLaunchKernelGGL((k<1, true, 3, 1>), dim3(1), dim3(512), 65536, 0, d_cyc, d_out, iters);
    hipDeviceSynchronize();
    long long h[8];
    hipMemcpy(h, d_cyc, 64, hipMemcpyDeviceToHost);
    long long rd = 0, mf = 0;
    for (int i = 0; i < 4; ++i) rd = h[i] > rd ? h[i] : rd;
    for (int i = 4; i < 8; ++i) mf = h[i] > mf ? h[i] : mf;
    const double per_iter = (double)rd / iters;
    printf("[prio read/mfma %s] ", mode < 2 ? "0/0" : (mode < 4 ? "0/1" : "3/1"));
    printf("%s  4 reading waves + 4 MFMA waves: %8.1f cycles per 24 KiB/wave iteration = %6.1f B/clk/CU (%.1f cycles per read "
           "instruction and wave); MFMA partner: %.1f cycles per v_mfma_f32_16x16x32_bf16 (16.0 alone)\n", names[mode & 1],
           per_iter, 4.0 * 24576.0 / per_iter, per_iter / ((mode & 1) == 0 ? 24.0 : 48.0), (double)mf / (iters * 64.0));
  }
  return 0;
}

// micro-benchmark: issue cost in shader-clock cycles of the VALU instructions an attention softmax is made of, of the 32x32x16
// bf16 MFMA, and of both together — in ONE wave (MFMA followed by independent VALU) and in two waves that share a SIMD.
//   hipcc --offload-arch=gfx950 -O3 tests/microbench_issue_rates.hip -o build/mir && build/mir
// One workgroup only (no power throttling); cycles from s_memtime around an inline-asm loop whose instruction order the
// compiler cannot change (volatile asm).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(ca), "v"(cb));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#define MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(ca), "v"(cb));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
#define CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(ca));
#define MFMA(j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(x), "v"(y));
#define R4(M, b) M(b) M(b + 1) M(b + 2) M(b + 3)
#define R8(M, b) R4(M, b) R4(M, b + 4)
#define R16(M) R4(M, 0) R4(M, 4) R4(M, 8) R4(M, 12)

enum { T_FMA, T_EXP, T_MAX3, T_PKFMA, T_PKMUL, T_CVT, T_MFMA, T_MFMA_FMA4, T_MFMA_EXP4, T_MFMA_FMA8, T_MFMA_PK4, T_MFMA_CVT4,
       T_IDLE, T_N };

template <int T>
__device__ __forceinline__ void body(float (&v)[16], f32x2 (&p)[16], unsigned (&u)[16], f32x16 (&acc)[4], bf16x8 x, bf16x8 y,
                                     float ca, float cb, f32x2 pa, f32x2 pb) {
  if constexpr (T == T_FMA) { R16(FMA) }
  if constexpr (T == T_EXP) { R16(EXP) }
  if constexpr (T == T_MAX3) { R16(MAX3) }
  if constexpr (T == T_PKFMA) { R16(PKFMA) }
  if constexpr (T == T_PKMUL) { R16(PKMUL) }
  if constexpr (T == T_CVT) { R16(CVT) }
  if constexpr (T == T_MFMA) { R4(MFMA, 0) }
  if constexpr (T == T_MFMA_FMA4) { MFMA(0) R4(FMA, 0) MFMA(1) R4(FMA, 4) MFMA(2) R4(FMA, 8) MFMA(3) R4(FMA, 12) }
  if constexpr (T == T_MFMA_EXP4) { MFMA(0) R4(EXP, 0) MFMA(1) R4(EXP, 4) MFMA(2) R4(EXP, 8) MFMA(3) R4(EXP, 12) }
  if constexpr (T == T_MFMA_FMA8) { MFMA(0) R8(FMA, 0) MFMA(1) R8(FMA, 8) MFMA(2) R8(FMA, 0) MFMA(3) R8(FMA, 8) }
  if constexpr (T == T_MFMA_PK4) { MFMA(0) R4(PKFMA, 0) MFMA(1) R4(PKFMA, 4) MFMA(2) R4(PKFMA, 8) MFMA(3) R4(PKFMA, 12) }
  if constexpr (T == T_MFMA_CVT4) { MFMA(0) R4(CVT, 0) MFMA(1) R4(CVT, 4) MFMA(2) R4(CVT, 8) MFMA(3) R4(CVT, 12) }
}

template <int TA, int TB>
__global__ __launch_bounds__(512) void k(long long* cyc, float* out, int iters, int noisy) {
  const int w = threadIdx.x >> 6;
  float v[16];
  f32x2 p[16];
  unsigned u[16];
  f32x16 acc[4];
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x & 3); y[i] = (__bf16)0.5f; }
  if (noisy) {   // operands with random mantissas / signs: realistic toggle rate for the power question
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u; x[i] = (__bf16)(((int)(h >> 9) & 0xffff) * (1.0f / 32768.f) - 1.0f);
      h = h * 1664525u + 1013904223u; y[i] = (__bf16)(((int)(h >> 9) & 0xffff) * (1.0f / 32768.f) - 1.0f);
    }
  }
  for (int i = 0; i < 16; ++i) { v[i] = 0.001f * (threadIdx.x + i); p[i] = f32x2{v[i], -v[i]}; u[i] = 0; }
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  const float ca = 0.999f, cb = 1e-3f;
  const f32x2 pa = {0.999f, 0.998f}, pb = {1e-3f, 2e-3f};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (w < 4) { for (int i = 0; i < iters; ++i) body<TA>(v, p, u, acc, x, y, ca, cb, pa, pb); }
  else { for (int i = 0; i < iters; ++i) body<TB>(v, p, u, acc, x, y, ca, cb, pa, pb); }
  asm volatile("s_nop 0" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i] + p[i][0] + p[i][1] + (float)u[i];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == gridDim.x - 1) cyc[w] = t1 - t0;
}

static long long* d_cyc; static float* d_out; static int g_grid = 1; static int g_noisy = 0;
template <int TA, int TB>
static void run(const char* name, double instrs_a, double instrs_b) {
  const int iters = 20000;
  long long h[8];
  static hipEvent_t e0, e1;
  if (!e0) { hipEventCreate(&e0); hipEventCreate(&e1); }
  hipLaunchKernelGGL((k<TA, TB>), dim3(g_grid), dim3(512), 0, 0, d_cyc, d_out, iters, g_noisy);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<TA, TB>), dim3(g_grid), dim3(512), 0, 0, d_cyc, d_out, iters, g_noisy);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 3;
  hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double longest = (double)(h[0] > h[4] ? h[0] : h[4]);
  printf("%-46s %7.3f ms  %5.2f GHz  wave0 %7.1f cyc/iter", name, ms, longest / (ms * 1e6), (double)h[0] / iters);
  if (instrs_a > 0) printf(" (%5.2f /instr)", (double)h[0] / iters / instrs_a);
  if (TB != T_IDLE) {
    printf("   wave4 %7.1f cyc/iter", (double)h[4] / iters);
    if (instrs_b > 0) printf(" (%5.2f /instr)", (double)h[4] / iters / instrs_b);
  }
  printf("\n");
}

int main(int argc, char** argv) {
  if (argc > 1) g_grid = atoi(argv[1]);
  if (argc > 2) g_noisy = atoi(argv[2]);
  hipMalloc(&d_cyc, 64); hipMalloc(&d_out, (size_t)g_grid * 512 * 4);
  printf("grid = %d workgroups of 8 waves, %s MFMA operands\n", g_grid, g_noisy ? "random" : "constant");
  printf("-- one wave per SIMD, solo --\n");
  run<T_FMA, T_IDLE>("16 v_fma_f32", 16, 0);
  run<T_EXP, T_IDLE>("16 v_exp_f32", 16, 0);
  run<T_MAX3, T_IDLE>("16 v_max3_f32", 16, 0);
  run<T_PKFMA, T_IDLE>("16 v_pk_fma_f32", 16, 0);
  run<T_PKMUL, T_IDLE>("16 v_pk_mul_f32", 16, 0);
  run<T_CVT, T_IDLE>("16 v_cvt_pk_bf16_f32", 16, 0);
  run<T_MFMA, T_IDLE>("4 v_mfma_f32_32x32x16_bf16", 4, 0);
  printf("-- one wave: MFMA followed by independent VALU --\n");
  run<T_MFMA_FMA4, T_IDLE>("4 x (MFMA32 + 4 v_fma)", 0, 0);
  run<T_MFMA_FMA8, T_IDLE>("4 x (MFMA32 + 8 v_fma)", 0, 0);
  run<T_MFMA_EXP4, T_IDLE>("4 x (MFMA32 + 4 v_exp)", 0, 0);
  run<T_MFMA_PK4, T_IDLE>("4 x (MFMA32 + 4 v_pk_fma)", 0, 0);
  run<T_MFMA_CVT4, T_IDLE>("4 x (MFMA32 + 4 v_cvt_pk)", 0, 0);
  printf("-- two waves on one SIMD (wave0: A, wave4: B) --\n");
  run<T_MFMA, T_MFMA>("A = 4 MFMA32, B = 4 MFMA32", 4, 4);
  run<T_FMA, T_FMA>("A = 16 v_fma, B = 16 v_fma", 16, 16);
  run<T_MFMA, T_FMA>("A = 4 MFMA32, B = 16 v_fma", 4, 16);
  run<T_MFMA, T_EXP>("A = 4 MFMA32, B = 16 v_exp", 4, 16);
  run<T_MFMA, T_PKFMA>("A = 4 MFMA32, B = 16 v_pk_fma", 4, 16);
  run<T_MFMA, T_CVT>("A = 4 MFMA32, B = 16 v_cvt_pk", 4, 16);
  run<T_MFMA_FMA4, T_MFMA_FMA4>("A = B = 4 x (MFMA32 + 4 v_fma)", 0, 0);
  return 0;
}

// bf16 MFMA NT GEMM, generation 8 (experimental, opt-in with SGL_GEMM_GEN=8): 256x256 tile, K-step 64, FOUR waves — one
// per SIMD — each owning a 128x128 quadrant (256 accumulator registers, 512-register budget).
//
// Why: generation 6 (gemm_bf16_v2.hip) gives every wave a 128x64 tile, so a K-step moves 8 x 24 KiB = 192 KiB of fragments
// out of LDS for 2060 cycles of matrix work: the LDS pipe is as busy as the matrix pipe.  A 128x128 wave tile needs
// 4 x 32 KiB = 128 KiB (-33 %).  The price is one wave per SIMD: nobody else hides this wave's LDS latency, so the loop is
// software-pipelined by hand — the fragments of the NEXT k-half are requested one at a time between groups of four MFMAs of
// the current one (the matrix pipe runs asynchronously: a ds_read issued behind an MFMA executes under it).
//
// Per K-step t (stage t & 1 of a two-stage LDS ring, same XOR-swizzled image and DMA addressing as generation 6):
//     phase 0:  64 MFMAs of k-half 0   |  16 fragment reads of k-half 1 (stage t & 1)
//     s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier      -> tile t+1 has landed, everyone is done reading stage t & 1
//     phase 1:  64 MFMAs of k-half 1   |  16 DMA instructions of tile t+2 -> stage t & 1,  16 fragment reads of tile t+1
// One barrier per K-step; a tile has 1.5 K-steps (~3000 cycles) to land.
//
// MEASURED (B = 128, the encoder's eight NT shapes, same box as generation 6): correct on all shapes, main loops 1.15 PF
// against generation 6's 1.28 PF; with the DMA requests deleted (wrong results) 1.35 PF.  Two lessons: (1) with one wave per
// SIMD the 64 DMA instructions of a K-step — 1024 cycles of the CU's texture path (16 cycles each, in-order issue) — stall
// the very waves that feed the matrix pipe, and they can only be issued in the half of the K-step after the stage is
// released; generation 6 issues them from the group that is NOT in its MFMA slot; (2) even the DMA-free loop is only 5 %
// above generation 6: with real operands the chip's clock under matrix load caps both.  Kept opt-in
// for A/B runs; the four-wave epilogue is also slower (half as many waves do the same store work).
#include <stdlib.h>

#include "common.hip.h"
#include "epilogue.hip.h"
#include "kernels.h"

namespace sgl {

constexpr int V_BM = 256, V_BN = 256, V_BK = 64;
constexpr int V_OP = V_BM * V_BK * 2;   // 32 KiB per operand per stage
constexpr int V_STAGE = 2 * V_OP;
constexpr int V_LDS = 2 * V_STAGE;      // 128 KiB
constexpr int V_CT_LD = 260;

__device__ __forceinline__ void v8_dma16(u32x4 desc, uint32_t lds_addr, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(desc)
               : "memory");
}
__device__ __forceinline__ u32x4 v8_desc(const void* base, uint32_t bytes) {
  const uint64_t q = (uint64_t)base;
  return u32x4{(uint32_t)q, (uint32_t)(q >> 32) & 0xffffu, bytes, 0x00020000u};
}
// same XCD-aware "B-stationary" order as generation 6 (gemm_bf16_v2.hip: tile_of_local)
__device__ __forceinline__ bool v8_tile(int xcd, int v, int tiles_m, int tiles_n, int cgw, int& tm, int& tn) {
  const int r_lo = (xcd * tiles_m) >> 3, r_hi = ((xcd + 1) * tiles_m) >> 3;
  const int R = r_hi - r_lo;
  if (v >= R * tiles_n) return false;
  const int full = R * cgw;
  const int g = v / full, rem = v - g * full;
  const int w = (tiles_n - g * cgw < cgw) ? tiles_n - g * cgw : cgw;
  const int r = rem / w;
  tm = r_lo + r;
  tn = g * cgw + (rem - r * w);
  return true;
}

// The 64 accumulator tiles fill the AGPR file exactly; left to itself the register allocator gives some MFMAs a destination
// different from their source and then shuffles accumulators through VGPRs (~800 v_accvgpr moves per K-step).  Inline asm
// pins every accumulator to one AGPR tuple, updated in place.
#define SGL_V8_MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

#define SGL_V8_LDS_BARRIER()                               \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                          \
    asm volatile("" ::: "memory");                         \
  } while (0)

// accumulators -> LDS (64 tile rows per pass) -> row-contiguous chunks -> fused epilogue (four waves)
template <int EPI, typename TOut>
__device__ __forceinline__ void store_tile256_w4(char* smem, f32x4 (&acc)[8][8], int wr, int wc, int lane, int t, int m0,
                                                 int n0, int M, int N, const EpiParams& p) {
  float* ct = reinterpret_cast<float*>(smem);
  const int g = lane >> 4, c16 = lane & 15;
  constexpr int NV = (sizeof(TOut) == 4) ? 4 : 8;
  constexpr int CPR = V_BN / NV;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    if (pass == 0) __syncthreads(); else SGL_V8_LDS_BARRIER();
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ct[(wr * 32 + i2 * 16 + g * 4 + r) * V_CT_LD + wc * 128 + j * 16 + c16] = acc[2 * pass + i2][j][r];
    SGL_V8_LDS_BARRIER();
    // ct row q holds tile row (q >> 5) * 128 + pass * 32 + (q & 31)
#pragma unroll
    for (int q = 0; q < (64 * CPR) / 256; ++q) {
      const int c = t + q * 256;
      const int row = c / CPR, col = (c % CPR) * NV;
      const int grow = m0 + (row >> 5) * 128 + pass * 32 + (row & 31), gcol = n0 + col;
      if (grow < M && gcol < N) {
        float v[NV];
        Vec<float, NV>::ld(ct + row * V_CT_LD + col, v);
        epi_apply<EPI, TOut, NV>(p, grow, gcol, N, v);
      }
    }
  }
}

template <int EPI, typename TOut>
__global__ __launch_bounds__(256, 1) void gemm_nt8_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ B,
                                                          int ldb, int M, int N, int K, int tiles_m, int tiles_n,
                                                          EpiParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tile_m, tile_n;
  if (!v8_tile(blockIdx.x & 7, blockIdx.x >> 3, tiles_m, tiles_n, 8, tile_m, tile_n)) return;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int m0 = tile_m * V_BM, n0 = tile_n * V_BN;
  const int rows_a = (M - m0 < V_BM) ? M - m0 : V_BM;
  const int rows_b = (N - n0 < V_BN) ? N - n0 : V_BN;
  const u32x4 da = v8_desc(A + (size_t)m0 * lda, (uint32_t)(((size_t)(rows_a - 1) * lda + K) * 2));
  const u32x4 db = v8_desc(B + (size_t)n0 * ldb, (uint32_t)(((size_t)(rows_b - 1) * ldb + K) * 2));
  const uint32_t lds0 = (uint32_t)(size_t)((SGL_LDS char*)smem);

  // DMA plan: wave w moves rows [64w, 64w+64) of each operand image, 8 rows (1 KiB) per instruction; slot c of row r holds
  // source chunk c ^ (r & 7)
  const int drow = lane >> 3, dchunk = (lane & 7) ^ drow;
  uint32_t voffa[8], voffb[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int row = w * 64 + q * 8 + drow;
    voffa[q] = (row < rows_a) ? (uint32_t)(row * lda + dchunk * 8) * 2u : SGL_OOB;
    voffb[q] = (row < rows_b) ? (uint32_t)(row * ldb + dchunk * 8) * 2u : SGL_OOB;
  }
  const int nk = (K + V_BK - 1) / V_BK;
  const uint32_t ldsw = lds0 + (uint32_t)(w * 64) * 128u;
  auto dma1 = [&](int idx, int kt) {  // instruction idx (0-7: A, 8-15: B) of K-step kt; past the end: zeros
    const int k0 = kt * V_BK;
    const bool kok = (kt < nk) && (k0 + dchunk * 8 < K);
    const uint32_t sb = (uint32_t)((kt & 1) * V_STAGE);
    const int q = idx & 7;
    if (idx < 8)
      v8_dma16(da, ldsw + sb + (uint32_t)q * 1024u, (kok && voffa[q] != SGL_OOB) ? voffa[q] + (uint32_t)k0 * 2u : SGL_OOB);
    else
      v8_dma16(db, ldsw + sb + (uint32_t)(V_OP + q * 1024), (kok && voffb[q] != SGL_OOB) ? voffb[q] + (uint32_t)k0 * 2u : SGL_OOB);
  };

  const int frow = lane & 15, fg = lane >> 4, fsw = frow & 7;
  const uint32_t fa_base = (uint32_t)((wr * 128 + frow) * 128);
  const uint32_t fb_base = (uint32_t)(V_OP + (wc * 128 + frow) * 128);
  const uint32_t ch[2] = {(uint32_t)(((0 + fg) ^ fsw) << 4), (uint32_t)(((4 + fg) ^ fsw) << 4)};
  // fragment f of a k-half: f < 8 -> B column block f, else A row block f - 8
  auto ldfrag = [&](const char* stage, int f, int h) -> bf16x8 {
    const uint32_t off = (f < 8 ? fb_base + (uint32_t)f * 2048u : fa_base + (uint32_t)(f - 8) * 2048u) + ch[h];
    return *reinterpret_cast<const bf16x8*>(stage + off);
  };

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int i = 0; i < 16; ++i) dma1(i, 0);
#pragma unroll
  for (int i = 0; i < 16; ++i) dma1(i, 1);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // tile 0 has landed (tile 1's sixteen may stay in flight)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  bf16x8 f0[16], f1[16];   // fragments of k-half 0 / k-half 1 (index: see ldfrag)
#pragma unroll
  for (int f = 0; f < 16; ++f) f0[f] = ldfrag(smem, f, 0);

  for (int kt = 0; kt < nk; ++kt) {
    const char* cur = smem + (kt & 1) * V_STAGE;
    const char* nxt = smem + ((kt + 1) & 1) * V_STAGE;
    // ---- phase 0: MFMAs of k-half 0, fragment reads of k-half 1
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      f1[g] = ldfrag(cur, g, 1);
      const int i = g >> 1, jb = (g & 1) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        SGL_V8_MFMA(acc[i][jb + j], f0[8 + i], f0[jb + j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    // ---- phase 1: MFMAs of k-half 1, DMA of tile kt+2 into the stage just released, first fragments of tile kt+1
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      dma1(g, kt + 2);
      f0[g] = ldfrag(nxt, g, 0);
      const int i = g >> 1, jb = (g & 1) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        SGL_V8_MFMA(acc[i][jb + j], f1[8 + i], f1[jb + j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // trailing (all-zero) requests must land before
                                                                            // the LDS is reused; the last MFMAs (inline asm:
                                                                            // invisible to the hazard recogniser) have retired
  if (p.atomic == 77) return;                       // developer experiment (SGL_NT6_SKIP_EPI): main loop only
  store_tile256_w4<EPI, TOut>(smem, acc, wr, wc, lane, t, m0, n0, M, N, p);
}

template <int EPI, typename TOut>
static hipError_t launch_nt8(const bf16* A, int lda, const bf16* B, int ldb, int M, int N, int K, const EpiParams& p,
                             hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt8_kernel<EPI, TOut>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, V_LDS);
    if (e != hipSuccess) return e;
    attr = true;
  }
  const int tiles_m = (M + V_BM - 1) / V_BM, tiles_n = (N + V_BN - 1) / V_BN;
  const int grid = 8 * (((tiles_m + 7) / 8) * tiles_n);
  static const bool skip_epi = getenv("SGL_NT6_SKIP_EPI") != nullptr;
  EpiParams pp = p;
  if (skip_epi) pp.atomic = 77;
  hipLaunchKernelGGL((gemm_nt8_kernel<EPI, TOut>), dim3(grid), dim3(256), V_LDS, s, A, lda, B, ldb, M, N, K, tiles_m, tiles_n,
                     pp);
  return hipGetLastError();
}

// hipErrorNotSupported: outside this generation's envelope (the dispatcher then uses generation 6)
hipError_t gemm_nt8_bf16(const void* A_, int lda, const void* B_, int ldb, int M, int N, int K, int epi, int out_dtype,
                         const EpiParams& p, hipStream_t s) {
  const bf16* A = (const bf16*)A_;
  const bf16* B = (const bf16*)B_;
  if (K < 128 || M < 2048) return hipErrorNotSupported;
  switch (epi) {
    case EPI_STORE:
      if (out_dtype == DT_BF16 && !p.colsum) return launch_nt8<EPI_STORE, bf16>(A, lda, B, ldb, M, N, K, p, s);
      return hipErrorNotSupported;
    case EPI_RES_F32: return launch_nt8<EPI_RES_F32, float>(A, lda, B, ldb, M, N, K, p, s);
    case EPI_BIAS_GELU: return launch_nt8<EPI_BIAS_GELU, bf16>(A, lda, B, ldb, M, N, K, p, s);
    case EPI_QKV: return launch_nt8<EPI_QKV, bf16>(A, lda, B, ldb, M, N, K, p, s);
    default: return hipErrorNotSupported;
  }
}

}  // namespace sgl

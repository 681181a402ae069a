// bf16 MFMA GEMMs for gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate), 64-lane wavefronts.
//
//   gemm_nt : C[M,N]  = A[M,K] · B[N,K]ᵀ      forward projections and dX (with pre-transposed weight shadows)
//   gemm_tn : C[N1,N2] (+)= Σ_m A[m,N1]·B[m,N2]   weight gradients dW = dYᵀ·X, reduction over tokens
//
// Tile 128x128, K-step 64, 256 threads = 4 waves in 2x2, each wave 64x64 = 4x4 MFMA tiles (64 acc VGPRs).
// Operands are staged global -> registers (buffer_load_dwordx4, hardware bounds check supplies zeros for the
// M/K tails) -> LDS, double buffered with one barrier per K-step (loads for step t+1 are issued before the
// MFMAs of step t and written to the other buffer after them).
//
// LDS images (no padding; XOR swizzles so every wide read is bank-conflict free on the 64-bank LDS):
//   NT: [row][64 k] = 128-B rows; 16-B chunk c of row r is stored at chunk c ^ (r & 7); fragments are read with
//       ds_read_b128 (lane = row, 8 consecutive k).
//   TN: [m][128 n] = 256-B rows; chunk c of row m is stored at c ^ (2*(m&3) + 8*((m>>3)&1)); fragments are read
//       with ds_read_b64_tr_b16 (hardware transpose: 4 rows x 16 columns -> lane i gets column i), two reads per
//       8-element fragment, so the token (reduction) index becomes the MFMA k index without any transposed copy
//       of the activations in HBM.
// The accumulator tile is transposed through LDS (fp32, row stride 132) so that the fused epilogue (bias, GELU,
// residual, head-major QKV scatter, ...) works on 8 consecutive columns per lane with 16-byte global accesses.
//
// Roofline: MFMA-bound (arithmetic intensity K/2... >> machine balance); algorithmic FLOPs = 2*M*N*K.
#include <stdlib.h>

#include "common.hip.h"
#include "epilogue.hip.h"
#include "kernels.h"

namespace sgl {

constexpr int G_BM = 128, G_BN = 128, G_BK = 64;
constexpr int G_STAGE = (G_BM + G_BN) * G_BK * 2;  // 32 KiB per stage
constexpr int G_CT_LD = 132;                       // fp32 staging row stride (floats)
constexpr int G_LDS = (2 * G_STAGE > G_BM * G_CT_LD * 4) ? 2 * G_STAGE : G_BM * G_CT_LD * 4;

__device__ __forceinline__ u32x4 ldg128(__amdgpu_buffer_rsrc_t r, uint32_t voff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
}

// Shared epilogue: acc[i][j] (wave sub-tile) -> LDS fp32 tile -> row-contiguous 8-column chunks -> epi_apply.
template <int EPI, typename TOut>
__device__ __forceinline__ void store_tile(char* smem, f32x4 (&acc)[4][4], int wr, int wc, int lane, int t, int m0,
                                           int n0, int M, int N, const EpiParams& p) {
  float* ct = reinterpret_cast<float*>(smem);
  __syncthreads();  // all waves are done reading the operand stages
  const int g = lane >> 4, c16 = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ct[(wr * 64 + i * 16 + g * 4 + r) * G_CT_LD + wc * 64 + j * 16 + c16] = acc[i][j][r];
  __syncthreads();
  if constexpr (EPI == EPI_F32) {
    if (p.atomic) {
      // split-K: fp32 atomics shaped as 256 contiguous bytes per wave instruction (one dword per lane)
      float* outp = reinterpret_cast<float*>(p.out);
      const int w = t >> 6;
#pragma unroll 4
      for (int rr = 0; rr < 32; ++rr) {
        const int row = w + rr * 4;
        const int grow = m0 + row;
        if (grow >= M) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int col = lane + 64 * h;
          if (n0 + col < N) atomicAdd(outp + (size_t)grow * p.ldo + n0 + col, ct[row * G_CT_LD + col] * p.alpha);
        }
      }
      return;
    }
  }
  // fp32 outputs: 4 columns (16 B) per lane so a wave instruction covers two whole 512-B rows;
  // bf16 outputs: 8 columns (16 B) per lane, four 256-B rows per wave instruction
  constexpr int NV = (sizeof(TOut) == 4) ? 4 : 8;
  constexpr int CPR = G_BN / NV;
  float csum[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) csum[j] = 0.f;
#pragma unroll
  for (int q = 0; q < (G_BM * CPR) / 256; ++q) {
    const int c = t + q * 256;
    const int row = c / CPR, col = (c % CPR) * NV;
    const int grow = m0 + row, gcol = n0 + col;
    if (grow < M && gcol < N) {
      float v[NV];
      Vec<float, NV>::ld(ct + row * G_CT_LD + col, v);
      epi_apply<EPI, TOut, NV>(p, grow, gcol, N, v);
      if constexpr (EPI == EPI_GELU_BWD) {
#pragma unroll
        for (int j = 0; j < NV; ++j) csum[j] += v[j];
      }
    }
  }
  if constexpr (EPI == EPI_GELU_BWD) {
    if (p.colsum) {  // fused bias gradient (see gemm_bf16_v2.hip)
      __syncthreads();
      constexpr int GROUPS = 256 / CPR;
#pragma unroll
      for (int j = 0; j < NV; ++j) ct[(t / CPR) * G_BN + (t % CPR) * NV + j] = csum[j];
      __syncthreads();
      if (t < G_BN && n0 + t < N) {
        float s = 0.f;
#pragma unroll
        for (int gI = 0; gI < GROUPS; ++gI) s += ct[gI * G_BN + t];
        if (p.colsum_ld > 0)
          p.colsum[(size_t)(m0 >> 7) * p.colsum_ld + n0 + t] = s;
        else
          atomicAdd(p.colsum + n0 + t, s);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// NT
// ------------------------------------------------------------------------------------------------------
template <int EPI, typename TOut>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const bf16* __restrict__ A, int lda,
                                                         const bf16* __restrict__ B, int ldb, int M, int N, int K,
                                                         EpiParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = (N + G_BN - 1) / G_BN;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
  const int m0 = tile_m * G_BM, n0 = tile_n * G_BN;
  const int rows_a = (M - m0 < G_BM) ? M - m0 : G_BM;
  const int rows_b = (N - n0 < G_BN) ? N - n0 : G_BN;
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(A + (size_t)m0 * lda, (uint32_t)(((size_t)(rows_a - 1) * lda + K) * 2));
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(B + (size_t)n0 * ldb, (uint32_t)(((size_t)(rows_b - 1) * ldb + K) * 2));

  // staging assignment: 4 chunks of A and 4 of B per thread; chunk = (row, 8 consecutive k)
  const int srow = t >> 3, skc = t & 7;
  uint32_t a_off[4], b_off[4], l_off[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = srow + q * 32;
    a_off[q] = (row < rows_a) ? (uint32_t)(row * lda + skc * 8) * 2u : SGL_OOB;
    b_off[q] = (row < rows_b) ? (uint32_t)(row * ldb + skc * 8) * 2u : SGL_OOB;
    l_off[q] = (uint32_t)(row * 128 + ((skc ^ (row & 7)) << 4));
  }
  // fragment read offsets (k-step s adds (4*s) to the chunk index before the XOR)
  const int frow = lane & 15, fg = lane >> 4;
  uint32_t fa_row[4], fb_row[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    fa_row[i] = (uint32_t)((wr * 64 + i * 16 + frow) * 128);
    fb_row[i] = (uint32_t)(G_BM * 128 + (wc * 64 + i * 16 + frow) * 128);
  }
  const int fsw = frow & 7;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (K + G_BK - 1) / G_BK;
  u32x4 sa[4], sb[4];
  auto load_tile = [&](int kt) {
    const int k0 = kt * G_BK;
    const bool kok = (k0 + skc * 8) < K;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      sa[q] = ldg128(ra, kok ? a_off[q] + (uint32_t)k0 * 2u : SGL_OOB);
      sb[q] = ldg128(rb, kok ? b_off[q] + (uint32_t)k0 * 2u : SGL_OOB);
    }
  };
  auto store_stage = [&](int stage) {
    char* base = smem + stage * G_STAGE;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<u32x4*>(base + l_off[q]) = sa[q];
      *reinterpret_cast<u32x4*>(base + G_BM * 128 + l_off[q]) = sb[q];
    }
  };

  load_tile(0);
  store_stage(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const char* base = smem + cur * G_STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint32_t coff = (uint32_t)(((4 * s + fg) ^ fsw) << 4);
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(base + fa_row[i] + coff);
        bfr[i] = *reinterpret_cast<const bf16x8*>(base + fb_row[i] + coff);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_stage(cur ^ 1);
    __syncthreads();
  }
  store_tile<EPI, TOut>(smem, acc, wr, wc, lane, t, m0, n0, M, N, p);
}

// ------------------------------------------------------------------------------------------------------
// TN
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x4 lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((SGL_LDS bf16x4*)(p));
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16* __restrict__ A, int lda,
                                                         const bf16* __restrict__ B, int ldb, int Mred, int N1, int N2,
                                                         int m_per_split, EpiParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles2 = (N2 + G_BN - 1) / G_BN;
  const int tile1 = blockIdx.x / tiles2, tile2 = blockIdx.x - tile1 * tiles2;
  const int n1_0 = tile1 * G_BM, n2_0 = tile2 * G_BN;
  const int m_begin = blockIdx.y * m_per_split;
  const int m_end = (m_begin + m_per_split < Mred) ? m_begin + m_per_split : Mred;
  const int rows = m_end - m_begin;  // > 0 by construction
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(A + (size_t)m_begin * lda, (uint32_t)((size_t)rows * lda * 2));
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(B + (size_t)m_begin * ldb, (uint32_t)((size_t)rows * ldb * 2));

  // staging: tile = 64 rows (m) x 128 columns = 16 chunks per row; 4 chunks per thread per operand
  const int srow = t >> 4, scc = t & 15;
  uint32_t a_off[4], b_off[4], l_off[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = srow + q * 16;
    const int ca = n1_0 + scc * 8, cb = n2_0 + scc * 8;
    a_off[q] = (ca < N1) ? (uint32_t)(row * lda + ca) * 2u : SGL_OOB;
    b_off[q] = (cb < N2) ? (uint32_t)(row * ldb + cb) * 2u : SGL_OOB;
    l_off[q] = (uint32_t)(row * 256 + ((scc ^ (2 * (row & 3) + 8 * ((row >> 3) & 1))) << 4));
  }
  // transposed fragment reads: lane (g = lane>>4, q = (lane>>2)&3, pp = lane&3) supplies row 8g+q(+4), cols 4pp..
  const int fg = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
  const uint32_t swz = (uint32_t)(32 * fq + 128 * (fg & 1));
  const uint32_t frow = (uint32_t)((8 * fg + fq) * 256);
  uint32_t fa_col[4], fb_col[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    fa_col[i] = ((uint32_t)(wr * 128 + i * 32 + 8 * fp)) ^ swz;
    fb_col[i] = ((uint32_t)(wc * 128 + i * 32 + 8 * fp)) ^ swz;
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (rows + G_BK - 1) / G_BK;
  u32x4 sa[4], sb[4];
  auto load_tile = [&](int kt) {
    const uint32_t r0 = (uint32_t)kt * G_BK;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // rows past m_end fall outside the descriptor and read as zero
      sa[q] = ldg128(ra, a_off[q] == SGL_OOB ? SGL_OOB : a_off[q] + r0 * (uint32_t)lda * 2u);
      sb[q] = ldg128(rb, b_off[q] == SGL_OOB ? SGL_OOB : b_off[q] + r0 * (uint32_t)ldb * 2u);
    }
  };
  constexpr int OPB = G_BK * 256;  // bytes per operand per stage
  auto store_stage = [&](int stage) {
    char* base = smem + stage * G_STAGE;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<u32x4*>(base + l_off[q]) = sa[q];
      *reinterpret_cast<u32x4*>(base + OPB + l_off[q]) = sb[q];
    }
  };

  load_tile(0);
  store_stage(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const char* base = smem + cur * G_STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const char* ba = base + s * 32 * 256 + frow;
      const char* bb = ba + OPB;
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x4 alo = lds_tr16(ba + fa_col[i]);
        const bf16x4 ahi = lds_tr16(ba + 4 * 256 + fa_col[i]);
        const bf16x4 blo = lds_tr16(bb + fb_col[i]);
        const bf16x4 bhi = lds_tr16(bb + 4 * 256 + fb_col[i]);
        af[i] = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
        bfr[i] = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_stage(cur ^ 1);
    __syncthreads();
  }
  EpiParams pq = p;
  if (p.split_stride) pq.out = reinterpret_cast<float*>(p.out) + (size_t)blockIdx.y * p.split_stride;  // private slab
  store_tile<EPI_F32, float>(smem, acc, wr, wc, lane, t, n1_0, n2_0, N1, N2, pq);
}

// ------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------
// 256x256-tile LDS-DMA generation (gemm_bf16_v2.hip); used whenever the problem is large enough to fill the chip
hipError_t gemm_nt2_bf16(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int epi, int out_dtype,
                         const EpiParams& p, hipStream_t s);
hipError_t gemm_tn2_bf16(const void* A, int lda, const void* B, int ldb, int Mred, int N1, int N2, int m_per,
                         int splits, const EpiParams& p, hipStream_t s);
#ifdef SGL_AB   // developer A/B build (make AB=1): the measured-slower generations 7 / 8 and the "generation 1 everywhere" switch
// persistent generation 7 (gemm_bf16_v3.hip); hipErrorNotSupported = outside its envelope, fall back to generation 6
hipError_t gemm_nt7_bf16(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int epi, int out_dtype,
                         const EpiParams& p, hipStream_t s);
// four-wave generation 8 (gemm_bf16_v4.hip), opt-in with SGL_GEMM_GEN=8; hipErrorNotSupported = fall back to generation 6
hipError_t gemm_nt8_bf16(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int epi, int out_dtype,
                         const EpiParams& p, hipStream_t s);
// SGL_GEMM_GEN=1 forces the 128x128 register-staged kernels (A/B comparisons)
static int gemm_generation() {
  static int gen = -1;
  if (gen < 0) {
    const char* e = getenv("SGL_GEMM_GEN");
    gen = (e && e[0] == '1') ? 1 : 2;
  }
  return gen;
}
#else
static inline int gemm_generation() { return 2; }
#endif
static bool g_attr_done = false;

template <int EPI, typename TOut>
static hipError_t launch_nt(const bf16* A, int lda, const bf16* B, int ldb, int M, int N, int K, const EpiParams& p,
                            hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<EPI, TOut>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS);
    if (e != hipSuccess) return e;
    attr = true;
  }
  const int tiles = ((M + G_BM - 1) / G_BM) * ((N + G_BN - 1) / G_BN);
  hipLaunchKernelGGL((gemm_nt_kernel<EPI, TOut>), dim3(tiles), dim3(256), G_LDS, s, A, lda, B, ldb, M, N, K, p);
  return hipGetLastError();
}

hipError_t gemm_nt_bf16(const void* A_, int lda, const void* B_, int ldb, int M, int N, int K, int epi, int out_dtype,
                        const EpiParams& p, hipStream_t s) {
  if (M == 0 || N == 0) return hipSuccess;
  if ((lda % 8) || (ldb % 8) || (K % 8) || K <= 0) return hipErrorInvalidValue;
  if (epi != EPI_F32 && (N % 8)) return hipErrorInvalidValue;
  if ((size_t)M * lda * 2 >= (1ull << 32) || (size_t)N * ldb * 2 >= (1ull << 32)) return hipErrorInvalidValue;
#ifndef SGL_NT6_MIN_N
#define SGL_NT6_MIN_N 256
#endif
  if (gemm_generation() != 1 && M >= 2048 && N >= SGL_NT6_MIN_N) {
#ifdef SGL_AB
    // generation 7 (persistent tile loop, gemm_bf16_v3.hip): measured equal-or-slower than generation 6 on the encoder's
    // shapes (negative results); A/B build only: SGL_GEMM_GEN=7
    static const bool gen7 = getenv("SGL_GEMM_GEN") && atoi(getenv("SGL_GEMM_GEN")) == 7;
    if (gen7) {
      const hipError_t e = gemm_nt7_bf16(A_, lda, B_, ldb, M, N, K, epi, out_dtype, p, s);
      if (e != hipErrorNotSupported) return e;
    }
    static const bool gen8 = getenv("SGL_GEMM_GEN") && atoi(getenv("SGL_GEMM_GEN")) == 8;
    if (gen8) {
      const hipError_t e = gemm_nt8_bf16(A_, lda, B_, ldb, M, N, K, epi, out_dtype, p, s);
      if (e != hipErrorNotSupported) return e;
    }
#endif
    return gemm_nt2_bf16(A_, lda, B_, ldb, M, N, K, epi, out_dtype, p, s);
  }
  const bf16* A = (const bf16*)A_;
  const bf16* B = (const bf16*)B_;
#define SGL_CASE(E)                                                                   \
  case E:                                                                             \
    return out_dtype == DT_BF16 ? launch_nt<E, bf16>(A, lda, B, ldb, M, N, K, p, s)   \
                                : launch_nt<E, float>(A, lda, B, ldb, M, N, K, p, s);
  switch (epi) {
    SGL_CASE(EPI_STORE)
    case EPI_BIAS_GELU:
      return out_dtype == DT_BF16 ? launch_nt<EPI_BIAS_GELU, bf16>(A, lda, B, ldb, M, N, K, p, s)
                                  : launch_nt<EPI_BIAS_GELU, float>(A, lda, B, ldb, M, N, K, p, s);   // bf16x3 strict mode
    case EPI_QKV:
      return out_dtype == DT_BF16 ? launch_nt<EPI_QKV, bf16>(A, lda, B, ldb, M, N, K, p, s)
                                  : launch_nt<EPI_QKV, float>(A, lda, B, ldb, M, N, K, p, s);   // bf16x3 strict mode
    case EPI_GELU_BWD:
      return out_dtype == DT_BF16 ? launch_nt<EPI_GELU_BWD, bf16>(A, lda, B, ldb, M, N, K, p, s)
                                  : launch_nt<EPI_GELU_BWD, float>(A, lda, B, ldb, M, N, K, p, s);   // bf16x3 strict mode
    case EPI_RES_F32: return launch_nt<EPI_RES_F32, float>(A, lda, B, ldb, M, N, K, p, s);
    case EPI_POS_F32: return launch_nt<EPI_POS_F32, float>(A, lda, B, ldb, M, N, K, p, s);
    case EPI_F32: return launch_nt<EPI_F32, float>(A, lda, B, ldb, M, N, K, p, s);
  }
#undef SGL_CASE
  return hipErrorInvalidValue;
}

hipError_t gemm_tn_bf16(const void* A_, int lda, const void* B_, int ldb, int Mred, int N1, int N2, int splits,
                        const EpiParams& p_, hipStream_t s, float* split_ws, size_t split_ws_bytes) {
  if (N1 == 0 || N2 == 0) return hipSuccess;
  if ((lda % 8) || (ldb % 8)) return hipErrorInvalidValue;
  if ((size_t)Mred * lda * 2 >= (1ull << 32) || (size_t)Mred * ldb * 2 >= (1ull << 32)) return hipErrorInvalidValue;
  if (!g_attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS);
    if (e != hipSuccess) return e;
    g_attr_done = true;
  }
  EpiParams p = p_;
  float* out = reinterpret_cast<float*>(p.out);
  if (Mred <= 0) {
    if (!p.accumulate)
      return hipMemset2DAsync(out, (size_t)p.ldo * sizeof(float), 0, (size_t)N2 * sizeof(float), N1, s);
    return hipSuccess;
  }
  if (gemm_generation() != 1 && N1 >= 512 && N2 >= 512 && Mred >= 2048) {
    // 256x256 tiles, one workgroup per CU: split the token reduction until ~256 workgroups exist
    // 256x256 tiles, one workgroup per CU (128 KiB LDS): keep tiles*splits <= 256 (a single full round)
    const int tiles = ((N1 + 255) / 256) * ((N2 + 255) / 256);
    int sp = 256 / tiles;
    if (sp < 1) sp = 1;
    const int max_sp = Mred / 1024 > 0 ? Mred / 1024 : 1;
    if (sp > max_sp) sp = max_sp;
    int mp = (Mred + sp - 1) / sp;
    mp = ((mp + G_BK - 1) / G_BK) * G_BK;
    sp = (Mred + mp - 1) / mp;
    if (sp > 1) {
      const size_t slab = (size_t)N1 * N2;
      if (split_ws && (size_t)sp * slab * sizeof(float) <= split_ws_bytes && p.alpha == 1.0f && !p.bias &&
          (N2 % 4 == 0) && (p.ldo % 4 == 0) && ((((uintptr_t)out) | ((uintptr_t)split_ws)) & 15) == 0) {
        // deterministic split-K: private slabs + a fixed-order reduction (also ~4x cheaper than the atomic epilogue)
        EpiParams q = p;
        q.out = split_ws;
        q.ldo = N2;
        q.accumulate = 0;
        q.atomic = 0;
        q.split_stride = slab;
        hipError_t e = gemm_tn2_bf16(A_, lda, B_, ldb, Mred, N1, N2, mp, sp, q, s);
        if (e != hipSuccess) return e;
        return reduce_splits(split_ws, sp, slab, N1, N2, out, p.ldo, p.accumulate, s);
      }
      if (!p.accumulate) {
        hipError_t e = hipMemset2DAsync(out, (size_t)p.ldo * sizeof(float), 0, (size_t)N2 * sizeof(float), N1, s);
        if (e != hipSuccess) return e;
      }
      p.atomic = 1;
    }
    return gemm_tn2_bf16(A_, lda, B_, ldb, Mred, N1, N2, mp, sp, p, s);
  }
  if (splits < 1) splits = 1;
  int m_per = (Mred + splits - 1) / splits;
  m_per = ((m_per + G_BK - 1) / G_BK) * G_BK;
  splits = (Mred + m_per - 1) / m_per;
  const int tiles = ((N1 + G_BM - 1) / G_BM) * ((N2 + G_BN - 1) / G_BN);
  if (splits > 1) {
    const size_t slab = (size_t)N1 * N2;
    if (split_ws && (size_t)splits * slab * sizeof(float) <= split_ws_bytes && p.alpha == 1.0f && !p.bias &&
        (N2 % 4 == 0) && (p.ldo % 4 == 0) && ((((uintptr_t)out) | ((uintptr_t)split_ws)) & 15) == 0) {
      EpiParams q = p;  // deterministic split-K, as in the 256x256 path above
      q.out = split_ws;
      q.ldo = N2;
      q.accumulate = 0;
      q.atomic = 0;
      q.split_stride = slab;
      hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, splits), dim3(256), G_LDS, s, (const bf16*)A_, lda,
                         (const bf16*)B_, ldb, Mred, N1, N2, m_per, q);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
      return reduce_splits(split_ws, splits, slab, N1, N2, out, p.ldo, p.accumulate, s);
    }
    if (!p.accumulate) {
      hipError_t e = hipMemset2DAsync(out, (size_t)p.ldo * sizeof(float), 0, (size_t)N2 * sizeof(float), N1, s);
      if (e != hipSuccess) return e;
    }
    p.atomic = 1;
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, splits), dim3(256), G_LDS, s, (const bf16*)A_, lda, (const bf16*)B_,
                     ldb, Mred, N1, N2, m_per, p);
  return hipGetLastError();
}

}  // namespace sgl

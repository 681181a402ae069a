// bf16 MFMA GEMMs for gfx950 on a 256x256 output tile, K-step 64, 512 threads = 8 waves, operands streamed
// global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction, no VGPR staging), two LDS stages
// (2 x 64 KiB).  Main loops (they share the images, tile order and epilogues):
//   nt6 / tn6: anti-phase wave groups, four barrier-separated slots per K-step (see the comment above them)
//   nt2 / tn2: developer A/B build only (make AB=1, SGL_GEMM_GEN=2): one barrier per K-step, the DMA for step t+1 is
//              issued right after the barrier that retires step t's DMA and runs under step t's 64 MFMAs per wave
//
//   nt : C[M,N]   = A[M,K] · B[N,K]ᵀ           (forward projections, dX with transposed weight shadows)
//   tn : C[N1,N2] (+)= Σ_m A[m,N1] · B[m,N2]    (dW; token index is the MFMA k index via ds_read_b64_tr_b16)
//
// Arithmetic intensity of the tile: 2*256*256*64 / (2*256*64*2 B) = 128 FLOP per LDS-staged byte (the 128x128
// tile of gemm_bf16.hip has 64 and is L2->LDS bandwidth bound near 0.6-0.7 PFLOP/s on this chip).
//
// LDS-DMA writes lane-linear (wave-uniform base + lane*16), so the bank-conflict swizzles are applied to each
// lane's SOURCE address and again on the fragment reads (same involution on both sides):
//   nt2 image [256 rows][64 k]   128-B rows: slot s of row r holds source chunk s ^ (r & 7)
//   tn2 image [64 m][256 n]      512-B rows: slot s of row m holds source chunk s ^ (2*(m&3) + 8*((m>>3)&1))
// Wave w -> (wr, wc) = ((w>>1)&1, (w&1) + 2*(w>>2)) so that the two waves sharing a SIMD (w, w+4) sit in different
// column halves: a half-empty last column tile (N = 1152 = 4.5 tiles) then costs half a tile, not a whole one.
// blockIdx -> tile is XCD-aware (unit_of_block / tile_of_unit below): each XCD works through a contiguous chunk of
// a grouped tile order, so its concurrent workgroups share A and B panels in its private L2.  Measured with
// rocprofv3 FETCH_SIZE on the fc1 shape: the earlier "row panel per XCD" map fetched 6x the algorithmic bytes
// because the 10 MB weight panel thrashed every 4 MiB L2.  Placement affects speed only.
#include <stdlib.h>
#ifdef SGL_TIMELINE
#include <stdio.h>
#include <string.h>
#endif

#include "common.hip.h"
#include "epilogue.hip.h"
#include "kernels.h"

namespace sgl {

constexpr int T_BM = 256, T_BN = 256, T_BK = 64;
constexpr int T_OP = T_BM * T_BK * 2;     // 32 KiB per operand per stage
constexpr int T_STAGE = 2 * T_OP;         // 64 KiB
constexpr int T_LDS = 2 * T_STAGE;        // 128 KiB
constexpr int T_CT_LD = 260;              // fp32 epilogue staging stride (64 rows x 260 floats = 66,560 B)

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds, uint32_t voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (SGL_LDS void*)lds, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ bf16x4 lds_tr16v2(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((SGL_LDS bf16x4*)(p));
}

// accumulators -> LDS (64 rows at a time) -> row-contiguous chunks -> fused epilogue
template <int EPI, typename TOut>
__device__ __forceinline__ void store_tile256(char* smem, f32x4 (&acc)[8][4], int wr, int wc, int lane, int t, int m0,
                                              int n0, int M, int N, const EpiParams& p) {
  float* ct = reinterpret_cast<float*>(smem);
  const int g = lane >> 4, c16 = lane & 15;
  constexpr int NV = (sizeof(TOut) == 4) ? 4 : 8;
  constexpr int CPR = T_BN / NV;
  float csum[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) csum[j] = 0.f;
  // Barriers here only order LDS traffic, so they are raw s_barrier + lgkmcnt(0): __syncthreads() would also emit
  // vmcnt(0) and make every pass wait for the previous pass's global stores to be acknowledged (microseconds under
  // load, 8 times per tile).  The stores stay in flight across passes instead.
#define SGL_LDS_BARRIER()                                  \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                          \
    asm volatile("" ::: "memory");                         \
  } while (0)
  // Per-thread constants of the chunk loop: 512 % CPR == 0, so a thread always owns the same NV columns.  Everything that
  // depends only on the column is loaded / decomposed ONCE here, and the per-chunk operands that come from memory (the fp32
  // residual, the saved pre-activation of GELU') are requested one pass ahead of their use, BEFORE the stores of the pass in
  // between: vmcnt retires in order, so a load issued behind a store cannot be waited for without waiting for that store's
  // acknowledgement first — with the loads inside the chunk loop (epi_apply) every chunk paid an HBM write round trip
  // (residual epilogue: 60 k cycles per tile against 25 k with this order; out_proj GEMM 410 -> 330 us at B = 128).
  constexpr int QN = (64 * CPR) / 512;     // chunks per thread and pass
  constexpr int RSTEP = 512 / CPR;         // ct rows between a thread's consecutive chunks
  const int trow = t / CPR, tcol = (t % CPR) * NV;
  const int gcol = n0 + tcol;
  const bool col_ok = gcol < N;
  constexpr bool HAS_BIAS = (EPI == EPI_BIAS_GELU || EPI == EPI_RES_F32 || EPI == EPI_QKV);
  constexpr bool FAST = HAS_BIAS || EPI == EPI_GELU_BWD;   // epilogues with an inlined chunk body below
  float bias[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) bias[j] = 0.f;
  if constexpr (HAS_BIAS) {
    if (col_ok) Vec<float, NV>::ld(p.bias + gcol, bias);
  }
  // QKV scatter: column -> (which, head, d) is fixed per thread; element offset of row (b, n) is qkv_c0 + (b*H*T + n) * P
  size_t qkv_c0 = 0;
  int qkv_pad = 0;                          // pad chunks this thread zeroes behind its head's last data chunk
  if constexpr (EPI == EPI_QKV) {
    const int dm = p.heads * p.head_dim;
    const int which = gcol / dm;
    const int hc = gcol - which * dm;
    const int h = hc / p.head_dim;
    const int d = hc - h * p.head_dim;
    qkv_c0 = ((size_t)which * p.batch * p.heads + h) * (size_t)p.tokens * p.head_dim_pad + d;
    qkv_pad = (d + NV == p.head_dim) ? (p.head_dim_pad - p.head_dim) / NV : 0;
  }
  using PreT = u32x4;                       // one raw 16-byte chunk: 4 fp32 residuals / 8 bf16 pre-activations
  constexpr bool HAS_PRE = (EPI == EPI_RES_F32 || EPI == EPI_GELU_BWD);
  PreT pre[2][QN];
  auto load_pre = [&](int pass, PreT (&dst)[QN]) {
    if constexpr (HAS_PRE) {
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        const int row = trow + q * RSTEP;
        const int grow = m0 + (row >> 5) * 128 + pass * 32 + (row & 31);
        const void* src = (EPI == EPI_RES_F32)
                              ? (const void*)(p.res + (size_t)grow * p.ldr + gcol)
                              : (const void*)(reinterpret_cast<const TOut*>(p.aux) + (size_t)grow * p.ldaux + gcol);
        dst[q] = (grow < M && col_ok) ? *reinterpret_cast<const PreT*>(src) : PreT{0u, 0u, 0u, 0u};
      }
    }
  };
  load_pre(0, pre[0]);
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    if (pass + 1 < 4) load_pre(pass + 1, pre[(pass + 1) & 1]);
    if (pass == 0) __syncthreads(); else SGL_LDS_BARRIER();
    // every wave contributes 32 of its 128 rows per pass (ct rows [32*wr, 32*wr+32)): LDS stores then come from both
    // SIMD halves at once (stores issued from SIMDs {0,1} only, as a "rows of wr==0 first" order would, run at half
    // rate: MI355X_MICROARCH.md, LDS section)
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ct[(wr * 32 + i2 * 16 + g * 4 + r) * T_CT_LD + wc * 64 + j * 16 + c16] = acc[2 * pass + i2][j][r];
    SGL_LDS_BARRIER();
    // ct row q holds tile row (q >> 5) * 128 + pass * 32 + (q & 31)
#define SGL_CT_GROW(q) (m0 + ((q) >> 5) * 128 + pass * 32 + ((q) & 31))
    if constexpr (EPI == EPI_F32) {
      if (p.atomic) {
        float* outp = reinterpret_cast<float*>(p.out);
        const int w = t >> 6;
#pragma unroll 2
        for (int rr = 0; rr < 8; ++rr) {
          const int row = w + rr * 8;
          const int grow = SGL_CT_GROW(row);
          if (grow >= M) continue;
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const int col = lane + 64 * h;
            if (n0 + col < N) atomicAdd(outp + (size_t)grow * p.ldo + n0 + col, ct[row * T_CT_LD + col] * p.alpha);
          }
        }
        continue;
      }
    }
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const int row = trow + q * RSTEP;
      const int grow = SGL_CT_GROW(row);
      if (grow < M && col_ok) {
        float v[NV];
        Vec<float, NV>::ld(ct + row * T_CT_LD + tcol, v);
        if constexpr (EPI == EPI_RES_F32) {
          const f32x4 r = __builtin_bit_cast(f32x4, pre[pass & 1][q]);
#pragma unroll
          for (int j = 0; j < NV; ++j) v[j] = r[j] + (v[j] + bias[j]);
          Vec<float, NV>::st_nt(reinterpret_cast<float*>(p.out) + (size_t)grow * p.ldo + gcol, v);
        } else if constexpr (EPI == EPI_BIAS_GELU) {
          float a[NV];
          if (p.out && p.gelu_grad_form) {   // training, derivative form: out := gelu'(u) (EpiParams::gelu_grad_form)
#pragma unroll
            for (int j = 0; j < NV; ++j) gelu_tanh_both(v[j] + bias[j], a[j], v[j]);
          } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
              v[j] += bias[j];
              a[j] = gelu_tanh(v[j]);
            }
          }
          // u / gelu'(u) is only read by the backward GELU': inference and frozen blocks pass out == nullptr
          if (p.out) Vec<TOut, NV>::st_nt(reinterpret_cast<TOut*>(p.out) + (size_t)grow * p.ldo + gcol, v);
          Vec<TOut, NV>::st_nt(reinterpret_cast<TOut*>(p.out2) + (size_t)grow * p.ldo2 + gcol, a);
        } else if constexpr (EPI == EPI_QKV) {
          // (Round 3 tried dealing a pass's chunks out head by head so that a wave store covers 6.4 whole 160-byte rows of
          // one head, 1 KiB contiguous instead of eight 144-byte pieces: QKV GEMM 853 -> 886 us, the per-chunk decode costs
          // more than the store pattern saves.  Not kept.)
          // row / tokens without the ~35-instruction integer division: float reciprocal estimate + one exact correction
          int b = (int)((float)grow * __builtin_amdgcn_rcpf((float)p.tokens));
          int n = grow - b * p.tokens;
          if (n < 0) { b -= 1; n += p.tokens; }
          else if (n >= p.tokens) { b += 1; n -= p.tokens; }
#pragma unroll
          for (int j = 0; j < NV; ++j) v[j] += bias[j];
          TOut* dst = reinterpret_cast<TOut*>(p.out) + qkv_c0 + ((size_t)b * p.heads * p.tokens + n) * p.head_dim_pad;
          Vec<TOut, NV>::st_nt(dst, v);
          if (qkv_pad) {   // zero the pad columns [head_dim, head_dim_pad): whole NV-chunks (head_dim % 8 == 0, pad = 0 or 8)
            float z[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) z[j] = 0.f;
            for (int k = 1; k <= qkv_pad; ++k) Vec<TOut, NV>::st_nt(dst + k * NV, z);
          }
        } else if constexpr (EPI == EPI_GELU_BWD) {
          float u[NV];
          if constexpr (NV == 8) {
            const bf16x8 ub = __builtin_bit_cast(bf16x8, pre[pass & 1][q]);
#pragma unroll
            for (int j = 0; j < NV; ++j) u[j] = (float)ub[j];
          } else {
            const f32x4 uf = __builtin_bit_cast(f32x4, pre[pass & 1][q]);
#pragma unroll
            for (int j = 0; j < NV; ++j) u[j] = uf[j];
          }
          if (p.gelu_grad_form) {
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] *= u[j];
          } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] *= gelu_tanh_grad(u[j]);
          }
          Vec<TOut, NV>::st_nt(reinterpret_cast<TOut*>(p.out) + (size_t)grow * p.ldo + gcol, v);
#pragma unroll
          for (int j = 0; j < NV; ++j) csum[j] += v[j];
        } else {
          epi_apply<EPI, TOut, NV>(p, grow, gcol, N, v);
        }
      }
    }
  }
#undef SGL_CT_GROW
  if constexpr (EPI == EPI_GELU_BWD) {
    // fused bias gradient: every thread always owns the same NV columns (512 % CPR == 0); fold the 512/CPR row
    // groups through LDS and add one value per column to p.colsum
    if (p.colsum) {
      SGL_LDS_BARRIER();
      constexpr int GROUPS = 512 / CPR;
#pragma unroll
      for (int j = 0; j < NV; ++j) ct[(t / CPR) * T_BN + (t % CPR) * NV + j] = csum[j];
      SGL_LDS_BARRIER();
      if (t < T_BN && n0 + t < N) {
        float s = 0.f;
#pragma unroll
        for (int gI = 0; gI < GROUPS; ++gI) s += ct[gI * T_BN + t];
        if (p.colsum_ld > 0)
          p.colsum[(size_t)(m0 >> 7) * p.colsum_ld + n0 + t] = s;
        else
          atomicAdd(p.colsum + n0 + t, s);
      }
    }
  }
}

#ifdef SGL_AB   // generation 2 (developer A/B build only: make AB=1)
// Compile-time interleave for one K-step: READS LDS reads per fragment (1 = ds_read_b128, 2 = two ds_read_b64_tr_b16).
// prologue 7 fragments (4 B + 3 A), then 16 x {4 MFMA, prefetch of A[f+3] (+ one B fragment of the 2nd k-half at f=3..6)}
template <int READS, int F>
__device__ __forceinline__ void sched_step() {
  __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
  if constexpr (F <= 12) __builtin_amdgcn_sched_group_barrier(0x100, READS, 0);
  if constexpr (F >= 3 && F < 7) __builtin_amdgcn_sched_group_barrier(0x100, READS, 0);
}
template <int READS>
__device__ __forceinline__ void sched_pipeline() {
  __builtin_amdgcn_sched_group_barrier(0x100, 7 * READS, 0);
  sched_step<READS, 0>();  sched_step<READS, 1>();  sched_step<READS, 2>();  sched_step<READS, 3>();
  sched_step<READS, 4>();  sched_step<READS, 5>();  sched_step<READS, 6>();  sched_step<READS, 7>();
  sched_step<READS, 8>();  sched_step<READS, 9>();  sched_step<READS, 10>(); sched_step<READS, 11>();
  sched_step<READS, 12>(); sched_step<READS, 13>(); sched_step<READS, 14>(); sched_step<READS, 15>();
}

#endif  // SGL_AB

// blockIdx -> work unit.  Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8 labels the XCD group), each
// with a private 4 MiB L2.  XCD x takes the CONTIGUOUS chunk [x*per, (x+1)*per) of a locality-ordered unit list, so
// the ~32 workgroups it runs at any moment are neighbours in that list.  Speed only; any placement is correct.
__device__ __forceinline__ int unit_of_block(int bid, int per) { return (bid & 7) * per + (bid >> 3); }

// NT tile order: bands of 8 row-tiles, column-major inside a band ("grouped" order): 32 consecutive tiles form an
// 8 x 4 block of the tile grid, i.e. 8 A panels + 4 B panels feed 32 tiles out of one L2.
__device__ __forceinline__ void tile_of_unit(int u, int tiles_m, int tiles_n, int& tile_m, int& tile_n, int bh = 8) {
  const int band = u / (bh * tiles_n);
  const int rows = (tiles_m - band * bh < bh) ? tiles_m - band * bh : bh;
  const int r = u - band * bh * tiles_n;
  tile_n = r / rows;
  tile_m = band * bh + (r - tile_n * rows);
}

// "B-stationary" alternative (band_h < 0 selects it, column-group width = -band_h): XCD x owns the contiguous range of row
// tiles [x*tiles_m/8, (x+1)*tiles_m/8) and walks it once per group of `cgw` column tiles — column group outermost, then
// rows, then the group's columns.  The ~32 workgroups an XCD runs at once are then (32/cgw) rows x cgw columns, and the
// NEXT 32 reuse the same cgw weight panels (4 x 590 KB at K = 1152: they stay in the XCD's 4-MiB L2) while only the
// activation panels stream: per 32 tiles 8 panel fetches instead of 12.  blockIdx b -> (XCD b & 7, local index b >> 3).
__device__ __forceinline__ bool tile_of_local(int xcd, int v, int tiles_m, int tiles_n, int cgw, int& tm, int& tn) {
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

#ifdef SGL_AB
// ------------------------------------------------------------------------------------------------------
template <int EPI, typename TOut>
__global__ __launch_bounds__(512, 2) void gemm_nt2_kernel(const bf16* __restrict__ A, int lda,
                                                          const bf16* __restrict__ B, int ldb, int M, int N, int K,
                                                          int tiles_m, int tiles_n, EpiParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntiles = tiles_m * tiles_n;
  const int unit = unit_of_block(blockIdx.x, (ntiles + 7) >> 3);
  if (unit >= ntiles) return;  // whole block exits together
  int tile_m, tile_n;
  tile_of_unit(unit, tiles_m, tiles_n, tile_m, tile_n);
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = (w >> 1) & 1, wc = (w & 1) + 2 * (w >> 2);
  const int m0 = tile_m * T_BM, n0 = tile_n * T_BN;
  const int rows_a = (M - m0 < T_BM) ? M - m0 : T_BM;
  const int rows_b = (N - n0 < T_BN) ? N - n0 : T_BN;
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(A + (size_t)m0 * lda, (uint32_t)(((size_t)(rows_a - 1) * lda + K) * 2));
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(B + (size_t)n0 * ldb, (uint32_t)(((size_t)(rows_b - 1) * ldb + K) * 2));

  // DMA assignment: wave w moves rows [w*32, w*32+32) of each operand, 8 rows (1 KiB) per instruction
  const int drow = lane >> 3;
  const int dchunk = (lane & 7) ^ drow;
  uint32_t a_off[4], b_off[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = w * 32 + q * 8 + drow;
    a_off[q] = (row < rows_a) ? (uint32_t)(row * lda + dchunk * 8) * 2u : SGL_OOB;
    b_off[q] = (row < rows_b) ? (uint32_t)(row * ldb + dchunk * 8) * 2u : SGL_OOB;
  }
  auto issue = [&](int kt, int stage) {
    const int k0 = kt * T_BK;
    const bool kok = (k0 + dchunk * 8) < K;
    char* base = smem + stage * T_STAGE + (w * 32) * 128;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      dma16(ra, base + q * 1024, (kok && a_off[q] != SGL_OOB) ? a_off[q] + (uint32_t)k0 * 2u : SGL_OOB);
      dma16(rb, base + T_OP + q * 1024, (kok && b_off[q] != SGL_OOB) ? b_off[q] + (uint32_t)k0 * 2u : SGL_OOB);
    }
  };

  const int frow = lane & 15, fg = lane >> 4, fsw = frow & 7;
  const uint32_t fa_base = (uint32_t)((wr * 128 + frow) * 128);
  const uint32_t fb_base = (uint32_t)(T_OP + (wc * 64 + frow) * 128);
  const bool active = (n0 + wc * 64 < N) && (m0 + wr * 128 < M);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (K + T_BK - 1) / T_BK;
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();  // drains this wave's DMA (vmcnt(0)) and orders every wave past step kt-1
    if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
    if (active) {
      // 16 fragment steps per K-step: step f = (k-half f>>3, 16-row tile f&7) feeds 4 MFMAs.  A fragments are read
      // three steps ahead, the second k-half's B fragments during steps 3..6, and sched_group_barrier pins the
      // ds_read/MFMA interleave so the LDS latency hides under the MFMAs of the same wave.
      const char* base = smem + (kt & 1) * T_STAGE;
      const char* pa = base + fa_base;
      const char* pb = base + fb_base;
      const uint32_t c0 = (uint32_t)(((0 + fg) ^ fsw) << 4), c1 = (uint32_t)(((4 + fg) ^ fsw) << 4);
      bf16x8 a[16], b[2][4];
#define SGL_LDA(f) (*reinterpret_cast<const bf16x8*>(pa + ((f) & 7) * 2048 + (((f) >> 3) ? c1 : c0)))
#define SGL_LDB(s_, j) (*reinterpret_cast<const bf16x8*>(pb + (j) * 2048 + ((s_) ? c1 : c0)))
#pragma unroll
      for (int j = 0; j < 4; ++j) b[0][j] = SGL_LDB(0, j);
      a[0] = SGL_LDA(0);
      a[1] = SGL_LDA(1);
      a[2] = SGL_LDA(2);
#pragma unroll
      for (int f = 0; f < 16; ++f) {
        if (f + 3 < 16) a[f + 3] = SGL_LDA(f + 3);
        if (f >= 3 && f < 7) b[1][f - 3] = SGL_LDB(1, f - 3);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[f & 7][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[f], b[f >> 3][j], acc[f & 7][j], 0, 0, 0);
      }
#undef SGL_LDA
#undef SGL_LDB
      sched_pipeline<1>();
    }
  }
  store_tile256<EPI, TOut>(smem, acc, wr, wc, lane, t, m0, n0, M, N, p);
}

#endif  // SGL_AB

// ------------------------------------------------------------------------------------------------------
// Anti-phase ("ping-pong") main loops, generation 6 (default).  The two wave groups of the workgroup (G0 = waves
// 0-3, G1 = waves 4-7; every SIMD holds one wave of each) alternate: time is cut into slots separated by workgroup
// barriers, and in every slot one group issues the LDS reads of its next half K-step (plus its share of the
// global->LDS DMA stream) while the other group issues 32 MFMAs, so each SIMD's matrix pipe always has exactly one
// wave feeding it and nobody's LDS latency is exposed.  G1 simply starts one barrier late.  Measured against the
// generation-2 loop (one barrier per K-step, rolling fragment prefetch) in the same process: +8..14 % on the
// encoder's NT shapes, 1.29 -> 1.45 PFLOP/s at 8192^3.  Variants tried and dropped: 8 slots of 16 MFMAs (+4 %; the
// barrier hand-off costs about as much per slot regardless of slot length), 2 slots of 64 MFMAs (needs all DMA
// issued one slot before use: the prefetch distance is too short, -12 %).
//
// LDS-DMA is issued through inline asm: with the builtin the compiler cannot prove that a ds_read_b64_tr_b16 does not
// alias the in-flight DMA and drains vmcnt(0) in front of every transposed read.  The kernels own the vmcnt
// accounting instead: every read slot ends with s_waitcnt vmcnt(8) (four younger 2-instruction units may stay in
// flight), and the schedule guarantees that (a) a unit is issued at least 4 slots = one K-step
// before the wait that publishes it, (b) both groups have passed that wait and a barrier before anyone reads it,
// (c) a unit is issued only after both groups finished (lgkmcnt(0) + barrier) reading the unit it overwrites.
__device__ __forceinline__ void pp_dma16(u32x4 desc, uint32_t lds_addr, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(desc) : "memory");
}
__device__ __forceinline__ u32x4 pp_desc(const void* base, uint32_t bytes) {
  const uint64_t q = (uint64_t)base;
  u32x4 d = {(uint32_t)q, (uint32_t)(q >> 32) & 0xffffu, bytes, 0x00020000u};
  return d;
}
#define SGL_PP_END_READ8()                                              \
  do {                                                                  \
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);                                  \
    __builtin_amdgcn_s_barrier();                                       \
    __builtin_amdgcn_sched_barrier(0);                                  \
  } while (0)
#define SGL_PP_END_MFMA()                 \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

// NT, four slots per K-step.  The global->LDS stream moves 16-KiB units in the fixed round-robin order
//     Aq02(t) = A rows {0-63,128-191},  BX(t) = first 32 columns of every wave column,  BY(t) = the other 32,
//     Aq13(t) = A rows {64-127,192-255},  Aq02(t+1), ...          (two wave-instructions per wave per unit)
// Per wave and K-step t:
//     R1(t): 16 fragment reads (A0 = rows 0-63 of the wave's 128, B0, B1 = all 64 columns) + DMA unit Aq13(t+1)
//     M1(t): A0 x (B0,B1)   32 MFMAs
//     R2(t):  8 fragment reads (A1 = rows 64-127)                 + DMA units Aq02(t+2), BX(t+2), BY(t+2)
//     M2(t): A1 x (B0,B1)   32 MFMAs
// G0 runs R1 in slot 4t, G1 in slot 4t+1.
// Developer build (make TIMELINE=1, then SGL_TIMELINE=1 in the environment): s_memtime stamps at kernel entry, end of the
// main loop and end of the epilogue, summed over all workgroups and printed per launch.  This is the tool behind the
// epilogue findings of DESIGN.md section 8.4, round 2 (loads queued behind stores; fc1's main loop slowed by its own output traffic).
#ifdef SGL_TIMELINE
__device__ unsigned long long g_nt6_tl[4];
__device__ unsigned long long g_tn6_tl[12];   // [group][phase 0..4, slot-pair count]
#define SGL_TL_STAMP(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define SGL_TL_STAMP(v)
#endif
template <int EPI, typename TOut>
__global__ __launch_bounds__(512, 2) void gemm_nt6_kernel(const bf16* __restrict__ A, int lda,
                                                          const bf16* __restrict__ B, int ldb, int M, int N, int K,
                                                          int tiles_m, int tiles_n, int band_h, EpiParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SGL_TL_STAMP(tl0);
  const int ntiles = tiles_m * tiles_n;
  int tile_m, tile_n;
  if (band_h < 0) {
    if (!tile_of_local(blockIdx.x & 7, blockIdx.x >> 3, tiles_m, tiles_n, -band_h, tile_m, tile_n)) return;
  } else {
    const int unit = unit_of_block(blockIdx.x, (ntiles + 7) >> 3);
    if (unit >= ntiles) return;  // whole block exits together
    tile_of_unit(unit, tiles_m, tiles_n, tile_m, tile_n, band_h);
  }
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = (w >> 1) & 1, wc = (w & 1) + 2 * (w >> 2), grp = w >> 2;
  const int m0 = tile_m * T_BM, n0 = tile_n * T_BN;
  const int rows_a = (M - m0 < T_BM) ? M - m0 : T_BM;
  const int rows_b = (N - n0 < T_BN) ? N - n0 : T_BN;
  const u32x4 da = pp_desc(A + (size_t)m0 * lda, (uint32_t)(((size_t)(rows_a - 1) * lda + K) * 2));
  const u32x4 db = pp_desc(B + (size_t)n0 * ldb, (uint32_t)(((size_t)(rows_b - 1) * ldb + K) * 2));
  const uint32_t lds0 = (uint32_t)(size_t)((SGL_LDS char*)smem);

  const int drow = lane >> 3, dchunk = (lane & 7) ^ drow;
  uint32_t voff[4][2], ldst[4][2];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ul = 16 * w + 8 * q;  // unit-local first row of the instruction (multiple of 8)
      int r0;
      if (u == 0) r0 = (ul < 64) ? ul : ul + 64;
      else if (u == 3) r0 = (ul < 64) ? ul + 64 : ul + 128;
      else r0 = (ul >> 5) * 64 + (ul & 31) + (u == 2 ? 32 : 0);
      const int row = r0 + drow;
      const bool isA = (u == 0 || u == 3);
      const int lim = isA ? rows_a : rows_b;
      const int ld = isA ? lda : ldb;
      voff[u][q] = (row < lim) ? (uint32_t)(row * ld + dchunk * 8) * 2u : SGL_OOB;
      ldst[u][q] = lds0 + (isA ? 0 : T_OP) + (uint32_t)r0 * 128u;
    }
  const int nk = (K + T_BK - 1) / T_BK;
  auto issue = [&](int u, int kt) {  // unit u of K-step kt (K-steps past the end: all lanes out of range -> zeros)
    const int k0 = kt * T_BK;
    const bool kok = (kt < nk) && (k0 + dchunk * 8 < K);
    const uint32_t sb = (uint32_t)((kt & 1) * T_STAGE);
    const u32x4 d = (u == 0 || u == 3) ? da : db;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      pp_dma16(d, ldst[u][q] + sb, (kok && voff[u][q] != SGL_OOB) ? voff[u][q] + (uint32_t)k0 * 2u : SGL_OOB);
  };

  const int frow = lane & 15, fg = lane >> 4, fsw = frow & 7;
  const uint32_t fa_base = (uint32_t)((wr * 128 + frow) * 128);
  const uint32_t fb_base = (uint32_t)(T_OP + (wc * 64 + frow) * 128);
  const uint32_t c0 = (uint32_t)(((0 + fg) ^ fsw) << 4), c1 = (uint32_t)(((4 + fg) ^ fsw) << 4);
  const bool active = (n0 + wc * 64 < N) && (m0 + wr * 128 < M);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0);
  issue(0, 1); issue(1, 1); issue(2, 1);
  SGL_PP_END_READ8();                // Aq02(0), BX(0), BY(0) of every wave have landed
  if (grp == 1) SGL_PP_END_MFMA();   // G1 runs one slot behind G0 from here on

  bf16x8 fa[8], fb[8];
  for (int kt = 0; kt < nk; ++kt) {
    const char* base = smem + (kt & 1) * T_STAGE;
    const char* pa = base + fa_base;
    const char* pb = base + fb_base;
    // ---- R1: A0 (row tiles 0-3), B (column tiles 0-3), both k-halves
    if (active) {
#pragma unroll
      for (int f = 0; f < 8; ++f) fb[f] = *reinterpret_cast<const bf16x8*>(pb + (f & 3) * 2048 + ((f >> 2) ? c1 : c0));
#pragma unroll
      for (int f = 0; f < 8; ++f) fa[f] = *reinterpret_cast<const bf16x8*>(pa + (f & 3) * 2048 + ((f >> 2) ? c1 : c0));
    }
    issue(3, kt + 1);
    SGL_PP_END_READ8();
    if (active) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[4 * h + i], fb[4 * h + j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    SGL_PP_END_MFMA();
    // ---- R2: A1 (row tiles 4-7)
    if (active) {
#pragma unroll
      for (int f = 0; f < 8; ++f) fa[f] = *reinterpret_cast<const bf16x8*>(pa + (4 + (f & 3)) * 2048 + ((f >> 2) ? c1 : c0));
    }
    issue(0, kt + 2); issue(1, kt + 2); issue(2, kt + 2);
    SGL_PP_END_READ8();
    if (active) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[4 * h + i], fb[4 * h + j], acc[4 + i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    SGL_PP_END_MFMA();
  }
  if (grp == 0) SGL_PP_END_MFMA();   // G0 waits for G1's last slot
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing (all-zero) units must land before the LDS is reused
#ifdef SGL_TIMELINE
  if (p.atomic == 77) return;   // developer build only (SGL_NT6_SKIP_EPI): main loop without its epilogue
#endif
  SGL_TL_STAMP(tl1);
  store_tile256<EPI, TOut>(smem, acc, wr, wc, lane, t, m0, n0, M, N, p);
#ifdef SGL_TIMELINE
  SGL_TL_STAMP(tl2);
  if (t == 0) {
    atomicAdd(&g_nt6_tl[0], tl1 - tl0);
    atomicAdd(&g_nt6_tl[1], tl2 - tl1);
    atomicAdd(&g_nt6_tl[2], 1ull);
  }
#endif
}

// ------------------------------------------------------------------------------------------------------
// TN (dW), four slots per K-step of 64 tokens, split by K-HALF: the token index is the MFMA k index, so the first 32
// image rows of both operands feed the first 32 MFMAs (every accumulator once) and the last 32 rows the other 32.
//     R1(t): k-half 0 fragments (8 A + 4 B, two transposed reads each) + DMA units U2(t+1), U3(t+1)
//     M1(t): 32 MFMAs      R2(t): k-half 1 fragments + DMA units U0(t+2), U1(t+2)      M2(t): 32 MFMAs
// units: U0 = A image rows 0-31, U1 = B rows 0-31, U2 = A rows 32-63, U3 = B rows 32-63 (16 KiB each).
__global__ __launch_bounds__(512, 2) void gemm_tn6_kernel(const bf16* __restrict__ A, int lda,
                                                          const bf16* __restrict__ B, int ldb, int Mred, int N1, int N2,
                                                          int m_per_split, int nsplits, int tiles_1, int tiles_2,
                                                          EpiParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntiles = tiles_1 * tiles_2;
  const int nunits = ntiles * nsplits;
  const int unit = unit_of_block(blockIdx.x, (nunits + 7) >> 3);
  if (unit >= nunits) return;
  const int split = unit / ntiles, trem = unit - split * ntiles;
  const int tile1 = trem / tiles_2, tile2 = trem - tile1 * tiles_2;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = (w >> 1) & 1, wc = (w & 1) + 2 * (w >> 2), grp = w >> 2;
  const int n1_0 = tile1 * T_BM, n2_0 = tile2 * T_BN;
  const int m_begin = split * m_per_split;
  const int m_end = (m_begin + m_per_split < Mred) ? m_begin + m_per_split : Mred;
  const int rows = m_end - m_begin;
  const u32x4 da = pp_desc(A + (size_t)m_begin * lda, (uint32_t)((size_t)rows * lda * 2));
  const u32x4 db = pp_desc(B + (size_t)m_begin * ldb, (uint32_t)((size_t)rows * ldb * 2));
  const uint32_t lds0 = (uint32_t)(size_t)((SGL_LDS char*)smem);

  // DMA plan: unit u = 2*khalf + operand; wave w moves image rows 32*khalf + 4w + {0..3}, 2 rows (1 KiB) per
  // instruction; slot s of row m holds source chunk s ^ (2*(m&3) + 8*((m>>3)&1)) (same image as tn2)
  const int dr2 = lane >> 5, dslot = lane & 31;
  uint32_t voff[4][2], ldst[4][2];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r0 = 32 * (u >> 1) + 4 * w + 2 * q;
      const int row = r0 + dr2;
      const int chunk = dslot ^ (2 * (row & 3) + 8 * ((row >> 3) & 1));
      const bool isA = (u & 1) == 0;
      const int col = (isA ? n1_0 : n2_0) + chunk * 8;
      const int lim = isA ? N1 : N2;
      const int ld = isA ? lda : ldb;
      voff[u][q] = (col < lim) ? (uint32_t)(row * ld + col) * 2u : SGL_OOB;
      ldst[u][q] = lds0 + (isA ? 0 : T_OP) + (uint32_t)r0 * 512u;
    }
  auto issue = [&](int u, int kt) {  // rows past the split's range are out of range for the descriptor -> zeros
    const uint32_t r0 = (uint32_t)kt * T_BK;
    const uint32_t sb = (uint32_t)((kt & 1) * T_STAGE);
    const bool isA = (u & 1) == 0;
    const u32x4 d = isA ? da : db;
    const uint32_t adv = r0 * (uint32_t)(isA ? lda : ldb) * 2u;
#pragma unroll
    for (int q = 0; q < 2; ++q) pp_dma16(d, ldst[u][q] + sb, voff[u][q] == SGL_OOB ? SGL_OOB : voff[u][q] + adv);
  };

  const int fg = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
  const uint32_t swz = (uint32_t)(32 * fq + 128 * (fg & 1));
  const uint32_t frow = (uint32_t)((8 * fg + fq) * 512);
  const uint32_t fa_col = ((uint32_t)(wr * 256 + 8 * fp)) ^ swz;
  const uint32_t fb_col = ((uint32_t)(wc * 128 + 8 * fp)) ^ swz;
  const bool active = (n2_0 + wc * 64 < N2) && (n1_0 + wr * 128 < N1);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (rows + T_BK - 1) / T_BK;
  issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0);
  issue(0, 1); issue(1, 1);
  SGL_PP_END_READ8();                // U0(0), U1(0) of every wave have landed
  if (grp == 1) SGL_PP_END_MFMA();   // G1 runs one slot behind G0 from here on

#define SGL_TR6(ptr) __builtin_shufflevector(lds_tr16v2(ptr), lds_tr16v2((ptr) + 4 * 512), 0, 1, 2, 3, 4, 5, 6, 7)
  bf16x8 fa[8], fb[4];
#ifdef SGL_TIMELINE
  // per-phase cycle sums of this wave over the main loop: [0] issue fragment reads + DMA, [1] wait for them (lgkmcnt/vmcnt),
  // [2] barrier after the read slot, [3] 32 MFMAs, [4] barrier after the MFMA slot
  unsigned long long tlp[5] = {0, 0, 0, 0, 0};
#define SGL_TLP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); tlp[i] += n_ - tl_last; tl_last = n_; } while (0)
  unsigned long long tl_last = __builtin_readcyclecounter();
#else
#define SGL_TLP(i)
#endif
  for (int kt = 0; kt < nk; ++kt) {
    const char* base = smem + (kt & 1) * T_STAGE + frow;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (active) {
#if !defined(SGL_TN_ABLATE) || SGL_TN_ABLATE != 1
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = SGL_TR6(base + T_OP + h * 32 * 512 + (fb_col ^ (uint32_t)(j * 32)));
#else
        if (kt == 0)   // ablation (wrong results): B fragments read once -> read slots carry A only
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[j] = SGL_TR6(base + T_OP + h * 32 * 512 + (fb_col ^ (uint32_t)(j * 32)));
#endif
#if !defined(SGL_TN_ABLATE) || SGL_TN_ABLATE != 2
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = SGL_TR6(base + h * 32 * 512 + (fa_col ^ (uint32_t)(i * 32)));
#else
        if (kt == 0)   // ablation (wrong results): A fragments read once -> read slots carry B only
#pragma unroll
          for (int i = 0; i < 8; ++i) fa[i] = SGL_TR6(base + h * 32 * 512 + (fa_col ^ (uint32_t)(i * 32)));
#endif
      }
      if (h == 0) { issue(2, kt + 1); issue(3, kt + 1); }
      else { issue(0, kt + 2); issue(1, kt + 2); }
#ifdef SGL_TIMELINE
      SGL_TLP(0);
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      SGL_TLP(1);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      SGL_TLP(2);
#else
      SGL_PP_END_READ8();
#endif
      if (active) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
#ifdef SGL_TIMELINE
      asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[7][3]));
      SGL_TLP(3);
#endif
      SGL_PP_END_MFMA();
      SGL_TLP(4);
    }
  }
#ifdef SGL_TIMELINE
  if (lane == 0 && active && (w == 0 || w == 4)) {
    for (int i = 0; i < 5; ++i) atomicAdd(&g_tn6_tl[(w >> 2) * 6 + i], tlp[i]);
    atomicAdd(&g_tn6_tl[(w >> 2) * 6 + 5], (unsigned long long)(2 * nk));
  }
#endif
#undef SGL_TR6
  if (grp == 0) SGL_PP_END_MFMA();   // G0 waits for G1's last slot
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing (all-zero) units must land before the LDS is reused
  EpiParams pq = p;
  if (p.split_stride) pq.out = reinterpret_cast<float*>(p.out) + (size_t)split * p.split_stride;  // private slab of this split
  store_tile256<EPI_F32, float>(smem, acc, wr, wc, lane, t, n1_0, n2_0, N1, N2, pq);
}

#ifdef SGL_AB
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void gemm_tn2_kernel(const bf16* __restrict__ A, int lda,
                                                          const bf16* __restrict__ B, int ldb, int Mred, int N1, int N2,
                                                          int m_per_split, int nsplits, int tiles_1, int tiles_2,
                                                          EpiParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // units ordered (split, tile1, tile2): an XCD's contiguous chunk shares one token range and neighbouring panels
  const int ntiles = tiles_1 * tiles_2;
  const int nunits = ntiles * nsplits;
  const int unit = unit_of_block(blockIdx.x, (nunits + 7) >> 3);
  if (unit >= nunits) return;
  const int split = unit / ntiles, trem = unit - split * ntiles;
  const int tile1 = trem / tiles_2, tile2 = trem - tile1 * tiles_2;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = (w >> 1) & 1, wc = (w & 1) + 2 * (w >> 2);
  const int n1_0 = tile1 * T_BM, n2_0 = tile2 * T_BN;
  const int m_begin = split * m_per_split;
  const int m_end = (m_begin + m_per_split < Mred) ? m_begin + m_per_split : Mred;
  const int rows = m_end - m_begin;
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(A + (size_t)m_begin * lda, (uint32_t)((size_t)rows * lda * 2));
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(B + (size_t)m_begin * ldb, (uint32_t)((size_t)rows * ldb * 2));

  // DMA: image [64 m][256 n] with 512-B rows; wave w moves rows [w*8, w*8+8), 2 rows (1 KiB) per instruction
  const int dr2 = lane >> 5, dslot = lane & 31;
  uint32_t a_off[4], b_off[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = w * 8 + q * 2 + dr2;
    const int chunk = dslot ^ (2 * (row & 3) + 8 * ((row >> 3) & 1));
    const int ca = n1_0 + chunk * 8, cb = n2_0 + chunk * 8;
    a_off[q] = (ca < N1) ? (uint32_t)(row * lda + ca) * 2u : SGL_OOB;
    b_off[q] = (cb < N2) ? (uint32_t)(row * ldb + cb) * 2u : SGL_OOB;
  }
  auto issue = [&](int kt, int stage) {
    const uint32_t r0 = (uint32_t)kt * T_BK;
    char* base = smem + stage * T_STAGE + (w * 8) * 512;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      dma16(ra, base + q * 1024, a_off[q] == SGL_OOB ? SGL_OOB : a_off[q] + r0 * (uint32_t)lda * 2u);
      dma16(rb, base + T_OP + q * 1024, b_off[q] == SGL_OOB ? SGL_OOB : b_off[q] + r0 * (uint32_t)ldb * 2u);
    }
  };

  const int fg = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
  const uint32_t swz = (uint32_t)(32 * fq + 128 * (fg & 1));
  const uint32_t frow = (uint32_t)((8 * fg + fq) * 512);
  const uint32_t fa_col = ((uint32_t)(wr * 256 + 8 * fp)) ^ swz;   // + i*32 bytes per 16-column tile (bits 5.. stay XOR-safe)
  const uint32_t fb_col = ((uint32_t)(wc * 128 + 8 * fp)) ^ swz;
  const bool active = (n2_0 + wc * 64 < N2) && (n1_0 + wr * 128 < N1);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (rows + T_BK - 1) / T_BK;
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
    if (active) {
      // same 16-step pipeline as nt2; every fragment is two transposed 8-byte reads
      const char* base = smem + (kt & 1) * T_STAGE + frow;
      bf16x8 a[16], b[2][4];
#define SGL_TR(ptr) __builtin_shufflevector(lds_tr16v2(ptr), lds_tr16v2((ptr) + 4 * 512), 0, 1, 2, 3, 4, 5, 6, 7)
#define SGL_LDA(f) SGL_TR(base + ((f) >> 3) * 32 * 512 + (fa_col ^ (uint32_t)(((f) & 7) * 32)))
#define SGL_LDB(s_, j) SGL_TR(base + T_OP + (s_) * 32 * 512 + (fb_col ^ (uint32_t)((j) * 32)))
#pragma unroll
      for (int j = 0; j < 4; ++j) b[0][j] = SGL_LDB(0, j);
      a[0] = SGL_LDA(0);
      a[1] = SGL_LDA(1);
      a[2] = SGL_LDA(2);
#pragma unroll
      for (int f = 0; f < 16; ++f) {
        if (f + 3 < 16) a[f + 3] = SGL_LDA(f + 3);
        if (f >= 3 && f < 7) b[1][f - 3] = SGL_LDB(1, f - 3);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[f & 7][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[f], b[f >> 3][j], acc[f & 7][j], 0, 0, 0);
      }
#undef SGL_LDA
#undef SGL_LDB
#undef SGL_TR
      sched_pipeline<2>();
    }
  }
  EpiParams pq = p;
  if (p.split_stride) pq.out = reinterpret_cast<float*>(p.out) + (size_t)split * p.split_stride;  // private slab of this split
  store_tile256<EPI_F32, float>(smem, acc, wr, wc, lane, t, n1_0, n2_0, N1, N2, pq);
}

#endif  // SGL_AB

// ------------------------------------------------------------------------------------------------------
template <int EPI, typename TOut>
static hipError_t launch_nt2(const bf16* A, int lda, const bf16* B, int ldb, int M, int N, int K, const EpiParams& p,
                             hipStream_t s) {
  const int tiles_m = (M + T_BM - 1) / T_BM, tiles_n = (N + T_BN - 1) / T_BN;
  const int grid = ((tiles_m * tiles_n + 7) / 8) * 8;
#ifdef SGL_AB
  static const int gen = getenv("SGL_GEMM_GEN") ? atoi(getenv("SGL_GEMM_GEN")) : 6;
  if (gen == 2) {
    static bool attr = false;
    if (!attr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt2_kernel<EPI, TOut>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
      if (e != hipSuccess) return e;
      attr = true;
    }
    hipLaunchKernelGGL((gemm_nt2_kernel<EPI, TOut>), dim3(grid), dim3(512), T_LDS, s, A, lda, B, ldb, M, N, K, tiles_m,
                       tiles_n, p);
    return hipGetLastError();
  }
#endif
  {
    static bool attr6 = false;
    if (!attr6) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt6_kernel<EPI, TOut>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
      if (e != hipSuccess) return e;
      attr6 = true;
    }
    // band height of the grouped tile order: 8 row-tiles for wide outputs, 4 when there are few column tiles (measured
    // on one box: qkv N=3456 +3 %, fc2 N=1152 +1.5 % with 4; fc1 N=4352 best with 8).  SGL_BAND overrides.
    // Tile order.  Default (round 2): B-stationary with column groups of 8 — same speed as the round-1 grouped bands
    // (+-1 % over the encoder's eight shapes) but 35 % fewer bytes leave the XCD L2s at the bench batch (rocprofv3
    // FETCH_SIZE: 2.42 -> 1.56 GB per fc1 launch at B = 128, profiles/r02_pmc_traffic.json).  SGL_BAND > 0 selects the
    // grouped bands of that height (8 / 4 were the round-1 choices), SGL_BAND < 0 another column-group width.
    static const int band_env = getenv("SGL_BAND") ? atoi(getenv("SGL_BAND")) : 0;
    const int band_h = band_env != 0 ? band_env : -8;
    int grid6 = grid;
    if (band_h < 0) grid6 = 8 * (((tiles_m + 7) / 8) * tiles_n);   // 8 XCDs x the largest per-XCD tile count
    EpiParams pp = p;
#ifdef SGL_TIMELINE   // measurement builds only (make TIMELINE=1): the product library never skips an epilogue
    static const bool skip_epi = getenv("SGL_NT6_SKIP_EPI") != nullptr;
    if (skip_epi && EPI != EPI_F32) pp.atomic = 77;
#endif
    hipLaunchKernelGGL((gemm_nt6_kernel<EPI, TOut>), dim3(grid6), dim3(512), T_LDS, s, A, lda, B, ldb, M, N, K,
                       tiles_m, tiles_n, band_h, pp);
#ifdef SGL_TIMELINE
    if (getenv("SGL_TIMELINE")) {   // synchronises: measurement builds only
      unsigned long long h[4];
      (void)hipStreamSynchronize(s);
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_nt6_tl), sizeof(h));
      if (h[2])
        fprintf(stderr, "[timeline] gemm_nt6 EPI %d M=%d N=%d K=%d: %llu tiles, main loop %.0f, epilogue %.0f cycles per tile\n", EPI, M,
                N, K, h[2], (double)h[0] / h[2], (double)h[1] / h[2]);
      memset(h, 0, sizeof(h));
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_nt6_tl), h, sizeof(h));
    }
#endif
    return hipGetLastError();
  }
}

hipError_t gemm_nt2_bf16(const void* A_, int lda, const void* B_, int ldb, int M, int N, int K, int epi, int out_dtype,
                         const EpiParams& p, hipStream_t s) {
  const bf16* A = (const bf16*)A_;
  const bf16* B = (const bf16*)B_;
  switch (epi) {
    case EPI_STORE:
      return out_dtype == DT_BF16 ? launch_nt2<EPI_STORE, bf16>(A, lda, B, ldb, M, N, K, p, s)
                                  : launch_nt2<EPI_STORE, float>(A, lda, B, ldb, M, N, K, p, s);
    case EPI_BIAS_GELU:
      return out_dtype == DT_BF16 ? launch_nt2<EPI_BIAS_GELU, bf16>(A, lda, B, ldb, M, N, K, p, s)
                                  : launch_nt2<EPI_BIAS_GELU, float>(A, lda, B, ldb, M, N, K, p, s);   // bf16x3 strict mode
    case EPI_QKV:
      return out_dtype == DT_BF16 ? launch_nt2<EPI_QKV, bf16>(A, lda, B, ldb, M, N, K, p, s)
                                  : launch_nt2<EPI_QKV, float>(A, lda, B, ldb, M, N, K, p, s);   // bf16x3 strict mode
    case EPI_GELU_BWD:
      return out_dtype == DT_BF16 ? launch_nt2<EPI_GELU_BWD, bf16>(A, lda, B, ldb, M, N, K, p, s)
                                  : launch_nt2<EPI_GELU_BWD, float>(A, lda, B, ldb, M, N, K, p, s);   // bf16x3 strict mode
    case EPI_RES_F32: return launch_nt2<EPI_RES_F32, float>(A, lda, B, ldb, M, N, K, p, s);
    case EPI_POS_F32: return launch_nt2<EPI_POS_F32, float>(A, lda, B, ldb, M, N, K, p, s);
    case EPI_F32: return launch_nt2<EPI_F32, float>(A, lda, B, ldb, M, N, K, p, s);
  }
  return hipErrorInvalidValue;
}

hipError_t gemm_tn2_bf16(const void* A_, int lda, const void* B_, int ldb, int Mred, int N1, int N2, int m_per,
                         int splits, const EpiParams& p, hipStream_t s) {
  const int tiles_1 = (N1 + T_BM - 1) / T_BM, tiles_2 = (N2 + T_BN - 1) / T_BN;
  const int grid = ((tiles_1 * tiles_2 * splits + 7) / 8) * 8;
#ifdef SGL_AB
  static const int gen = getenv("SGL_GEMM_GEN") ? atoi(getenv("SGL_GEMM_GEN")) : 6;
  static const int tngen = getenv("SGL_TN_GEN") ? atoi(getenv("SGL_TN_GEN")) : gen;
  if (tngen == 2) {
    static bool attr = false;
    if (!attr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn2_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
      if (e != hipSuccess) return e;
      attr = true;
    }
    hipLaunchKernelGGL(gemm_tn2_kernel, dim3(grid), dim3(512), T_LDS, s, (const bf16*)A_, lda, (const bf16*)B_, ldb,
                       Mred, N1, N2, m_per, splits, tiles_1, tiles_2, p);
    return hipGetLastError();
  }
#endif
  static bool attr6 = false;
  if (!attr6) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn6_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
    if (e != hipSuccess) return e;
    attr6 = true;
  }
  hipLaunchKernelGGL(gemm_tn6_kernel, dim3(grid), dim3(512), T_LDS, s, (const bf16*)A_, lda, (const bf16*)B_, ldb,
                     Mred, N1, N2, m_per, splits, tiles_1, tiles_2, p);
#ifdef SGL_TIMELINE
  if (getenv("SGL_TIMELINE")) {   // synchronises: measurement builds only
    unsigned long long h[12];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tn6_tl), sizeof(h));
    for (int g = 0; g < 2; ++g)
      if (h[g * 6 + 5])
        fprintf(stderr, "[timeline] gemm_tn6 N1=%d N2=%d Mred=%d splits=%d group %d, cycles per slot pair (read slot + MFMA slot): issue "
                "reads+DMA %.0f, wait %.0f, barrier %.0f, 32 MFMAs %.0f, barrier %.0f\n", N1, N2, Mred, splits, g,
                (double)h[g * 6 + 0] / h[g * 6 + 5], (double)h[g * 6 + 1] / h[g * 6 + 5], (double)h[g * 6 + 2] / h[g * 6 + 5],
                (double)h[g * 6 + 3] / h[g * 6 + 5], (double)h[g * 6 + 4] / h[g * 6 + 5]);
    memset(h, 0, sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tn6_tl), h, sizeof(h));
  }
#endif
  return hipGetLastError();
}

}  // namespace sgl

// Generation 7 of the bf16 NT GEMM (C[M,N] = A[M,K] · B[N,K]ᵀ + fused epilogue): the generation-6 anti-phase main loop
// (gemm_bf16_v2.hip) inside a PERSISTENT tile loop, one workgroup per CU.
//
// What generation 6 left on the table (round-1 measurements): with K = 1152 a 256x256 tile has only 18 K-steps, so the
// pipeline ramp (first DMA units ≈1-2 µs from a cold start) and the epilogue (4 passes, each behind two workgroup
// barriers and a dependent round trip for the residual / GELU' operand) were 20-45 % of a tile: bare store 1.03-1.15 PF,
// fused epilogues 0.61-0.99 PF.  Here:
//   * the K-step stream never drains: while a tile's last K-steps run, the DMA units of the NEXT tile's first two
//     K-steps are already being issued (same round-robin unit order, stage parity carried across tiles), so the next
//     main loop starts on landed data the moment the epilogue is done;
//   * the epilogue is wave-private: each wave transposes its own 128x64 accumulator block through a private 4-KiB LDS
//     region (16 rows at a time, outside the two 64-KiB stage buffers, 160 KiB total) — no workgroup barrier, the eight
//     waves drift, one wave's memory latency hides behind the others' LDS/VALU/store work; residual / GELU' operands
//     are requested before the slab goes through LDS;
//   * blockIdx -> tile order is "B-stationary" per XCD: XCD x owns a contiguous range of row tiles and sweeps it once
//     per group of `cgw` column tiles, so the group's weight panels (4 x 590 KB at K = 1152) stay in its 4-MiB L2 while
//     activation panels stream through (rocprofv3 FETCH_SIZE: profiles/r02_pmc_traffic.json).
// vmcnt accounting across a tile boundary: every wave drains its DMA (`vmcnt(0)`) at epilogue entry — the units in
// flight there were issued at least one MFMA slot earlier — so all units of the next tile's K-steps 0 and 1 have
// landed before anybody reads them, and the two read slots of that K-step end WITHOUT a vmcnt wait (a counted wait
// there would also wait for this wave's epilogue stores).  From K-step 1 on the generation-6 invariants hold unchanged:
// by then more than one K-step (≈0.5 µs) separates the wait from the epilogue's last stores.
#include <stdlib.h>

#include "common.hip.h"
#include "kernels.h"

namespace sgl {

namespace {

constexpr int P_BM = 256, P_BN = 256, P_BK = 64;
constexpr int P_OP = P_BM * P_BK * 2;     // 32 KiB per operand per stage
constexpr int P_STAGE = 2 * P_OP;         // 64 KiB
constexpr int P_MAIN = 2 * P_STAGE;       // 128 KiB
constexpr int P_EPI_WAVE = 16 * 64 * 4;   // 4 KiB per wave: 16 rows x 64 columns fp32
constexpr int P_LDS = P_MAIN + 8 * P_EPI_WAVE;   // 160 KiB

__device__ __forceinline__ void p7_dma16(u32x4 desc, uint32_t lds_addr, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(desc) : "memory");
}
__device__ __forceinline__ u32x4 p7_desc(const void* base, uint32_t bytes) {
  const uint64_t q = (uint64_t)base;
  u32x4 d = {(uint32_t)q, (uint32_t)(q >> 32) & 0xffffu, bytes, 0x00020000u};
  return d;
}
#define P7_END_READ8()                                                  \
  do {                                                                  \
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);                                  \
    __builtin_amdgcn_s_barrier();                                       \
    __builtin_amdgcn_sched_barrier(0);                                  \
  } while (0)
#define P7_END_READ_NOVM()                                              \
  do {                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  \
    __builtin_amdgcn_sched_barrier(0);                                  \
    __builtin_amdgcn_s_barrier();                                       \
    __builtin_amdgcn_sched_barrier(0);                                  \
  } while (0)
#define P7_END_MFMA()                     \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

// XCD-local, B-stationary tile order.  XCD x owns row tiles [x*tiles_m/8, (x+1)*tiles_m/8); its local unit v walks column
// groups of width cgw outermost, then rows, then the columns of the group: 32 consecutive units (what the XCD's 32
// workgroups run together) are (32/w) rows x w columns.  Placement affects speed only.
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

// ---- wave-private epilogue ---------------------------------------------------------------------------------------
// The wave's accumulators cover rows [row0, row0+128) x columns [col0, col0+64).  Pass i moves the 16-row slab i through
// the wave's LDS region and every lane finishes Q row-contiguous chunks of NV columns (NV*sizeof(TOut) = 16 bytes).
//
// STRAIGHT-LINE code on purpose: every global access is a buffer load/store whose out-of-range lanes (rows >= M through
// the descriptor's size, columns >= N through an explicit out-of-range offset) are dropped by the hardware, so there is
// no branch anywhere in the eight passes.  vmcnt retires in issue order, loads and stores alike; with branches between
// them the compiler's waitcnt pass falls back to vmcnt(0) before every use of a loaded value, i.e. every pass waits for
// the previous pass's STORES to be acknowledged (measured: 16 us per 256x256 tile, 8 B/clk/CU).  Without branches it
// counts exactly and the residual / GELU' operand of pass i+1 is in flight while pass i's stores drain.
constexpr uint32_t P7_OOB = 0xFFFFFFF0u;

template <typename T, int NV>
__device__ __forceinline__ void buf_ld(__amdgpu_buffer_rsrc_t r, uint32_t off, float (&v)[NV]) {
  const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  if constexpr (sizeof(T) == 4) {
    const f32x4 f = __builtin_bit_cast(f32x4, raw);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = f[j];
  } else {
    const bf16x8 h = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
  }
}
template <typename T, int NV>
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, uint32_t off, const float (&v)[NV]) {
  u32x4 raw;
  if constexpr (sizeof(T) == 4) {
    f32x4 f = {v[0], v[1], v[2], v[3]};
    raw = __builtin_bit_cast(u32x4, f);
  } else {
    bf16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (bf16)v[j];
    raw = __builtin_bit_cast(u32x4, h);
  }
  __builtin_amdgcn_raw_buffer_store_b128(raw, r, off, 0, 0);
}
__device__ __forceinline__ uint32_t clamp_u32(size_t b) { return b > 0xFFFFFFE0ull ? 0xFFFFFFE0u : (uint32_t)b; }

template <int EPI, typename TOut>
__device__ __forceinline__ void store_wave_tile(float* ct, f32x4 (&acc)[8][4], int lane, int row0, int col0, int M,
                                                int N, int wr_tile_row, const EpiParams& p, int dbg) {
  constexpr int NV = (sizeof(TOut) == 4) ? 4 : 8;
  constexpr int ES = sizeof(TOut);
  constexpr int LPR = 64 / NV;   // lanes per 64-column row: 16 / 8
  constexpr int RPI = 64 / LPR;  // rows covered by one wave instruction: 4 / 8
  constexpr int Q = 16 / RPI;    // chunks per lane per pass: 4 / 2
  const int g = lane >> 4, c16 = lane & 15;
  const int lr = lane / LPR, lc = (lane % LPR) * NV;
  const int gcol = col0 + lc;
  const bool col_ok = (gcol < N) && !(dbg & 2);      // dbg&2: experiment, drop every global access of the epilogue
  constexpr bool HAS_AUX = (EPI == EPI_RES_F32 || EPI == EPI_GELU_BWD);

  // descriptors: sizes end exactly after row M-1, so rows >= M are out of range by themselves
  const __amdgpu_buffer_rsrc_t r_out =
      make_rsrc(p.out, p.out ? clamp_u32(EPI == EPI_QKV ? (size_t)3 * p.batch * p.heads * p.tokens * p.head_dim_pad * ES
                                                       : (size_t)M * p.ldo * ES) : 0u);
  const __amdgpu_buffer_rsrc_t r_out2 = make_rsrc(p.out2, (EPI == EPI_BIAS_GELU) ? clamp_u32((size_t)M * p.ldo2 * ES) : 0u);
  const __amdgpu_buffer_rsrc_t r_aux =
      make_rsrc(EPI == EPI_RES_F32 ? (const void*)p.res : p.aux,
                EPI == EPI_RES_F32 ? clamp_u32((size_t)M * p.ldr * 4) : (EPI == EPI_GELU_BWD ? clamp_u32((size_t)M * p.ldaux * ES) : 0u));

  float bias[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) bias[j] = 0.f;
  if constexpr (EPI != EPI_GELU_BWD) {
    const __amdgpu_buffer_rsrc_t r_bias = make_rsrc(p.bias, p.bias ? (uint32_t)N * 4u : 0u);
    if constexpr (NV == 4) {
      buf_ld<float, 4>(r_bias, col_ok ? (uint32_t)gcol * 4u : P7_OOB, bias);
    } else {
      float lo[4], hi[4];
      buf_ld<float, 4>(r_bias, col_ok ? (uint32_t)gcol * 4u : P7_OOB, lo);
      buf_ld<float, 4>(r_bias, col_ok ? (uint32_t)gcol * 4u + 16u : P7_OOB, hi);
#pragma unroll
      for (int j = 0; j < 4; ++j) { bias[j] = lo[j]; bias[4 + j] = hi[j]; }
    }
    // consume the loaded values here, once, so that no later wait is attributed to them
#pragma unroll
    for (int j = 0; j < NV; ++j) asm volatile("" : "+v"(bias[j]));
  }
  // QKV scatter: column -> (which, head, d) is fixed per lane
  int qkv_which = 0, qkv_h = 0, qkv_d = 0;
  if constexpr (EPI == EPI_QKV) {
    const int dm = p.heads * p.head_dim;
    qkv_which = gcol / dm;
    const int hc = gcol - qkv_which * dm;
    qkv_h = hc / p.head_dim;
    qkv_d = hc - qkv_h * p.head_dim;
  }
  float csum[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) csum[j] = 0.f;

  u32x4 aux[2][Q];   // raw 16-byte chunks (4 fp32 residuals / 8 bf16 pre-activations), converted where they are used
  auto load_aux = [&](int pass, u32x4 (&dst)[Q]) {
    if constexpr (HAS_AUX) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int grow = row0 + pass * 16 + lr + q * RPI;
        const uint32_t off = (EPI == EPI_RES_F32) ? (uint32_t)grow * (uint32_t)p.ldr * 4u + (uint32_t)gcol * 4u
                                                  : (uint32_t)grow * (uint32_t)p.ldaux * ES + (uint32_t)gcol * ES;
        dst[q] = __builtin_amdgcn_raw_buffer_load_b128(r_aux, col_ok ? off : P7_OOB, 0, 0);
      }
    }
  };
  load_aux(0, aux[0]);

#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    if (pass + 1 < 8) load_aux(pass + 1, aux[(pass + 1) & 1]);   // one pass ahead of its use, issued before this pass's stores
    // accumulator slab -> LDS (row-major 16 x 64 fp32); two lanes per bank on the ds_write_b32: free (MI355X_MICROARCH)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) ct[(g * 4 + r) * 64 + j * 16 + c16] = acc[pass][j][r];
    // no s_waitcnt between the writes and the reads: one wave's DS instructions execute in issue order, and the slab is
    // private to this wave; the empty asm only keeps the compiler from reordering across it
    asm volatile("" ::: "memory");
    float v[Q][NV];
#pragma unroll
    for (int q = 0; q < Q; ++q) Vec<float, NV>::ld(ct + (lr + q * RPI) * 64 + lc, v[q]);
    asm volatile("" ::: "memory");   // (same argument: the next pass's writes are issued behind these reads)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int grow = row0 + pass * 16 + lr + q * RPI;
      float (&x)[NV] = v[q];
      if constexpr (EPI == EPI_STORE) {
#pragma unroll
        for (int j = 0; j < NV; ++j) x[j] = x[j] * p.alpha + bias[j];
        buf_st<TOut, NV>(r_out, col_ok ? (uint32_t)grow * (uint32_t)p.ldo * ES + (uint32_t)gcol * ES : P7_OOB, x);
      } else if constexpr (EPI == EPI_BIAS_GELU) {
        float a[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          x[j] += bias[j];
          a[j] = gelu_tanh(x[j]);
        }
        // the pre-activation is only read by the backward GELU': inference and frozen blocks pass out == nullptr, which
        // gives r_out zero records (the store is issued and dropped: no branch)
        buf_st<TOut, NV>(r_out, col_ok ? (uint32_t)grow * (uint32_t)p.ldo * ES + (uint32_t)gcol * ES : P7_OOB, x);
        buf_st<TOut, NV>(r_out2, col_ok ? (uint32_t)grow * (uint32_t)p.ldo2 * ES + (uint32_t)gcol * ES : P7_OOB, a);
      } else if constexpr (EPI == EPI_RES_F32) {
        const f32x4 r = __builtin_bit_cast(f32x4, aux[pass & 1][q]);
#pragma unroll
        for (int j = 0; j < NV; ++j) x[j] = r[j] + (x[j] + bias[j]);
        buf_st<float, NV>(r_out, col_ok ? (uint32_t)grow * (uint32_t)p.ldo * 4u + (uint32_t)gcol * 4u : P7_OOB, x);
      } else if constexpr (EPI == EPI_QKV) {
        const int b = grow / p.tokens;
        const int n = grow - b * p.tokens;
#pragma unroll
        for (int j = 0; j < NV; ++j) x[j] += bias[j];
        const uint32_t dst = (uint32_t)((((qkv_which * p.batch + b) * p.heads + qkv_h) * p.tokens + n) * p.head_dim_pad +
                                        qkv_d) * ES;
        const bool ok = col_ok && grow < M;
        buf_st<TOut, NV>(r_out, ok ? dst : P7_OOB, x);
        // zero the pad columns [head_dim, head_dim_pad): head_dim % 8 == 0 and head_dim_pad = round_up(head_dim, 16), so
        // the pad is zero or one NV-chunk; the lane that holds a head's last chunk writes it (others: dropped)
        float z[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) z[j] = 0.f;
        const bool pad = ok && (qkv_d + NV == p.head_dim) && (p.head_dim_pad > p.head_dim);
        buf_st<TOut, NV>(r_out, pad ? dst + NV * ES : P7_OOB, z);
      } else {  // EPI_GELU_BWD
        const bf16x8 u = __builtin_bit_cast(bf16x8, aux[pass & 1][q]);
        const bool rok = grow < M;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          x[j] *= gelu_tanh_grad((float)u[j]);
          csum[j] += rok ? x[j] : 0.f;
        }
        buf_st<TOut, NV>(r_out, col_ok ? (uint32_t)grow * (uint32_t)p.ldo * ES + (uint32_t)gcol * ES : P7_OOB, x);
      }
    }
  }
  if constexpr (EPI == EPI_GELU_BWD) {
    if (p.colsum) {   // fused bias gradient: fold the lanes that own the same columns (lane bits above log2(LPR))
#pragma unroll
      for (int j = 0; j < NV; ++j) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) csum[j] += __shfl_xor(csum[j], o, 64);
      }
      if (lr == 0 && col_ok) {
        if (p.colsum_ld > 0) {   // deterministic: one row of partial sums per 128 output rows, folded by the host in order
          float* dst = p.colsum + (size_t)wr_tile_row * p.colsum_ld + gcol;
          Vec<float, NV>::st(dst, csum);
        } else {
#pragma unroll
          for (int j = 0; j < NV; ++j) atomicAdd(p.colsum + gcol + j, csum[j]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
template <int EPI, typename TOut>
__global__ __launch_bounds__(512, 2) void gemm_nt7_kernel(const bf16* __restrict__ A, int lda,
                                                          const bf16* __restrict__ B, int ldb, int M, int N, int K,
                                                          int tiles_m, int tiles_n, int cgw, int dbg, EpiParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  int v = slot, tile_m, tile_n;
  if (!tile_of_local(xcd, v, tiles_m, tiles_n, cgw, tile_m, tile_n)) return;   // whole block exits together
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = (w >> 1) & 1, wc = (w & 1) + 2 * (w >> 2), grp = w >> 2;
  if ((dbg >> 8) > 0 && false) {   // experiment: de-phase the workgroups (slot-dependent start delay, units of 0.1 us per phase step)
    const int phases = (dbg >> 4) & 15;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t wait = (uint64_t)(dbg >> 8) * (uint64_t)(slot % (phases > 0 ? phases : 2));
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
  }
  const u32x4 da = p7_desc(A, (uint32_t)(((size_t)(M - 1) * lda + K) * 2));
  const u32x4 db = p7_desc(B, (uint32_t)(((size_t)(N - 1) * ldb + K) * 2));
  const uint32_t lds0 = (uint32_t)(size_t)((SGL_LDS char*)smem);
  float* ct = reinterpret_cast<float*>(smem + P_MAIN + w * P_EPI_WAVE);

  const int nk = (K + P_BK - 1) / P_BK;
  // Per-lane DMA plan and fragment addresses.  They are pure functions of the lane id, recomputed at the top of every tile
  // from an opaque copy of it (P7_LANE_STATE): the epilogue needs the registers (128 accumulators + a 16-row slab in
  // flight), and ~40 VGPRs of loop constants kept alive across it are what pushes the kernel into scratch.
#define P7_LANE_STATE()                                                                                         \
  int lane_o = lane;                                                                                             \
  asm volatile("" : "+v"(lane_o));                                                                               \
  const int drow = lane_o >> 3, dchunk = (lane_o & 7) ^ drow;                                                    \
  uint32_t voff[4][2], ldst[4][2];                                                                               \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) _Pragma("unroll") for (int q = 0; q < 2; ++q) {                  \
    const int ul = 16 * w + 8 * q;                                                                               \
    int r0;                                                                                                      \
    if (u == 0) r0 = (ul < 64) ? ul : ul + 64;                                                                   \
    else if (u == 3) r0 = (ul < 64) ? ul + 64 : ul + 128;                                                        \
    else r0 = (ul >> 5) * 64 + (ul & 31) + (u == 2 ? 32 : 0);                                                    \
    const int row = r0 + drow;                                                                                   \
    const bool isA = (u == 0 || u == 3);                                                                         \
    voff[u][q] = (uint32_t)(row * (isA ? lda : ldb) + dchunk * 8) * 2u;                                          \
    ldst[u][q] = lds0 + (isA ? 0 : P_OP) + (uint32_t)r0 * 128u;                                                  \
  }                                                                                                              \
  auto issue = [&](int u, int kt, int stage, uint32_t aoff_, uint32_t boff_, bool ok) {                          \
    const int k0 = kt * P_BK;                                                                                    \
    const bool kok = ok && (k0 + dchunk * 8 < K);                                                                \
    const uint32_t sb = (uint32_t)(stage * P_STAGE);                                                             \
    const bool isA = (u == 0 || u == 3);                                                                         \
    const u32x4 d = isA ? da : db;                                                                               \
    const uint32_t base = (isA ? aoff_ : boff_) + (uint32_t)k0 * 2u;                                             \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) p7_dma16(d, ldst[u][q] + sb, kok ? voff[u][q] + base : SGL_OOB); \
  };                                                                                                             \
  const int frow = lane_o & 15, fg = lane_o >> 4, fsw = frow & 7;                                                \
  const uint32_t fa_base = (uint32_t)((wr * 128 + frow) * 128);                                                  \
  const uint32_t fb_base = (uint32_t)(P_OP + (wc * 64 + frow) * 128);                                            \
  const uint32_t c0 = (uint32_t)(((0 + fg) ^ fsw) << 4), c1 = (uint32_t)(((4 + fg) ^ fsw) << 4)

  uint32_t aoff = (uint32_t)tile_m * P_BM * (uint32_t)lda * 2u, boff = (uint32_t)tile_n * P_BN * (uint32_t)ldb * 2u;
  int par = 0;   // stage of the current tile's K-step 0
  {
    P7_LANE_STATE();
    (void)fa_base; (void)fb_base; (void)c0; (void)c1;
    issue(0, 0, 0, aoff, boff, true); issue(1, 0, 0, aoff, boff, true); issue(2, 0, 0, aoff, boff, true);
    issue(3, 0, 0, aoff, boff, true);
    issue(0, 1, 1, aoff, boff, true); issue(1, 1, 1, aoff, boff, true); issue(2, 1, 1, aoff, boff, true);
  }
  P7_END_READ8();                // Aq02(0), BX(0), BY(0) of every wave have landed
  if (grp == 1) P7_END_MFMA();   // G1 runs one slot behind G0 from here on
  bool first_tile = true;

  for (;;) {
    int ntm = 0, ntn = 0;
    const bool nvalid = tile_of_local(xcd, v + nslots, tiles_m, tiles_n, cgw, ntm, ntn);
    const uint32_t naoff = (uint32_t)ntm * P_BM * (uint32_t)lda * 2u, nboff = (uint32_t)ntn * P_BN * (uint32_t)ldb * 2u;
    const int m0 = tile_m * P_BM, n0 = tile_n * P_BN;
    const bool active = (n0 + wc * 64 < N) && (m0 + wr * 128 < M);
    P7_LANE_STATE();

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 fa[8], fb[8];
    for (int kt = 0; kt < nk; ++kt) {
      const int stage = (par + kt) & 1;
      const char* base = smem + stage * P_STAGE;
      const char* pa = base + fa_base;
      const char* pb = base + fb_base;
      const bool novm = (kt == 0) && !first_tile;   // see the header: the epilogue already drained this wave's DMA
      // ---- R1: A0 (row tiles 0-3), B (column tiles 0-3), both k-halves; DMA unit Aq13 of the next K-step
      if (active) {
#pragma unroll
        for (int f = 0; f < 8; ++f) fb[f] = *reinterpret_cast<const bf16x8*>(pb + (f & 3) * 2048 + ((f >> 2) ? c1 : c0));
#pragma unroll
        for (int f = 0; f < 8; ++f) fa[f] = *reinterpret_cast<const bf16x8*>(pa + (f & 3) * 2048 + ((f >> 2) ? c1 : c0));
      }
      if (kt + 1 < nk) issue(3, kt + 1, stage ^ 1, aoff, boff, true);
      else issue(3, kt + 1 - nk, stage ^ 1, naoff, nboff, nvalid);
      if (novm) P7_END_READ_NOVM(); else P7_END_READ8();
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
      P7_END_MFMA();
      // ---- R2: A1 (row tiles 4-7); DMA units Aq02, BX, BY of K-step kt+2
      if (active) {
#pragma unroll
        for (int f = 0; f < 8; ++f) fa[f] = *reinterpret_cast<const bf16x8*>(pa + (4 + (f & 3)) * 2048 + ((f >> 2) ? c1 : c0));
      }
      if (kt + 2 < nk) {
        issue(0, kt + 2, stage, aoff, boff, true); issue(1, kt + 2, stage, aoff, boff, true);
        issue(2, kt + 2, stage, aoff, boff, true);
      } else {
        issue(0, kt + 2 - nk, stage, naoff, nboff, nvalid); issue(1, kt + 2 - nk, stage, naoff, nboff, nvalid);
        issue(2, kt + 2 - nk, stage, naoff, nboff, nvalid);
      }
      if (novm) P7_END_READ_NOVM(); else P7_END_READ8();
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
      P7_END_MFMA();
    }
    par = (par + nk) & 1;
    // ---- tile boundary: drain this wave's DMA (next tile's K-steps 0 and 1), then the wave-private epilogue
    if (!(dbg & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // dbg&4: experiment (results are garbage)
    if (grp == 1 && (dbg & 0xf0)) {   // experiment: offset the two waves of each SIMD inside the epilogue
      for (int z = 0; z < ((dbg >> 4) & 15); ++z) __builtin_amdgcn_s_sleep(8);   // 8 x 64 cycles each
    }
    if (active && !(dbg & 1))
      store_wave_tile<EPI, TOut>(ct, acc, lane, m0 + wr * 128, n0 + wc * 64, M, N, (m0 >> 7) + wr, p, dbg);
    if (!nvalid) break;
    v += nslots;
    tile_m = ntm; tile_n = ntn; aoff = naoff; boff = nboff;
    first_tile = false;
  }
  if (grp == 0) P7_END_MFMA();   // G0 waits for G1's last slot
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EPI, typename TOut>
hipError_t launch_nt7(const bf16* A, int lda, const bf16* B, int ldb, int M, int N, int K, const EpiParams& p,
                      hipStream_t s) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  static bool attr[64] = {};
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!attr[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt7_kernel<EPI, TOut>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
    if (e != hipSuccess) return e;
    attr[dev] = true;
  }
  const int tiles_m = (M + P_BM - 1) / P_BM, tiles_n = (N + P_BN - 1) / P_BN;
  // one workgroup per CU (160 KiB LDS): 32 per XCD; fewer when an XCD owns fewer tiles than that
  const int per_xcd = ((tiles_m + 7) / 8) * tiles_n;
  const int slots = per_xcd < 32 ? per_xcd : 32;
  static const int cg_env = getenv("SGL_NT7_CGW") ? atoi(getenv("SGL_NT7_CGW")) : 0;
  const int cgw = cg_env > 0 ? cg_env : 4;
  static const int dbg = getenv("SGL_NT7_DBG") ? atoi(getenv("SGL_NT7_DBG")) : 0;
  hipLaunchKernelGGL((gemm_nt7_kernel<EPI, TOut>), dim3(slots * 8), dim3(512), P_LDS, s, A, lda, B, ldb, M, N, K,
                     tiles_m, tiles_n, cgw, dbg, p);
  return hipGetLastError();
}

}  // namespace

// Returns hipErrorNotSupported when the problem is outside what generation 7 covers (the caller falls back to 6).
hipError_t gemm_nt7_bf16(const void* A_, int lda, const void* B_, int ldb, int M, int N, int K, int epi, int out_dtype,
                         const EpiParams& p, hipStream_t s) {
  if (p.gelu_grad_form) return hipErrorNotSupported;   // only generation 6 and the generic epilogue implement it
  const bf16* A = (const bf16*)A_;
  const bf16* B = (const bf16*)B_;
  if (K < 2 * P_BK || M < 8 * P_BM) return hipErrorNotSupported;
  if (((size_t)(M - 1) * lda + K) * 2 >= 0x80000000ull || ((size_t)(N - 1) * ldb + K) * 2 >= 0x80000000ull)
    return hipErrorNotSupported;   // whole-matrix buffer descriptors with 0x80000000 as the out-of-range offset
  switch (epi) {
    case EPI_STORE:
      return out_dtype == DT_BF16 ? launch_nt7<EPI_STORE, bf16>(A, lda, B, ldb, M, N, K, p, s)
                                  : launch_nt7<EPI_STORE, float>(A, lda, B, ldb, M, N, K, p, s);
    case EPI_BIAS_GELU: return launch_nt7<EPI_BIAS_GELU, bf16>(A, lda, B, ldb, M, N, K, p, s);
    case EPI_QKV: return launch_nt7<EPI_QKV, bf16>(A, lda, B, ldb, M, N, K, p, s);
    case EPI_GELU_BWD: return launch_nt7<EPI_GELU_BWD, bf16>(A, lda, B, ldb, M, N, K, p, s);
    case EPI_RES_F32: return launch_nt7<EPI_RES_F32, float>(A, lda, B, ldb, M, N, K, p, s);
    default: return hipErrorNotSupported;
  }
}

}  // namespace sgl

// bf16 flash attention for gfx950: softmax(QKᵀ/√dh)·V, non-causal, no mask, dropout 0
// (SiglipAttention, TF:models/siglip/modeling_siglip.py:273-306; softmax in fp32 as in :241).
// v_mfma_f32_32x32x16_bf16 throughout, 64-lane wavefronts, 4 waves per workgroup, each wave owns 32 rows.
//
// Layouts: q,k,v head-major [B][H][N][DP] (ld = 0; DP = head_dim rounded up to 16, pad columns zero in memory; what the
// QKV GEMM's EPI_QKV epilogue writes and what the encoder uses), or (ld > 0) straight out of a token-major projection
// output [B*N][ld]: head h of token row r is the 2*dh-byte segment at r*ld + h*dh (144 B = nine 16-byte chunks for
// dh = 72), gathered chunk by chunk by the LDS-DMA; the chunks that pad dh up to DP never exist in HBM — their DMA lanes
// point out of range and the hardware writes zeros into the LDS image.  Measured (round 3, B = 128): the gather costs the
// three kernels 5-6 % (14 cache lines per DMA instruction instead of 8), more than the plain-store QKV epilogue saves.
// out / dout token-major [B*N][H*dh]; dqkv token-major [B*N][3*H*dh].
//
// Forward ("query on the lane"): Sᵀ = K·Qᵀ so each lane owns one query column and 16 of a 32-key block's scores;
// the row max / row sum are lane-local plus one cross-half shuffle, the running rescale factor is a per-lane
// scalar, and Oᵀ = Vᵀ·Pᵀ consumes the Sᵀ accumulator registers directly as the MFMA B operand (no LDS round
// trip for P).  Vᵀ fragments come from the row-major V tile in LDS through ds_read_b64_tr_b16.
// Backward recomputes P from the saved log-sum-exp.  Two kernels, no atomics, bitwise reproducible (the dQ kernel runs first
// and also produces delta = rowsum(dO ∘ O) for the dK/dV kernel):
//   kv-kernel ("key on the lane"):  S = Q·Kᵀ, dP = dO·Vᵀ, dVᵀ += dOᵀ·P, dKᵀ += Qᵀ·dS   (wave owns 32 keys)
//   q-kernel  ("query on the lane"): Sᵀ = K·Qᵀ, dPᵀ = V·dOᵀ, dQᵀ += Kᵀ·dSᵀ             (wave owns 32 queries)
//
// Roofline: MFMA-bound; algorithmic FLOPs fwd 4·N²·dh per (b,h), bwd 10·N²·dh (the two-kernel split executes
// 14·N²·dh; the extra recompute buys determinism and no dQ atomics).
#include <stdlib.h>

#include "common.hip.h"
#include "kernels.h"

namespace sgl {

hipError_t attn_f32_fwd(const float*, const float*, const float*, float*, float*, int, int, int, int, int, int, hipStream_t);
hipError_t attn_f32_bwd(const float*, const float*, const float*, const float*, const float*, const float*, float*,
                        float*, int, int, int, int, int, int, hipStream_t);
hipError_t attn_ref_fwd(const float*, const float*, const float*, float*, float*, int, int, int, int, int, int, hipStream_t);
hipError_t attn_ref_bwd(const float*, const float*, const float*, const float*, const float*, const float*, float*,
                        float*, int, int, int, int, int, int, hipStream_t);

template <int DP>
struct AttnCfg {
  static constexpr int KS = DP / 16;          // MFMA k-steps across the head dim
  static constexpr int DT = (DP + 31) / 32;   // 32-wide head-dim tiles of the output
  static constexpr int CPR = DP / 8;          // 16-byte chunks per row
  static constexpr int RSTR = DP * 2 + 16;    // row-read image stride: odd multiple of 16 B -> b128 conflict-free
  static constexpr int TSTR = 192;            // transposed-read image stride: 4 rows x 64 B cover all 64 banks
  static constexpr int DSTR = 208;            // dual-use image (row reads + transposed reads), 96 columns
};

__device__ __forceinline__ u32x4 a_ldg(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ bf16x8 lds_row8(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x4 lds_tr4(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((SGL_LDS bf16x4*)(p));
}
// A-operand fragment of Xᵀ for a 32x32x16 MFMA whose B operand is an accumulator tile: element j of lane half h
// is row 16*kk + 8*(j>>2) + 4*h + (j&3) of the row-major LDS image, column = d0 + (lane & 31).
__device__ __forceinline__ bf16x8 lds_trfrag(const char* img, int stride, int kk, int dcol0, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const char* a = img + (16 * kk + 4 * (g >> 1) + q) * stride + (dcol0 + 16 * (g & 1) + 4 * p) * 2;
  const bf16x4 lo = lds_tr4(a);
  const bf16x4 hi = lds_tr4(a + 8 * stride);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& s, int base) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16)s[base + j];
  return r;
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Workgroup -> (head, 128-row block of that head).  The dispatcher deals workgroups to the 8 XCDs round-robin in linear-id
// order, and every XCD has its own L2: with the natural (block, head) grid the `nb` workgroups of one head — which all
// stream the SAME Q/dO (or K/V) tiles of that head — sat on `nb` different XCDs, so every tile was an L2 miss `nb` times
// over.  Here XCD x owns heads x, x+8, x+16, ... and runs their `nb` blocks back to back: one miss, nb-1 L2 hits.
// 1-D launch of 8 * ceil(BH / 8) * nb workgroups (head_grid); workgroups of the padding heads exit at once.
__device__ __forceinline__ bool head_block(int nb, int BH, int& bh, int& xb) {
  const int pid = blockIdx.x, xcd = pid & 7, slot = pid >> 3;
  const int hq = slot / nb;
  bh = hq * 8 + xcd;
  xb = slot - hq * nb;
  return bh < BH;
}
static inline dim3 head_grid(int nb, int BH) { return dim3((unsigned)(8 * ((BH + 7) / 8) * nb)); }

// Where head `hd` of batch `b` lives in a q/k/v source: byte stride between token rows, data chunks per row, buffer extent.
struct HeadSrc {
  uint32_t rowbytes, nc, bytes;
  size_t base;   // element offset of (token 0, column 0) of this head
};
__device__ __forceinline__ HeadSrc head_src(int ld, int b, int hd, int H, int N, int dh, int DP) {
  HeadSrc r;
  if (ld > 0) {
    r.rowbytes = (uint32_t)ld * 2u;
    r.nc = (uint32_t)dh / 8u;
    r.bytes = (uint32_t)(((size_t)(N - 1) * ld + dh) * 2);
    r.base = (size_t)b * N * ld + (size_t)hd * dh;
  } else {
    r.rowbytes = (uint32_t)DP * 2u;
    r.nc = (uint32_t)DP / 8u;
    r.bytes = (uint32_t)((size_t)N * DP * 2);
    r.base = ((size_t)b * H + hd) * (size_t)N * DP;
  }
  return r;
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// ======================================================================================================
// tile staging
// ======================================================================================================
// Staging (all three kernels): global -> LDS by DMA (buffer_load ... lds, 16 B per lane, 1 KiB per instruction); the two
// backward kernels use three LDS stages, tile t+2 requested while tile t is computed.  One raw s_barrier per tile; no staging registers, no
// ds_write, no address arithmetic in the loop beyond one add per DMA instruction.  The register-staged version it
// replaces (load tile t+1 at the top, ds_write + barrier at the bottom) cost 24 % of the kernel: an ablation without the
// staging ran 134 us faster of 560 (B = 64), half of it the ds_write -> lgkmcnt(0) -> barrier chain at the end of every tile.
// An image row holds `SC` 16-byte chunks of which the first `NC` are data; DMA lanes of the pad chunks (and of rows past N)
// point out of range and write zeros, so the pad columns the 80 -> 96 wide transposed reads touch are always clean.
// The DMA goes through inline asm (see gemm_bf16_v2.hip): the kernel owns the vmcnt accounting for it.
__device__ __forceinline__ void att_dma16(u32x4 desc, uint32_t lds_addr, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(desc)
               : "memory");
}
__device__ __forceinline__ void att_dma4(u32x4 desc, uint32_t lds_addr, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(desc)
               : "memory");
}
__device__ __forceinline__ u32x4 att_desc(const void* base, uint32_t bytes) {
  const uint64_t q = (uint64_t)base;
  return u32x4{(uint32_t)q, (uint32_t)(q >> 32) & 0xffffu, bytes, 0x00020000u};
}
// global byte offset (tile 0) of image chunk `cidx` for a row-major source with `rowbytes` per row
__device__ __forceinline__ uint32_t att_chunk_off(int cidx, int rows, int SC, int NC, uint32_t rowbytes) {
  const int row = cidx / SC, col = cidx - row * SC;
  return (row < rows && col < NC) ? (uint32_t)row * rowbytes + (uint32_t)col * 16u : SGL_OOB;
}
#define SGL_ATT_WAIT_BARRIER(n)                                   \
  do {                                                            \
    asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                            \
    __builtin_amdgcn_s_barrier();                                 \
    __builtin_amdgcn_sched_barrier(0);                            \
    asm volatile("" ::: "memory");                                \
  } while (0)

// ======================================================================================================
// forward
// ======================================================================================================
template <int DP, bool TOK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_fwd_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                          const bf16* __restrict__ V, bf16* __restrict__ out,
                                                          float* __restrict__ lse, int B, int H, int N, int dh, int ld,
                                                          float c, float scale) {
  using C = AttnCfg<DP>;
  constexpr int KT = 64;
  constexpr int KBYTES = KT * C::RSTR, VBYTES = KT * C::TSTR, STAGE = KBYTES + VBYTES;
  constexpr int SCK = C::RSTR / 16, SCV = C::TSTR / 16;                       // 16-byte chunks per image row
  constexpr int NIK = KT * SCK / 64, NIV = KT * SCV / 64, NSLOT = NIK + NIV;  // DMA instructions per image / per tile
  constexpr int NPW = (NSLOT + 3) / 4;                                        // slots per wave
  static_assert(KT * SCK % 64 == 0 && KT * SCV % 64 == 0, "whole DMA instructions per image");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  int bh, xb;
  if (!head_block((N + 127) / 128, B * H, bh, xb)) return;
  const int b = bh / H, hd = bh - b * H;
  const int q0 = xb * 128 + w * 32;
  HeadSrc src = head_src(TOK ? ld : 0, b, hd, H, N, dh, DP);
  if constexpr (!TOK) {   // head-major: compile-time row geometry (the encoder's path keeps its constant-folded addressing)
    src.rowbytes = (uint32_t)DP * 2u;
    src.nc = (uint32_t)DP / 8u;
  }
  const __amdgpu_buffer_rsrc_t rq = make_rsrc(Q + src.base, src.bytes);
  const u32x4 dk_ = att_desc(K + src.base, src.bytes);
  const u32x4 dv_ = att_desc(V + src.base, src.bytes);
  const uint32_t lds0 = (uint32_t)(size_t)((SGL_LDS char*)smem);

  const int qi = lane & 31, hh = lane >> 5;
  bf16x8 qf[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    const uint32_t ch = (uint32_t)(2 * ks + hh);
    qf[ks] = as_bf16x8(a_ldg(rq, (q0 + qi < N && ch < src.nc) ? (uint32_t)(q0 + qi) * src.rowbytes + ch * 16u : SGL_OOB));
  }

  // K/V tiles: global -> LDS by DMA (see the dK/dV kernel), two stages (three workgroups per CU leave no room for a third):
  // tile+1 is requested right after the barrier that frees its stage and has the whole tile to land.  Slots sl = w + 4j;
  // sl < NIK: K image (row reads), else V image (transposed reads); pad chunks of a row point out of range -> zeros.
  uint32_t voff[NPW];
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    const int sl = w + 4 * j;
    voff[j] = (sl < NIK) ? att_chunk_off(sl * 64 + lane, KT, SCK, (int)src.nc, src.rowbytes)
                         : att_chunk_off((sl - NIK) * 64 + lane, KT, SCV, (int)src.nc, src.rowbytes);
  }
  const uint32_t tile_adv = (uint32_t)KT * src.rowbytes;
  auto issue = [&](int stage) {
    const uint32_t sb = lds0 + (uint32_t)(stage * STAGE);
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int sl = w + 4 * j;
      if (sl < NSLOT) att_dma16(sl < NIK ? dk_ : dv_, sb + (uint32_t)sl * 1024u, voff[j]);
      voff[j] += tile_adv;
    }
  };

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o[C::DT];
#pragma unroll
  for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;

  const int ntiles = (N + KT - 1) / KT;
  // retire the Q fragment loads before the first DMA: the compiler's vmcnt bookkeeping does not see the DMA
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) asm volatile("" ::"v"(qf[ks]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  issue(0);
  for (int tile = 0; tile < ntiles; ++tile) {
    const int cur = tile & 1;
    SGL_ATT_WAIT_BARRIER(0);   // tile landed (own DMA; everyone's behind the barrier); the other stage is free
    if (tile + 1 < ntiles) issue(cur ^ 1);
    const char* kb = smem + cur * STAGE;
    const char* vb = kb + KBYTES;
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const bf16x8 k0 = lds_row8(kb + qi * C::RSTR + (16 * ks + 8 * hh) * 2);
      const bf16x8 k1 = lds_row8(kb + (qi + 32) * C::RSTR + (16 * ks + 8 * hh) * 2);
      s0 = MFMA32(k0, qf[ks], s0);
      s1 = MFMA32(k1, qf[ks], s1);
    }
    if (tile == ntiles - 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = tile * KT + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (key >= N) s0[r] = -INFINITY;
        if (key + 32 >= N) s1[r] = -INFINITY;
      }
    }
    float mx = s0[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s0[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s1[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const bool grew = __any(m_new > m_run);  // wave-uniform: alpha == 1 exactly in every lane when nothing grew
    const float alpha = grew ? fast_exp2(c * (m_run - m_new)) : 1.0f;
    const float mc = m_new * c;
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s0[r] = fast_exp2(fmaf(s0[r], c, -mc));
      s1[r] = fast_exp2(fmaf(s1[r], c, -mc));
      rs += s0[r] + s1[r];
    }
    l_run = l_run * alpha + rs;
    m_run = m_new;
    if (grew) {
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    bf16x8 pb[4];
    pb[0] = pack8(s0, 0);
    pb[1] = pack8(s0, 8);
    pb[2] = pack8(s1, 0);
    pb[3] = pack8(s1, 8);
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) o[dt] = MFMA32(lds_trfrag(vb, C::TSTR, kk, dt * 32, lane), pb[kk], o[dt]);
  }
  const float l = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l;
  const int q = q0 + qi;
  if (q < N) {
    bf16* orow = out + ((size_t)b * N + q) * ((size_t)H * dh) + hd * dh;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d = dt * 32 + 8 * g4 + 4 * hh;
        if (d < dh) {
          bf16x4 v4;
#pragma unroll
          for (int r = 0; r < 4; ++r) v4[r] = (bf16)(o[dt][4 * g4 + r] * inv);
          *reinterpret_cast<bf16x4*>(orow + d) = v4;
        }
      }
    if (hh == 0) lse[(size_t)bh * N + q] = m_run * scale + __logf(l);
  }
}

// ======================================================================================================
// backward: dK, dV   (wave owns 32 keys; sweeps 32-query tiles staged in LDS)
// ======================================================================================================
template <int DP, bool TOK>
__global__ __launch_bounds__(256, 2) void attn_bwd_kv_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                             const bf16* __restrict__ V, const bf16* __restrict__ dO,
                                                             const float* __restrict__ aux, bf16* __restrict__ dqkv,
                                                             int B, int H, int N, int dh, int ld, float c, float scale) {
  using C = AttnCfg<DP>;
  constexpr int QT = 32;
  constexpr int SC = C::DSTR / 16;                         // chunks per image row (13)
  constexpr int NIS = (QT * SC + 63) / 64;                 // DMA instructions per image (7)
  constexpr int IMG = NIS * 1024;                          // one dual-use image incl. slack
  constexpr int STAGE = 2 * IMG + QT * 8;                  // Q image, dO image, {-lse*log2e, -delta*scale}[32]
  constexpr int NSLOT = 2 * NIS + 1;                       // DMA instructions per tile, dealt to the 4 waves round-robin
  static_assert(NSLOT <= 16 && NSLOT > 12, "four DMA slots per wave (three for the last wave)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  int bh, xb;
  if (!head_block((N + 127) / 128, B * H, bh, xb)) return;
  const int b = bh / H, hd = bh - b * H;
  const int D = H * dh;
  const int key0 = xb * 128 + w * 32;
  HeadSrc src = head_src(TOK ? ld : 0, b, hd, H, N, dh, DP);
  if constexpr (!TOK) {   // head-major: compile-time row geometry (the encoder's path keeps its constant-folded addressing)
    src.rowbytes = (uint32_t)DP * 2u;
    src.nc = (uint32_t)DP / 8u;
  }
  const __amdgpu_buffer_rsrc_t rk = make_rsrc(K + src.base, src.bytes);
  const __amdgpu_buffer_rsrc_t rv = make_rsrc(V + src.base, src.bytes);
  const bf16* dOb = dO + (size_t)b * N * D + hd * dh;
  const u32x4 dq_ = att_desc(Q + src.base, src.bytes);
  const u32x4 ddo = att_desc(dOb, (uint32_t)(((size_t)(N - 1) * D + dh) * 2));
  const u32x4 dax = att_desc(aux + (size_t)bh * N * 2, (uint32_t)((size_t)N * 8));
  const uint32_t lds0 = (uint32_t)(size_t)((SGL_LDS char*)smem);

  const int li = lane & 31, hh = lane >> 5;
  bf16x8 kfr[C::KS], vfr[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    const uint32_t ch = (uint32_t)(2 * ks + hh);
    const uint32_t off = (key0 + li < N && ch < src.nc) ? (uint32_t)(key0 + li) * src.rowbytes + ch * 16u : SGL_OOB;
    kfr[ks] = as_bf16x8(a_ldg(rk, off));
    vfr[ks] = as_bf16x8(a_ldg(rv, off));
  }

  // this wave's DMA slots: sl = w + 4j.  sl < NIS: Q image, sl < 2*NIS: dO image, sl == 2*NIS: the 32 {lse, delta} pairs.
  // Slots j = 0..2 are 16-byte image slots for every wave (straight-line code, descriptor picked once); j = 3 is an image
  // slot for waves 0-1, the pair slot for wave 2 and nothing for wave 3.  Tiles are requested in order, so the per-lane
  // offsets just advance by one tile per issue().
  uint32_t voff[4], vadv[4];
  u32x4 dsc[3];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int sl = w + 4 * j;
    if (sl < NIS) {
      voff[j] = att_chunk_off(sl * 64 + lane, QT, SC, (int)src.nc, src.rowbytes);
      vadv[j] = (uint32_t)QT * src.rowbytes;
    } else if (sl < 2 * NIS) {
      voff[j] = att_chunk_off((sl - NIS) * 64 + lane, QT, SC, dh / 8, (uint32_t)D * 2u);
      vadv[j] = (uint32_t)(QT * D * 2);
    } else {
      voff[j] = (uint32_t)lane * 4u;
      vadv[j] = (uint32_t)(QT * 8);
    }
    if (j < 3) dsc[j] = (sl < NIS) ? dq_ : ddo;
  }
  // one DMA instruction of this wave (j = 0..3) into `stage`; tiles past the end read out of range (zeros into a stage
  // nobody reads again), so the loop issues unconditionally and its vmcnt accounting is the same for every tile
  auto issue1 = [&](int j, int stage) {
    const uint32_t sb = lds0 + (uint32_t)(stage * STAGE) + (uint32_t)w * 1024u;
    if (j < 3) {
      att_dma16(dsc[j], sb + (uint32_t)j * 4096u, voff[j]);   // out-of-range lanes stay >= 2^31 as they advance
    } else {
      if (w < 2) att_dma16(ddo, sb + 3u * 4096u, voff[3]);
      else if (w == 2) att_dma4(dax, lds0 + (uint32_t)(stage * STAGE + 2 * IMG), voff[3]);
    }
    voff[j] += vadv[j];
  };
  auto issue = [&](int stage) {
#pragma unroll
    for (int j = 0; j < 4; ++j) issue1(j, stage);
  };

  f32x16 dk[C::DT], dv[C::DT];
#pragma unroll
  for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }

  const int ntiles = (N + QT - 1) / QT;
  // retire the K/V fragment loads before the first DMA: the compiler's own vmcnt bookkeeping does not see the DMA
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) asm volatile("" ::"v"(kfr[ks]), "v"(vfr[ks]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  issue(0);
  issue(1);
  int st = 0;                                  // stage of tile qt
  for (int qt = 0; qt < ntiles; ++qt) {
    // tile qt has landed (own DMA: all but the younger tile's instructions; everyone's: barrier) and every wave is done
    // reading the stage tile qt+2 will overwrite
    if (w == 3) SGL_ATT_WAIT_BARRIER(3); else SGL_ATT_WAIT_BARRIER(4);
    const int st_next = (st == 0) ? 2 : st - 1;   // stage of tile qt+2 (= of tile qt-1)
    const char* qimg = smem + st * STAGE;
    const char* dimg = qimg + IMG;
    const float* ld = reinterpret_cast<const float*>(qimg + 2 * IMG);
    st = (st == 2) ? 0 : st + 1;
    f32x16 S, dP;
#pragma unroll
    for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
    // ---- phase A: S = Q·Kᵀ, dP = dO·Vᵀ.  All row fragments are requested up front and the schedule is pinned
    // (sched_group_barrier): left alone, hipcc issued two LDS reads and waited for them in front of EVERY MFMA, so the
    // matrix pipe idled for an LDS round trip 22 times per tile (PMC: waves parked 43 % of the time).
    bf16x8 qa[C::KS], da[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      qa[ks] = lds_row8(qimg + li * C::DSTR + (16 * ks + 8 * hh) * 2);
      da[ks] = lds_row8(dimg + li * C::DSTR + (16 * ks + 8 * hh) * 2);
    }
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      S = MFMA32(qa[ks], kfr[ks], S);
      dP = MFMA32(da[ks], vfr[ks], dP);
    }
    if constexpr (C::KS == 5) {   // 4 fragments in flight ahead of the MFMA chain
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- the first two transposed fragments of phase B depend only on the staged tile: their latency runs under the
    // softmax arithmetic.  Fragment order: (dv, dk) for (dt, kk) = (0,0) (0,1) (1,0) ...
    constexpr int NF = 2 * 2 * C::DT;
    bf16x8 fr[NF];
    auto frag = [&](int i) {
      const int which = i & 1, kk = (i >> 1) & 1, dt = i >> 2;
      return lds_trfrag(which ? qimg : dimg, C::DSTR, kk, dt * 32, lane);
    };
    constexpr int PRE = NF < 2 ? NF : 2;
#pragma unroll
    for (int i = 0; i < PRE; ++i) fr[i] = frag(i);
    __builtin_amdgcn_sched_barrier(0);
    // S[r], dP[r]: query row (r&3) + 8*(r>>2) + 4*hh of the tile, key = lane & 31;  ld[2*row] = -lse*log2e, ld[2*row+1] =
    // -delta*scale (written by the dQ kernel): p = exp2(S*c - lse*log2e), dS = p * (dP*scale - delta*scale)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 A0 = *reinterpret_cast<const f32x4*>(ld + 2 * (8 * g4 + 4 * hh));
      const f32x4 A1 = *reinterpret_cast<const f32x4*>(ld + 2 * (8 * g4 + 4 * hh) + 4);
      const float L4[4] = {A0[0], A0[2], A1[0], A1[2]};
      const float D4[4] = {A0[1], A0[3], A1[1], A1[3]};
#pragma unroll
      for (int r3 = 0; r3 < 4; ++r3) {
        const int r = 4 * g4 + r3;
        const float p = fast_exp2(fmaf(S[r], c, L4[r3]));
        S[r] = p;
        dP[r] = p * fmaf(dP[r], scale, D4[r3]);
      }
    }
    bf16x8 pb[2], dsb[2];
    pb[0] = pack8(S, 0);
    pb[1] = pack8(S, 8);
    dsb[0] = pack8(dP, 0);
    dsb[1] = pack8(dP, 8);
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase B: dVᵀ += dOᵀ·P, dKᵀ += Qᵀ·dS, two fragments ahead of the MFMA that consumes them
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      if (i + PRE < NF) fr[i + PRE] = frag(i + PRE);
      const int which = i & 1, kk = (i >> 1) & 1, dt = i >> 2;
      if (which) dk[dt] = MFMA32(fr[i], dsb[kk], dk[dt]);
      else dv[dt] = MFMA32(fr[i], pb[kk], dv[dt]);
      // the request for tile qt+2, one DMA instruction behind every third MFMA: the texture addresser takes ~16 cycles per
      // 64-lane 16-byte instruction and a wave that cannot hand its instruction over stalls IN ORDER — issued as a burst
      // after the barrier, four waves' sixteen instructions held every wave's phase A back (15 % of the kernel); here
      // the hand-over runs under the matrix pipe's 32 cycles
      if (i % 3 == 1) issue1(i / 3, st_next);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = (NF + 1) / 3; j < 4; ++j) issue1(j, st_next);   // head dims with fewer than 11 MFMAs in phase B
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing out-of-range requests
  const int key = key0 + li;
  if (key < N) {
    bf16* krow = dqkv + ((size_t)b * N + key) * (3 * (size_t)D) + D + hd * dh;
    bf16* vrow = krow + D;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d = dt * 32 + 8 * g4 + 4 * hh;
        if (d < dh) {
          bf16x4 a, g;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            a[r] = (bf16)dk[dt][4 * g4 + r];
            g[r] = (bf16)dv[dt][4 * g4 + r];
          }
          *reinterpret_cast<bf16x4*>(krow + d) = a;
          *reinterpret_cast<bf16x4*>(vrow + d) = g;
        }
      }
  }
}

// ======================================================================================================
// backward: dQ   (wave owns 32 queries; sweeps 32-key tiles staged in LDS)
// ======================================================================================================
template <int DP, bool TOK>
__global__ __launch_bounds__(256, 2) void attn_bwd_q_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                            const bf16* __restrict__ V, const bf16* __restrict__ O,
                                                            const bf16* __restrict__ dO, const float* __restrict__ lse,
                                                            float* __restrict__ aux, bf16* __restrict__ dqkv,
                                                            int B, int H, int N, int dh, int ld, float c, float scale) {
  using C = AttnCfg<DP>;
  constexpr int KT = 32;
  constexpr int SCK = C::DSTR / 16, SCV = C::RSTR / 16;   // chunks per image row: K (dual-use image), V (row reads only)
  constexpr int NIK = (KT * SCK + 63) / 64, NIV = (KT * SCV + 63) / 64;   // DMA instructions per image
  constexpr int KIMG = NIK * 1024, VIMG = NIV * 1024, STAGE = KIMG + VIMG;
  constexpr int NSLOT = NIK + NIV;
  static_assert(NSLOT <= 16, "at most four DMA slots per wave");
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  int bh, xb;
  if (!head_block((N + 127) / 128, B * H, bh, xb)) return;
  const int b = bh / H, hd = bh - b * H;
  const int D = H * dh;
  const int q0 = xb * 128 + w * 32;
  HeadSrc src = head_src(TOK ? ld : 0, b, hd, H, N, dh, DP);
  if constexpr (!TOK) {   // head-major: compile-time row geometry (the encoder's path keeps its constant-folded addressing)
    src.rowbytes = (uint32_t)DP * 2u;
    src.nc = (uint32_t)DP / 8u;
  }
  const __amdgpu_buffer_rsrc_t rq = make_rsrc(Q + src.base, src.bytes);
  const bf16* dOb = dO + (size_t)b * N * D + hd * dh;
  const __amdgpu_buffer_rsrc_t rdo = make_rsrc(dOb, (uint32_t)(((size_t)(N - 1) * D + dh) * 2));
  const u32x4 dk_ = att_desc(K + src.base, src.bytes);
  const u32x4 dv_ = att_desc(V + src.base, src.bytes);
  const uint32_t lds0 = (uint32_t)(size_t)((SGL_LDS char*)smem);

  const int li = lane & 31, hh = lane >> 5;
  const int q = q0 + li;
  bf16x8 qf[C::KS], dof[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    const int col = 16 * ks + 8 * hh;
    qf[ks] = as_bf16x8(a_ldg(rq, (q < N && (uint32_t)(col >> 3) < src.nc) ? (uint32_t)q * src.rowbytes + (uint32_t)col * 2u
                                                                           : SGL_OOB));
    dof[ks] = as_bf16x8(a_ldg(rdo, (q < N && col < dh) ? (uint32_t)(((size_t)q * D + col) * 2) : SGL_OOB));
  }
  // delta = rowsum(dO ∘ O) of this lane's query, computed here (each lane holds half of its row of dO; O's half is loaded
  // once) instead of in a pre-pass over O and dO; the pair {-lse*log2e, -delta*scale} is what both backward kernels add
  // inside their fused multiply-adds, and the dK/dV kernel (launched after this one) streams it from `aux`.
  float dl = 0.f;
  {
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(O + (size_t)b * N * D + hd * dh, (uint32_t)(((size_t)(N - 1) * D + dh) * 2));
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const int col = 16 * ks + 8 * hh;
      const bf16x8 of = as_bf16x8(a_ldg(ro, (q < N && col < dh) ? (uint32_t)(((size_t)q * D + col) * 2) : SGL_OOB));
#pragma unroll
      for (int j = 0; j < 8; ++j) dl = fmaf((float)of[j], (float)dof[ks][j], dl);
    }
    dl += __shfl_xor(dl, 32, 64);
  }
  const float Lq = (q < N) ? -lse[(size_t)bh * N + q] * LOG2E : -INFINITY;
  const float Dq = (q < N) ? -dl * scale : 0.f;
  if (q < N && hh == 0) *reinterpret_cast<float2*>(aux + 2 * ((size_t)bh * N + q)) = make_float2(Lq, Dq);

  // K/V tiles: global -> LDS by DMA, three stages, tile kt+2 requested while tile kt is computed (see the dK/dV kernel).
  // This wave's slots: sl = w + 4j; sl < NIK: K image, else V image.
  uint32_t voff[4];
  u32x4 dsc[4];
  uint32_t ldst[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int sl = w + 4 * j;
    if (sl < NIK) {
      voff[j] = att_chunk_off(sl * 64 + lane, KT, SCK, (int)src.nc, src.rowbytes);
      dsc[j] = dk_;
      ldst[j] = (uint32_t)sl * 1024u;
    } else {
      voff[j] = att_chunk_off((sl - NIK) * 64 + lane, KT, SCV, (int)src.nc, src.rowbytes);
      dsc[j] = dv_;
      ldst[j] = (uint32_t)(KIMG + (sl - NIK) * 1024);
    }
  }
  auto issue1 = [&](int j, int stage) {   // tiles past the end read out of range: zeros into a stage nobody reads again
    if (w + 4 * j < NSLOT) att_dma16(dsc[j], lds0 + (uint32_t)(stage * STAGE) + ldst[j], voff[j]);
    voff[j] += (uint32_t)KT * src.rowbytes;
  };

  f32x16 dq[C::DT];
#pragma unroll
  for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

  const int ntiles = (N + KT - 1) / KT;
  // retire the register loads before the first DMA: the compiler's vmcnt bookkeeping does not see the DMA
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) asm volatile("" ::"v"(qf[ks]), "v"(dof[ks]));
  asm volatile("" ::"v"(Lq), "v"(Dq));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < 4; ++j) issue1(j, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) issue1(j, 1);
  int st = 0;
  for (int kt = 0; kt < ntiles; ++kt) {
    // tile kt has landed (own DMA: all but the younger tile's; everyone's: barrier); the stage of tile kt-1 is free
    const int nw = (NSLOT - w + 3) / 4;           // this wave's DMA instructions per tile
    if (nw >= 4) SGL_ATT_WAIT_BARRIER(4);
    else if (nw == 3) SGL_ATT_WAIT_BARRIER(3);
    else if (nw == 2) SGL_ATT_WAIT_BARRIER(2);
    else SGL_ATT_WAIT_BARRIER(1);
    const int st_next = (st == 0) ? 2 : st - 1;
    const char* kimg = smem + st * STAGE;
    const char* vimg = kimg + KIMG;
    st = (st == 2) ? 0 : st + 1;
    f32x16 S, dP;
#pragma unroll
    for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
    // phase A: Sᵀ = K·Qᵀ, dPᵀ = V·dOᵀ with the row fragments requested up front and the schedule pinned (see the kv kernel)
    bf16x8 ka[C::KS], va[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      ka[ks] = lds_row8(kimg + li * C::DSTR + (16 * ks + 8 * hh) * 2);
      va[ks] = lds_row8(vimg + li * C::RSTR + (16 * ks + 8 * hh) * 2);
    }
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      S = MFMA32(ka[ks], qf[ks], S);
      dP = MFMA32(va[ks], dof[ks], dP);
    }
    if constexpr (C::KS == 5) {
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // S[r], dP[r]: key row (r&3) + 8*(r>>2) + 4*hh of the tile, query = lane & 31
    // the transposed K fragments of the dQ product depend only on the staged tile: request them now, so that their LDS
    // latency runs under the softmax arithmetic below instead of in front of every MFMA
    bf16x8 kt_frag[C::DT][2];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) kt_frag[dt][kk] = lds_trfrag(kimg, C::DSTR, kk, dt * 32, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = fast_exp2(fmaf(S[r], c, Lq));
      dP[r] = p * fmaf(dP[r], scale, Dq);
    }
    if (kt == ntiles - 1) {   // keys past N exist only in the last tile (wave-uniform branch: 4 VALU per score saved elsewhere)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * KT + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (key >= N) dP[r] = 0.f;
      }
    }
    bf16x8 dsb[2];
    dsb[0] = pack8(dP, 0);
    dsb[1] = pack8(dP, 8);
    // dQᵀ += Kᵀ·dSᵀ, with the request for tile kt+2 spread behind the MFMAs (see the dK/dV kernel)
#pragma unroll
    for (int i = 0; i < 2 * C::DT; ++i) {
      dq[i >> 1] = MFMA32(kt_frag[i >> 1][i & 1], dsb[i & 1], dq[i >> 1]);
      if (i < 4) issue1(i, st_next);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 2 * C::DT; j < 4; ++j) issue1(j, st_next);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing out-of-range requests
  if (q < N) {
    bf16* qrow = dqkv + ((size_t)b * N + q) * (3 * (size_t)D) + hd * dh;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d = dt * 32 + 8 * g4 + 4 * hh;
        if (d < dh) {
          bf16x4 a;
#pragma unroll
          for (int r = 0; r < 4; ++r) a[r] = (bf16)dq[dt][4 * g4 + r];
          *reinterpret_cast<bf16x4*>(qrow + d) = a;
        }
      }
  }
}

// ======================================================================================================
// launchers
// ======================================================================================================
template <int DP>
static hipError_t fwd_launch(const bf16* q, const bf16* k, const bf16* v, bf16* out, float* lse, int B, int H, int N,
                             int dh, int ld, hipStream_t s) {
  using C = AttnCfg<DP>;
  constexpr int smem = 2 * (64 * C::RSTR + 64 * C::TSTR);
  const float scale = 1.0f / sqrtf((float)dh);
  if (ld > 0)
    hipLaunchKernelGGL((attn_fwd_kernel<DP, true>), head_grid((N + 127) / 128, B * H), dim3(256), smem, s, q, k, v, out, lse,
                       B, H, N, dh, ld, scale * 1.4426950408889634f, scale);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<DP, false>), head_grid((N + 127) / 128, B * H), dim3(256), smem, s, q, k, v, out,
                       lse, B, H, N, dh, 0, scale * 1.4426950408889634f, scale);
  return hipGetLastError();
}

template <int DP>
static hipError_t bwd_launch(const bf16* q, const bf16* k, const bf16* v, const bf16* out, const bf16* dout,
                             const float* lse, bf16* dqkv, float* delta, int B, int H, int N, int dh, int ld,
                             hipStream_t s) {
  using C = AttnCfg<DP>;
  const float scale = 1.0f / sqrtf((float)dh);
  const float c = scale * 1.4426950408889634f;
  // dQ first: it also computes delta and leaves the {-lse*log2e, -delta*scale} pairs in `delta` for the dK/dV kernel
  constexpr int smem_kv = 3 * (2 * ((32 * (C::DSTR / 16) + 63) / 64) * 1024 + 32 * 8);
  constexpr int smem_q = 3 * (((32 * (C::DSTR / 16) + 63) / 64) + ((32 * (C::RSTR / 16) + 63) / 64)) * 1024;
  const dim3 grid = head_grid((N + 127) / 128, B * H), block(256);
  if (ld > 0)
    hipLaunchKernelGGL((attn_bwd_q_kernel<DP, true>), grid, block, smem_q, s, q, k, v, out, dout, lse, delta, dqkv, B, H, N,
                       dh, ld, c, scale);
  else
    hipLaunchKernelGGL((attn_bwd_q_kernel<DP, false>), grid, block, smem_q, s, q, k, v, out, dout, lse, delta, dqkv, B, H, N,
                       dh, 0, c, scale);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (ld > 0)
    hipLaunchKernelGGL((attn_bwd_kv_kernel<DP, true>), grid, block, smem_kv, s, q, k, v, dout, delta, dqkv, B, H, N, dh, ld, c,
                       scale);
  else
    hipLaunchKernelGGL((attn_bwd_kv_kernel<DP, false>), grid, block, smem_kv, s, q, k, v, dout, delta, dqkv, B, H, N, dh, 0,
                       c, scale);
  return hipGetLastError();
}

static bool attn_shape_ok(int N, int dh, int DP, int H, int ld) {
  if (dh % 8 || DP != ((dh + 15) / 16) * 16) return false;
  if (DP != 16 && DP != 32 && DP != 48 && DP != 64 && DP != 80 && DP != 96) return false;
  if (ld < 0 || (ld > 0 && (ld % 8 || ld < H * dh))) return false;          // 16-byte aligned token rows
  if ((size_t)N * (ld > 0 ? ld : DP) * 2 >= (1ull << 31)) return false;
  if ((size_t)N * H * dh * 2 >= (1ull << 31)) return false;
  return true;
}

hipError_t attn_fwd(const void* q, const void* k, const void* v, int dtype, void* out, float* lse, int B, int H, int N,
                    int dh, int DP, int ld, hipStream_t s) {
  if (B * H == 0 || N == 0) return hipSuccess;
  if (dtype == DT_F32)
    return attn_ref_fwd((const float*)q, (const float*)k, (const float*)v, (float*)out, lse, B, H, N, dh, DP, ld, s);
  if (dtype == DT_F32_MFMA)
    return attn_f32_fwd((const float*)q, (const float*)k, (const float*)v, (float*)out, lse, B, H, N, dh, DP, ld, s);
  if (!attn_shape_ok(N, dh, DP, H, ld)) return hipErrorInvalidValue;
  if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v)) & 15) return hipErrorInvalidValue;
#define SGL_F(DPV) \
  case DPV: return fwd_launch<DPV>((const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)out, lse, B, H, N, dh, ld, s);
  switch (DP) {
    SGL_F(16) SGL_F(32) SGL_F(48) SGL_F(64) SGL_F(80) SGL_F(96)
  }
#undef SGL_F
  return hipErrorInvalidValue;
}

hipError_t attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                    int dtype, void* dqkv, float* delta, float* /*unused*/, int B, int H, int N, int dh, int DP,
                    int ld, hipStream_t s) {
  if (B * H == 0 || N == 0) return hipSuccess;
  if (dtype == DT_F32)
    return attn_ref_bwd((const float*)q, (const float*)k, (const float*)v, (const float*)out, (const float*)dout, lse,
                        (float*)dqkv, delta, B, H, N, dh, DP, ld, s);
  if (dtype == DT_F32_MFMA)
    return attn_f32_bwd((const float*)q, (const float*)k, (const float*)v, (const float*)out, (const float*)dout, lse,
                        (float*)dqkv, delta, B, H, N, dh, DP, ld, s);
  if (!attn_shape_ok(N, dh, DP, H, ld)) return hipErrorInvalidValue;
  if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v)) & 15) return hipErrorInvalidValue;
#define SGL_B(DPV)                                                                                             \
  case DPV:                                                                                                    \
    return bwd_launch<DPV>((const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)out, (const bf16*)dout, \
                           lse, (bf16*)dqkv, delta, B, H, N, dh, ld, s);
  switch (DP) {
    SGL_B(16) SGL_B(32) SGL_B(48) SGL_B(64) SGL_B(80) SGL_B(96)
  }
#undef SGL_B
  return hipErrorInvalidValue;
}

// delta_scratch of attn_bwd: {-lse*log2e, -delta*scale} PAIRS per (batch, head, query) = 2*B*H*N floats (since round 2)
size_t attn_bwd_scratch_bytes(int, int B, int H, int N, int, int) { return (size_t)2 * B * H * N * sizeof(float); }

}  // namespace sgl

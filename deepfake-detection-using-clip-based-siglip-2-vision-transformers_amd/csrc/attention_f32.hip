// fp32 attention ON the matrix cores (compute mode SGL_DTYPE_BF16X3, the strict mode that runs on MFMA): softmax(QKᵀ/√dh)·V
// and its backward with v_mfma_f32_32x32x2_f32 — fp32 operands, fp32 products, fp32 accumulate, so the two attention
// contractions are as exact as the plain-FMA reference kernels of attention_ref.hip (same math, TF:modeling_siglip.py:227-247)
// at matrix-core speed (fp32 MFMA peak 157 TFLOP/s; the reference kernels run at a few TFLOP/s).
//
// Same dataflow as the bf16 kernels of attention.hip, 4 waves per workgroup, each wave owns 32 rows:
//   forward   ("query on the lane")  Sᵀ = K·Qᵀ ; online softmax per lane ; Oᵀ += Vᵀ·Pᵀ
//   dQ kernel ("query on the lane")  Sᵀ = K·Qᵀ, dPᵀ = V·dOᵀ, dSᵀ = Pᵀ∘(dPᵀ − δ)·scale, dQᵀ += Kᵀ·dSᵀ ; also writes δ
//   dK/dV     ("key on the lane")    S = Q·Kᵀ, dP = dO·Vᵀ, dVᵀ += dOᵀ·P, dKᵀ += Qᵀ·dS
// A 32x32x2 MFMA takes A[m][k] from lane (m = lane & 31, k = lane >> 5) and B[k][n] from lane (n = lane & 31, k = lane >> 5);
// the accumulator register i of lane l holds C[8*(i>>2) + 4*(l>>5) + (i&3)][l & 31].  The second product of each kernel
// contracts over the index the accumulators are spread over, and since a contraction index may be visited in any order the
// accumulator registers feed it DIRECTLY as the B operand: k-step (a, b') pairs register i = 4a + b' of the two lane halves,
// i.e. rows 8a + b' (lower half) and 8a + 4 + b' (upper half), and the A operand is read from the same two rows.
// No atomics; bitwise reproducible.  Tiles are fetched one ahead into registers (float4) and written to LDS between two
// barriers (B = 32 per layer: forward 2.48 -> 1.65 ms, dQ 4.74 -> 2.09 ms, dK/dV 5.38 -> 2.79 ms against synchronous scalar
// staging; 47 TFLOP/s of fp32 MFMA in the forward.  Strict mode is not the benchmarked path: no LDS-DMA, no anti-phase).
// q,k,v: fp32, head-major [B][H][N][DP] (ld = 0) or token-major column blocks (ld > 0), as attention_ref.hip.
#include "common.hip.h"
#include "kernels.h"

namespace sgl {

#define MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr int AF_MAXD = 96;          // head_dim <= 96 (3 output tiles of 32)
constexpr int AF_RS = AF_MAXD + 1;   // LDS row stride in floats (odd: the 32 rows of a column read hit 32 banks)
constexpr int AF_TILE = 32 * AF_RS;  // one staged 32-row tile

__device__ __forceinline__ size_t af_base(int ld, int b, int h, int H, int N, int dh, int DP, int& rs) {
  if (ld > 0) { rs = ld; return (size_t)b * N * ld + (size_t)h * dh; }
  rs = DP;
  return ((size_t)b * H + h) * (size_t)N * DP;
}

// Tile staging: rows [r0, r0+32) x dh of a row-major source (row stride rs floats, 16-byte aligned rows) are fetched as
// float4 into registers one tile AHEAD (the loads fly under the current tile's MFMAs) and written to the LDS tile between two
// barriers.  Rows past N arrive as zeros; the columns that pad dh up to the next multiple of 32 are zeroed once (af_zero).
constexpr int AF_NLD = 3;   // float4 per thread and tile: 32 rows x (dh / 4 <= 24) chunks / 256 threads
__device__ __forceinline__ void af_load(f32x4 (&r)[AF_NLD], const float* src, int rs, int r0, int N, int nc4, int t) {
#pragma unroll
  for (int i = 0; i < AF_NLD; ++i) {
    const int idx = t + 256 * i;
    const int row = idx / nc4, c4 = idx - row * nc4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    r[i] = (idx < 32 * nc4 && r0 + row < N) ? *reinterpret_cast<const f32x4*>(src + (size_t)(r0 + row) * rs + 4 * c4) : z;
  }
}
__device__ __forceinline__ void af_store(float* tile, const f32x4 (&r)[AF_NLD], int nc4, int t) {
#pragma unroll
  for (int i = 0; i < AF_NLD; ++i) {
    const int idx = t + 256 * i;
    const int row = idx / nc4, c4 = idx - row * nc4;
    if (idx < 32 * nc4) {
      float* d = tile + row * AF_RS + 4 * c4;
      d[0] = r[i][0]; d[1] = r[i][1]; d[2] = r[i][2]; d[3] = r[i][3];
    }
  }
}
__device__ __forceinline__ void af_zero(float* tile, int t) {
  for (int i = t; i < AF_TILE; i += 256) tile[i] = 0.f;
}

// ======================================================================================================
__global__ __launch_bounds__(256) void attn_f32_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                           const float* __restrict__ V, float* __restrict__ out,
                                                           float* __restrict__ lse, int H, int N, int dh, int DP, int ld,
                                                           float scale) {
  __shared__ float ktile[AF_TILE], vtile[AF_TILE];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int bh = blockIdx.y, h = bh % H, b = bh / H;
  int rs;
  const size_t hb = af_base(ld, b, h, H, N, dh, DP, rs);
  const int q = blockIdx.x * 128 + w * 32 + li;
  const int ks = (dh + 1) / 2;             // k-steps over the head dim (dh is a multiple of 8)
  const int ndt = (dh + 31) / 32;          // 32-row tiles of Oᵀ
  const int dpad = ndt * 32;
  float qf[AF_MAXD / 2];
#pragma unroll
  for (int s = 0; s < AF_MAXD / 2; ++s)
    qf[s] = (s < ks && q < N && 2 * s + hh < dh) ? Q[hb + (size_t)q * rs + 2 * s + hh] : 0.f;
  f32x16 o[3];
#pragma unroll
  for (int dt = 0; dt < 3; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int nc4 = dh >> 2;
  f32x4 pk[AF_NLD], pv[AF_NLD];
  af_zero(ktile, t);
  af_zero(vtile, t);
  af_load(pk, K + hb, rs, 0, N, nc4, t);
  af_load(pv, V + hb, rs, 0, N, nc4, t);
  for (int k0 = 0; k0 < N; k0 += 32) {
    __syncthreads();
    af_store(ktile, pk, nc4, t);
    af_store(vtile, pv, nc4, t);
    __syncthreads();
    if (k0 + 32 < N) {
      af_load(pk, K + hb, rs, k0 + 32, N, nc4, t);
      af_load(pv, V + hb, rs, k0 + 32, N, nc4, t);
    }
    f32x16 s16;
#pragma unroll
    for (int i = 0; i < 16; ++i) s16[i] = 0.f;
#pragma unroll
    for (int s = 0; s < AF_MAXD / 2; ++s)
      if (s < ks) s16 = MFMA_F32(ktile[li * AF_RS + 2 * s + hh], qf[s], s16);
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = k0 + 8 * (i >> 2) + 4 * hh + (i & 3);
      s16[i] = (key < N) ? s16[i] * scale : -INFINITY;
      mx = fmaxf(mx, s16[i]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);      // exp(-inf) = 0 on the first tile
    float rsum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s16[i] = expf(s16[i] - m_new);
      rsum += s16[i];
    }
    l_run = l_run * alpha + rsum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
      if (dt < ndt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int krow = 8 * (i >> 2) + 4 * hh + (i & 3);
          o[dt] = MFMA_F32(vtile[krow * AF_RS + dt * 32 + li], s16[i], o[dt]);
        }
      }
  }
  const float l = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l;
  if (q < N) {
    float* orow = out + ((size_t)b * N + q) * ((size_t)H * dh) + h * dh;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
      if (dt < ndt) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d = dt * 32 + 8 * g4 + 4 * hh;
          if (d < dh) {
            f32x4 v4;
#pragma unroll
            for (int r = 0; r < 4; ++r) v4[r] = o[dt][4 * g4 + r] * inv;
            *reinterpret_cast<f32x4*>(orow + d) = v4;
          }
        }
      }
    if (hh == 0) lse[(size_t)bh * N + q] = m_run + logf(l);
  }
}

// ======================================================================================================
// dQ (+ delta = rowsum(dO ∘ O) into `delta`, plain [B][H][N] floats for the dK/dV kernel)
__global__ __launch_bounds__(256) void attn_f32_bwd_q_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                             const float* __restrict__ V, const float* __restrict__ O,
                                                             const float* __restrict__ dO, const float* __restrict__ lse,
                                                             float* __restrict__ dqkv, float* __restrict__ delta, int H,
                                                             int N, int dh, int DP, int ld, float scale) {
  __shared__ float ktile[AF_TILE], vtile[AF_TILE];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int bh = blockIdx.y, h = bh % H, b = bh / H;
  const int D = H * dh;
  int rs;
  const size_t hb = af_base(ld, b, h, H, N, dh, DP, rs);
  const int q = blockIdx.x * 128 + w * 32 + li;
  const int ks = (dh + 1) / 2, ndt = (dh + 31) / 32, dpad = ndt * 32;
  float qf[AF_MAXD / 2], dof[AF_MAXD / 2];
  float dl = 0.f;
  const float* orow = O + ((size_t)b * N + q) * D + h * dh;
  const float* dorow = dO + ((size_t)b * N + q) * D + h * dh;
#pragma unroll
  for (int s = 0; s < AF_MAXD / 2; ++s) {
    const bool ok = s < ks && q < N && 2 * s + hh < dh;
    qf[s] = ok ? Q[hb + (size_t)q * rs + 2 * s + hh] : 0.f;
    dof[s] = ok ? dorow[2 * s + hh] : 0.f;
    if (ok) dl = fmaf(dof[s], orow[2 * s + hh], dl);
  }
  dl += __shfl_xor(dl, 32, 64);
  const float Lq = (q < N) ? lse[(size_t)bh * N + q] : INFINITY;
  if (q < N && hh == 0) delta[(size_t)bh * N + q] = dl;
  f32x16 dq[3];
#pragma unroll
  for (int dt = 0; dt < 3; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[dt][i] = 0.f;
  const int nc4 = dh >> 2;
  f32x4 pk[AF_NLD], pv[AF_NLD];
  af_zero(ktile, t);
  af_zero(vtile, t);
  af_load(pk, K + hb, rs, 0, N, nc4, t);
  af_load(pv, V + hb, rs, 0, N, nc4, t);
  for (int k0 = 0; k0 < N; k0 += 32) {
    __syncthreads();
    af_store(ktile, pk, nc4, t);
    af_store(vtile, pv, nc4, t);
    __syncthreads();
    if (k0 + 32 < N) {
      af_load(pk, K + hb, rs, k0 + 32, N, nc4, t);
      af_load(pv, V + hb, rs, k0 + 32, N, nc4, t);
    }
    f32x16 S, dP;
#pragma unroll
    for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
    for (int s = 0; s < AF_MAXD / 2; ++s)
      if (s < ks) {
        S = MFMA_F32(ktile[li * AF_RS + 2 * s + hh], qf[s], S);
        dP = MFMA_F32(vtile[li * AF_RS + 2 * s + hh], dof[s], dP);
      }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = k0 + 8 * (i >> 2) + 4 * hh + (i & 3);
      const float p = (key < N) ? expf(S[i] * scale - Lq) : 0.f;
      dP[i] = p * (dP[i] - dl) * scale;
    }
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
      if (dt < ndt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int krow = 8 * (i >> 2) + 4 * hh + (i & 3);
          dq[dt] = MFMA_F32(ktile[krow * AF_RS + dt * 32 + li], dP[i], dq[dt]);
        }
      }
  }
  if (q < N) {
    float* qrow = dqkv + ((size_t)b * N + q) * (3 * (size_t)D) + h * dh;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
      if (dt < ndt) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d = dt * 32 + 8 * g4 + 4 * hh;
          if (d < dh) {
            f32x4 v4;
#pragma unroll
            for (int r = 0; r < 4; ++r) v4[r] = dq[dt][4 * g4 + r];
            *reinterpret_cast<f32x4*>(qrow + d) = v4;
          }
        }
      }
  }
}

// ======================================================================================================
// dK, dV (needs lse and the delta written by the dQ kernel)
__global__ __launch_bounds__(256) void attn_f32_bwd_kv_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                              const float* __restrict__ V, const float* __restrict__ dO,
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ delta, float* __restrict__ dqkv,
                                                              int H, int N, int dh, int DP, int ld, float scale) {
  __shared__ float qtile[AF_TILE], dotile[AF_TILE], lrow[32], drow[32];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int bh = blockIdx.y, h = bh % H, b = bh / H;
  const int D = H * dh;
  int rs;
  const size_t hb = af_base(ld, b, h, H, N, dh, DP, rs);
  const int key = blockIdx.x * 128 + w * 32 + li;
  const int ks = (dh + 1) / 2, ndt = (dh + 31) / 32, dpad = ndt * 32;
  float kf[AF_MAXD / 2], vf[AF_MAXD / 2];
#pragma unroll
  for (int s = 0; s < AF_MAXD / 2; ++s) {
    const bool ok = s < ks && key < N && 2 * s + hh < dh;
    kf[s] = ok ? K[hb + (size_t)key * rs + 2 * s + hh] : 0.f;
    vf[s] = ok ? V[hb + (size_t)key * rs + 2 * s + hh] : 0.f;
  }
  f32x16 dk[3], dv[3];
#pragma unroll
  for (int dt = 0; dt < 3; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[dt][i] = 0.f; dv[dt][i] = 0.f; }
  const float* dOb = dO + (size_t)b * N * D + h * dh;
  const int nc4 = dh >> 2;
  f32x4 pq[AF_NLD], pd[AF_NLD];
  af_zero(qtile, t);
  af_zero(dotile, t);
  af_load(pq, Q + hb, rs, 0, N, nc4, t);
  af_load(pd, dOb, D, 0, N, nc4, t);
  for (int q0 = 0; q0 < N; q0 += 32) {
    __syncthreads();
    af_store(qtile, pq, nc4, t);
    af_store(dotile, pd, nc4, t);
    if (t < 32) {
      lrow[t] = (q0 + t < N) ? lse[(size_t)bh * N + q0 + t] : INFINITY;
      drow[t] = (q0 + t < N) ? delta[(size_t)bh * N + q0 + t] : 0.f;
    }
    __syncthreads();
    if (q0 + 32 < N) {
      af_load(pq, Q + hb, rs, q0 + 32, N, nc4, t);
      af_load(pd, dOb, D, q0 + 32, N, nc4, t);
    }
    f32x16 S, dP;
#pragma unroll
    for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
    for (int s = 0; s < AF_MAXD / 2; ++s)
      if (s < ks) {
        S = MFMA_F32(qtile[li * AF_RS + 2 * s + hh], kf[s], S);
        dP = MFMA_F32(dotile[li * AF_RS + 2 * s + hh], vf[s], dP);
      }
    // S[i], dP[i]: query row 8*(i>>2) + 4*hh + (i&3) of the tile, key = this lane's key
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int qr = 8 * (i >> 2) + 4 * hh + (i & 3);
      const float p = (q0 + qr < N && key < N) ? expf(S[i] * scale - lrow[qr]) : 0.f;
      S[i] = p;
      dP[i] = p * (dP[i] - drow[qr]) * scale;
    }
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
      if (dt < ndt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int qr = 8 * (i >> 2) + 4 * hh + (i & 3);
          dv[dt] = MFMA_F32(dotile[qr * AF_RS + dt * 32 + li], S[i], dv[dt]);
          dk[dt] = MFMA_F32(qtile[qr * AF_RS + dt * 32 + li], dP[i], dk[dt]);
        }
      }
  }
  if (key < N) {
    float* krow = dqkv + ((size_t)b * N + key) * (3 * (size_t)D) + D + h * dh;
    float* vrow = krow + D;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
      if (dt < ndt) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d = dt * 32 + 8 * g4 + 4 * hh;
          if (d < dh) {
            f32x4 a, g;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              a[r] = dk[dt][4 * g4 + r];
              g[r] = dv[dt][4 * g4 + r];
            }
            *reinterpret_cast<f32x4*>(krow + d) = a;
            *reinterpret_cast<f32x4*>(vrow + d) = g;
          }
        }
      }
  }
}

// ======================================================================================================
static bool af_shape_ok(int N, int dh, int DP, int H, int ld) {
  if (dh % 8 || dh > AF_MAXD || DP < dh || DP % 4) return false;
  if (ld < 0 || (ld > 0 && (ld < H * dh || ld % 4))) return false;   // float4 row fetches
  return N > 0;
}

hipError_t attn_f32_fwd(const float* q, const float* k, const float* v, float* out, float* lse, int B, int H, int N,
                        int dh, int DP, int ld, hipStream_t s) {
  if (!af_shape_ok(N, dh, DP, H, ld)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(attn_f32_fwd_kernel, dim3((N + 127) / 128, B * H), dim3(256), 0, s, q, k, v, out, lse, H, N, dh, DP,
                     ld, 1.0f / sqrtf((float)dh));
  return hipGetLastError();
}

// delta: scratch of B*H*N floats (the caller's 2*B*H*N buffer is more than enough)
hipError_t attn_f32_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout,
                        const float* lse, float* dqkv, float* delta, int B, int H, int N, int dh, int DP, int ld,
                        hipStream_t s) {
  if (!af_shape_ok(N, dh, DP, H, ld)) return hipErrorInvalidValue;
  const dim3 grid((N + 127) / 128, B * H), block(256);
  const float scale = 1.0f / sqrtf((float)dh);
  hipLaunchKernelGGL(attn_f32_bwd_q_kernel, grid, block, 0, s, q, k, v, out, dout, lse, dqkv, delta, H, N, dh, DP, ld, scale);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(attn_f32_bwd_kv_kernel, grid, block, 0, s, q, k, v, dout, lse, delta, dqkv, H, N, dh, DP, ld, scale);
  return hipGetLastError();
}

}  // namespace sgl

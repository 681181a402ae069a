// Strict-fp32 attention (compute mode "fp32"): explicit softmax(QKᵀ/√dh)·V and its backward, one wave per
// query row (forward, dQ) or per key row (dK, dV).  Deterministic (no atomics).  Follows the eager path of
// TF:models/siglip/modeling_siglip.py:227-247.  Not a performance path; the bf16 path is attention.hip.
// q,k,v fp32: token-major [B*N][ld] column blocks of the QKV projection's output (ld > 0; head h of row r at r*ld + h*dh)
// or, ld = 0, head-major [B][H][N][DP]; out / dout token-major [B*N][H*dh]; dqkv token-major [B*N][3*H*dh].
#include "common.hip.h"
#include "kernels.h"

namespace sgl {

constexpr int AR_MAXD = 2;  // dh <= 128

// element offset of (token 0, column 0) of head (b, h) and the element stride between its token rows
__device__ __forceinline__ size_t ar_base(int ld, int b, int h, int H, int N, int dh, int DP, int& rs) {
  if (ld > 0) { rs = ld; return (size_t)b * N * ld + (size_t)h * dh; }
  rs = DP;
  return ((size_t)b * H + h) * (size_t)N * DP;
}

__global__ __launch_bounds__(256) void attn_ref_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                           const float* __restrict__ V, float* __restrict__ out,
                                                           float* __restrict__ lse, int H, int N, int dh, int DP,
                                                           int ld) {
  extern __shared__ __attribute__((aligned(16))) float ar_smem[];  // per wave: [N] scores + [128] q
  const int w = wave_id(), lane = lane_id();
  float* sc = ar_smem + (size_t)w * (N + 128);
  float* qs = sc + N;
  const int bh = blockIdx.y, h = bh % H, b = bh / H;
  const int i = blockIdx.x * 4 + w;
  if (i >= N) return;
  int rs;
  const size_t hb = ar_base(ld, b, h, H, N, dh, DP, rs);
  const float* Qb = Q + hb + (size_t)i * rs;
  const float* Kb = K + hb;
  const float* Vb = V + hb;
  const float scale = 1.0f / sqrtf((float)dh);
  for (int d = lane; d < dh; d += 64) qs[d] = Qb[d];
  __builtin_amdgcn_wave_barrier();
  float mx = -INFINITY;
  for (int n = lane; n < N; n += 64) {
    const float* kr = Kb + (size_t)n * rs;
    float s = 0.f;
    for (int d = 0; d < dh; ++d) s = fmaf(qs[d], kr[d], s);
    s *= scale;
    sc[n] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int n = lane; n < N; n += 64) {
    const float e = expf(sc[n] - mx);
    sc[n] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  __builtin_amdgcn_wave_barrier();
  float* orow = out + ((size_t)b * N + i) * (H * dh) + h * dh;
  for (int d = lane; d < dh; d += 64) {
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc = fmaf(sc[n], Vb[(size_t)n * rs + d], acc);
    orow[d] = acc * inv;
  }
  if (lane == 0) lse[(size_t)bh * N + i] = mx + logf(sum);
}

// dQ (+ delta): one wave per query
__global__ __launch_bounds__(256) void attn_ref_bwd_q_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                             const float* __restrict__ V, const float* __restrict__ O,
                                                             const float* __restrict__ dO, const float* __restrict__ lse,
                                                             float* __restrict__ dqkv, float* __restrict__ delta, int H,
                                                             int N, int dh, int DP, int ld) {
  extern __shared__ __attribute__((aligned(16))) float ar_smem[];  // per wave: [N] ds + [128] q + [128] do
  const int w = wave_id(), lane = lane_id();
  float* ds = ar_smem + (size_t)w * (N + 256);
  float* qs = ds + N;
  float* dos = qs + 128;
  const int bh = blockIdx.y, h = bh % H, b = bh / H;
  const int i = blockIdx.x * 4 + w;
  if (i >= N) return;
  const int D = H * dh;
  int rs;
  const size_t hb = ar_base(ld, b, h, H, N, dh, DP, rs);
  const float* Qb = Q + hb + (size_t)i * rs;
  const float* Kb = K + hb;
  const float* Vb = V + hb;
  const float* orow = O + ((size_t)b * N + i) * D + h * dh;
  const float* dorow = dO + ((size_t)b * N + i) * D + h * dh;
  const float scale = 1.0f / sqrtf((float)dh);
  float dl = 0.f;
  for (int d = lane; d < dh; d += 64) {
    qs[d] = Qb[d];
    dos[d] = dorow[d];
    dl += dorow[d] * orow[d];
  }
  dl = wave_sum(dl);
  __builtin_amdgcn_wave_barrier();
  const float L = lse[(size_t)bh * N + i];
  for (int n = lane; n < N; n += 64) {
    const float* kr = Kb + (size_t)n * rs;
    const float* vr = Vb + (size_t)n * rs;
    float s = 0.f, dp = 0.f;
    for (int d = 0; d < dh; ++d) {
      s = fmaf(qs[d], kr[d], s);
      dp = fmaf(dos[d], vr[d], dp);
    }
    const float p = expf(s * scale - L);
    ds[n] = p * (dp - dl) * scale;
  }
  __builtin_amdgcn_wave_barrier();
  float* dq = dqkv + ((size_t)b * N + i) * (3 * D) + h * dh;
  for (int d = lane; d < dh; d += 64) {
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc = fmaf(ds[n], Kb[(size_t)n * rs + d], acc);
    dq[d] = acc;
  }
  if (lane == 0) delta[(size_t)bh * N + i] = dl;
}

// dK, dV: one wave per key (needs delta from the kernel above)
__global__ __launch_bounds__(256) void attn_ref_bwd_kv_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                              const float* __restrict__ V,
                                                              const float* __restrict__ dO,
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ delta, float* __restrict__ dqkv,
                                                              int H, int N, int dh, int DP, int ld) {
  extern __shared__ __attribute__((aligned(16))) float ar_smem[];  // per wave: [N] p + [N] ds + [128] k + [128] v
  const int w = wave_id(), lane = lane_id();
  float* ps = ar_smem + (size_t)w * (2 * N + 256);
  float* ds = ps + N;
  float* ks = ds + N;
  float* vs = ks + 128;
  const int bh = blockIdx.y, h = bh % H, b = bh / H;
  const int j = blockIdx.x * 4 + w;
  if (j >= N) return;
  const int D = H * dh;
  int rs;
  const size_t hb = ar_base(ld, b, h, H, N, dh, DP, rs);
  const float* Qb = Q + hb;
  const float* Kr = K + hb + (size_t)j * rs;
  const float* Vr = V + hb + (size_t)j * rs;
  const float scale = 1.0f / sqrtf((float)dh);
  for (int d = lane; d < dh; d += 64) {
    ks[d] = Kr[d];
    vs[d] = Vr[d];
  }
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < N; i += 64) {
    const float* qr = Qb + (size_t)i * rs;
    const float* dor = dO + ((size_t)b * N + i) * D + h * dh;
    float s = 0.f, dp = 0.f;
    for (int d = 0; d < dh; ++d) {
      s = fmaf(qr[d], ks[d], s);
      dp = fmaf(dor[d], vs[d], dp);
    }
    const float p = expf(s * scale - lse[(size_t)bh * N + i]);
    ps[i] = p;
    ds[i] = p * (dp - delta[(size_t)bh * N + i]) * scale;
  }
  __builtin_amdgcn_wave_barrier();
  float* dk = dqkv + ((size_t)b * N + j) * (3 * D) + D + h * dh;
  float* dv = dk + D;
  for (int d = lane; d < dh; d += 64) {
    float ak = 0.f, av = 0.f;
    for (int i = 0; i < N; ++i) {
      ak = fmaf(ds[i], Qb[(size_t)i * rs + d], ak);
      av = fmaf(ps[i], dO[((size_t)b * N + i) * D + h * dh + d], av);
    }
    dk[d] = ak;
    dv[d] = av;
  }
}

hipError_t attn_ref_fwd(const float* q, const float* k, const float* v, float* out, float* lse, int B, int H, int N,
                        int dh, int DP, int ld, hipStream_t s) {
  if (dh > 128) return hipErrorInvalidValue;
  const size_t smem = (size_t)4 * (N + 128) * sizeof(float);
  if (smem > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_ref_fwd_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(attn_ref_fwd_kernel, dim3((N + 3) / 4, B * H), dim3(256), smem, s, q, k, v, out, lse, H, N, dh,
                     DP, ld);
  return hipGetLastError();
}

hipError_t attn_ref_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout,
                        const float* lse, float* dqkv, float* delta, int B, int H, int N, int dh, int DP, int ld,
                        hipStream_t s) {
  if (dh > 128) return hipErrorInvalidValue;
  const size_t smem_q = (size_t)4 * (N + 256) * sizeof(float);
  const size_t smem_kv = (size_t)4 * (2 * N + 256) * sizeof(float);
  if (smem_kv > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_ref_bwd_q_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_q);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_ref_bwd_kv_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_kv);
  if (e != hipSuccess) return e;
  dim3 grid((N + 3) / 4, B * H), block(256);
  hipLaunchKernelGGL(attn_ref_bwd_q_kernel, grid, block, smem_q, s, q, k, v, out, dout, lse, dqkv, delta, H, N, dh, DP,
                     ld);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(attn_ref_bwd_kv_kernel, grid, block, smem_kv, s, q, k, v, dout, lse, delta, dqkv, H, N, dh, DP, ld);
  return hipGetLastError();
}

}  // namespace sgl

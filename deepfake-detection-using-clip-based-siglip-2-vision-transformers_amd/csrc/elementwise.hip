// HBM-bound helper kernels around the encoder GEMMs (gfx950): patch gather (im2col), weight-shadow casts,
// bias-gradient column sums, position-table bicubic resize, and the 1-query attention-pool head.
#include "common.hip.h"
#include "kernels.h"

namespace sgl {

// ---- im2col: Conv2d(3->D, k=s=P, 'valid') becomes a GEMM over K = 3*P*P (TF:modeling_siglip.py:124-130,178).
// out[m][k], m = (b, gy, gx), k = c*P*P + ky*P + kx  (== weight.reshape(D,-1) order); k >= 3P² is zero pad.
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ pix, int channels_last,
                                                     T* __restrict__ out, int B, int H, int W, int P, int gh, int gw,
                                                     int Kp) {
  const size_t total = (size_t)B * gh * gw * Kp;
  const int K = 3 * P * P;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int k = (int)(idx % Kp);
    const size_t m = idx / Kp;
    float v = 0.f;
    if (k < K) {
      const int gx = (int)(m % gw);
      const int gy = (int)((m / gw) % gh);
      const int b = (int)(m / ((size_t)gw * gh));
      const int c = k / (P * P);
      const int r = k - c * P * P;
      const int ky = r / P, kx = r - ky * P;
      const int yy = gy * P + ky, xx = gx * P + kx;
      const size_t src = channels_last ? (((size_t)b * H + yy) * W + xx) * 3 + c
                                       : (((size_t)b * 3 + c) * H + yy) * W + xx;
      v = pix[src];
    }
    Elem<T>::st(out + idx, v);
  }
}

hipError_t im2col(const float* pix, int channels_last, void* out, int out_dtype, int B, int H, int W, int P, int Kp,
                  hipStream_t s) {
  const int gh = H / P, gw = W / P;
  const size_t total = (size_t)B * gh * gw * Kp;
  if (total == 0) return hipSuccess;
  int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (out_dtype == DT_BF16)
    hipLaunchKernelGGL(im2col_kernel<bf16>, dim3(blocks), dim3(256), 0, s, pix, channels_last, (bf16*)out, B, H, W, P,
                       gh, gw, Kp);
  else
    hipLaunchKernelGGL(im2col_kernel<float>, dim3(blocks), dim3(256), 0, s, pix, channels_last, (float*)out, B, H, W,
                       P, gh, gw, Kp);
  return hipGetLastError();
}

// ---- weight shadows: dst[Rp][Cp] = pad(cast(src[R][C])) and the transposed form dst[Cp][Rp] ------------
template <typename T>
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ src, int R, int C, int ld,
                                                       T* __restrict__ dst, int Rp, int Cp, int ldd) {
  const size_t total = (size_t)Rp * Cp;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int c = (int)(idx % Cp);
    const int r = (int)(idx / Cp);
    const float v = (r < R && c < C) ? src[(size_t)r * ld + c] : 0.f;
    Elem<T>::st(dst + (size_t)r * ldd + c, v);
  }
}

hipError_t cast_pad(const float* src, int R, int C, int ld, void* dst, int dst_dtype, int Rp, int Cp, int ldd,
                    hipStream_t s) {
  const size_t total = (size_t)Rp * Cp;
  if (total == 0) return hipSuccess;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dst_dtype == DT_BF16)
    hipLaunchKernelGGL(cast_pad_kernel<bf16>, dim3(blocks), dim3(256), 0, s, src, R, C, ld, (bf16*)dst, Rp, Cp, ldd);
  else
    hipLaunchKernelGGL(cast_pad_kernel<float>, dim3(blocks), dim3(256), 0, s, src, R, C, ld, (float*)dst, Rp, Cp, ldd);
  return hipGetLastError();
}

// ---- bf16x3 operand splits (compute mode SGL_DTYPE_BF16X3: strict arithmetic on the matrix cores) ------------------
// x = hi + lo + O(2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi).  A product a*b is taken as hi*hi + hi*lo + lo*hi (the
// lo*lo term, 2^-18 relative, is dropped) by running ONE bf16 MFMA GEMM over a reduction dimension three times as long:
//   NT (reduction along the row):   A' row = [ hi | hi | lo ],  B' row = [ hi | lo | hi ]      (split3_rows)
//   TN (reduction down the rows):   A' = [ hi ; hi ; lo ],      B' = [ hi ; lo ; hi ]          (split3_stack)
// so the generation-6 kernels, their tile order and every fused epilogue are reused unchanged; accumulation is fp32.
__device__ __forceinline__ void split_bf16(float x, bf16& hi, bf16& lo) {
  hi = (bf16)x;
  lo = (bf16)(x - (float)hi);
}

// dst [R][3*Cs] bf16 (Cs = C rounded up to 8, zero filled); b_side selects the B-operand segment order
__global__ __launch_bounds__(256) void split3_rows_kernel(const float* __restrict__ src, int R, int C, int ld,
                                                          bf16* __restrict__ dst, int Cs, int b_side) {
  const int cpr = Cs >> 3;
  const size_t total = (size_t)R * cpr;
  const bool vec = ((ld & 3) == 0) && ((((uintptr_t)src) & 15) == 0);
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int r = (int)(idx / cpr), c0 = (int)(idx - (size_t)r * cpr) * 8;
    const float* sp = src + (size_t)r * ld + c0;
    float v[8];
    if (vec && c0 + 8 <= C) {
      Vec<float, 4>::ld(sp, v);
      Vec<float, 4>::ld(sp + 4, v + 4);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (c0 + j < C) ? sp[j] : 0.f;
    }
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bf16 h, l;
      split_bf16(v[j], h, l);
      hi[j] = h;
      lo[j] = l;
    }
    bf16* dp = dst + (size_t)r * (3 * (size_t)Cs) + c0;
    *reinterpret_cast<bf16x8*>(dp) = hi;
    *reinterpret_cast<bf16x8*>(dp + Cs) = b_side ? lo : hi;
    *reinterpret_cast<bf16x8*>(dp + 2 * (size_t)Cs) = b_side ? hi : lo;
  }
}

hipError_t split3_rows(const float* src, int R, int C, int ld, void* dst, int Cs, int b_side, hipStream_t s) {
  const size_t total = (size_t)R * (Cs / 8);
  if (total == 0) return hipSuccess;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(split3_rows_kernel, dim3(blocks), dim3(256), 0, s, src, R, C, ld, (bf16*)dst, Cs, b_side);
  return hipGetLastError();
}

// dst [3*R][Cs] bf16: rows [0,R) hi, [R,2R) hi (A) / lo (B), [2R,3R) lo (A) / hi (B)
__global__ __launch_bounds__(256) void split3_stack_kernel(const float* __restrict__ src, int R, int C, int ld,
                                                           bf16* __restrict__ dst, int Cs, int b_side) {
  const int cpr = Cs >> 3;
  const size_t total = (size_t)R * cpr;
  const bool vec = ((ld & 3) == 0) && ((((uintptr_t)src) & 15) == 0);
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int r = (int)(idx / cpr), c0 = (int)(idx - (size_t)r * cpr) * 8;
    const float* sp = src + (size_t)r * ld + c0;
    float v[8];
    if (vec && c0 + 8 <= C) {
      Vec<float, 4>::ld(sp, v);
      Vec<float, 4>::ld(sp + 4, v + 4);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (c0 + j < C) ? sp[j] : 0.f;
    }
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bf16 h, l;
      split_bf16(v[j], h, l);
      hi[j] = h;
      lo[j] = l;
    }
    bf16* dp = dst + (size_t)r * Cs + c0;
    const size_t plane = (size_t)R * Cs;
    *reinterpret_cast<bf16x8*>(dp) = hi;
    *reinterpret_cast<bf16x8*>(dp + plane) = b_side ? lo : hi;
    *reinterpret_cast<bf16x8*>(dp + 2 * plane) = b_side ? hi : lo;
  }
}

hipError_t split3_stack(const float* src, int R, int C, int ld, void* dst, int Cs, int b_side, hipStream_t s) {
  const size_t total = (size_t)R * (Cs / 8);
  if (total == 0) return hipSuccess;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(split3_stack_kernel, dim3(blocks), dim3(256), 0, s, src, R, C, ld, (bf16*)dst, Cs, b_side);
  return hipGetLastError();
}

// dst[c][r] = src[r][c]; dst is [Cp][Rp] zero padded.  32x32 LDS tile so both sides stay coalesced.
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, int R, int C, int ld,
                                                             T* __restrict__ dst, int Cp, int Rp, int ldd) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + i * 8, c = c0 + tx;
    tile[ty + i * 8][tx] = (r < R && c < C) ? src[(size_t)r * ld + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + i * 8, r = r0 + tx;
    if (c < Cp && r < Rp) Elem<T>::st(dst + (size_t)c * ldd + r, tile[tx][ty + i * 8]);
  }
}

hipError_t cast_transpose_pad(const float* src, int R, int C, int ld, void* dst, int dst_dtype, int Cp, int Rp,
                              int ldd, hipStream_t s) {
  if ((size_t)Rp * Cp == 0) return hipSuccess;
  dim3 grid((Cp + 31) / 32, (Rp + 31) / 32), block(256);
  if (dst_dtype == DT_BF16)
    hipLaunchKernelGGL(cast_transpose_kernel<bf16>, grid, block, 0, s, src, R, C, ld, (bf16*)dst, Cp, Rp, ldd);
  else
    hipLaunchKernelGGL(cast_transpose_kernel<float>, grid, block, 0, s, src, R, C, ld, (float*)dst, Cp, Rp, ldd);
  return hipGetLastError();
}

// ---- all weight shadows of one transformer block (or of the pooling head) in ONE launch -----------------------------------
// The per-step refresh of the compute-dtype weight copies (what autocast re-does every step in the reference,
// Siglip2sidafrozen.py:1375) used to be 13 launches per block (cast_pad + cast_transpose_pad per matrix, copy_f32 per bias):
// 350 launches of 5-7 us per so400m step, 153 per base-224 step.  Here a block's matrices are walked in 64x64 tiles of their
// PADDED destination (zeros outside the source), each tile read once as fp32 and written twice: row-major and, through an LDS
// transpose, column-major, both with >= 128-byte row segments; one extra workgroup copies / pads the bias vectors.
template <typename T> __device__ __forceinline__ T cj_cvt(float x);
template <> __device__ __forceinline__ float cj_cvt<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 cj_cvt<bf16>(float x) { return (bf16)x; }

template <typename T>
__global__ __launch_bounds__(256) void cast_job_kernel(CastJob job) {
  __shared__ float lds_raw[64 * 66];
  T* lt = reinterpret_cast<T*>(lds_raw);
  const int t = threadIdx.x;
  int tile = blockIdx.x;
  if (tile >= job.ntiles) {   // the vectors: dst[i] = i < n ? src[i] : 0 for i < np  (fp32)
    for (int v = 0; v < job.nvec; ++v)
      for (int i = t; i < job.vnp[v]; i += 256) job.vdst[v][i] = (i < job.vn[v]) ? job.vsrc[v][i] : 0.f;
    return;
  }
  int mi = 0;
#pragma unroll
  for (int k = 1; k < 6; ++k) mi += (k < job.nmat && tile >= job.m[k].tile0) ? 1 : 0;
  CastMat m = job.m[0];
#pragma unroll
  for (int k = 1; k < 6; ++k)
    if (mi == k) m = job.m[k];
  tile -= m.tile0;
  const int tr = tile / m.tiles_c, tc = tile - tr * m.tiles_c;
  const int r0 = tr * 64, c0 = tc * 64;
  const int tx = t & 15, ty = t >> 4;
  T* dst = reinterpret_cast<T*>(m.dst);
  const bool vsrc_ok = ((m.lds & 3) == 0) && ((((uintptr_t)m.src) & 15) == 0);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 16 * k, cc = c0 + tx * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < m.R && cc < m.C) {
      const float* sp = m.src + (size_t)r * m.lds + cc;
      if (vsrc_ok && cc + 3 < m.C) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(sp);
        v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
      } else {
        for (int j = 0; j < 4 && cc + j < m.C; ++j) v[j] = sp[j];
      }
    }
    T o4[4] = {cj_cvt<T>(v[0]), cj_cvt<T>(v[1]), cj_cvt<T>(v[2]), cj_cvt<T>(v[3])};
    if (dst && r < m.Rp && cc < m.Cp) {
      T* d = dst + (size_t)r * m.ldd + cc;
      if (cc + 3 < m.Cp && ((((uintptr_t)d) & (4 * sizeof(T) - 1)) == 0)) {
        if constexpr (sizeof(T) == 2) {
          u32x2 w;
          __builtin_memcpy(&w, o4, 8);
          *reinterpret_cast<u32x2*>(d) = w;
        } else {
          u32x4 w;
          __builtin_memcpy(&w, o4, 16);
          *reinterpret_cast<u32x4*>(d) = w;
        }
      } else {
        for (int j = 0; j < 4 && cc + j < m.Cp; ++j) d[j] = o4[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) lt[(ty + 16 * k) * 66 + tx * 4 + j] = o4[j];
  }
  if (!m.dst_t) return;
  __syncthreads();
  // transposed copy [Cp][Rp]: output row = source column c0 + oc, 16 consecutive elements = source rows r0 + 16*seg ..
  T* dt = reinterpret_cast<T*>(m.dst_t);
  const int oc = t >> 2, seg = t & 3;
  if (c0 + oc < m.Cp) {
    const int rb = r0 + seg * 16;
    T* drow = dt + (size_t)(c0 + oc) * m.ldt + rb;
    T vals[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) vals[j] = lt[(seg * 16 + j) * 66 + oc];
    if (rb + 15 < m.Rp && ((((uintptr_t)drow) & 15) == 0)) {
      constexpr int PER = 16 / sizeof(T);
#pragma unroll
      for (int q = 0; q < 16 / PER; ++q) {
        u32x4 w;
        __builtin_memcpy(&w, &vals[q * PER], 16);
        *reinterpret_cast<u32x4*>(drow + q * PER) = w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (rb + j < m.Rp) drow[j] = vals[j];
    }
  }
}

void cast_job_add(CastJob& job, const float* src, int R, int C, int lds_, void* dst, int Rp, int Cp, int ldd, void* dst_t,
                  int ldt) {
  CastMat& m = job.m[job.nmat++];
  m.src = src; m.dst = dst; m.dst_t = dst_t;
  m.R = R; m.C = C; m.lds = lds_; m.Rp = Rp; m.Cp = Cp; m.ldd = ldd; m.ldt = ldt;
  m.tiles_c = (Cp + 63) / 64;
  m.tile0 = job.ntiles;
  job.ntiles += ((Rp + 63) / 64) * m.tiles_c;
}
void cast_job_add_vec(CastJob& job, const float* src, int n, float* dst, int np) {
  job.vsrc[job.nvec] = src; job.vdst[job.nvec] = dst; job.vn[job.nvec] = n; job.vnp[job.nvec] = np;
  ++job.nvec;
}
hipError_t cast_job_run(const CastJob& job, int dst_dtype, hipStream_t s) {
  if (job.ntiles == 0 && job.nvec == 0) return hipSuccess;
  const dim3 grid((unsigned)(job.ntiles + (job.nvec ? 1 : 0)));
  if (dst_dtype == DT_BF16)
    hipLaunchKernelGGL(cast_job_kernel<bf16>, grid, dim3(256), 0, s, job);
  else
    hipLaunchKernelGGL(cast_job_kernel<float>, grid, dim3(256), 0, s, job);
  return hipGetLastError();
}

// ---- column sums (bias gradients): two deterministic stages ------------------------------------------
int colsum_chunks(int M) {
  int c = (M + 511) / 512;
  return c < 1 ? 1 : (c > 256 ? 256 : c);
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ in, int ld, int M, int N, int rows_per,
                                                     float* __restrict__ partial) {
  __shared__ float red[8][257];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col0 = blockIdx.x * 256 + cg * 8;
  const int r_begin = blockIdx.y * rows_per;
  const int r_end = (r_begin + rows_per < M) ? r_begin + rows_per : M;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col0 < N) {
    for (int r = r_begin + rl; r < r_end; r += 8) {
      float v[8];
      Vec<T, 8>::ld(in + (size_t)r * ld + col0, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][cg * 8 + j] = acc[j];
  __syncthreads();
  const int c = threadIdx.x;
  if (blockIdx.x * 256 + c < N) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r][c];
    partial[(size_t)blockIdx.y * N + blockIdx.x * 256 + c] = s;
  }
}

hipError_t colsum(const void* in, int dtype, int ld, int M, int N, int n_out, float* partial, float* out,
                  int accumulate, hipStream_t s) {
  if (N % 8 || ld % 8) return hipErrorInvalidValue;
  if (N == 0) return hipSuccess;
  const int chunks = colsum_chunks(M);
  const int rows_per = (M + chunks - 1) / chunks;
  dim3 grid((N + 255) / 256, chunks), block(256);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16>, grid, block, 0, s, (const bf16*)in, ld, M, N, rows_per, partial);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, s, (const float*)in, ld, M, N, rows_per, partial);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return reduce_partials(partial, chunks, N, out, n_out < N ? n_out : N, accumulate, s);
}

// out[j] (+)= sum_b in[b*n + j]   (d pos-table = sum over images of d embeddings)
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* __restrict__ in, int B, size_t n,
                                                        float* __restrict__ out, int accumulate) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += in[(size_t)b * n + j];
  out[j] = accumulate ? out[j] + s : s;
}
hipError_t batch_sum(const float* in, int B, size_t n, float* out, int accumulate, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, B, n, out, accumulate);
  return hipGetLastError();
}

// out[j] (+)= sum_i v[i] * W[i*cols + j]: a row vector times a row-major fp32 matrix.  Block (x, y) sums row chunk y of 64
// columns (4 row groups folded through LDS) into partial[y][j]; reduce_partials folds the chunks in fixed order.
// Used for the v_proj bias gradient, see encoder.hip.  scratch: >= VECMAT_CHUNKS * cols floats.
constexpr int VECMAT_CHUNKS = 16;
__global__ __launch_bounds__(256) void vecmat_f32_kernel(const float* __restrict__ v, const float* __restrict__ W, int rows,
                                                         int cols, float* __restrict__ partial) {
  __shared__ float part[4][64];
  const int jj = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jj;
  const int per = (rows + VECMAT_CHUNKS - 1) / VECMAT_CHUNKS;
  const int r0 = blockIdx.y * per, r1 = (r0 + per < rows) ? r0 + per : rows;
  float s = 0.f;
  if (j < cols)
    for (int i = r0 + rg; i < r1; i += 4) s = fmaf(v[i], W[(size_t)i * cols + j], s);
  part[rg][jj] = s;
  __syncthreads();
  if (rg == 0 && j < cols) partial[(size_t)blockIdx.y * cols + j] = (part[0][jj] + part[1][jj]) + (part[2][jj] + part[3][jj]);
}
hipError_t vecmat_f32(const float* v, const float* W, int rows, int cols, float* scratch, float* out, int accumulate,
                      hipStream_t s) {
  if (rows <= 0 || cols <= 0) return hipSuccess;
  hipLaunchKernelGGL(vecmat_f32_kernel, dim3((unsigned)((cols + 63) / 64), VECMAT_CHUNKS), dim3(256), 0, s, v, W, rows, cols,
                     scratch);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return reduce_partials(scratch, VECMAT_CHUNKS, cols, out, cols, accumulate, s);
}

// ---- position-table bicubic resize (F.interpolate bicubic, align_corners=False, A=-0.75;
//      TF:modeling_siglip.py:137-173).  Runs once per resolution; tiny. ---------------------------------
__device__ __forceinline__ void cubic_coeffs(float t, float* w) {
  const float A = -0.75f;
  float x = t + 1.0f;
  w[0] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
  x = t;
  w[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
  x = 1.0f - t;
  w[2] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
  x = 2.0f - t;
  w[3] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
}
__device__ __forceinline__ void cubic_taps(int o, int n_in, int n_out, int* idx, float* w) {
  const float scale = (float)n_in / (float)n_out;
  const float src = scale * ((float)o + 0.5f) - 0.5f;
  const float fl = floorf(src);
  cubic_coeffs(src - fl, w);
  const int i0 = (int)fl;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int i = i0 - 1 + k;
    idx[k] = i < 0 ? 0 : (i > n_in - 1 ? n_in - 1 : i);
  }
}

__global__ __launch_bounds__(256) void pos_resize_kernel(const float* __restrict__ table, int g0,
                                                         float* __restrict__ out, int gh, int gw, int D) {
  const size_t total = (size_t)gh * gw * D;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int d = (int)(idx % D);
  const int o = (int)(idx / D);
  const int oy = o / gw, ox = o - oy * gw;
  int iy[4], ix[4];
  float wy[4], wx[4];
  cubic_taps(oy, g0, gh, iy, wy);
  cubic_taps(ox, g0, gw, ix, wx);
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    float rowv = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) rowv += wx[b] * table[((size_t)iy[a] * g0 + ix[b]) * D + d];
    acc += wy[a] * rowv;
  }
  out[idx] = acc;
}
hipError_t pos_resize(const float* table, int g0, float* out, int gh, int gw, int D, hipStream_t s) {
  const size_t total = (size_t)gh * gw * D;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(pos_resize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, table, g0, out, gh, gw,
                     D);
  return hipGetLastError();
}

// transpose of the above: dtable[iy][ix][d] (+)= sum over outputs (oy, ox) of wy(oy -> iy) * wx(ox -> ix) * dout[oy][ox][d].
// GATHER form, one thread per table element, contributions added in a fixed (oy, ox) order: bitwise reproducible (the
// round-1 scatter used 16 fp32 atomics per output element).  The outputs that can touch table row iy lie in a short range
// around (iy + 0.5) * gh / g0; it is scanned conservatively (+-3 source rows plus the border clamp) and the exact test is
// "one of the output's four clamped taps equals iy".
__device__ __forceinline__ void tap_range(int i, int g0, int gout, int& lo, int& hi) {
  const float inv = (float)gout / (float)g0;
  lo = (int)floorf(((float)i - 3.0f + 0.5f) * inv - 0.5f) - 1;
  hi = (int)ceilf(((float)i + 3.0f + 0.5f) * inv - 0.5f) + 1;
  if (i == 0) lo = 0;                 // border entries also collect the clamped taps of every output near the edge
  if (i == g0 - 1) hi = gout - 1;
  if (lo < 0) lo = 0;
  if (hi > gout - 1) hi = gout - 1;
}
__global__ __launch_bounds__(256) void pos_resize_bwd_kernel(const float* __restrict__ dout, int gh, int gw,
                                                             float* __restrict__ dtable, int g0, int D) {
  const size_t total = (size_t)g0 * g0 * D;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int d = (int)(idx % D);
  const int e = (int)(idx / D);
  const int ty = e / g0, tx = e - ty * g0;
  int ylo, yhi, xlo, xhi;
  tap_range(ty, g0, gh, ylo, yhi);
  tap_range(tx, g0, gw, xlo, xhi);
  float acc = 0.f;
  for (int oy = ylo; oy <= yhi; ++oy) {
    int iy[4];
    float wy[4];
    cubic_taps(oy, g0, gh, iy, wy);
    float wyy = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) wyy += (iy[a] == ty) ? wy[a] : 0.f;
    if (wyy == 0.f) continue;
    float rowacc = 0.f;
    for (int ox = xlo; ox <= xhi; ++ox) {
      int ix[4];
      float wx[4];
      cubic_taps(ox, g0, gw, ix, wx);
      float wxx = 0.f;
#pragma unroll
      for (int b2 = 0; b2 < 4; ++b2) wxx += (ix[b2] == tx) ? wx[b2] : 0.f;
      if (wxx != 0.f) rowacc += wxx * dout[((size_t)oy * gw + ox) * D + d];
    }
    acc += wyy * rowacc;
  }
  dtable[idx] += acc;
}
hipError_t pos_resize_bwd(const float* dout, int gh, int gw, float* dtable, int g0, int D, hipStream_t s) {
  const size_t total = (size_t)g0 * g0 * D;
  if (total == 0 || gh * gw == 0) return hipSuccess;
  hipLaunchKernelGGL(pos_resize_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dout, gh, gw,
                     dtable, g0, D);
  return hipGetLastError();
}

// ---- small fp32 utilities ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       float* __restrict__ out, size_t n4, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 v = reinterpret_cast<const f32x4*>(a)[i];
    if (b) v += reinterpret_cast<const f32x4*>(b)[i];
    reinterpret_cast<f32x4*>(out)[i] = v;
  }
  if (blockIdx.x == 0) {
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) out[i] = a[i] + (b ? b[i] : 0.f);
  }
}
hipError_t add_f32(const float* a, const float* b, float* out, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, b, out, n4, n);
  return hipGetLastError();
}
hipError_t copy_f32(const float* src, float* dst, size_t n, hipStream_t s) { return add_f32(src, nullptr, dst, n, s); }

template <typename T>
__global__ __launch_bounds__(256) void cast_f32_kernel(const float* __restrict__ a, T* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    Elem<T>::st(out + i, a[i]);
}
hipError_t cast_f32(const float* src, void* dst, int dst_dtype, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (dst_dtype == DT_BF16)
    hipLaunchKernelGGL(cast_f32_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, s, src, (bf16*)dst, n);
  else
    hipLaunchKernelGGL(cast_f32_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, src, (float*)dst, n);
  return hipGetLastError();
}

// ---- attention-pool head: one probe query per image (TF:modeling_siglip.py:633-637, nn.MultiheadAttention
//      with q = probe, k = v = tokens).  One block per (b, h).  < 0.1 % of the FLOPs, HBM-bound: K and V of the
//      head are read once, in 16-byte (bf16) / 32-byte (fp32) chunks with consecutive lanes on consecutive chunks.
//      Thread t of the first RP*CPR (RP = 256 / CPR rows per pass, CPR = DP/8 chunks per row) owns chunk column
//      t % CPR for rows t / CPR, + RP, + 2 RP, ...: its slice of q / dO stays in registers. ---------------------
template <typename T>
__device__ __forceinline__ float block_reduce_256(float v, float* red, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if (lane_id() == 0) red[wave_id()] = v;
  __syncthreads();
  return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename T>
__global__ __launch_bounds__(256) void pool_attn_fwd_kernel(const float* __restrict__ q, const T* __restrict__ K,
                                                            const T* __restrict__ V, T* __restrict__ out,
                                                            float* __restrict__ probs, int H, int N, int dh, int DP) {
  extern __shared__ __attribute__((aligned(16))) float pa_smem[];  // [N*CPR] chunk partials, [N] probabilities, [8] scratch
  const int CPR = DP >> 3, RP = 256 / CPR;
  float* part = pa_smem;
  float* pr = part + (size_t)N * CPR;
  float* red = pr + N;
  const int bh = blockIdx.x, h = bh % H, b = bh / H;
  const T* Kb = K + (size_t)bh * N * DP;
  const T* Vb = V + (size_t)bh * N * DP;
  const float scale = rsqrtf((float)dh);
  const int t = threadIdx.x, c = t % CPR, r0 = t / CPR;
  const bool act = t < RP * CPR;
  float qv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) qv[j] = (c * 8 + j < dh) ? q[h * dh + c * 8 + j] : 0.f;
  if (act)
    for (int n = r0; n < N; n += RP) {
      float kv[8];
      Vec<T, 8>::ld(Kb + (size_t)n * DP + c * 8, kv);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s = fmaf(qv[j], kv[j], s);
      part[n * CPR + c] = s;
    }
  __syncthreads();
  float mx = -INFINITY;
  for (int n = t; n < N; n += 256) {
    float s = 0.f;
    for (int j = 0; j < CPR; ++j) s += part[n * CPR + j];
    s *= scale;
    pr[n] = s;
    mx = fmaxf(mx, s);
  }
  mx = block_reduce_256<T>(mx, red, true);
  float sum = 0.f;
  for (int n = t; n < N; n += 256) {
    const float e = __expf(pr[n] - mx);
    pr[n] = e;
    sum += e;
  }
  sum = block_reduce_256<T>(sum, red, false);
  const float inv = 1.0f / sum;
  for (int n = t; n < N; n += 256) {
    const float pv = pr[n] * inv;
    pr[n] = pv;
    probs[(size_t)bh * N + n] = pv;
  }
  __syncthreads();
  // out[d] = sum_n p_n V[n, d]: per-thread partial over its rows, then the RP threads of a chunk column through LDS
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (act)
    for (int n = r0; n < N; n += RP) {
      float vv[8];
      Vec<T, 8>::ld(Vb + (size_t)n * DP + c * 8, vv);
      const float pv = pr[n];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(pv, vv[j], acc[j]);
    }
  __syncthreads();   // part[] is free again
  if (act) {
#pragma unroll
    for (int j = 0; j < 8; ++j) part[(r0 * CPR + c) * 8 + j] = acc[j];
  }
  __syncthreads();
  for (int d = t; d < dh; d += 256) {
    float a = 0.f;
    for (int r = 0; r < RP; ++r) a += part[(r * CPR + (d >> 3)) * 8 + (d & 7)];
    Elem<T>::st(out + (size_t)b * H * dh + h * dh + d, a);
  }
}

hipError_t pool_attn_fwd(const float* q, const void* K, const void* V, int dtype, void* out, float* probs, int B,
                         int H, int N, int dh, int DP, hipStream_t s) {
  if (B * H == 0) return hipSuccess;
  if (DP % 8 || DP < 8 || DP > 2048) return hipErrorInvalidValue;
  const int CPR = DP / 8;
  const size_t need = (size_t)N * CPR > (size_t)256 * 8 ? (size_t)N * CPR : (size_t)256 * 8;
  const size_t smem = (need + N + 8) * sizeof(float);
  if (smem > 64 * 1024) return hipErrorInvalidValue;   // N * DP / 8 chunk partials must fit the default LDS window
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(pool_attn_fwd_kernel<bf16>, dim3(B * H), dim3(256), smem, s, q, (const bf16*)K, (const bf16*)V,
                       (bf16*)out, probs, H, N, dh, DP);
  else
    hipLaunchKernelGGL(pool_attn_fwd_kernel<float>, dim3(B * H), dim3(256), smem, s, q, (const float*)K,
                       (const float*)V, (float*)out, probs, H, N, dh, DP);
  return hipGetLastError();
}

// backward: dV[n,d] = p_n * do[d];  dp_n = do·V[n];  ds_n = p_n (dp_n - Σ p dp) * scale;
//           dK[n,d] = ds_n q[d];   dq[d] (per image) = Σ_n ds_n K[n,d]
// dkv is token-major [B*N][2*H*dh] (k block then v block) — the A operand of the kv-projection backward GEMMs.
template <typename T>
__global__ __launch_bounds__(256) void pool_attn_bwd_kernel(const float* __restrict__ q, const T* __restrict__ K,
                                                            const T* __restrict__ V, const float* __restrict__ probs,
                                                            const float* __restrict__ dout, T* __restrict__ dkv,
                                                            float* __restrict__ dq_partial, int H, int N, int dh,
                                                            int DP) {
  extern __shared__ __attribute__((aligned(16))) float pa_smem[];  // [N*CPR] chunk partials, [N] ds, [N] p, [8] scratch
  const int CPR = DP >> 3, RP = 256 / CPR;
  float* part = pa_smem;
  float* ds = part + (size_t)N * CPR;
  float* pr = ds + N;
  float* red = pr + N;
  const int bh = blockIdx.x, h = bh % H, b = bh / H;
  const int D = H * dh;
  const T* Kb = K + (size_t)bh * N * DP;
  const T* Vb = V + (size_t)bh * N * DP;
  const float scale = rsqrtf((float)dh);
  const int t = threadIdx.x, c = t % CPR, r0 = t / CPR;
  const bool act = t < RP * CPR;
  const bool real = c * 8 < dh;     // dh % 8 == 0: a chunk is entirely data or entirely pad
  float qv[8], dov[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    qv[j] = real ? q[h * dh + c * 8 + j] : 0.f;
    dov[j] = real ? dout[(size_t)b * D + h * dh + c * 8 + j] : 0.f;
  }
  if (act)
    for (int n = r0; n < N; n += RP) {
      float vv[8];
      Vec<T, 8>::ld(Vb + (size_t)n * DP + c * 8, vv);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s = fmaf(dov[j], vv[j], s);
      part[n * CPR + c] = s;
    }
  __syncthreads();
  float psum = 0.f;
  for (int n = t; n < N; n += 256) {
    float dp = 0.f;
    for (int j = 0; j < CPR; ++j) dp += part[n * CPR + j];
    const float pv = probs[(size_t)bh * N + n];
    ds[n] = dp;
    pr[n] = pv;
    psum += pv * dp;
  }
  const float dot = block_reduce_256<T>(psum, red, false);
  for (int n = t; n < N; n += 256) ds[n] = pr[n] * (ds[n] - dot) * scale;
  __syncthreads();
  // dK / dV rows (16-byte chunks, 9 consecutive lanes per 144-byte row segment) and the per-thread part of dq
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (act)
    for (int n = r0; n < N; n += RP) {
      float kv[8];
      Vec<T, 8>::ld(Kb + (size_t)n * DP + c * 8, kv);
      const float g = ds[n], pv = pr[n];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(g, kv[j], acc[j]);
      if (real) {
        float ok[8], ov[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          ok[j] = g * qv[j];
          ov[j] = pv * dov[j];
        }
        T* row = dkv + ((size_t)b * N + n) * (2 * (size_t)D) + h * dh + c * 8;
        Vec<T, 8>::st(row, ok);
        Vec<T, 8>::st(row + D, ov);
      }
    }
  __syncthreads();   // part[] is free again
  if (act) {
#pragma unroll
    for (int j = 0; j < 8; ++j) part[(r0 * CPR + c) * 8 + j] = acc[j];
  }
  __syncthreads();
  for (int d = t; d < dh; d += 256) {
    float a = 0.f;
    for (int r = 0; r < RP; ++r) a += part[(r * CPR + (d >> 3)) * 8 + (d & 7)];
    dq_partial[(size_t)b * D + h * dh + d] = a;
  }
}

hipError_t pool_attn_bwd(const float* q, const void* K, const void* V, int dtype, const float* probs,
                         const float* dout, void* dkv, float* dq_partial, int B, int H, int N, int dh, int DP,
                         hipStream_t s) {
  if (B * H == 0) return hipSuccess;
  if (DP % 8 || dh % 8 || DP < 8 || DP > 2048) return hipErrorInvalidValue;
  const int CPR = DP / 8;
  const size_t need = (size_t)N * CPR > (size_t)256 * 8 ? (size_t)N * CPR : (size_t)256 * 8;
  const size_t smem = (need + 2 * (size_t)N + 8) * sizeof(float);
  if (smem > 64 * 1024) return hipErrorInvalidValue;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(pool_attn_bwd_kernel<bf16>, dim3(B * H), dim3(256), smem, s, q, (const bf16*)K, (const bf16*)V,
                       probs, dout, (bf16*)dkv, dq_partial, H, N, dh, DP);
  else
    hipLaunchKernelGGL(pool_attn_bwd_kernel<float>, dim3(B * H), dim3(256), smem, s, q, (const float*)K,
                       (const float*)V, probs, dout, (float*)dkv, dq_partial, H, N, dh, DP);
  return hipGetLastError();
}

// ---- split-K slabs -> result, fixed order -----------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_splits_kernel(const float* __restrict__ ws, int splits, size_t stride,
                                                            int N1, int N2, float* __restrict__ out, int ldo,
                                                            int accumulate) {
  const size_t total4 = ((size_t)N1 * N2) >> 2;  // N2 % 4 == 0 (checked by the host)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
    const size_t e = i << 2;
    const int r = (int)(e / N2), c = (int)(e - (size_t)r * N2);
    f32x4 a = *reinterpret_cast<const f32x4*>(ws + e);
    for (int s = 1; s < splits; ++s) a += *reinterpret_cast<const f32x4*>(ws + (size_t)s * stride + e);
    float* o = out + (size_t)r * ldo + c;
    if (accumulate) a += *reinterpret_cast<const f32x4*>(o);
    *reinterpret_cast<f32x4*>(o) = a;
  }
}

hipError_t reduce_splits(const float* ws, int splits, size_t stride, int N1, int N2, float* out, int ldo, int accumulate,
                         hipStream_t s) {
  if ((N2 & 3) || (ldo & 3) || (((uintptr_t)out) & 15) || (((uintptr_t)ws) & 15) || (stride & 3))
    return hipErrorInvalidValue;
  const size_t total4 = ((size_t)N1 * N2) >> 2;
  if (total4 == 0) return hipSuccess;
  const int blocks = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(reduce_splits_kernel, dim3(blocks), dim3(256), 0, s, ws, splits, stride, N1, N2, out, ldo,
                     accumulate);
  return hipGetLastError();
}

}  // namespace sgl

// Internal launch API shared by the kernel translation units and the C-ABI host code (encoder.cpp).
// Every launcher enqueues on the stream it is given, allocates nothing and never synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace sgl {

enum DType : int { DT_F32 = 0, DT_BF16 = 1, DT_F32_MFMA = 2 /* attention only: fp32 operands on v_mfma_f32_32x32x2_f32 */ };
static inline size_t dtype_size(int dt) { return dt == DT_BF16 ? 2 : 4; }

// ---- GEMM epilogues (shared by the MFMA kernels and the strict-fp32 generic kernel) ----------------
enum Epi : int {
  EPI_STORE = 0,      // out[T]   = alpha*acc (+bias)
  EPI_BIAS_GELU = 1,  // out[T]   = u = acc+bias ; out2[T] = gelu_tanh(u)
  EPI_RES_F32 = 2,    // out[f32] = res + acc + bias
  EPI_QKV = 3,        // head-major scatter of acc+bias: out[which][B][H][tokens][head_dim_pad]  (T)
  EPI_GELU_BWD = 4,   // out[T]   = acc * gelu_tanh'(aux[row,col])
  EPI_POS_F32 = 5,    // out[f32] = acc + bias + pos[row % pos_rows, col]
  EPI_F32 = 6,        // out[f32] = alpha*acc (+bias) (+out if accumulate) ; atomicAdd when atomic != 0
};

struct EpiParams {
  void* out = nullptr;   int ldo = 0;
  void* out2 = nullptr;  int ldo2 = 0;
  const float* bias = nullptr;
  const float* res = nullptr;  int ldr = 0;
  const void* aux = nullptr;   int ldaux = 0;
  const float* pos = nullptr;  int pos_rows = 1;
  int tokens = 1, heads = 1, head_dim = 8, head_dim_pad = 8, batch = 1;
  float alpha = 1.0f;
  int accumulate = 0;
  int atomic = 0;
  // EPI_GELU_BWD on the MFMA kernels: when non-null, column sums of the fp32 output tile are atomically added here
  // (= the bias gradient of the GEMM whose output gradient is being produced); must be zeroed/initialised by the host
  float* colsum = nullptr;
  // colsum_ld == 0: atomicAdd into colsum[col].  colsum_ld > 0 (deterministic): the workgroup whose tile starts at row m0
  // STORES its column sums at colsum[(m0 / 128) * colsum_ld + col]; the host zeroes the [ceil(M/128)][colsum_ld] buffer
  // first and folds the rows in a fixed order afterwards (reduce_partials)
  int colsum_ld = 0;
  // EPI_BIAS_GELU / EPI_GELU_BWD pair: when set, the forward stores gelu'(u) in `out` instead of the pre-activation u and
  // the backward multiplies by the saved value instead of re-evaluating the derivative (two transcendentals per element
  // less in the backward epilogue, six plain VALU operations more in the forward one); both launches must agree
  int gelu_grad_form = 0;
  // split-K TN GEMM, deterministic form: split s stores its partial tile at out + s*split_stride (plain stores, ldo = N2)
  // instead of atomically adding into the result; reduce_splits() then sums the slabs in a fixed order
  size_t split_stride = 0;
};

// ---- layernorm.hip -----------------------------------------------------------------------------------
hipError_t layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_dtype, int ldy,
                         float* mean, float* rstd, int M, int D, float eps, hipStream_t s);
int layernorm_bwd_blocks(int M);
// partial: [nblk][3*D] floats per block: dgamma, dbeta, column sums of the dx this call wrote
hipError_t layernorm_bwd(const void* dy, int dy_dtype, int lddy, const float* x, const float* mean,
                         const float* rstd, const float* gamma, const float* dres, float* dx, void* dx_lp,
                         int lp_dtype, float* partial, int nblk, int M, int D, hipStream_t s);
// out[j] (+)= sum_b partial[b*stride + j], j < n
hipError_t reduce_partials(const float* partial, int nblk, int stride, float* out, int n, int accumulate,
                           hipStream_t s);
// o_k[j] (+)= sum_b partial[b*stride + k*n + j] for k = 0,1,2 in one launch (null outputs skipped)
hipError_t reduce_partials3(const float* partial, int nblk, int stride, float* o0, float* o1, float* o2, int n, int a0,
                            int a1, int a2, hipStream_t s);

// ---- elementwise.hip ---------------------------------------------------------------------------------
hipError_t im2col(const float* pix, int channels_last, void* out, int out_dtype, int B, int H, int W, int P,
                  int Kp, hipStream_t s);
// dst[r][c] (row stride ldd) = cast(src[r][c]) for r < Rp, c < Cp, zero outside src's R x C
hipError_t cast_pad(const float* src, int R, int C, int lds_, void* dst, int dst_dtype, int Rp, int Cp, int ldd,
                    hipStream_t s);
// dst[c][r] (row stride ldd) = cast(src[r][c]) for c < Cp, r < Rp, zero outside
hipError_t cast_transpose_pad(const float* src, int R, int C, int lds_, void* dst, int dst_dtype, int Cp, int Rp,
                              int ldd, hipStream_t s);
// One launch for all weight shadows of a block: up to 6 matrices (row-major copy [Rp][Cp] with ld ldd and, optionally, the
// transposed copy [Cp][Rp] with ld ldt, both zero padded outside the R x C source) and up to 4 fp32 vectors (copied, zero padded)
struct CastMat {
  const float* src;
  void* dst;
  void* dst_t;
  int R, C, lds, Rp, Cp, ldd, ldt, tiles_c, tile0;
};
struct CastJob {
  CastMat m[6];
  const float* vsrc[4];
  float* vdst[4];
  int vn[4], vnp[4];
  int nmat = 0, nvec = 0, ntiles = 0;
};
void cast_job_add(CastJob& job, const float* src, int R, int C, int lds_, void* dst, int Rp, int Cp, int ldd, void* dst_t,
                  int ldt);
void cast_job_add_vec(CastJob& job, const float* src, int n, float* dst, int np);
hipError_t cast_job_run(const CastJob& job, int dst_dtype, hipStream_t s);
// bf16x3 operand splits (elementwise.hip): Cs = round_up(C, 8); dst holds 3 * R * Cs bf16
hipError_t split3_rows(const float* src, int R, int C, int ld, void* dst, int Cs, int b_side, hipStream_t s);
hipError_t split3_stack(const float* src, int R, int C, int ld, void* dst, int Cs, int b_side, hipStream_t s);
int colsum_chunks(int M);
// out[j] (+)= sum_r in[r][j] for j < n_out (n_out <= N; N, ld multiples of 8; columns up to N are read)
hipError_t colsum(const void* in, int dtype, int ld, int M, int N, int n_out, float* partial /*[chunks][N]*/,
                  float* out, int accumulate, hipStream_t s);
hipError_t batch_sum(const float* in, int B, size_t n, float* out, int accumulate, hipStream_t s);
// out[j] (+)= sum_i v[i] * W[i*cols + j]; scratch: >= 16 * cols floats
hipError_t vecmat_f32(const float* v, const float* W, int rows, int cols, float* scratch, float* out, int accumulate,
                      hipStream_t s);
hipError_t pos_resize(const float* table, int g0, float* out, int gh, int gw, int D, hipStream_t s);
hipError_t pos_resize_bwd(const float* dout, int gh, int gw, float* dtable, int g0, int D, hipStream_t s);
hipError_t copy_f32(const float* src, float* dst, size_t n, hipStream_t s);
hipError_t cast_f32(const float* src, void* dst, int dst_dtype, size_t n, hipStream_t s);
hipError_t add_f32(const float* a, const float* b, float* out, size_t n, hipStream_t s);  // b may be null
// attention-pool (1 query per image): q [H*dh] fp32, K/V head-major [B][H][N][DP]
hipError_t pool_attn_fwd(const float* q, const void* K, const void* V, int dtype, void* out /*[B][H*dh]*/,
                         float* probs /*[B][H][N]*/, int B, int H, int N, int dh, int DP, hipStream_t s);
hipError_t pool_attn_bwd(const float* q, const void* K, const void* V, int dtype, const float* probs,
                         const float* dout /*[B][H*dh] fp32*/, void* dkv /*[B*N][2*H*dh] T*/,
                         float* dq_partial /*[B][H*dh]*/, int B, int H, int N, int dh, int DP, hipStream_t s);

// ---- gemm_f32.hip: strict-mode generic strided GEMM  C[m,n] = sum_k A(m,k)*B(n,k) ---------------------
hipError_t gemm_f32_generic(const float* A, long sam, long sak, const float* B, long sbn, long sbk, int M, int N,
                            int K, int epi, int out_dtype, const EpiParams& p, hipStream_t s);

// ---- gemm_bf16.hip: MFMA kernels ----------------------------------------------------------------------
// NT: C[M,N] = A[M,K] * B[N,K]^T ; A,B bf16 K-contiguous, lda/ldb multiples of 8, K multiple of 8
hipError_t gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int epi,
                        int out_dtype, const EpiParams& p, hipStream_t s);
// TN: C[N1,N2] (+)= sum_m A[m,n1] * B[m,n2] ; fp32 output via EPI_F32 (accumulate / atomic split-K)
// out[r*ldo + c] (+)= sum_s ws[s*stride + r*N2 + c]   (fixed summation order)
hipError_t reduce_splits(const float* ws, int splits, size_t stride, int N1, int N2, float* out, int ldo, int accumulate,
                         hipStream_t s);
// split_ws (optional device scratch): when given and large enough, the splits of the 256x256-tile kernel write private
// slabs and reduce_splits() sums them (bitwise reproducible); otherwise they add into `out` with fp32 atomics.
hipError_t gemm_tn_bf16(const void* A, int lda, const void* B, int ldb, int Mred, int N1, int N2, int splits,
                        const EpiParams& p, hipStream_t s, float* split_ws = nullptr, size_t split_ws_bytes = 0);

// ---- attention.hip -------------------------------------------------------------------------------------
// q,k,v: ld_qkv > 0: token-major [B*N][ld_qkv] column blocks (head h of row r at r*ld_qkv + h*dh; the QKV projection's
// plain output, q/k/v = its three D-wide column blocks); ld_qkv == 0: head-major [B][H][N][DP] with zero pad columns.
// out: token-major [B*N][H*dh]; lse: [B][H][N] (natural log units)
hipError_t attn_fwd(const void* q, const void* k, const void* v, int dtype, void* out, float* lse, int B, int H,
                    int N, int dh, int DP, int ld_qkv, hipStream_t s);
// dout: token-major [B*N][H*dh] (T); dqkv: token-major [B*N][3*H*dh] (T); delta: scratch of 2*B*H*N floats
hipError_t attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                    const float* lse, int dtype, void* dqkv, float* delta, float* reserved, int B, int H, int N,
                    int dh, int DP, int ld_qkv, hipStream_t s);
size_t attn_bwd_scratch_bytes(int dtype, int B, int H, int N, int dh, int DP);

}  // namespace sgl

/* siglip_hip.h — C ABI of libsiglip_hip.so: the MI355X (gfx950) SigLIP-2 vision-encoder hot path.
 *
 * The reference has no FFI of its own: its "plugin API" for this path is two Python call surfaces on an
 * nn.Module obtained from third-party libraries:
 *   surface H  self.encoder(pixel_values=..., output_hidden_states=True, interpolate_pos_encoding=True)
 *              -> .pooler_output / .last_hidden_state / .hidden_states        Siglip2sidafrozen.py:753,787-793
 *   surface O  backbone.encode_image(x) -> (B, D)                             cifake_binary_classifier.py:721,
 *                                                                             hidf_video_classifier.py:307
 * The entry points below are what a binding for that path binds instead of the HuggingFace / open_clip ViT:
 * plain pointers and sizes, caller-owned memory, the caller's HIP stream.  INTEGRATION.md shows the ctypes
 * stub.  Rules common to every call:
 *   - returns 0 (SGL_OK) or a negative sgl_status; never throws, aborts, prints, allocates device memory or
 *     synchronises the device; sgl_last_hip_error(ctx) holds the hipError_t behind SGL_ERR_HIP;
 *   - all pointers are device pointers on the current device unless stated; work is enqueued on `stream`;
 *   - re-entrant per ctx as long as the calls on one ctx are stream-ordered; no thread-local state (PyTorch runs
 *     backward on an autograd worker thread);
 *   - fp32 master parameters use the HuggingFace layouts (Linear W[out,in] row-major, patch conv W[D,3,p,p],
 *     nn.MultiheadAttention in_proj_weight [3D,D] in q,k,v order).
 */
#ifndef SIGLIP_HIP_H
#define SIGLIP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sgl_ctx sgl_ctx;
typedef void* sgl_stream; /* hipStream_t */

enum { SGL_DTYPE_F32 = 0, SGL_DTYPE_BF16 = 1, SGL_DTYPE_BF16X3 = 2 };

typedef enum {
  SGL_OK = 0,
  SGL_ERR_BAD_SHAPE = -1,   /* image not divisible into patches, non-square grid, dims not supported */
  SGL_ERR_UNSUPPORTED = -2, /* dtype / config outside what the kernels implement */
  SGL_ERR_WORKSPACE = -3,   /* saved / workspace / shadow buffer smaller than sgl_query_sizes reports */
  SGL_ERR_HIP = -4,         /* a HIP call failed: see sgl_last_hip_error */
  SGL_ERR_NULL = -5         /* a required pointer is NULL */
} sgl_status;

/* Immutable model description (HF SiglipVisionConfig fields; TF:models/siglip/configuration_siglip.py:90-99). */
typedef struct {
  int hidden_size;        /* D */
  int intermediate_size;  /* I */
  int num_layers;         /* L */
  int num_heads;          /* H, head_dim = D / H must be a multiple of 8 and <= 96 */
  int patch_size;         /* p */
  int native_grid;        /* image_size / p : side of the stored position table */
  float layer_norm_eps;   /* 1e-6 */
  int compute_dtype;      /* SGL_DTYPE_BF16: bf16 MFMA operands, fp32 accumulate / residual stream / statistics;
                             SGL_DTYPE_F32 : strict fp32 everywhere (parity mode; plain fp32 FMAs, no matrix cores);
                             SGL_DTYPE_BF16X3: strict mode ON the matrix cores: activations, weights and buffers exactly as
                                in SGL_DTYPE_F32, but every GEMM runs as one bf16 MFMA GEMM over split operands
                                (x = hi + lo; hi*hi + hi*lo + lo*hi, fp32 accumulate: ~2^-17 relative per product) and
                                attention as fp32 MFMA; meets "logits within 1e-3" at a fraction of SGL_DTYPE_F32's cost */
  int use_head;           /* attention-pool head present (vision_use_head) */
} sgl_config;

/* fp32 master parameters of one encoder block (TF:modeling_siglip.py:268-271,315-316,329-331). */
typedef struct {
  const float *ln1_w, *ln1_b;
  const float *q_w, *q_b, *k_w, *k_b, *v_w, *v_b, *o_w, *o_b;
  const float *ln2_w, *ln2_b;
  const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} sgl_layer_weights;

typedef struct {
  const float *patch_w, *patch_b; /* [D, 3*p*p], [D] */
  const float* pos;               /* [native_grid^2, D] */
  const sgl_layer_weights* layers; /* HOST array of num_layers entries (device pointers inside) */
  const float *post_ln_w, *post_ln_b;
  /* attention-pool head (TF:modeling_siglip.py:622-643); ignored when use_head == 0 */
  const float *probe;                     /* [D] */
  const float *in_proj_w, *in_proj_b;     /* [3D, D], [3D] */
  const float *out_proj_w, *out_proj_b;   /* [D, D], [D] */
  const float *head_ln_w, *head_ln_b;
  const float *head_fc1_w, *head_fc1_b, *head_fc2_w, *head_fc2_b;
} sgl_weights;

/* Gradient destinations, same layouts as sgl_weights.  A NULL pointer means "frozen: do not compute".
 * accumulate != 0 adds into the buffers (gradient accumulation), otherwise they are overwritten. */
typedef struct {
  float *ln1_w, *ln1_b;
  float *q_w, *q_b, *k_w, *k_b, *v_w, *v_b, *o_w, *o_b;
  float *ln2_w, *ln2_b;
  float *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} sgl_layer_grads;

typedef struct {
  float *patch_w, *patch_b, *pos;
  const sgl_layer_grads* layers; /* HOST array of num_layers entries */
  float *post_ln_w, *post_ln_b;
  float *probe, *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b, *head_ln_w, *head_ln_b;
  float *head_fc1_w, *head_fc1_b, *head_fc2_w, *head_fc2_b;
  int accumulate;
} sgl_grads;

/* ---- lifetime ------------------------------------------------------------------------------------- */
sgl_ctx* sgl_create(const sgl_config* cfg); /* NULL if the config is unsupported */
void sgl_destroy(sgl_ctx* ctx);
int sgl_last_hip_error(const sgl_ctx* ctx);
const char* sgl_status_string(int status);
int sgl_abi_version(void);

/* ---- sizes (bytes) for caller-allocated buffers ----------------------------------------------------- */
/* shadow: compute-dtype copies of the weight matrices (padded, plus pre-transposed forms for dX GEMMs);
 * saved:  activations kept from forward for backward (0 when train == 0);
 * ws:     scratch; must stay untouched between sgl_backward_begin and the last sgl_backward_* call. */
int sgl_query_sizes(const sgl_ctx* ctx, int B, int H, int W, int train, size_t* shadow_bytes, size_t* saved_bytes,
                    size_t* ws_bytes);

/* Refresh the shadow arena from the fp32 masters (call after every optimizer step / load_state_dict).
 * Replaces the per-step autocast weight casts of the reference (Siglip2sidafrozen.py:1375). */
int sgl_prepare_weights(sgl_ctx* ctx, const sgl_weights* w, void* shadow, size_t shadow_bytes, sgl_stream stream);
/* Same, restricted to what changed since the last call: layer_dirty[l] != 0 re-casts block l (NULL = all blocks),
 * globals_dirty != 0 re-casts the patch-embedding and pooling-head matrices.  For frozen-prefix fine-tuning. */
int sgl_prepare_weights_dirty(sgl_ctx* ctx, const sgl_weights* w, void* shadow, size_t shadow_bytes,
                              const unsigned char* layer_dirty, int globals_dirty, sgl_stream stream);

/* ---- forward ---------------------------------------------------------------------------------------- */
/* pixels: fp32 (B,3,H,W) NCHW, or NHWC storage when channels_last == 1 (reference .to(channels_last),
 *         Siglip2sidafrozen.py:1191,1365).  Grid = (H / p, W / p) as in a "valid" conv (384 / 14 = 27).
 *         channels_last == 2: `pixels` is instead the ready patch-major operand [B*grid^2][round_up(3p^2,64)] in the
 *         compute dtype (sgl_op_preprocess, patch_major): the gather pass is skipped (H, W still give the geometry).
 * hidden_states: fp32 [hs_slots][B*N][D]; slot l holds hidden_states[l] of the HF output (0 = embeddings,
 *         L = last block output before post_layernorm).  hs_slots = L+1 keeps all of them (required when
 *         saved != NULL); hs_slots = 2 ping-pongs (inference without taps).
 * last_hidden: fp32 [B*N][D] (post_layernorm output).   pooled: fp32 [B][D] or NULL.
 * interpolate_pos != 0: bicubic-resize the position table when the grid differs from native
 *         (TF:modeling_siglip.py:137-173); with 0 the grid must equal the native grid. */
int sgl_forward(sgl_ctx* ctx, const sgl_weights* w, const void* shadow, const float* pixels, int channels_last, int B,
                int H, int W, int interpolate_pos, float* hidden_states, int hs_slots, float* last_hidden,
                float* pooled, void* saved, size_t saved_bytes, void* ws, size_t ws_bytes, sgl_stream stream);
/* sgl_forward with a frozen prefix declared: blocks < first_trainable_block will not be differentiated
 * (sgl_backward_layer is never called for them), so their GELU pre-activations are not saved. */
int sgl_forward_ex(sgl_ctx* ctx, const sgl_weights* w, const void* shadow, const float* pixels, int channels_last, int B,
                   int H, int W, int interpolate_pos, float* hidden_states, int hs_slots, float* last_hidden,
                   float* pooled, void* saved, size_t saved_bytes, void* ws, size_t ws_bytes, int first_trainable_block,
                   sgl_stream stream);

/* sgl_forward_ex with one pointer per hidden-state slot (HOST array of L+1 device pointers, each [B*N][D] fp32) instead
 * of one [slots][B*N][D] block: the PyTorch custom op (torch.ops.siglip_hip.encoder_fwd) hands the requested taps out as
 * tensors of their own and keeps the other slots in a private buffer, so no output aliases another.  Inference callers
 * may point several entries at the same buffer (ping-pong) as long as slots l and l+1 differ; with saved != NULL all
 * L+1 pointers must be distinct. */
int sgl_forward_slots(sgl_ctx* ctx, const sgl_weights* w, const void* shadow, const float* pixels, int channels_last,
                      int B, int H, int W, int interpolate_pos, float* const* hs_slots, float* last_hidden,
                      float* pooled, void* saved, size_t saved_bytes, void* ws, size_t ws_bytes,
                      int first_trainable_block, sgl_stream stream);

/* ---- backward (stepwise so that a data-parallel caller can all-reduce each block's gradients while the
 *      next block's backward runs; sgl_backward is the plain loop over the three steps) ------------------ */
/* d_last_hidden [B*N][D], d_pooled [B][D], d_tap_last [B*N][D] (gradient w.r.t. hidden_states[L]); any may be
 * NULL.  Computes head + post_layernorm gradients and leaves d hidden_states[L] in the workspace. */
int sgl_backward_begin(sgl_ctx* ctx, const sgl_weights* w, const void* shadow, const sgl_grads* g, int B, int H, int W,
                       const float* hidden_states, const float* d_last_hidden, const float* d_pooled,
                       const float* d_tap_last, const void* saved, size_t saved_bytes, void* ws, size_t ws_bytes,
                       sgl_stream stream);
/* Block `layer`: parameter gradients of that block, then workspace gradient := d hidden_states[layer]
 * (+ d_tap, the external gradient w.r.t. hidden_states[layer], may be NULL).  need_dx == 0 skips the input
 * gradient (first trainable block of a frozen prefix, Siglip2sidafrozen.py:762-768). */
int sgl_backward_layer(sgl_ctx* ctx, const sgl_weights* w, const void* shadow, const sgl_grads* g, int layer, int B,
                       int H, int W, const float* hidden_states, const float* d_tap, int need_dx, const void* saved,
                       size_t saved_bytes, void* ws, size_t ws_bytes, sgl_stream stream);
/* The same two steps taking just the hidden state they read (hidden_states[L] / hidden_states[layer]) instead of the
 * base of a contiguous [L+1][B*N][D] block: for callers that keep the slots in separate buffers (sgl_forward_slots). */
int sgl_backward_begin_p(sgl_ctx* ctx, const sgl_weights* w, const void* shadow, const sgl_grads* g, int B, int H, int W,
                         const float* hs_last, const float* d_last_hidden, const float* d_pooled,
                         const float* d_tap_last, const void* saved, size_t saved_bytes, void* ws, size_t ws_bytes,
                         sgl_stream stream);
int sgl_backward_layer_p(sgl_ctx* ctx, const sgl_weights* w, const void* shadow, const sgl_grads* g, int layer, int B,
                         int H, int W, const float* hs_in, const float* d_tap, int need_dx, const void* saved,
                         size_t saved_bytes, void* ws, size_t ws_bytes, sgl_stream stream);
/* Patch-embedding / position-table gradients from the workspace gradient (d hidden_states[0]). */
int sgl_backward_embed(sgl_ctx* ctx, const sgl_weights* w, const sgl_grads* g, int B, int H, int W, int interpolate_pos,
                       const void* saved, size_t saved_bytes, void* ws, size_t ws_bytes, sgl_stream stream);
/* d_taps: HOST array of L+1 device pointers (NULL entries allowed) or NULL.  Stops above first_trainable_block. */
int sgl_backward(sgl_ctx* ctx, const sgl_weights* w, const void* shadow, const sgl_grads* g, int B, int H, int W,
                 int interpolate_pos, const float* hidden_states, const float* const* d_taps,
                 const float* d_last_hidden, const float* d_pooled, int first_trainable_block, int train_embeddings,
                 const void* saved, size_t saved_bytes, void* ws, size_t ws_bytes, sgl_stream stream);

/* ---- single-kernel entry points (unit parity tests, micro-benchmarks, roofline measurement) ----------- */
int sgl_op_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_dtype, float* mean,
                         float* rstd, int M, int D, float eps, sgl_stream stream);
int sgl_op_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd,
                         const float* gamma, const float* dres, float* dx, void* dx_lp, int lp_dtype, float* dgamma,
                         float* dbeta, float* scratch, size_t scratch_bytes, int M, int D, sgl_stream stream);
/* epilogue selectors for sgl_op_gemm_nt */
enum { SGL_EPI_STORE = 0, SGL_EPI_BIAS_GELU = 1, SGL_EPI_RES_F32 = 2, SGL_EPI_QKV = 3, SGL_EPI_GELU_BWD = 4,
       SGL_EPI_POS_F32 = 5, SGL_EPI_F32 = 6 };
/* C[M,N] = A[M,K] * B[N,K]^T with a fused epilogue; dtype is the operand dtype (bf16 -> MFMA kernel). */
int sgl_op_gemm_nt(int dtype, const void* A, int lda, const void* B, int ldb, int M, int N, int K, int epi, void* out,
                   int ldo, void* out2, int ldo2, const float* bias, const float* res, int ldr, const void* aux,
                   int ldaux, const float* pos, int pos_rows, int tokens, int heads, int head_dim, int head_dim_pad,
                   int batch, sgl_stream stream);
/* C[N1,N2] (+)= sum_m A[m,N1] * B[m,N2]  (fp32 output). */
int sgl_op_gemm_tn(int dtype, const void* A, int lda, const void* B, int ldb, int Mred, int N1, int N2, int splits,
                   float* out, int ldo, int accumulate, sgl_stream stream);
/* Same with device scratch for the split-K partial tiles: with scratch >= splits_used * N1 * N2 * 4 bytes (64 MiB always
 * suffices for the 256x256-tile kernel) the splits are summed in a fixed order -> bitwise reproducible, no fp32 atomics. */
int sgl_op_gemm_tn_ws(int dtype, const void* A, int lda, const void* B, int ldb, int Mred, int N1, int N2, int splits,
                      float* out, int ldo, int accumulate, float* scratch, size_t scratch_bytes, sgl_stream stream);
/* Scaled-dot-product attention of one block (TF:modeling_siglip.py:227-247,288-301), all heads and images in one launch.
 * dtype: SGL_DTYPE_BF16 (bf16 operands, bf16 MFMA, fp32 softmax), SGL_DTYPE_F32 (fp32 operands, plain FMAs: the reference
 *   kernels) or SGL_DTYPE_BF16X3 (fp32 operands on v_mfma_f32_32x32x2_f32: what the strict MFMA mode uses).
 * ld_qkv > 0 (what the encoder uses since ABI 3): q, k, v point at the three column blocks of the QKV projection's
 *   token-major output [B*N][ld_qkv]; head h of token row r is the head_dim elements at r*ld_qkv + h*head_dim (16-byte
 *   aligned: head_dim % 8 == 0, ld_qkv % 8 == 0, pointers 16-byte aligned).  Nothing is padded in memory.
 * ld_qkv == 0: legacy head-major [B][H][N][head_dim_pad] matrices whose pad columns are zero (EPI_QKV's layout).
 * out: token-major [B*N][H*head_dim]; lse: [B][H][N]. */
int sgl_op_attn_fwd(int dtype, const void* q, const void* k, const void* v, void* out, float* lse, int B, int H, int N,
                    int head_dim, int head_dim_pad, int ld_qkv, sgl_stream stream);
/* dqkv: token-major [B*N][3*H*head_dim].  delta_scratch: 2 * B * H * N floats (per query and head the pair
 * {-lse * log2 e, -rowsum(dO * O) / sqrt(head_dim)}; ABI 1 took B * H * N floats here). */
int sgl_op_attn_bwd(int dtype, const void* q, const void* k, const void* v, const void* out, const void* dout,
                    const float* lse, void* dqkv, float* delta_scratch, int B, int H, int N, int head_dim,
                    int head_dim_pad, int ld_qkv, sgl_stream stream);
int sgl_op_colsum(int dtype, const void* in, int ld, int M, int N, float* out, int accumulate, float* scratch,
                  size_t scratch_bytes, sgl_stream stream);
int sgl_op_im2col(const float* pixels, int channels_last, void* out, int out_dtype, int B, int H, int W, int P, int Kp,
                  sgl_stream stream);
int sgl_op_pos_resize(const float* table, int native_grid, float* out, int gh, int gw, int D, sgl_stream stream);

/* ---- SID mask-decoder tail ---------------------------------------------------------------
 * Depthwise 3x3 convolution, zero padding 1, of SegFormerStrongDecoder's per-tap smoothing block
 * (nn.Conv2d(E, E, 3, padding=1, groups=E), Siglip2sidafrozen.py:713-718) on channels-last (B, gh, gw, E) data of
 * dtype f32 or bf16.  w9 holds the nine taps TAP-MAJOR in fp32: w9[k*E + e] = Conv2d.weight[e][0][k/3][k%3].
 * flip = 1 applies the 180-degree rotated taps (the data gradient: dx = sgl_op_dwconv3x3(dy, w9, NULL, flip = 1)).
 * E % 8 == 0 (bf16) / E % 4 == 0 (f32), E <= 1024, 256 % (E / 8 or 4) == 0. */
int sgl_op_dwconv3x3(const void* x, int dtype, const float* w9, const float* bias, void* y, int B, int gh, int gw, int E,
                     int flip, sgl_stream stream);
/* dw10[k*E + e] (+)= sum over pixels of x(shifted by tap k) * dy for k < 9, and dw10[9*E + e] (+)= sum dy (the bias
 * gradient); two deterministic stages, scratch >= sgl_op_dwconv3x3_wgrad_scratch_bytes(). */
size_t sgl_op_dwconv3x3_wgrad_scratch_bytes(int B, int gh, int gw, int E);
int sgl_op_dwconv3x3_wgrad(const void* x, const void* dy, int dtype, float* dw10, int accumulate, float* scratch,
                           size_t scratch_bytes, int B, int gh, int gw, int E, sgl_stream stream);

/* ---- GPU input pipeline (first slice) ----------------------------------------------------------
 * K.Resize(S, antialias=True) -> [MixUp] -> K.Normalize(mean, std) of the reference's per-batch GPU transform
 * (cifake_binary_classifier.py:1791-1794,812-817), optionally fused with the patch gather of the patch-embedding
 * convolution: src is uint8 NHWC (B,Hs,Ws,3) decoded bytes (src_is_u8_nhwc != 0, scaled by 1/255) or float32 NCHW
 * (B,3,Hs,Ws) in [0,1]; the image is resampled to S x S with torch's antialiased bilinear filter
 * (upsample_bilinear2d(antialias=True)); mix_index != NULL blends image b with image mix_index[b] (device int32[B]):
 * lam*img[b] + (1-lam)*img[mix_index[b]].
 *   patch_major != 0: out is the patch GEMM's A operand [B*(S/P)^2][Kp] in out_dtype (k = c*P*P + ky*P + kx, columns
 *                     >= 3*P*P zero), Kp as the encoder uses it (round_up(3*P*P, 64)): pass it to sgl_forward_slots with
 *                     channels_last = 2 and the im2col pass is skipped;
 *   patch_major == 0: out is (B,3,S,S) NCHW in out_dtype (the tensor the reference's transform returns). */
int sgl_op_preprocess(const void* src, int src_is_u8_nhwc, int B, int Hs, int Ws, void* out, int out_dtype, int S, int P,
                      int Kp, int patch_major, float mean, float std, const int* mix_index, float lam,
                      sgl_stream stream);

/* Augmentation branch of the video trainer's GPU transform (hidf_video_classifier.py:2868-2874): K.Resize(S, antialias) ->
 * RandomHorizontalFlip -> RandomRotation(+-5 deg, bilinear, zeros outside) -> ColorJitter -> K.Normalize, one pass, same
 * sources / outputs as sgl_op_preprocess.  Random draws stay with the caller: aug is a DEVICE table of B samples.
 *   flip != 0: mirror x.  (cos_a, sin_a): rotation about the image centre, counter-clockwise positive; (1, 0) = none.
 *   order[]: permutation of {0 brightness (x*f), 1 contrast ((x-m)*f+m, m = mean grey level of the image at that point),
 *   2 saturation ((x-grey)*f+grey), 3 hue (h += hue, fraction of the circle)}, each result clamped to [0,1];
 *   order[0] < 0 = no colour jitter for this sample.  grey_mean: device scratch of B floats.
 * kornia itself is not installed in the build image: parity with it is unpinned; oracle/preprocess_oracle.py restates
 * exactly the operators above (torchvision's definitions). */
typedef struct sgl_aug_sample {
  float flip, cos_a, sin_a, brightness, contrast, saturation, hue;
  int order[4];
  int reserved;
} sgl_aug_sample;
int sgl_op_preprocess_aug(const void* src, int src_is_u8_nhwc, int B, int Hs, int Ws, void* out, int out_dtype, int S,
                          int P, int Kp, int patch_major, float mean, float std, const sgl_aug_sample* aug,
                          float* grey_mean, sgl_stream stream);

/* Video tail (hidf_video_classifier.py:304-316): per-frame embeddings f (B*T, D) fp32 -> each frame L2-normalised ->
 * mean over the T frames of a clip -> out (B, D); inv_norm (B*T) keeps 1/|f_t| for the backward
 * d f_t = (g - fhat_t (fhat_t . g)) / (T |f_t|), g = d out[b]. */
int sgl_op_l2norm_tmean_fwd(const float* f, float* out, float* inv_norm, int B, int T, int D, sgl_stream stream);
int sgl_op_l2norm_tmean_bwd(const float* f, const float* inv_norm, const float* dout, float* df, int B, int T, int D,
                            sgl_stream stream);

/* ---- SID mask-decoder tail, second half (Siglip2sidafrozen.py:731-745,174-181) -------------------
 * y = sigmoid(g) * x on n elements (the channel gate applied to the concatenated taps, `gate * x` at :741-742), and its
 * backward: dx = dy * sigmoid(g), dg = dy * x * s(1-s) (gradient w.r.t. the PRE-sigmoid gate; dg / dx may be NULL).
 * dtype f32 or bf16; n % 4 (f32) / n % 8 (bf16) == 0; 16-byte aligned pointers. */
int sgl_op_gate_mul(const void* g, const void* x, void* y, size_t n, int dtype, sgl_stream stream);
int sgl_op_gate_mul_bwd(const void* dy, const void* g, const void* x, void* dg, void* dx, size_t n, int dtype,
                        sgl_stream stream);
/* bce_dice_loss (:174-181) evaluated straight from the LOW-RESOLUTION logit map: logits_lr (B,g,g) fp32 is what the 1x1
 * head produces on the token grid; every one of the S*S output pixels is its bilinear (align_corners=False, as
 * F.interpolate at :743) interpolation, computed in registers, so the (B,1,S,S) logits never exist in HBM.
 * fwd: partial[b][chunk][4] = {sum bce, sum p*t, sum p, sum t} over 8 output rows (chunks = sgl_op_seg_loss_chunks(S));
 *      the caller folds the chunks (fixed order) and forms  bce_w * mean(bce) + dice_w * (1 - mean_b(2I/(P+T+eps))).
 * bwd: dlogits_lr[b] = transposed interpolation of  coef[b][0]*(p - t) + coef[b][1]*p(1-p)*(2tD - 2I)/D^2,
 *      sums[b] = the folded forward sums {.., I, P, T}, D = P + T + eps; gathered per low-res pixel, no atomics. */
int sgl_op_seg_loss_chunks(int S);
int sgl_op_seg_loss_fwd(const float* logits_lr, const float* targets, float* partial, int B, int g, int S,
                        sgl_stream stream);
int sgl_op_seg_loss_bwd(const float* logits_lr, const float* targets, const float* sums, const float* coef,
                        float* dlogits_lr, int B, int g, int S, float eps, sgl_stream stream);

/* ---- optimizer step tail -----------------------------------------------------------------
 * Replaces, for a list of fp32 tensors, the reference's per-step pair
 *     torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)   Siglip2sidafrozen.py:1396
 *     torch.optim.AdamW(...).step()                                  Siglip2sidafrozen.py:1241-1244,1398
 * without a host synchronisation: the clip coefficient stays in device memory.
 * One table entry per parameter tensor; the table, the block map and the scratch live in DEVICE memory owned by the
 * caller.  g == NULL marks a parameter without a gradient this step (skipped, as torch does). */
typedef struct {
  float* p;        /* fp32 parameter, updated in place */
  const float* g;  /* fp32 gradient (not modified: the clip coefficient is applied on the fly) */
  float* m;        /* exp_avg */
  float* v;        /* exp_avg_sq */
  uint64_t n;      /* elements */
  float lr, weight_decay; /* of the parameter group the tensor belongs to */
} sgl_adamw_tensor;
/* HOST helper: cuts tensor t into ceil(numel[t]/4096) chunks and writes (tensor, chunk) int32 pairs into blockmap
 * (host memory, capacity in pairs; may be NULL to size it).  Returns the number of pairs (= workgroups). */
int64_t sgl_adamw_plan(const uint64_t* numel, int ntensors, int32_t* blockmap, int64_t capacity_pairs);
/* norm_and_coef[0] = sqrt(sum g^2) over the table; [1] = min(1, max_norm/(norm+1e-6)) (1 when max_norm <= 0).
 * partials: nblocks floats of scratch.  Reduction order is fixed: bitwise reproducible. */
int sgl_op_grad_norm(const sgl_adamw_tensor* table, const int32_t* blockmap, int64_t nblocks, float max_norm,
                     float* partials, float* norm_and_coef, sgl_stream stream);
/* One AdamW step (decoupled weight decay, torch/optim/adamw.py operation order) on every tensor of the table;
 * step >= 1 is the bias-correction exponent; norm_and_coef (may be NULL) is the output of sgl_op_grad_norm. */
int sgl_op_adamw(const sgl_adamw_tensor* table, const int32_t* blockmap, int64_t nblocks, double beta1, double beta2,
                 double eps, int step, const float* norm_and_coef, sgl_stream stream);
/* ---- AdamW that also leaves behind everything a parameter update invalidates (kernel work-list k11) -----------------
 * Per table entry, optional destinations written in the same pass as the update:
 *   dst / dst_t : compute-dtype (dtype) row-major copy dst[(r-row0)*ld + c] and transposed copy dst_t[c*ld_t + (r-row0)]
 *                 of the rows r >= row0 of the [rows, cols] parameter: the encoder's weight shadows (sgl_prepare_weights
 *                 layouts; pad regions are never touched), so no re-cast is needed before the next forward;
 *   dst_f32     : fp32 copy of a 1-D tensor (the fused qkv / fc1 bias vectors of the shadow arena);
 *   ema         : ExponentialMovingAverage shadow (cifake_binary_classifier.py:222-225): ema = ema*decay + p*(1-decay);
 *   group       : index into the per-launch (lr, weight_decay) list (changing the learning rate every step then needs no
 *                 table upload); -1 = use the table entry's own lr / weight_decay.
 * Entries with dst or dst_t are walked in 64x64 tiles: plan them as ceil(rows/64)*ceil(cols/64) chunks, i.e. pass
 * that count * 4096 as the tensor's numel to sgl_adamw_plan. */
typedef struct {
  void* dst;
  void* dst_t;
  float* dst_f32;
  float* ema;
  int ld, ld_t, rows, cols, row0, dtype, group, reserved;
} sgl_adamw_aux;
/* HOST helper: fills aux[i].{dst, dst_t, dst_f32, ld, ld_t, rows, cols, row0, dtype} for every table entry whose .p is one
 * of the master tensors in `w` that has a copy in the shadow arena `shadow` of `ctx` (table and aux are HOST arrays of
 * ntensors entries; other aux fields are left untouched).  Returns the number of entries bound. */
int sgl_adamw_bind_shadows(const sgl_ctx* ctx, const sgl_weights* w, void* shadow, const sgl_adamw_tensor* table_host,
                           sgl_adamw_aux* aux_host, int ntensors);
/* sgl_op_grad_norm with a gradient scale s (gradients are rank sums, s = 1/world): [0] = s*norm,
 * [1] = s*min(1, max_norm/(s*norm + 1e-6)) (= s when max_norm <= 0). */
int sgl_op_grad_norm_scaled(const sgl_adamw_tensor* table, const int32_t* blockmap, int64_t nblocks, float max_norm,
                            float grad_scale, float* partials, float* norm_and_coef, sgl_stream stream);
/* group_lr_wd_host: HOST array of ngroups (<= 16) {lr, weight_decay} pairs passed by value with the launch, or NULL with
 * ngroups = 0; ema_decay is used by entries with aux.ema != NULL. */
int sgl_op_adamw_ex(const sgl_adamw_tensor* table, const sgl_adamw_aux* aux, const int32_t* blockmap, int64_t nblocks,
                    double beta1, double beta2, double eps, int step, const float* norm_and_coef,
                    const float* group_lr_wd_host, int ngroups, double ema_decay, sgl_stream stream);
/* Weight EMA of the CiFake trainer (ExponentialMovingAverage.update, cifake_binary_classifier.py:222-225):
 * shadow = shadow*decay + p*(1-decay) for every table entry, with p = entry.p (read only) and shadow = entry.m. */
int sgl_op_ema(const sgl_adamw_tensor* table, const int32_t* blockmap, int64_t nblocks, double decay,
               sgl_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* SIGLIP_HIP_H */

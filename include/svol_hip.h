/*
 * svol_hip.h — C-ABI of libsvol_hip.so, the MI355X (gfx950) kernel library behind
 * the SVOL hot path (sketch<->video cross-modal DETR head + Hungarian/GIoU loss).
 *
 * The reference (sangminwoo/SVOL) is pure Python and has NO FFI: its boundary
 * is lib/modeling/{model,svanet,loss,matcher}.py.  This header
 * is the build's own inner boundary; every entry point cites the reference
 * arithmetic it replaces (paths relative to the reference root).  The Python
 * host (svol_amd/) binds it with ctypes.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types; every pointer is DEVICE memory
 *    owned by the caller (no ownership transfer, no allocation inside);
 *  - `dtype` selects the activation/compute element type: SVOL_F32 (the
 *    reference's precision) or SVOL_BF16 (fp32 accumulate, fp32 statistics);
 *  - matrices are row-major with explicit leading dimensions (elements);
 *  - every function enqueues on `stream` (a hipStream_t passed as void*) and
 *    returns immediately: 0 on success, <0 = SVOL_E_* (never throws, never
 *    syncs).  No hidden global mutable state: callable concurrently from the
 *    forward thread and the autograd thread.
 */
#ifndef SVOL_HIP_H
#define SVOL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVOL_F32 0
#define SVOL_BF16 1
#define SVOL_F16 2  /* fp16 operands (v_mfma_*_f16), fp32 accumulate / statistics: the GEMM, LayerNorm, gate and attention entry
                       points of the SVANet path; the ViT / ResNet / enc-dec-only entry points take SVOL_F32 / SVOL_BF16 */

#define SVOL_OK 0
#define SVOL_E_INVALID (-1)     /* null pointer / bad size / misaligned          */
#define SVOL_E_UNSUPPORTED (-2) /* shape outside what the kernels implement      */
#define SVOL_E_LAUNCH (-3)      /* hipGetLastError() != hipSuccess after launch  */

/* epilogue / activation selectors */
#define SVOL_ACT_NONE 0
#define SVOL_ACT_RELU 1
#define SVOL_ACT_GELU 2    /* exact erf GELU: cross_modal_transformer.py:189-190 */
#define SVOL_ACT_SIGMOID 3 /* svanet.py:127 */
#define SVOL_ACT_GELU_D 5  /* GELU whose saved tensor is the DERIVATIVE: svol_gemm_nt writes gelu'(pre-activation) into `pre`
                            * instead of the pre-activation (same Phi / exp as the activation itself: two more VALU ops), and
                            * svol_gemm_nt_dact / svol_act_bwd multiply by `aux` as it is — the MLP backward's epilogue loses its
                            * erf + exp per element (the K = 256 kernels are VALU-issue bound: profiles/round2_ws_gemm_lab.md) */
#define SVOL_ACT_RELU_RES 4 /* relu(A*B^T + bias + residual): the ReLU AFTER the identity add of a ResNet BasicBlock (svol_gemm_nt only) */

int svol_abi_version(void);
const char* svol_strerror(int code);

/* ---- dtype plumbing ----------------------------------------------------- */
/* dst[i] = (dtype_dst) src[i];  n elements. */
int svol_cast(const void* src, int dtype_src, void* dst, int dtype_dst, int64_t n, void* stream);
/* src fp32 [R,C] -> dst (dtype) [R,C] (may be null) and dstT (dtype) [C,R] (may be null).
 * Used once per step per weight: nn.Linear weights are [out,in]; dX = dY*W needs W^T K-contiguous. */
int svol_cast_transpose(const float* src, void* dst, void* dstT, int dtype, int64_t R, int64_t C, void* stream);
/* The same for MANY weights in one launch (one per training step instead of one per weight).  `descs` is a DEVICE
 * array of n_desc records, sorted by tile_begin:
 *   struct { const float* src; void* dst; void* dstT; void* dstS; int32 R, C, tiles_c, tile_begin; }   (48 bytes)
 * tiles_c = ceil(C/32); a weight owns ceil(R/32)*tiles_c consecutive 32x32 tiles starting at tile_begin;
 * total_tiles = their sum.  dst, dstT, dstS may each be NULL per record; dstS (bf16 only) receives the SPLIT copy
 * [R, 2C] = [hi | lo] of svol_cast_split. */
int svol_cast_transpose_multi(const void* descs, int32_t n_desc, int64_t total_tiles, int dtype, void* stream);
/* Split bf16 copy of an fp32 weight: dst_hilo [R, 2C] bf16 = [hi | lo], hi = bf16(w), lo = bf16(w - hi) (src rows ld_src
 * floats apart).  hi + lo carries 16 mantissa bits of w: rounding the value-projection WEIGHTS to bf16 was 95 % of the
 * bf16 logit error at the benchmark depth (it shifts every token's value the same way; attention over 6272 keys and
 * LayerNorm do not average it out, profiles/round2_bf16_output_error.md), so those products use both halves. */
int svol_cast_split(const float* src, int64_t ld_src, void* dst_hilo, int64_t R, int64_t C, void* stream);

/* One AdamW step (torch.optim.AdamW semantics: decoupled weight decay, amsgrad / maximize off — the reference's optimizer,
 * train.py:98-99) over a FLAT fp32 range of parameters p with gradients g and moment buffers m, v, all 16-byte aligned:
 *   g' = g * grad_scale;  p *= 1 - lr*wd;  m += (g' - m)(1 - b1);  v = b2*v + (1 - b2) g'^2;
 *   p -= lr/(1 - b1^step) * m / (sqrt(v)/sqrt(1 - b2^step) + eps).          step counts from 1. */
int svol_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int64_t step, float grad_scale, void* stream);
/* The same update; the gradient range is ZEROED behind its read (round 6, ABI 7): optimizer.zero_grad() of the next iteration
 * (train.py:222) folded into the step, so that the step boundary — a chain of small dependent launches with the chip idle — loses
 * one fill launch per gradient bucket.  For a caller whose loop reads no gradient between step() and zero_grad(). */
int svol_adamw_flat_zero(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int64_t step, float grad_scale, void* stream);
/* Dynamic loss scaling for fp16 operands (the reference's fp16 mode is apex amp: dynamic scale, overflowed steps skipped,
 * configs.py:52-61, train.py:111-114,231-232), without a host synchronisation.  scaler_state: four floats on the DEVICE —
 * [0] the loss scale (the caller multiplies the loss by it ON THE DEVICE), [1] overflow flag of the current step, [2] clean steps since
 * the scale last changed, [3] optimizer steps really taken.  Per step: svol_grad_finite over every gradient range (sets [1] on an
 * inf / NaN), svol_adamw_flat_scaled over every range (g' = g * grad_mul / state[0]; a no-op when [1] is set; the bias corrections use
 * step = state[3] + 1), then svol_loss_scaler_update once: overflow -> scale *= backoff_factor (>= min_scale), else state[3]++ and
 * after growth_interval clean steps scale *= growth_factor (<= max_scale); clears [1]. */
int svol_grad_finite(const float* g, int64_t n, float* scaler_state, void* stream);
int svol_adamw_flat_scaled(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                           float weight_decay, float grad_mul, const float* scaler_state, void* stream);
int svol_loss_scaler_update(float* scaler_state, float growth_factor, float backoff_factor, int64_t growth_interval, float min_scale,
                            float max_scale, void* stream);

/* ---- GEMMs (nn.Linear and its backward) --------------------------------- */
/* C[M,N] = act((A[M,K] * B[N,K]^T + bias[N]) * colscale[N]) + residual[M,N]
 *   A2/n_split: output columns n >= n_split read their A operand from A2 instead of A
 *               (fused q/k-with-pos vs v-without-pos projection, cross_modal_transformer.py:137-138);
 *               pass A2=NULL, n_split=0 when unused.  n_split must be a multiple of 128.
 *   bias fp32 or NULL; colscale fp32 or NULL (per-output-column factor: the attention query columns are
 *   emitted pre-multiplied by d_h^-1/2 * log2(e) so the attention kernels exponentiate raw MFMA results);
 *   pre_act_out (dtype, ld = ldp) or NULL receives the pre-activation (saved
 *   for backward); residual (ld = ldr) or NULL.  out_f32 != 0: C and residual are fp32 regardless of
 *   dtype (the fp32 residual stream that feeds the post-norms).  Replaces nn.Linear/F.relu/F.gelu:
 *   svanet.py:174-181, cross_modal_transformer.py:163-179, svanet.py:144-156. */
int svol_gemm_nt(const void* A, int64_t lda, const void* A2, int64_t n_split, const void* B, int64_t ldb,
                 void* C, int64_t ldc, const float* bias, const float* colscale, int act, void* pre_act_out,
                 int64_t ldp, const void* residual, int64_t ldr, int out_f32, int64_t M, int64_t N, int64_t K,
                 int dtype, void* stream);
/* C[M,N] (bf16) = A[M,K] * (W_hi + W_lo)[N,K]^T + bias[N] with W_hilo [N, 2K] from svol_cast_split (ld = ldw): ONE
 * K-concatenated product [A | A] * [W_hi | W_lo]^T — the kernels wrap A's contraction index, [A | A] is never materialised;
 * fp32 accumulation over both halves, one rounding of the result.  The V projections of nn.MultiheadAttention in bf16 mode
 * (cross_modal_transformer.py:137-139,151-154; transformer.py:183-186,237-246). */
int svol_gemm_nt_split(const void* A, int64_t lda, const void* W_hilo, int64_t ldw, void* C, int64_t ldc, const float* bias,
                       int64_t M, int64_t N, int64_t K, void* stream);
/* Fused MLP backward step: C[M,N] = (A[M,K] * B[N,K]^T) .* gelu'(pre[M,N]);  colsum[N] (fp32, may be NULL,
 * caller zeroes) += column sums of C (= the fc1 bias gradient).  Replaces dh = ds W2, dpre = dh *
 * gelu'(pre), db1 = sum(dpre) of the MLP backward (cross_modal_transformer.py:163-179). */
int svol_gemm_nt_dgelu(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                       const void* pre, int64_t ldp, float* colsum, int64_t M, int64_t N, int64_t K, int dtype,
                       void* stream);
/* The same step for either MLP activation: C = (A * B^T) .* act'(aux), act = SVOL_ACT_GELU (aux = the saved
 * PRE-activation), SVOL_ACT_GELU_D (aux = the derivative itself, saved by svol_gemm_nt under the same code) or SVOL_ACT_RELU
 * (aux = the saved POST-activation, relu' = [aux > 0]: the F.relu FFN of the enc/dec Transformer, transformer.py:191,245). */
int svol_gemm_nt_dact(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                      const void* aux, int64_t ldaux, int act, float* colsum, int64_t M, int64_t N, int64_t K,
                      int dtype, void* stream);

/* dW[N,K] (fp32, ld = ldc) (+)= A[Mc,N]^T * B[Mc,K]   (contraction over the Mc rows; weight gradient).
 * The output is accumulated with fp32 atomics: the caller zeroes C first unless it wants accumulation.
 * colsum (fp32 [N], may be NULL, caller zeroes) += column sums of A — the bias gradient, computed on the
 * matrix pipe (A^T * ones) by the workgroups that already hold the A tiles. */
int svol_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, float* colsum,
                 int64_t Mc, int64_t N, int64_t K, int dtype, void* stream);
/* Up to SVOL_TN_GROUP_MAX svol_gemm_tn problems of ONE element type in one launch: the weight gradients of one backward block
 * (dW1, dW2, dWo, dW_in of a transformer block).  `problems` is a HOST array of n records (copied into the kernel arguments);
 * more than SVOL_TN_GROUP_MAX problems, or shapes the grouped kernels do not take, are issued one by one. */
#define SVOL_TN_GROUP_MAX 6
typedef struct svol_tn_problem {
    const void* A; int64_t lda;   /* [Mc, N] */
    const void* B; int64_t ldb;   /* [Mc, K] */
    float* C; int64_t ldc;        /* [N, K] fp32, accumulated into */
    float* colsum;                /* [N] fp32 or NULL, accumulated into */
    int64_t Mc, N, K;
} svol_tn_problem;
int svol_gemm_tn_grouped(const svol_tn_problem* problems, int32_t n, int dtype, void* stream);
/* out[N] (fp32) += sum_m X[m,n]   (bias gradient).  Caller zeroes `out`. */
int svol_colsum(const void* X, int64_t ldx, float* out, int64_t M, int64_t N, int dtype, void* stream);
/* dpre[i] = dy[i] * act'(aux[i]); aux = post-activation for RELU/SIGMOID, pre-activation for GELU. */
int svol_act_bwd(const void* dy, const void* aux, void* dpre, int act, int64_t n, int dtype, void* stream);

/* ---- LayerNorm (+ dropout, + positional add) ----------------------------
 * The post-norm residual stream is kept in fp32 even when dtype is bf16 (x_f32 != 0: x is fp32);
 * the kernel emits the fp32 value (y32, feeds the next residual add) and the compute-dtype copies the
 * GEMMs consume (y, and ypos = y + pos).  Any of y32 / y may be NULL (not both); ypos/pos go together.
 * pos has `pos_rows` rows and is indexed row % pos_rows so a [N,d] query embedding broadcasts over
 * the batch.  Dropout (p, seed) is a stateless counter-based mask applied to LN's output; the effective
 * seed is seed + *seed_offset_dev (device int64, may be NULL) so that a captured hipGraph draws a fresh
 * mask on every replay.  mean/rstd
 * fp32 [M] are saved for backward.  nn.LayerNorm + nn.Dropout of svanet.py:168-178; post-norms
 * cross_modal_transformer.py:127-158. */
int svol_layernorm_fwd(const void* x, int x_f32, const float* gamma, const float* beta, float* y32, void* y,
                       void* ypos, const void* pos, int64_t pos_rows, float* mean, float* rstd, int64_t M, int64_t D,
                       float dropout_p, uint64_t seed, const int64_t* seed_offset_dev, int dtype, void* stream);
/* dx = LN'(dy32 + dy + dy2) (each may be NULL, not all); outputs dx32 (fp32) and/or dx (dtype);
 * dgamma/dbeta (fp32) accumulated with atomics (caller zeroes).  dx_colsum (fp32 [D], may be NULL,
 * caller zeroes) += column sums of dx — the bias gradient of the Linear that produced LN's input. */
int svol_layernorm_bwd(const float* dy32, const void* dy, const void* dy2, const void* x, int x_f32,
                       const float* gamma, const float* mean, const float* rstd, float* dx32, void* dx,
                       float* dgamma, float* dbeta, float* dx_colsum, int64_t M, int64_t D, float dropout_p,
                       uint64_t seed, const int64_t* seed_offset_dev, int dtype, void* stream);

/* ---- stand-alone dropout (the enc/dec Transformer in training mode: transformer.py:165-215 encoder, :225-295 decoder) ----
 * y[r, k] = x[r, k] * keep(seed, r, k) over the [n / row_len, row_len] view, keep = 0 with probability p else 1/(1-p): a stateless
 * counter-based mask of (seed, row, column), so the backward pass (and a test that wants the mask itself: x = ones) regenerates it.
 * y may alias x.  The FFN dropout after the activation (:168,:238) and the backward of the residual dropouts.  (Round 4: the mask is
 * keyed by row and column instead of the flat element index — same contract, a several times cheaper generator; the drop probability
 * is p rounded up to a multiple of 2^-16; row_len <= 2^32.) */
int svol_dropout(const void* x, void* y, int64_t n, int64_t row_len, float p, uint64_t seed, int dtype, void* stream);
/* out32 = res32 + t32 * keep(seed, r, k): the residual dropouts src + dropout1(src2) (:171,:177,:232-247) on the fp32 stream;
 * out32 may alias t32. */
int svol_dropout_add(const float* t32, const float* res32, float* out32, int64_t n, int64_t row_len, float p, uint64_t seed, void* stream);

/* ---- sine positional encoding (position_encoding.py:51-71) -------------- */
/* mask [B,L] float (1 = valid) -> pos [B,L,D] (dtype). */
int svol_posenc_sine(const float* mask, void* pos, int64_t B, int64_t L, int64_t D, int dtype, void* stream);

/* ---- multi-head attention core (flash style, never materialises Lq x Lk) -
 * q/k/v/o are [B*L, ld] row-major, head h occupies columns [h*dh, (h+1)*dh).  softmax(q k^T * scale
 * + kbias) v; kbias [B,Lk] fp32 additive (0 / -inf = key_padding_mask, cross_modal_transformer.py:154)
 * or NULL.  lse2 [B,H,Lq] fp32 = log2-domain log-sum-exp saved for backward.  dh <= 32, dh % 8 == 0.
 * q_premul != 0: the caller already multiplied q by q_premul = scale*log2(e) (fused into the projection
 * GEMM epilogue); the kernels then exponentiate the raw MFMA results and dq is still the gradient with
 * respect to the UNscaled q.  Pass 0 for plain q.
 * ws / ws_bytes: optional 16-byte aligned scratch (may be NULL / 0).  Launches with few queries and many keys
 * (the N = 100 object queries attending to L = 6272 video tokens: 64 workgroups on 256 CUs) are split over the
 * keys when the scratch is large enough — svol_attn_ws_bytes() says how much that takes; partial results are
 * merged by a small second kernel.  Launches with a key bias or a key count that is not a multiple of 128 and MANY queries
 * use the scratch for one int per (batch, 128-key tile) — the tile's class: no bias / mixed / fully masked (skipped) — so
 * that the fast kernels can serve them; without scratch such launches run on the general (slower) kernels.
 * Unmasked bf16 launches with dh == 32 and pre-multiplied q use it for one int per workgroup (batch, head, 128-query tile): the
 * fast forward anchors the softmax once per query and FLAGS a workgroup whose row sums overflowed; the per-tile-maximum kernel
 * launched right behind it recomputes exactly the flagged ones.  Without scratch only the per-tile-maximum kernel runs.
 * Unmasked 16-bit launches with H == 8, dh == 32, pre-multiplied q, Lq and Lk multiples of 128 and Lk >= 1024 (the video
 * self-attention) run the BACKWARD as one pass when the scratch holds B*Lq*H*dh + B*H*4*2*(Lk % 512)*dh floats: an fp32 dQ image that
 * key-stationary 512-key workgroups add to with fp32 atomics (rounded into dq at the end) and the dK / dV partials of a head's tail key
 * group (csrc/attention_bf16.hip: attn_bwd_sp_bf16; 10 instead of 16 matrix products per score block and one exp pass instead of two).
 * dK / dV are bit-reproducible, dQ to fp32 summation order.  Without that much scratch the two-pass kernels run.
 * svol_attn_ws_bytes() returns the largest of these needs for the shape.  Nothing is allocated inside the library.
 * Replaces the core of nn.MultiheadAttention (cross_modal_transformer.py:139,147,154). */
int64_t svol_attn_ws_bytes(int64_t B, int64_t H, int64_t Lq, int64_t Lk, int64_t dh);
int svol_attn_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                  int64_t ldo, float* lse2, const float* kbias, int64_t B, int64_t H, int64_t Lq, int64_t Lk,
                  int64_t dh, float scale, float q_premul, void* ws, int64_t ws_bytes, int dtype, void* stream);
/* delta: fp32 scratch of 3*B*H*Lq elements (rowsum(dO * O), then the row constants of the second pass re-encoded for its
 * LDS-DMA loads); then dq / dk / dv (same layouts as q/k/v). */
int svol_attn_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                  int64_t ldo, const void* d_o, int64_t lddo, const float* lse2, float* delta, const float* kbias,
                  void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, int64_t B, int64_t H,
                  int64_t Lq, int64_t Lk, int64_t dh, float scale, float q_premul, void* ws, int64_t ws_bytes,
                  int dtype, void* stream);

/* Round 6 (ABI 7): the zero fill of the single pass's fp32 dQ image off the critical path.  The image (B*Lq*H*dh floats at the head
 * of ws) must be zero when the key-stationary kernel starts; svol_attn_bwd zeroes it in its prologue kernel, 51 MB of stores per cfg2
 * launch in front of an issue-bound kernel that leaves the memory side idle.  A caller that alternates between TWO workspaces can
 * have the other one zeroed beside that kernel instead:
 *   svol_attn_bwd_sp_image_bytes: bytes of the image if an unmasked, pre-multiplied launch of this shape with a workspace of ws_bytes
 *     runs the single pass, else 0 (then the two entries below add nothing).
 *   svol_attn_bwd_zero_ws: zeroes that image on `stream` (SVOL_E_UNSUPPORTED when the shape is not the single pass's).
 *   svol_attn_bwd_ex = svol_attn_bwd plus  flags: SVOL_ATTN_DQ_PREZEROED = the image of ws is zero already (honoured by the single
 *     pass only; every other path ignores it)  ·  ev_prep: optional hipEvent_t recorded on `stream` behind the prologue kernel (other
 *     paths: behind the last launch) — the point after which work on the caller's OTHER workspace may start on another stream.
 * Same values as svol_attn_bwd (the zero fill moves, nothing else); replaces the same reference lines. */
#define SVOL_ATTN_DQ_PREZEROED 1
int64_t svol_attn_bwd_sp_image_bytes(int64_t B, int64_t H, int64_t Lq, int64_t Lk, int64_t dh, int64_t ws_bytes, int dtype);
int svol_attn_bwd_zero_ws(void* ws, int64_t ws_bytes, int64_t B, int64_t H, int64_t Lq, int64_t Lk, int64_t dh, int dtype, void* stream);
int svol_attn_bwd_ex(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                     int64_t ldo, const void* d_o, int64_t lddo, const float* lse2, float* delta, const float* kbias,
                     void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, int64_t B, int64_t H,
                     int64_t Lq, int64_t Lk, int64_t dh, float scale, float q_premul, void* ws, int64_t ws_bytes,
                     int dtype, int flags, void* ev_prep, void* stream);

/* The same with attention-probability dropout (nn.MultiheadAttention(dropout = p) in training mode, transformer.py:158-160): the softmax
 * numerators that feed P V are multiplied by keep(seed, row = (b*H + h)*Lq + q, column = key) — svol_dropout over a ones tensor
 * [B,H,Lq,Lk] with row_len = Lk and the same seed IS the mask; lse2 stays the undropped softmax's.  dropout_p > 0 runs on the general kernels (no anchored fast path). */
int svol_attn_fwd_dropout(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                          int64_t ldo, float* lse2, const float* kbias, int64_t B, int64_t H, int64_t Lq, int64_t Lk,
                          int64_t dh, float scale, float q_premul, void* ws, int64_t ws_bytes, float dropout_p, uint64_t seed,
                          int dtype, void* stream);
int svol_attn_bwd_dropout(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                          int64_t ldo, const void* d_o, int64_t lddo, const float* lse2, float* delta, const float* kbias,
                          void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, int64_t B, int64_t H,
                          int64_t Lq, int64_t Lk, int64_t dh, float scale, float q_premul, void* ws, int64_t ws_bytes,
                          float dropout_p, uint64_t seed, int dtype, void* stream);

/* ---- sketch->video gate (cross_modal_transformer.py:122-127) ------------
 * Only the head-averaged attention weights of the 1-query MHA are used by the reference, so the
 * K projection collapses to one d-vector per (batch, head): u[b,h,:] = scale * W_k,h^T q_{b,h}
 * (the k-bias term is constant over keys and cancels in the softmax).
 *   scores[b,h,l] = (x[b,l,:] + pos[b,l,:]) . u[b,h,:]
 *   a[b,l]        = mean_h softmax_l(scores[b,h,:])
 *   y             = LN1(x * (1 + a)) ;  ypos = y + pos
 * x32 is the fp32 residual stream [B*L, D]; pos / y / ypos are `dtype`; y32 fp32 (may be NULL).
 * ws: fp32 workspace of B*H*(L+2) floats (scores, then per-(b,h) max and sum). */
int svol_gate_fwd(const float* x32, const void* pos, const float* u, const float* gamma, const float* beta,
                  float* y32, void* y, void* ypos, float* a, float* mean, float* rstd, float* ws, int64_t B,
                  int64_t L, int64_t D, int64_t H, int dtype, void* stream);
/* (dy32 + dy + dy2) -> dx32 (fp32), du (fp32, caller zeroes), dgamma/dbeta (fp32 atomics, caller
 * zeroes).  ws: the forward's workspace (scores/max/sum); ws2: B*L + B*H fp32 scratch. */
int svol_gate_bwd(const float* dy32, const void* dy, const void* dy2, const float* x32, const void* pos,
                  const float* u, const float* gamma, const float* a, const float* mean, const float* rstd,
                  const float* ws, float* ws2, float* dx32, float* du, float* dgamma, float* dbeta, int64_t B,
                  int64_t L, int64_t D, int64_t H, int dtype, void* stream);

/* The gate of layer i + 1 with its score pass folded into layer i's last LayerNorm (round 6; cross_modal_transformer.py:122-127 behind
 * :143): the gate vectors of every layer depend only on the sketch token, so they exist before the layer loop.
 *   svol_layernorm_gate_scores_fwd: y = LN(x32) exactly as svol_layernorm_fwd (no dropout; y32 / y / ypos = y + pos, each may be
 *     NULL, not all; pos REQUIRED, one row per token) and scores_next[b,h,l] = (y + pos)[b,l,:] . u_next[b,h,:] — the bits of
 *     svol_gate_fwd's own score pass over the stored fp32 row.  scores_next = the first B*H*L floats of the NEXT gate's ws.
 *     L % 4 != 0 or D > 256 -> SVOL_E_UNSUPPORTED (run svol_layernorm_fwd, and svol_gate_fwd for the next layer).
 *   svol_gate_fwd_scored: svol_gate_fwd with ws[0 .. B*H*L) already filled that way (statistics + apply only). */
int svol_layernorm_gate_scores_fwd(const float* x32, const float* gamma, const float* beta, float* y32, void* y, void* ypos,
                                   const void* pos, float* mean, float* rstd, const float* u_next, float* scores_next,
                                   int64_t B, int64_t L, int64_t D, int64_t H, int dtype, void* stream);
int svol_gate_fwd_scored(const float* x32, const void* pos, const float* u, const float* gamma, const float* beta,
                         float* y32, void* y, void* ypos, float* a, float* mean, float* rstd, float* ws, int64_t B,
                         int64_t L, int64_t D, int64_t H, int dtype, void* stream);

/* Gate VECTORS (the B*d-sized algebra in front of the gate): q[b,:] = W_q s_b + b_q, u[b,h,:] = d_h^-1/2 * W_k,h^T q[b,h,:],
 * with W_in [3d,d] / b_in [3d] the packed nn.MultiheadAttention in_proj parameters (rows 0..d-1 = W_q, d..2d-1 = W_k),
 * skch [B,d] the projected sketch token, all fp32.  Backward: du [B,H,d] -> dskch [B,d] (may be NULL), and dW_in / db_in are
 * ACCUMULATED INTO (rows 0..2d-1 / entries 0..d-1; every element has one owner, no atomics), dq_ws: B*d fp32 scratch. */
int svol_gate_vectors_fwd(const float* skch, const float* W_in, const float* b_in, float* q_out, float* u_out, int64_t B, int64_t D,
                          int64_t H, void* stream);
int svol_gate_vectors_bwd(const float* du, const float* skch, const float* W_in, const float* q, float* dq_ws, float* dskch,
                          float* dW_in, float* db_in, int64_t B, int64_t D, int64_t H, void* stream);
/* The gate vectors of up to SVOL_GATE_VEC_MAX_LAYERS layers that read the SAME sketch token, one launch forward and two backward for
 * all of them (the reference computes them layer by layer inside nn.MultiheadAttention, cross_modal_transformer.py:122-123; they
 * depend on the sketch token and the layer's parameters only).  Every pointer argument except skch is a HOST array of n_layers device
 * pointers with the meaning of the single-layer entry; dskch[l] receives layer l's sketch gradient (the caller adds them up), the
 * array or an entry of it may be NULL. */
#define SVOL_GATE_VEC_MAX_LAYERS 8
int svol_gate_vectors_fwd_multi(const float* skch, const float* const* W_in, const float* const* b_in, float* const* q_out,
                                float* const* u_out, int64_t n_layers, int64_t B, int64_t D, int64_t H, void* stream);
int svol_gate_vectors_bwd_multi(const float* const* du, const float* skch, const float* const* W_in, const float* const* q,
                                float* const* dq_ws, float* const* dskch, float* const* dW_in, float* const* db_in, int64_t n_layers,
                                int64_t B, int64_t D, int64_t H, void* stream);

/* ---- set matching + criterion (matcher.py:38-159, loss.py:39-157) -------
 * A "problem" is one LSAP block the reference solves with scipy: one video for
 * HungarianMatcher (matcher.py:158), one (video, frame) for PerFrameMatcher (matcher.py:92-96),
 * replicated for every decoder layer (loss.py:148-155).  Problem p covers prediction rows
 * [pred_off[p], pred_off[p]+pred_cnt[p]) of logits[R,2]/boxes[R,4] (fp32, R = layers*B*N) and target
 * boxes [tgt_off[p], tgt_off[p]+tgt_cnt[p]) of tgt_boxes[sumM,4] (fp32 cxcywh).
 *   cost = w_bbox * L1 + w_giou * (-GIoU) + w_class * (-softmax(logits)[0])   (matcher.py:76-85)
 * written to cost[cost_off[p] + i*tgt_cnt[p] + j].
 * box_status (int32 [n_problems], may be NULL): box_status[p] = 1 when a prediction or target box of problem p fails
 * generalized_box_iou's early check x1 >= x0 and y1 >= y0 (box_utils.py:51-52; NaN coordinates fail it), else 0.
 * The reference raises AssertionError there; the host side of this build raises it lazily from the flag. */
int svol_match_cost(const float* logits, const float* boxes, const float* tgt_boxes, const int32_t* pred_off,
                    const int32_t* pred_cnt, const int32_t* tgt_off, const int32_t* tgt_cnt,
                    const int64_t* cost_off, float* cost, int32_t n_problems, float w_bbox, float w_giou,
                    float w_class, int32_t* box_status, void* stream);
/* Batched rectangular LSAP, bit-exact restatement of scipy.optimize.linear_sum_assignment
 * (Crouse 2016 shortest augmenting path, fp64 duals, tall blocks solved transposed, reverse column
 * scan, tie rule "lower, or equal and unassigned").  match[r] = matched target index (global row of
 * tgt_boxes) for prediction row r, or -1.  status[p] = 0 ok, 1 invalid entries (NaN/-inf), 2 infeasible.
 * max_dim = max over problems of max(pred_cnt, tgt_cnt) (sizes the per-problem LDS). */
int svol_lsap_batched(const float* cost, const int64_t* cost_off, const int32_t* pred_off, const int32_t* pred_cnt,
                      const int32_t* tgt_off, const int32_t* tgt_cnt, int32_t* match, int32_t* status,
                      int32_t n_problems, int32_t max_dim, void* stream);
/* ---- the heads and the criterion's autograd glue as single launches (the forward -> backward turn of the training step) --------------
 * svol_heads_fwd: SVANet's class head Linear(d, 2) (svanet.py:44,125) and box head MLP(d, d, 4, 3) -> sigmoid (svanet.py:42,126-127,144-156)
 * on the stacked decoder states hs [R = layers * B * N, D] (fp32, exact-fp32 MFMA whatever the compute dtype): logits [R,2], boxes [R,4],
 * and the two hidden activations h1, h2 [R,D] (post-ReLU) for the backward.  D a multiple of 32, <= 512.
 * svol_heads_bwd: its backward from dlogits [R,2] / dboxes [R,4]: dhs [R,D]; gp0, gp1 [R,D] = the gradients w.r.t. the pre-activations
 * of MLP layers 0 / 1 — the caller forms dW0 = gp0^T hs, db0 = colsum(gp0), dW1 = gp1^T h1, db1 = colsum(gp1) with svol_gemm_tn, off the
 * critical path —; the 2- and 4-row weight gradients dWc [2,D], dbc [2], dW2 [4,D], db2 [4] are ADDED in place (fp32 atomics: the caller
 * zeroes them or passes running sums; SVOL_E_UNSUPPORTED under SVOL_DETERMINISTIC).
 * svol_set_loss_bwd: the criterion's backward (loss.py:39-60,76-103 differentiated): dlogits = g_label * dl[layer][0], dboxes = g_bbox *
 * dl[layer][1] + g_giou * dl[layer][2] from svol_set_loss's unit gradients and dl [layers,4] (device) = d(objective) / d(loss table).
 * svol_weighted_total / _bwd: out[0] = sum_i x[i] w[i] (train.py:227-228 on the [layers,4] loss table, w = weight_dict laid out
 * likewise), dx[i] = w[i] dout[0]; n <= 1024. */
int svol_heads_fwd(const float* hs, const float* Wc, const float* bc, const float* W0, const float* b0, const float* W1, const float* b1,
                   const float* W2, const float* b2, float* logits, float* h1, float* h2, float* boxes, int64_t R, int64_t D, void* stream);
int svol_heads_bwd(const float* dlogits, const float* dboxes, const float* hs, const float* h1, const float* h2, const float* boxes,
                   const float* Wc, const float* W0, const float* W1, const float* W2, float* dhs, float* gp0, float* gp1, float* dWc,
                   float* dbc, float* dW2, float* db2, int64_t R, int64_t D, void* stream);
int svol_set_loss_bwd(const float* g_label, const float* g_bbox, const float* g_giou, const float* dl, float* dlogits, float* dboxes,
                      int64_t n_layers, int64_t rows_per_layer, void* stream);
int svol_weighted_total(const float* x, const float* w, int64_t n, float* out, void* stream);
int svol_weighted_total_bwd(const float* w, const float* dout, int64_t n, float* dx, void* stream);
/* HOST entry (no device, no stream): one rectangular LSAP with scipy.optimize.linear_sum_assignment's semantics, for a
 * caller that keeps the reference's host-side matching (matcher.py:86 `C.cpu()`, :93 / :158 `linear_sum_assignment(c)`).  cost = row-major
 * [nr, nc] fp64 in HOST memory; rows / cols = caller-owned int64[min(nr, nc)] (host).  Returns the number of pairs (= min(nr, nc); rows
 * ascending, (rows[k], cols[k]) the k-th pair), -1 when the matrix holds NaN / -inf (scipy: ValueError "matrix contains invalid numeric
 * entries"; +inf is accepted), -2 when it is infeasible.  The training / evaluation path of this build does not call it — all layers'
 * problems are solved on the device by svol_lsap_batched; the two are independent implementations tested against each other. */
int svol_lsap_solve(const double* cost, int64_t nr, int64_t nc, int64_t* rows, int64_t* cols);
/* Per decoder layer (rows_per_layer = B*N prediction rows each):
 *   losses[layer*4 + {0,1,2,3}] = loss_label (weighted CE, plain mean over B*N; loss.py:54-55),
 *                                 loss_bbox (L1 mean over matched*4; loss.py:93-94),
 *                                 loss_giou (mean(1-GIoU) over matched; loss.py:96-102),
 *                                 class_error (100 - top1 acc on matched; loss.py:59)
 * and the unit gradients g_label[R,2] = d loss_label/d logits, g_bbox[R,4] = d loss_bbox/d boxes,
 * g_giou[R,4] = d loss_giou/d boxes (the backward pass scales them by the upstream loss weights).
 * rebase_vid_off (int32 [B], first global target row of each video) non-NULL reproduces
 * PerFrameMatcher's target-id re-basing (matcher.py:114-115 + loss.py:87): the loss target of a matched
 * query is row vid_off[b] + (match - min matched id of video b); rows_per_video = N.  NULL for
 * HungarianMatcher.
 * status / box_status (the outputs of svol_lsap_batched / svol_match_cost, layer-major with problems_per_layer entries
 * per layer; either may be NULL): a layer with any non-zero flag gets NaN in its four losses — where the reference
 * raises (scipy ValueError, box_utils AssertionError) this build stays asynchronous but cannot pass for a healthy step. */
int svol_set_loss(const float* logits, const float* boxes, const float* tgt_boxes, const int32_t* match,
                  float* losses, float* g_label, float* g_bbox, float* g_giou, int32_t n_layers,
                  int32_t rows_per_layer, float eos_coef, const int32_t* rebase_vid_off, int32_t rows_per_video,
                  const int32_t* status, const int32_t* box_status, int32_t problems_per_layer, void* stream);

/* ---- post-processing + evaluation (the step after the hot path: test.py:133-169, lib/evaluate/eval.py,
 * lib/evaluate/utils.py) ------------------------------------------------------------------
 * svol_postprocess: logits [B,N,2], boxes [B,N,4] (cxcywh) fp32 -> out [B,N,5] fp32 = (x0,y0,x1,y1 clamped to [0,1],
 *   foreground score softmax(logits)[...,0]); rows are grouped in consecutive chunks of `chunk` = ceil(N/num_frames)
 *   (torch.chunk, test.py:148) and sorted by score, descending and stable, inside each chunk (test.py:150-153). */
int svol_postprocess(const float* logits, const float* boxes, float* out, int64_t B, int64_t N, int64_t chunk, void* stream);
/* recall@k / mIoU@k (eval.py:72-99): one record = one (video, frame); pred_box [P,4], gt_box [G,4] xyxy fp64;
 * pred_off / gt_off [R+1] record offsets; gt_rec [G] record of each ground-truth box.  out[g] = max IoU of box g over the
 * first k predictions of its record, with the reference's pair layout (utils.py:88-96) and numpy's NaN propagation. */
int svol_eval_max_iou(const double* pred_box, const int32_t* pred_off, const double* gt_box, const int32_t* gt_off,
                      const int32_t* gt_rec, double* out, int64_t n_gt, int32_t k, void* stream);
/* AP per (video, sketch) group at every IoU threshold (utils.py:121-201).  Predictions / ground truths of a group are
 * contiguous: grp_pred_off / grp_gt_off [n_groups+1]; *_frame are integer ids of the frame keys (equal id <=> same
 * frame).  Scratch (caller-owned): ws_order int32 [n_pred]; ws_u8 bytes [K*(n_pred + n_gt)];
 * ws_f64 fp64 [2*K*(n_pred + 2*n_groups)].  ap [n_groups, K] fp64, bit-identical to the reference's numpy result. */
int svol_eval_ap(const double* pred_box, const double* pred_score, const int32_t* pred_frame, const int32_t* grp_pred_off,
                 const double* gt_box, const int32_t* gt_frame, const int32_t* grp_gt_off, const double* thresholds,
                 int32_t n_thresholds, int32_t* ws_order, unsigned char* ws_u8, double* ws_f64, double* ap, int64_t n_pred,
                 int64_t n_gt, int64_t n_groups, void* stream);

/* ---- ViT-B/16 feature extractor pieces (the Hugging Face ViTModel the reference's ViT backbone calls,
 * lib/modeling/backbone.py:30,48; inference only) ----------------------------------------------------------------
 * svol_patchify: pixel_values [n,C,H,W] fp32 -> out [n*(H/p)*(W/p), C*p*p] (dtype), the im2col of the stride-p patch
 *   convolution in the conv weight's (c, ky, kx) order: the patch embedding is then svol_gemm_nt with W [d, C*p*p]. */
int svol_patchify(const float* pixel_values, void* out, int64_t n, int64_t C, int64_t H, int64_t W, int64_t p, int dtype,
                  void* stream);
/* tokens[img, 0] = cls + pos[0]; tokens[img, 1+j] = patch_proj[img*P + j] + pos[1+j]  (patch_proj, cls, pos fp32).
 * x32 [n, P+1, D] fp32 residual stream; x (dtype, may be NULL) its compute-dtype copy. */
int svol_vit_embed(const float* patch_proj, const float* cls_token, const float* pos_embed, float* x32, void* x, int64_t n,
                   int64_t P, int64_t D, int dtype, void* stream);
/* ---- convolutional backbone pieces (torchvision ResNet-18/34, backbone.py:65-89,133-152) --------------
 * A convolution is svol_im2col followed by svol_gemm_nt (weights [Cout, kh*kw*Cin] in (ky, kx, c) order with the eval-mode
 * BatchNorm scale folded in, BatchNorm shift as bias, SVOL_ACT_RELU / SVOL_ACT_RELU_RES epilogue); activations are NHWC.
 * cols[(n,ho,wo), (ky,kx,c)] = x[n, ho*stride-pad+ky, wo*stride-pad+kx, c] (0 outside the image); columns kh*kw*C .. ldcols-1
 * are zero-filled.  x is addressed by element strides (sn, sh, sw, sc), so NCHW pixel tensors and NHWC activations both fit;
 * src_dtype / dtype: element types of x / cols (fp32 -> bf16 conversion happens here for the stem). */
int svol_im2col(const void* x, int64_t sn, int64_t sh, int64_t sw, int64_t sc, int src_dtype, void* cols, int64_t ldcols,
                int64_t N, int64_t H, int64_t W, int64_t C, int64_t kh, int64_t kw, int64_t stride, int64_t pad, int dtype,
                void* stream);
/* Convolution without the im2col matrix (bf16, C % 32 == 0): y[(n,ho,wo), co] = act(sum x[n, ho*stride-pad+ky, wo*stride-pad+kx, c]
 * * w[co, (ky*kw + kx)*C + c] + bias[co] (+ residual[(n,ho,wo), co])); x NHWC contiguous, w [Cout, >= kh*kw*C] with leading
 * dimension ldw, y / residual [N*Ho*Wo, Cout] contiguous; act in {SVOL_ACT_NONE, SVOL_ACT_RELU, SVOL_ACT_RELU_RES}.  The GEMM
 * tile kernel gathers its A operand from the activation tensor inside its LDS-DMA loads (zero padding = out-of-range buffer
 * offsets).  Returns SVOL_E_UNSUPPORTED for shapes it does not take (the caller then uses svol_im2col + svol_gemm_nt). */
int svol_conv_nhwc(const void* x, const void* w, int64_t ldw, void* y, const float* bias, int act, const void* residual,
                   int64_t N, int64_t H, int64_t W, int64_t C, int64_t Cout, int64_t kh, int64_t kw, int64_t stride,
                   int64_t pad, int dtype, void* stream);
/* nn.MaxPool2d(k, stride, pad) on NHWC activations, C % 8 == 0 (resnet stem: 3, 2, 1). */
int svol_maxpool_nhwc(const void* x, void* y, int64_t N, int64_t H, int64_t W, int64_t C, int64_t k, int64_t stride,
                      int64_t pad, int dtype, void* stream);
/* nn.AdaptiveAvgPool2d(1) on NHWC activations: y[n, c] (fp32) = mean over the HW positions. */
int svol_avgpool_nhwc(const void* x, float* y, int64_t N, int64_t HW, int64_t C, int dtype, void* stream);
/* ---- the backbone in TRAINING mode (the reference optimises every parameter of build_model(args), train.py:72; its backbone is
 * torchvision's ResNet-34 / ResNet-18 with train-mode BatchNorm, backbone.py:133-152).  Activations NHWC [M = n*h*w, C], 16-bit; C a
 * multiple of 8 (the column reductions: C / 8 a divisor of 256).  The convolutions stay GEMMs: forward svol_conv_nhwc / svol_im2col +
 * svol_gemm_nt without bias; weight gradient svol_gemm_tn(dz, svol_im2col(x)); data gradient svol_gemm_nt(dz, W^T) -> svol_col2im_nhwc.
 *   svol_bn_colstats    sum[c] += sum_m (z[m,c] - s_c), sumsq[c] += sum_m (z[m,c] - s_c)^2, s_c = shift[c] * shift_scale (shift may be NULL;
 *                       caller zeroes): pass 1 with NULL gives the sums, pass 2 with (sums, 1 / M) the second moment about the mean
 *   svol_bn_apply       y = z * scale[c] + shift[c] (+ residual) (relu != 0: max(., 0))  — nn.BatchNorm2d in train mode with
 *                       scale = gamma * rstd, shift = beta - mean * scale from the BATCH statistics
 *   svol_bn_bwd_reduce  g = dy * [y > 0] (y NULL: g = dy); sum_g[c] += sum g, sum_gx[c] += sum g * (z - mean[c]) * rstd[c]  (caller zeroes)
 *                       = the gradients of beta and gamma
 *   svol_bn_bwd_apply   dz = gamma * rstd * (g - sum_g / M - xhat * sum_gx / M); dres (may be NULL) = g (the identity branch's gradient)
 *   svol_col2im_nhwc    dx[n,iy,ix,c] = sum of dcols[(n,oy,ox), (ky,kx,c)] over the windows that hold (iy,ix): the transpose of svol_im2col
 *   svol_maxpool_idx_nhwc / svol_maxpool_bwd_nhwc   nn.MaxPool2d with the window position (ky*k + kx, one byte per output element) of the
 *                       first maximum of the (ky, kx) scan, and the backward that routes dy to it */
int svol_bn_colstats(const void* z, const float* shift, float shift_scale, float* sum, float* sumsq, int64_t M, int64_t C, int dtype,
                     void* stream);
/* the [C]-sized step between the two column passes and svol_bn_apply, one launch: mean = sum / M, rstd = (sumsq_centered / M + eps)^-1/2,
 * scale = gamma * rstd, shift = beta - mean * scale, and nn.BatchNorm2d's running update (running_* may be NULL):
 * running_mean = (1 - momentum) running_mean + momentum mean, running_var likewise with the UNBIASED batch variance.
 * pivot (may be NULL): the sums were taken in ONE pass about a per-channel pivot s (svol_bn_colstats(z, s, 1.0, ...), s = any sample of
 * the channel, e.g. the first row of z): then mean = s + sum / M and var = sumsq / M - (sum / M)^2. */
int svol_bn_finalize(const float* sum, const float* sumsq_centered, const float* pivot, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, int64_t M, int64_t C, float* mean, float* rstd,
                     float* scale, float* shift, void* stream);
/* nn.Conv2d weights fp32 [Cout, Cin, kh, kw] -> the GEMMs' 16-bit layout: flip = 0: out [Cout, Kp], column (ky*kw + kx)*Cin + c, zero
 * beyond kh*kw*Cin; flip = 1 (the data gradient of a stride-1 convolution as a convolution of dz): out [Cin, Kp], column
 * (ky*kw + kx)*Cout + co = w[co, ci, kh-1-ky, kw-1-kx].  svol_conv_weight_unpack_add: grad [Cout, Cin, kh, kw] (fp32) += dWp [Cout, Kp]
 * (fp32, the flip = 0 layout): the weight gradient of svol_gemm_tn(dz, im2col(x)) back into the parameter's layout. */
int svol_conv_weight_pack(const float* w, void* out, int64_t Cout, int64_t Cin, int64_t kh, int64_t kw, int64_t Kp, int flip, int dtype,
                          void* stream);
int svol_conv_weight_unpack_add(const float* dwp, float* grad, int64_t Cout, int64_t Cin, int64_t kh, int64_t kw, int64_t Kp, void* stream);
/* The weight gradient of a convolution without the im2col matrix: dwp [Cout, Kp] (fp32, caller zeroes; the flip = 0 layout) +=
 * dz[(n,oy,ox), co]^T * x[n, oy*stride-pad+ky, ox*stride-pad+kx, c] — the split-M weight-gradient GEMM gathers its second operand from the
 * NHWC activation inside its LDS-DMA loads (zero outside the image).  C, Cout, Kp multiples of 8; SVOL_E_UNSUPPORTED otherwise (the
 * caller then uses svol_im2col + svol_gemm_tn). */
int svol_conv_wgrad_nhwc(const void* dz, const void* x, float* dwp, int64_t N, int64_t H, int64_t W, int64_t C, int64_t Cout, int64_t kh, int64_t kw,
                         int64_t stride, int64_t pad, int64_t Kp, int dtype, void* stream);
int svol_bn_apply(const void* z, const float* scale, const float* shift, const void* residual, int relu, void* y, int64_t M, int64_t C,
                  int dtype, void* stream);
int svol_bn_bwd_reduce(const void* dy, const void* y, const void* z, const float* mean, const float* rstd, float* sum_g, float* sum_gx,
                       int64_t M, int64_t C, int dtype, void* stream);
int svol_bn_bwd_apply(const void* dy, const void* y, const void* z, const float* mean, const float* rstd, const float* gamma,
                      const float* sum_g, const float* sum_gx, void* dz, void* dres, int64_t M, int64_t C, int dtype, void* stream);
int svol_col2im_nhwc(const void* dcols, int64_t ldcols, void* dx, int64_t N, int64_t H, int64_t W, int64_t C, int64_t kh, int64_t kw,
                     int64_t stride, int64_t pad, int dtype, void* stream);
int svol_maxpool_idx_nhwc(const void* x, void* y, uint8_t* idx, int64_t N, int64_t H, int64_t W, int64_t C, int64_t k, int64_t stride,
                          int64_t pad, int dtype, void* stream);
int svol_maxpool_bwd_nhwc(const void* dy, const uint8_t* idx, void* dx, int64_t N, int64_t H, int64_t W, int64_t C, int64_t k,
                          int64_t stride, int64_t pad, int dtype, void* stream);
/* att[B,Lq,Lk] (fp32) = 1/H * sum_h softmax_l(q_h k_h^T * scale + kbias): the head-averaged attention weights that
 * nn.MultiheadAttention returns with need_weights=True and the reference's TransformerDecoder stacks per layer
 * (transformer.py:139-152, 258-262).  Recomputed from q, k (layouts as svol_attn_fwd, q_premul as there) and the
 * lse2 [B,H,Lq] that svol_attn_fwd wrote; dh in {8,16,32,64}.  No gradient. */
int svol_attn_weights_mean(const void* q, int64_t ldq, const void* k, int64_t ldk, const float* lse2, const float* kbias,
                           float* att, int64_t B, int64_t H, int64_t Lq, int64_t Lk, int64_t dh, float scale,
                           float q_premul, int dtype, void* stream);
/* softmax(q k^T * scale) v for n_seq independent SHORT sequences (L <= 256), H heads of dh = 32 or 64, bf16, forward only;
 * layouts as svol_attn_fwd (rows = tokens, head h in columns [h*dh, (h+1)*dh)). */
int svol_attn_small_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo,
                        int64_t n_seq, int64_t H, int64_t L, int64_t dh, float scale, int dtype, void* stream);


/* ---- composite block programs (round 3: the host off the critical path) ---------------------------------------------
 * One C call enqueues a whole block of a CrossModalTransformerLayer (cross_modal_transformer.py:105-160) — the same kernels,
 * in the same order, that the per-op entry points above would launch — so the Python host issues ~3 calls per layer and
 * direction instead of ~50 (a ctypes call + a torch.empty + an autograd node per kernel was 19.6 ms of host time for a 20.8 ms
 * step).  Nothing is allocated inside: every buffer (inputs, parameters, compute-dtype weight copies, tensors saved for
 * backward, scratch, gradient targets) is a caller-owned DEVICE pointer in a slot table.
 *
 *   dims  : int64[SVOL_DIM_COUNT], see SVOL_DIM_* ;  slots : void*[SVOL_xx_COUNT], see the SVOL_xx_SLOTS lists below.
 *   *_fwd  : forward.      *_bwd : the dX chain of backward (phase: 0 = all, 1 = the part in front of the large attention
 *   backward, 2 = the rest — so the caller can release queued weight-gradient work right before that launch).
 *   *_wgrad: the weight-gradient GEMMs of the block (dW += dY^T X with fp32 atomics into the gradient targets), separate so
 *   that the caller can put them on another stream, later.  Gradient targets are ACCUMULATED INTO (caller zeroes).
 * Layouts: activations [B*L, D] / [B*N, D] row-major; "dt" = element type SVOL_DIM_DTYPE of the video tokens, "qdt" =
 * SVOL_DIM_QDTYPE of the object-query stream (fp32 in the bf16 model: ops.QUERY_FP32); f32 = float whatever the dtypes.
 * Optional slots may be NULL where noted.  EV_* slots: optional hipEvent_t handles recorded around the large attention launch
 * (bench.py's live roofline figure). */
#define SVOL_DIM_B 0
#define SVOL_DIM_L 1        /* video tokens per sample */
#define SVOL_DIM_N 2        /* object queries per sample */
#define SVOL_DIM_D 3
#define SVOL_DIM_H 4
#define SVOL_DIM_F 5        /* MLP hidden width */
#define SVOL_DIM_DTYPE 6
#define SVOL_DIM_QDTYPE 7
#define SVOL_DIM_ATTN_WS_BYTES 8   /* size of the ATTN_WS slot */
#define SVOL_DIM_COUNT 9

#define SVOL_BLK_VIDEO_HALF 0
#define SVOL_BLK_QUERY_SELF 1
#define SVOL_BLK_QUERY_CROSS 2
/* comma-separated slot names of a block, in index order (the host builds its name -> index map from this) */
const char* svol_block_slot_names(int block);
/* Measurement aid.  With SVOL_BLOCK_TRACE=1 in the environment every block program records a HIP event behind each entry point it
 * calls; this synchronises the device, writes "program | call site  calls  avg_us  total_ms" lines for everything recorded since the
 * last dump into buf (NUL-terminated, truncated to cap) and forgets them.  Without the variable: writes "" and returns. */
int svol_block_trace_dump(char* buf, int64_t cap);
/* ---- measurement aid: the shader clock the chip holds while other streams work (bench.py `sclk_ghz`) ----
 * One wave samples (shader-clock ticks, wall-counter ticks) pairs in windows of period_us until *stop != 0 (the caller sets it with a
 * stream-ordered fill on another stream), max_samples windows, or max_ms (<= 60000) have passed; then writes *count.
 * GHz of a window = samples[2i] / samples[2i+1] * wall_khz / 1e6 (*wall_khz: hipDeviceAttributeWallClockRate, returned at once).
 * Not part of the hot path: the reference has no counterpart (it reads nvidia-smi by hand). */
int svol_clock_probe(uint64_t* samples, int32_t* count, int32_t max_samples, const int32_t* stop, int64_t period_us,
                     int64_t max_ms, int32_t* wall_khz, void* stream);


/* video half (:122-143): gate -> LN1 ; q|k, v projections, self-attention, out-proj + residual -> LN2 ; fc1+GELU, fc2 + residual
 * -> LN3 (+pos).  M = B*L rows.
 *  in : X32 f32 [M,D] · POS dt [M,D] · U f32 [B,H,D] (svol_gate_vectors_fwd)
 *       optional (round 6, all three NULL = the plain program): GATE_PRE non-NULL = GATE_WS already holds this layer's scores (the
 *       layer before wrote them through ITS GATE_WS_NEXT: the caller points both layers' slots at one buffer) · U_NEXT f32 [B,H,D]
 *       + GATE_WS_NEXT f32 [B*H*(L+2)]: LN3 also writes the NEXT layer's gate scores (svol_layernorm_gate_scores_fwd; L % 4 == 0, D <= 256)
 *  par: G1,BT1,G2,BT2,G3,BT3 f32 [D] (norm weight, bias) · B_IN f32 [3D] · B_O [D] · B_FC1 [F] · B_FC2 [D] · QSCALE f32 [2D] or NULL
 *  w  : W_IN dt [3D,D] · WV_HILO bf16 [D,2D] or NULL · W_O dt [D,D] · W_FC1 dt [F,D] · W_FC2 dt [D,F] and the transposes
 *       W_IN_T [D,3D] · W_O_T · W_FC1_T [D,F] · W_FC2_T [F,D]
 *  sav: Y1, Y1POS dt [M,D] · A, MEAN1, RSTD1 f32 [M] · GATE_WS f32 [B*H*(L+2)] · QKV dt [M,3D] · O dt [M,D] · LSE f32 [B,H,L] ·
 *       S2 f32 [M,D] · Y2 dt [M,D] · MEAN2, RSTD2 · PRE (= gelu'(fc1 output), SVOL_ACT_GELU_D), HID dt [M,F] · S3 f32 [M,D] · MEAN3, RSTD3
 *  out: M32 f32, M dt, MPOS dt [M,D]          scr: Y1_32, Y2_32 f32 [M,D] · ATTN_WS
 *  bwd in : DM32 f32, DM dt, DMPOS dt (each may be NULL, not all)
 *  bwd tmp: DS32_3 f32, DS3 dt [M,D] · DPRE dt [M,F] · DY2 dt · DS32_2 f32 · G2D dt · DO dt [M,D] · DQKV dt [M,3D] ·
 *           DELTA f32 [3,B,H,L] · DXQP, DXQ dt [M,D] · GATE_WS2 f32 [B*L + B*H]
 *  bwd out: DX32 f32 [M,D] · DU f32 [B,H,D] (caller zeroes)
 *  grads  : DG1,DBT1,DG2,DBT2,DG3,DBT3 · DW_IN [3D,D], DB_IN · DW_O, DB_O · DW_FC1, DB_FC1 · DW_FC2, DB_FC2 (f32, accumulated)
 *  backward, optional (all NULL = svol_attn_bwd as before): EV_CLEAN_IN hipEvent_t = ATTN_WS's dQ image was zeroed on another stream
 *       (the program waits for it and passes SVOL_ATTN_DQ_PREZEROED) · EV_PREP hipEvent_t + ATTN_WS_NEXT + ZERO_STREAM hipStream_t +
 *       EV_CLEAN_OUT hipEvent_t: behind the attention backward's prologue the program zeroes ATTN_WS_NEXT's image on ZERO_STREAM
 *       (svol_attn_bwd_zero_ws) and records EV_CLEAN_OUT there — the caller hands that buffer / event to the next layer's backward */
#define SVOL_VH_SLOTS(X) \
    X(X32) X(POS) X(U) X(G1) X(BT1) X(G2) X(BT2) X(G3) X(BT3) X(B_IN) X(B_O) X(B_FC1) X(B_FC2) X(QSCALE) \
    X(W_IN) X(WV_HILO) X(W_O) X(W_FC1) X(W_FC2) X(W_IN_T) X(W_O_T) X(W_FC1_T) X(W_FC2_T) \
    X(Y1) X(Y1POS) X(A) X(MEAN1) X(RSTD1) X(GATE_WS) X(GATE_PRE) X(U_NEXT) X(GATE_WS_NEXT) \
    X(QKV) X(O) X(LSE) X(S2) X(Y2) X(MEAN2) X(RSTD2) X(PRE) X(HID) X(S3) \
    X(MEAN3) X(RSTD3) X(M32) X(M) X(MPOS) X(Y1_32) X(Y2_32) X(ATTN_WS) \
    X(DM32) X(DM) X(DMPOS) X(DS32_3) X(DS3) X(DPRE) X(DY2) X(DS32_2) X(G2D) X(DO) X(DQKV) X(DELTA) X(DXQP) X(DXQ) X(GATE_WS2) \
    X(DX32) X(DU) \
    X(DG1) X(DBT1) X(DG2) X(DBT2) X(DG3) X(DBT3) X(DW_IN) X(DB_IN) X(DW_O) X(DB_O) X(DW_FC1) X(DB_FC1) X(DW_FC2) X(DB_FC2) \
    X(EV_A0) X(EV_A1) X(EV_CLEAN_IN) X(EV_PREP) X(ATTN_WS_NEXT) X(ZERO_STREAM) X(EV_CLEAN_OUT)
/* query self-attention (:145-147): packed q|k, v projections of the N queries, self-attention, out-proj + residual -> LN4 (+query_pos).
 * R = B*N rows, everything in qdt.
 *  in : O32 f32, O qdt, OPOS qdt [R,D] · QPOS qdt [N,D]
 *  par: B_IN f32 [3D], B_O, G4, BT4 · QSCALE or NULL        w: W_IN qdt [3D,D], WV_HILO (bf16 queries only) or NULL, W_O, W_IN_T, W_O_T
 *  sav: QKV qdt [R,3D] · OA qdt [R,D] · LSE f32 [B,H,N] · S4 f32 [R,D] · MEAN4, RSTD4 [R]      out: Y32 f32, Y qdt, YPOS qdt [R,D]
 *  scr: ATTN_WS       bwd in: DY32, DY, DYPOS (each may be NULL)
 *  bwd tmp: G qdt [R,D] · DOA qdt · DQKV qdt [R,3D] · DELTA f32 [3,B,H,N]
 *  bwd out: DO32 f32 [R,D] (residual path) · DXQ qdt (through v) · DXQP qdt (through q, k) · DQPOS f32 [N,D] or NULL (caller zeroes)
 *  grads  : DW_IN, DB_IN, DW_O, DB_O, DG4, DBT4 */
#define SVOL_QS_SLOTS(X) \
    X(O32) X(O) X(OPOS) X(QPOS) X(B_IN) X(B_O) X(G4) X(BT4) X(QSCALE) X(W_IN) X(WV_HILO) X(W_O) X(W_IN_T) X(W_O_T) \
    X(QKV) X(OA) X(LSE) X(S4) X(MEAN4) X(RSTD4) X(Y32) X(Y) X(YPOS) X(ATTN_WS) \
    X(DY32) X(DY) X(DYPOS) X(G) X(DOA) X(DQKV) X(DELTA) X(DO32) X(DXQ) X(DXQP) X(DQPOS) \
    X(DW_IN) X(DB_IN) X(DW_O) X(DB_O) X(DG4) X(DBT4)
/* query -> video cross-attention + MLP2 (:151-158): q projection of the queries (qdt), k / v projections of the L video tokens
 * (dt; v with split weights), attention with the additive key mask, out-proj + residual -> LN5 ; fc1+GELU, fc2 + residual ->
 * LN6 (+query_pos).  R = B*N, M = B*L.  qdt != dt ("mixed": fp32 queries over bf16 video tokens): q and the attention output
 * cross the boundary through one cast each.
 *  in : O32 f32, O qdt, OPOS qdt [R,D] · MV dt [M,D] (values from) · MPOS dt [M,D] (keys from) · KBIAS f32 [B,L] · QPOS qdt [N,D]
 *  par: B_IN f32 [3D] · B_O · G5, BT5 · B_FC1 [F] · B_FC2 · G6, BT6 · QSCALE f32 [>= D] or NULL
 *  w  : W_INQ qdt [3D,D] (rows 0..D-1 used) · W_INQ_T qdt [D,3D] · W_KV dt [3D,D] (rows D.. used) · W_KV_T dt [D,3D] · WV_HILO or NULL ·
 *       W_O, W_O_T qdt [D,D] · W_FC1 qdt [F,D] · W_FC1_T [D,F] · W_FC2 qdt [D,F] · W_FC2_T [F,D]
 *  sav: Q qdt [R,D] (mixed only) · QC dt [R,D] · KV dt [M,2D] · OA dt [R,D] · OAQ qdt [R,D] (mixed only) · LSE f32 [B,H,N] · S5 f32 ·
 *       Y5_32 f32, Y5 qdt · MEAN5, RSTD5 · PRE, HID qdt [R,F] · S6 f32 · MEAN6, RSTD6        out: Y32 f32, Y qdt, YPOS qdt
 *  scr: ATTN_WS       bwd in: DY32, DY, DYPOS
 *  bwd tmp: DS32_6 f32, DS6 qdt [R,D] · DPRE qdt [R,F] · DY5 qdt · G5D qdt · DOAQ qdt · DOA dt · DQC dt [R,D] · DQ qdt (mixed only) ·
 *           DKV dt [M,2D] · DELTA f32 [3,B,H,N]
 *  bwd out: DO32 f32 [R,D] · DXQP qdt [R,D] · DMPOS dt [M,D] (through k) · DMV dt [M,D] (through v) · DQPOS f32 [N,D] or NULL
 *  grads  : DW_IN [3D,D], DB_IN · DW_O, DB_O · DG5, DBT5 · DW_FC1, DB_FC1 · DW_FC2, DB_FC2 · DG6, DBT6 */
#define SVOL_QC_SLOTS(X) \
    X(O32) X(O) X(OPOS) X(MV) X(MPOS) X(KBIAS) X(QPOS) X(B_IN) X(B_O) X(G5) X(BT5) X(B_FC1) X(B_FC2) X(G6) X(BT6) X(QSCALE) \
    X(W_INQ) X(W_INQ_T) X(W_KV) X(W_KV_T) X(WV_HILO) X(W_O) X(W_O_T) X(W_FC1) X(W_FC1_T) X(W_FC2) X(W_FC2_T) \
    X(Q) X(QC) X(KV) X(OA) X(OAQ) X(LSE) X(S5) X(Y5_32) X(Y5) X(MEAN5) X(RSTD5) X(PRE) X(HID) X(S6) X(MEAN6) X(RSTD6) \
    X(Y32) X(Y) X(YPOS) X(ATTN_WS) \
    X(DY32) X(DY) X(DYPOS) X(DS32_6) X(DS6) X(DPRE) X(DY5) X(DS32_5) X(G5D) X(DOAQ) X(DOA) X(DQC) X(DQ) X(DKV) X(DELTA) \
    X(DO32) X(DXQP) X(DMPOS) X(DMV) X(DQPOS) \
    X(DW_IN) X(DB_IN) X(DW_O) X(DB_O) X(DG5) X(DBT5) X(DW_FC1) X(DB_FC1) X(DW_FC2) X(DB_FC2) X(DG6) X(DBT6)
#define SVOL_SLOT_ENUM_VH(n) SVOL_VH_##n,
#define SVOL_SLOT_ENUM_QS(n) SVOL_QS_##n,
#define SVOL_SLOT_ENUM_QC(n) SVOL_QC_##n,
enum { SVOL_VH_SLOTS(SVOL_SLOT_ENUM_VH) SVOL_VH_COUNT };
enum { SVOL_QS_SLOTS(SVOL_SLOT_ENUM_QS) SVOL_QS_COUNT };
enum { SVOL_QC_SLOTS(SVOL_SLOT_ENUM_QC) SVOL_QC_COUNT };

int svol_video_half_fwd(const int64_t* dims, void* const* slots, void* stream);
int svol_video_half_bwd(const int64_t* dims, void* const* slots, int phase, void* stream);
int svol_video_half_wgrad(const int64_t* dims, void* const* slots, void* stream);
/* the same in two launches: part 1 = fc2 / fc1 / out-proj (operands exist after svol_video_half_bwd phase 1), part 2 = in_proj
 * (after phase 2); part 0 = all five */
int svol_video_half_wgrad_part(const int64_t* dims, void* const* slots, int part, void* stream);
int svol_query_self_fwd(const int64_t* dims, void* const* slots, void* stream);
int svol_query_self_bwd(const int64_t* dims, void* const* slots, void* stream);
int svol_query_self_wgrad(const int64_t* dims, void* const* slots, void* stream);
int svol_query_cross_fwd(const int64_t* dims, void* const* slots, void* stream);
int svol_query_cross_bwd(const int64_t* dims, void* const* slots, void* stream);
int svol_query_cross_wgrad(const int64_t* dims, void* const* slots, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SVOL_HIP_H */

/*
 * rlt_hip.h - C ABI of librlt_hip.so: the MI355X (gfx950) implementation of the
 * ranked-list-truncation forward/backward hot path.
 *
 * The reference (Woody5962/Ranked-List-Truncation) has no FFI: its hot path sits behind Python
 * classes (models/__init__.py:1-12, utils/losses.py, utils/metrics.py).  This header is the
 * boundary those classes are re-implemented on: every entry point below replaces the arithmetic
 * of the reference lines it cites, and the Python mirror of the reference's classes
 * (ranked-list-truncation_amd/{models,utils}) binds exactly these symbols through ctypes.
 *
 * Conventions
 *   - all tensors are fp32 device pointers owned by the caller (labels are fp32 0/1 like the
 *     reference's), contiguous unless a leading dimension is given; nothing is allocated or
 *     freed here, workspaces are passed in; no host synchronisation, everything is ordered on
 *     `stream` (a hipStream_t passed as void*; NULL = the default stream);
 *   - return value: 0 ok; <0 argument error (RLT_E_*); >0 a hipError_t from the launch;
 *   - "position-major" activations: token row index t = s*B + b (position s outermost, list b
 *     innermost), i.e. a (S, B, E) tensor.  The reference keeps (B, S, E) and lets
 *     nn.TransformerEncoderLayer(batch_first=False) attend over axis 0 = the B lists at each
 *     position (models/AttnCut.py:9,17-18); position-major makes that attention axis contiguous.
 *     User-facing inputs/outputs stay in the reference's (B, S, F) / (B, S, 1) layout.
 */
#ifndef RLT_HIP_H
#define RLT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLT_E_ARG      (-1)   /* null pointer / non-positive dimension            */
#define RLT_E_SHAPE    (-2)   /* dimension outside what the kernels support        */
#define RLT_E_WORKSPACE (-3)  /* workspace too small                               */
#define RLT_E_ALIGN    (-4)   /* pointer / leading dimension not 16-byte aligned   */

#define RLT_ABI_VERSION 5
int rlt_abi_version(void);
/* Precision of the MFMA contractions (the rlt_gemm* family, rlt_list_attention_*, the BiLSTM recurrences); inputs, outputs,
 * softmax, LayerNorm, gate nonlinearities, losses and every accumulator are fp32 in all modes.
 *   RLT_PRECISION_FP32   exact fp32 products on the f32 MFMA (157 TFLOP/s peak) - bit-for-bit fp32 fma chains
 *   RLT_PRECISION_BF16X6 fp32-FAITHFUL products on the bf16 MFMA: every operand split exactly into three bf16 values
 *                        (8 + 8 + 8 significand bits: all 24 operand bits enter), six of the nine partial products kept,
 *                        each exact in the matrix pipe, fp32 accumulation; what is dropped is < 2^-23 of the product
 *                        in the worst case (under one fp32 ulp; 2^-29 typical).  Held to the FP32 mode's tolerances in
 *                        the tests.  Measured against fp64 (tools/gpu_probe.py x6_adversarial, profiles/r05_notes.md): at
 *                        or below the f32 MFMA kernels' error on random operands and on the adversarial classes `ones`
 *                        (low significand bits all ones, one sign) and `cancel` (sums cancelling by 1e4); on operands
 *                        that all share the SAME worst-case low bits (`worst-split`: 0x7F40, one sign) the roundings of
 *                        the small plane products into one running sum add coherently - 17x the f32 MFMA kernels' error
 *                        in a K ~ 10^6 contraction of the GEMM family, always inside the a-priori bound K 2^-24 of an
 *                        fp32 chain and two orders inside BASELINE.json's 1e-4; list attention keeps those products in an
 *                        accumulator of their own (head dim 16: all kernels; head dim 64: the forward at 512 lists and
 *                        more) and stays at or below the f32 kernels on that class too, its head-dim-64 gradients within
 *                        2x on every class (asserted; 4-8x until ABI 4).  Head dim 128 and shapes off the tile grid
 *                        run the exact-fp32 kernels in this mode.  THE DEFAULT: the reference computes in fp32 end to
 *                        end (models/AttnCut.py:8-14).
 *   RLT_PRECISION_BF16X3 opt-in fast mode: every operand split into bf16 hi + bf16 lo, a*b = hi*hi + hi*lo + lo*hi
 *                        (16 operand bits, ~2^-16 relative error per product, ~2x faster end to end; inside
 *                        BASELINE.json's 1e-4 bound but narrower than the reference's arithmetic)
 * Every entry point whose arithmetic or buffer layout depends on the mode takes an `int precision` argument: one of the
 * three codes above for THAT call, or RLT_PRECISION_DEFAULT = the process default (environment variable
 * RLT_PRECISION=fp32|bf16x6|bf16x3 at first use, else BF16X6; rlt_set_precision changes it and does nothing else).
 * The calls are re-entrant: two models in one process - or two threads - may run different modes side by side.  A
 * backward call and the workspace queries of a forward / backward pair must be given the forward call's precision (the
 * stash and workspace layouts depend on it).  Any other code: RLT_E_ARG (workspace queries: 0 bytes). */
#define RLT_PRECISION_DEFAULT (-1)
#define RLT_PRECISION_FP32   0
#define RLT_PRECISION_BF16X3 1
#define RLT_PRECISION_BF16X6 2
int rlt_set_precision(int mode);    /* sets the process default (not RLT_PRECISION_DEFAULT) */
int rlt_get_precision(void);        /* the process default */
/* human-readable name of an RLT_E_* / hipError_t code (static storage) */
const char* rlt_error_string(int code);

/* ------------------------------------------------------------------ reward losses (L1-L6)
 * utils/losses.py:48-68 (ChoopyLoss), :71-96 (AttnCutLoss), :194-233 (DivLoss) with the
 * reward matrix of :57-65/:81-89/:217-225 built from utils/metrics.py:85-101 in closed form
 * (prefix sums), fused: one pass over p and labels yields the per-list loss terms and
 * d(loss)/dp.
 *   p, labels: (B,S).  metric: RLT_METRIC_*.  kind: RLT_LOSS_*.  tau: reward temperature
 *   (DivLoss: 0.85 augmented / 1.0; AttnCutLoss: 0.95; ignored for EXPECT).
 *   dcg_coef: S floats log2(j+2) (utils/metrics.py:7), required for RLT_METRIC_DCG.
 *   loss_per_list: (B) un-normalised per-list terms; *loss_out = sum(loss_per_list)/B
 *   (KLDivLoss 'batchmean' / the reference's .div(B)).
 *   dp: (B,S) or NULL; receives d(loss_out)/dp (already divided by B).
 *   S <= 1024.
 */
#define RLT_METRIC_F1  0
#define RLT_METRIC_DCG 1
#define RLT_LOSS_EXPECT 0   /* ChoopyLoss:  -E_p[r]                 */
#define RLT_LOSS_CE     1   /* AttnCutLoss: -sum q log p            */
#define RLT_LOSS_KL     2   /* DivLoss kl:  KL(q || p)              */
#define RLT_LOSS_JS     3   /* DivLoss js:  JS(q, p)                */
int rlt_reward_loss(const float* p, const float* labels, const float* dcg_coef, int B, int S,
                    int metric, int kind, float tau,
                    float* loss_per_list, float* loss_out, float* dp, void* stream);
/* the same with the `penalty` argument of Metric_for_Loss.dcg (utils/metrics.py:94): the DCG gain of a non-relevant
 * document is penalty / log2(j+2); rlt_reward_loss is this call with the reference's default -1.  Ignored for F1. */
int rlt_reward_loss_ex(const float* p, const float* labels, const float* dcg_coef, int B, int S,
                       int metric, float penalty, int kind, float tau,
                       float* loss_per_list, float* loss_out, float* dp, void* stream);
/* reward matrix r (B,S) and its distribution q = softmax(r/tau) (either may be NULL):
 * the B*S python loop of utils/losses.py:217-228 on its own (tests, plots). */
int rlt_reward_matrix(const float* labels, const float* dcg_coef, int B, int S, int metric, float tau,
                      float* r_out, float* q_out, void* stream);
int rlt_reward_matrix_ex(const float* labels, const float* dcg_coef, int B, int S, int metric, float penalty, float tau,
                         float* r_out, float* q_out, void* stream);
/* Training-step form (run.py:126 + :141-145 in ONE pass over p and labels): everything rlt_reward_loss_ex produces, plus
 * the cut metrics of rlt_cut_metrics_ex on the same rows while they are in registers - k_out (B) int32 = argmax_j p + 1 (first
 * maximum), f1_out / dcg_out (B) float64 (DCG with `metric_penalty`, utils/metrics.py:27), sums[0..1] = their batch sums.
 * loss_out = (sum of the per-list terms)/B from a float64 sum.  All outputs required except dp.  Two launches: the pass
 * (two ranked lists per wavefront - one per 32-lane half - when S % 4 == 0 and S <= 384, one per wavefront otherwise; a grid
 * sized to the chip striding over the lists; every wavefront leaves its float64 partial sums in ws) and a one-workgroup
 * fixed-order reduction of those partials (deterministic).  The float64 DCG coefficients 1/log2(j+2) (utils/metrics.py:7) and
 * their prefix sums are read from `dcg_table`, CALLER memory of rlt_dcg_table_bytes() bytes (8-byte aligned) that
 * rlt_dcg_table_init has filled once (one small launch on `stream`; the table does not depend on B, S or the metric and may be
 * shared by every later call on that device): like every entry point of this library the call allocates nothing, never
 * synchronises with the host and may be captured into a graph.  (ABI 3 built the table itself on first use: a hidden
 * hipMalloc + synchronous copy.)
 * ws: rlt_loss_metrics_workspace(B) bytes.
 * Algorithmic bytes per list: read p, labels 8S, write dp 4S + 24 B of results (3.6 KB at S = 300). */
size_t rlt_dcg_table_bytes(void);
int rlt_dcg_table_init(void* table, size_t table_bytes, void* stream);
size_t rlt_loss_metrics_workspace(int B);
int rlt_loss_metrics(const float* p, const float* labels, const float* dcg_coef, int B, int S,
                     int metric, float penalty, int kind, float tau, double metric_penalty,
                     float* loss_per_list, float* loss_out, float* dp,
                     int32_t* k_out, double* f1_out, double* dcg_out, double* sums,
                     const void* dcg_table, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ multi-task terms (L7-L8)
 * utils/losses.py:99-141 (RerankLoss) and nn.BCELoss of :177,:187 (MtCutLoss).
 * rlt_mt_terms: one pass over the rerank scores and/or class probabilities (either may be NULL)
 * accumulating the batch-wide sums; then finalises on device (no host sync):
 *   terms[0] = rerank hinge  max(0, mean_{y==0} s - mean_{y==1} s + margin), 0 if a class is empty
 *   terms[1] = BCE mean over B*S (log clamped at -100 like torch)
 *   terms[2] = d hinge / d s for y==1 entries (= -1/n_pos if active else 0)
 *   terms[3] = d hinge / d s for y==0 entries (= +1/n_neg if active else 0)
 * ws: at least rlt_mt_terms_workspace(B,S) bytes.
 * rlt_mt_terms_bwd: d_rerank (B,S) = w_r * terms[2|3]; d_class (B,S) = w_c * dBCE/dc / (B*S);
 * both scaled by *gscale (device scalar, NULL = 1).
 */
size_t rlt_mt_terms_workspace(int B, int S);
int rlt_mt_terms(const float* rerank, const float* cls, const float* labels, int B, int S, float margin,
                 float* terms, void* ws, size_t ws_bytes, void* stream);
int rlt_mt_terms_bwd(const float* cls, const float* labels, const float* terms, int B, int S,
                     float w_rerank, float w_class, const float* gscale,
                     float* d_rerank, float* d_class, void* stream);
/* out[0] = sum_i w[i] * x[i] for n <= 8 device scalars (combining cut / rerank / class terms) */
int rlt_weighted_sum(const float* const* x, const float* w, int n, float* out, void* stream);

/* ------------------------------------------------------------------ cut metrics (E1-E3)
 * run.py:137-142 (k = argmax+1) and utils/metrics.py:15-38 (Metric.f1 / Metric.dcg at k).
 * p, labels: (B,S).  k_out (B) int32; f1_out, dcg_out (B) float64 per list (the reference
 * averages them over the batch on the host); any output may be NULL.
 * If k_in != NULL the metrics are evaluated at those cut positions instead of the argmax.
 * sums (2 doubles, may be NULL) receives sum_i f1_i, sum_i dcg_i.
 */
int rlt_cut_metrics(const float* p, const float* labels, const int32_t* k_in, int B, int S,
                    int32_t* k_out, double* f1_out, double* dcg_out, double* sums, void* stream);
/* the same with Metric.dcg's `penalty` argument (utils/metrics.py:27; rlt_cut_metrics uses the default -1) */
int rlt_cut_metrics_ex(const float* p, const float* labels, const int32_t* k_in, int B, int S, double penalty,
                       int32_t* k_out, double* f1_out, double* dcg_out, double* sums, void* stream);

/* Task metrics of utils/metrics.py:40-76 (section 8f row N4), per list in float64:
 *   dcg_out[b] = taskr_metric's DCG of list b re-ordered by descending pred (relevant +1/log2(i+2), else -1/log2(i+2));
 *   auc_out[b] = taskc_metric's ROC AUC of pred against labels (ties 1/2), or -1 for a list with a single class (the
 *                reference skips those);
 *   sums (3 doubles, may be NULL) = sum of dcg_out, sum of the valid auc_out, number of valid lists.
 * labels, pred (B,S), S <= 1024. */
int rlt_task_metrics(const float* labels, const float* pred, int B, int S,
                     double* dcg_out, double* auc_out, double* sums, void* stream);

/* ------------------------------------------------------------------ dense contraction (M2-M7)
 * C[M,N] (+)= op(A) * op(B) (+ bias[N] + bias2[N]), optional ReLU.  fp32 in, fp32 accumulate on
 * the f32 MFMA (exact fp32 products, v_mfma_f32_32x32x2_f32).
 *   ta = 0: A stored [M,K] (lda >= K);  ta = 1: A stored [K,M] (lda >= M)
 *   tb = 0: B stored [K,N] (ldb >= N);  tb = 1: B stored [N,K] (ldb >= K)   (nn.Linear: tb = 1)
 *   flags: RLT_GEMM_RELU, RLT_GEMM_ACCUMULATE (C += ...).
 * Replaces the nn.Linear / in_proj / out_proj / LSTM input-projection matmuls of
 * models/AttnCut.py:8-14 and their backward.  ws: rlt_gemm_workspace(...) bytes (split-K partials).
 */
#define RLT_GEMM_RELU       1
#define RLT_GEMM_ACCUMULATE 2
size_t rlt_gemm_workspace(int ta, int tb, int M, int N, int K);
int rlt_gemm(int ta, int tb, int M, int N, int K,
             const float* A, int lda, const float* B, int ldb, float* C, int ldc,
             const float* bias, const float* bias2, int flags,
             void* ws, size_t ws_bytes, int precision, void* stream);
/* rlt_gemm plus two fused side products of the backward pass:
 *   relu_mask (M x N, ldmask) : C = relu_mask > 0 ? C * mask_scale : 0 after the epilogue
 *                               (dH = (dY W2) * (H > 0) [/ (1-p) when the forward dropped H])
 *   drop_p, seed              : dropout on the output after bias/ReLU: keep(seed,row,col) ? C/(1-p) : 0
 *                               (the FFN hidden dropout of nn.TransformerEncoderLayer), 0 = off
 *   colsum_a  (M)             : = sum_k op(A)[m][k]; requires ta = 1.  With A = dY stored [T,N_out] this is
 *                               the bias gradient, produced by the dW = dY^T X product at no extra HBM pass. */
int rlt_gemm_ex(int ta, int tb, int M, int N, int K,
                const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                const float* bias, const float* bias2, int flags,
                const float* relu_mask, int ldmask, float mask_scale, float* colsum_a,
                float drop_p, uint32_t seed,
                void* ws, size_t ws_bytes, int precision, void* stream);

/* The FFN pair of rlt_gemm_ex epilogues with a 1-bit-per-element mask instead of the fp32 activation
 * (nn.TransformerEncoderLayer's linear1 -> ReLU -> dropout forward and the dH = (dY W2) * mask backward):
 *   relu_bits_out != NULL (flags must hold RLT_GEMM_RELU): C = dropout(relu(op(A) op(B) + bias)) with the keep mask of
 *     (drop_p, seed) as in rlt_gemm_ex (drop_p = 0: no dropout), and the mask bit of element (row, col) = (C > 0), i.e.
 *     the element passed the ReLU and was kept;
 *   mask_bits_in  != NULL: C = bit ? (op(A) op(B) + bias) * mask_scale : 0   (mask_scale = 1/(1-p); drop_p must be 0).
 * Mask layout (private to this pair of calls): packed along ROWS - bit (row & 31) of word [(row >> 5) * N + col];
 * rlt_gemm_bits_words(M, N) = ceil(M/32) * N words.  Exactly one of the two pointers; N % 32 == 0; no split-K, no
 * workspace.  The backward product then reads M*N/8 bytes of mask instead of 4*M*N (10 GB at 1,228,800 x 2048). */
size_t rlt_gemm_bits_words(int M, int N);
int rlt_gemm_bits(int ta, int tb, int M, int N, int K,
                  const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                  const float* bias, int flags, float drop_p, uint32_t seed,
                  uint32_t* relu_bits_out, const uint32_t* mask_bits_in, float mask_scale,
                  int precision, void* stream);
/* out[N] (+)= sum over the T rows of X[T,N] (ldx) - bias gradients.  ws: rlt_colsum_workspace bytes. */
size_t rlt_colsum_workspace(int T, int N);
int rlt_colsum(const float* X, int ldx, int T, int N, float* out, int accumulate,
               void* ws, size_t ws_bytes, void* stream);
/* out[g][n] (+)= sum_{r<R} X[(g*R + r)*ldx + n] for g < G: per-position sums (Choopy dPE) */
int rlt_segment_colsum(const float* X, int ldx, int G, int R, int N, float* out, int ldo,
                       int accumulate, void* stream);

/* Narrow weight gradient of an input projection: dW[M][I] = A^T X and db[M] = column sums of A, for 1 <= I <= 3
 * (the LSTM layer-0 input weights, input_size = 3, models/AttnCut.py:8; replaces the N = 3 case of the dW GEMM and
 * its bias-gradient side sum with ONE streaming pass over A for both directions).  A (T, lda >= M, M % 4 == 0),
 * X (T, ldx >= I); db may be NULL. */
size_t rlt_narrow_dw_workspace(int T, int M);
int rlt_narrow_dw(const float* A, int lda, const float* X, int ldx, int I, int T, int M,
                  float* dW, float* db, void* ws, size_t ws_bytes, void* stream);
/* y = max(x,0) backward etc. are fused in the kernels; dX *= (Y > 0) in place (FFN backward) */
int rlt_relu_bwd(float* dX, const float* Y, size_t n, void* stream);
/* x[i] *= *scale (device scalar) */
int rlt_scale(float* x, const float* scale, size_t n, void* stream);

/* ------------------------------------------------------------------ residual + LayerNorm (M3)
 * y = LayerNorm(x + r) * gamma + beta, eps inside the sqrt, biased variance
 * (nn.TransformerEncoderLayer post-norm, models/AttnCut.py:9).  stats: (T,2) mean, rstd.
 * bwd: dz (T,E) = gradient w.r.t. (x + r) (identical for x and r); dgamma/dbeta (E) (+)=.
 * E multiple of 64, E <= 1024.
 * Dropout (train mode, dropout1/dropout2 of the encoder layer): y = LN(x + drop(r)) with
 * drop(r) = keep(seed,t,c) ? r/(1-p) : 0; bwd then also writes dr = dz * keep/(1-p) (dr may be NULL
 * when drop_p == 0: the gradient of r is dz itself).
 */
int rlt_add_layernorm_fwd(const float* x, const float* r, const float* gamma, const float* beta,
                          int T, int E, float eps, float drop_p, uint32_t seed,
                          float* y, float* stats, void* stream);
size_t rlt_add_layernorm_bwd_workspace(int T, int E);
int rlt_add_layernorm_bwd(const float* x, const float* r, const float* gamma, const float* stats,
                          const float* dy, int T, int E, float drop_p, uint32_t seed,
                          float* dz, float* dr, float* dgamma, float* dbeta,
                          int accumulate, void* ws, size_t ws_bytes, void* stream);
/* the dropout keep-mask of the kernels above as data (tests): out[r][c] = keep(seed,r,c) ? 1/(1-p) : 0 */
int rlt_dropout_mask(uint32_t seed, size_t rows, int cols, float p, float* out, void* stream);

/* ------------------------------------------------------------------ list-axis attention (M3)
 * Multi-head self-attention where, at every position s and head h, the B lists of the
 * mini-batch attend to each other (F.multi_head_attention_forward on a (L=B, N=S, E) input,
 * models/AttnCut.py:9,18).  Flash-style: the B x B score matrix is never
 * materialised.
 *   qkv: (S*B, 3E) position-major, columns [q | k | v], head h = columns h*HD..h*HD+HD of each.
 *   out: (S*B, E) concatenated heads (input of out_proj).  lse: (S,H,B) log-sum-exp of the
 *   scaled scores (saved for backward).  scale = 1/sqrt(HD).
 *   drop_p, seed: dropout on the attention probabilities (train mode, the `dropout` of
 *   nn.MultiheadAttention); the mask is a pure function of (seed, s, h, query, key), recomputed in backward.
 * bwd: dqkv (S*B,3E) from dout; ws: rlt_list_attention_bwd_workspace bytes.
 * HD in {16, 32, 64}; E = H*HD.
 */
/* In the split-bf16 mode the forward first writes Q (pre-scaled), K and V as pre-split bf16 hi/lo tile
 * records (64-row tiles in the kernels' LDS layout) into `images` (rlt_list_attention_fwd_workspace bytes; 0 in
 * fp32 mode, then `images` may be NULL); the caller keeps `images` for the backward pass.  The records are written
 * for the dropout rate of THIS call (without dropout the head-dim-64 backward kernels read the transposed operands
 * straight from the row images and the transposed images of Q and K are not written): the backward entry points must
 * be given the same drop_p and seed as the forward call whose `images` they use. */
/* BF16X6, 512 lists and more in whole tiles (head dim 64, with or without dropout: csrc/attention6h.hip; head dim 16 without dropout:
 * csrc/attention6n.hip):
 * the pipelined forward kernels stage pre-split K / V tile images that the call itself writes into `images` (+ a flag word per
 * workgroup for its fix-up launch); the backward pass does not read them - rlt_list_attention_images_retained() tells whether the
 * caller has to keep `images` for the backward (1) or may treat it as scratch of the forward call (0).  `drop_p` of the workspace
 * query = the drop_p of the forward call (at head dim 16 a train-mode call has no pipelined form and needs no such buffer). */
size_t rlt_list_attention_fwd_workspace(int S, int B, int H, int HD, float drop_p, int precision);
int rlt_list_attention_images_retained(int S, int B, int H, int HD, int precision);
int rlt_list_attention_fwd(const float* qkv, int S, int B, int H, int HD, float drop_p, uint32_t seed,
                           float* out, float* lse, void* images, size_t images_bytes, int precision, void* stream);
/* backward: ws = [delta (S,H,B) | tile records], rlt_list_attention_bwd_workspace bytes.
 *   _bwd_prepare: delta = rowsum(dout*out) and (split-bf16 mode) the dO records;
 *   _bwd_dkv:     dK, dV columns of dqkv;   _bwd_dq: dQ columns of dqkv;
 *   _bwd:         the three in sequence.  `images` = the forward's buffer (NULL => exact-fp32 kernels).
 * BF16X6 at head dim 16 with 512 lists and more (Choopy / MtChoopy at scale; csrc/attention6n.hip): the backward kernels stage
 * pre-split 128-row tile images of Q, K, V, dO and the rows' seeds (-lse, -delta) from ws - _bwd_prepare writes the dO images and
 * the seeds, _bwd_dkv the Q images, _bwd_dq the K and V images, each before its kernel (so the three entry points stay callable
 * on their own; ws is scratch of ONE backward pass: the parts of a pass run on one stream, in any order after _bwd_prepare). */
/* `drop_p` of the workspace query and of _bwd_prepare = the drop_p of the forward call (train-mode kernels stage their tiles
 * themselves: ws is delta only).  The parts check ws_bytes against the same rule in their OWN precision scope and ws for 16-byte
 * alignment: RLT_E_WORKSPACE / RLT_E_ALIGN instead of a device write beyond a buffer sized under another mode. */
size_t rlt_list_attention_bwd_workspace(int S, int B, int H, int HD, float drop_p, int precision);
int rlt_list_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse,
                           int S, int B, int H, int HD, float drop_p, uint32_t seed, const void* images, float* dqkv,
                           void* ws, size_t ws_bytes, int precision, void* stream);
int rlt_list_attention_bwd_prepare(const float* out, const float* dout, const float* lse, int S, int B, int H, int HD, float drop_p,
                                   const void* images, void* ws, size_t ws_bytes, int precision, void* stream);
int rlt_list_attention_bwd_dkv(const float* qkv, const float* dout, const float* lse, const void* images, void* ws, size_t ws_bytes,
                               int S, int B, int H, int HD, float drop_p, uint32_t seed, float* dqkv, int precision, void* stream);
int rlt_list_attention_bwd_dq(const float* qkv, const float* dout, const float* lse, const void* images, void* ws, size_t ws_bytes,
                              int S, int B, int H, int HD, float drop_p, uint32_t seed, float* dqkv, int precision, void* stream);
/* keep-mask of the attention-probability dropout as data (tests, small B): out (S,H,B,B) =
 * keep ? 1/(1-p) : 0 for (position, head, query, key) */
int rlt_attention_dropout_mask(uint32_t seed, int S, int B, int H, float p, float* out, void* stream);
/* the same for the (position, head) pairs pair0 .. pair0 + npair - 1 only, pair = position * H + head: out (npair,B,B)
 * (tests at benchmark sizes, where the whole (S,H,B,B) mask would not fit) */
int rlt_attention_dropout_mask_range(uint32_t seed, int pair0, int npair, int B, float p, float* out, void* stream);

/* ------------------------------------------------------------------ BiLSTM recurrence (M2)
 * One bidirectional LSTM layer with hidden size 128 (nn.LSTM(..., hidden_size=128,
 * bidirectional=True, batch_first=True), models/AttnCut.py:8): gate order i,f,g,o, h0=c0=0.
 * The input projections x_t W_ih^T + b_ih + b_hh of both directions are computed beforehand by
 * rlt_gemm into `gates` (S*B, 2, 512) position-major (dir 0 = forward in s, dir 1 = reverse).
 * fwd: adds h_{t-1} W_hh^T, applies the nonlinearities, overwrites `gates` in place with the
 *      ACTIVATED gates (i,f,g,o), writes the cell states c (S*B,2,128) and h_out (S*B,256) =
 *      [forward | reverse] hidden states.
 * bwd: from d_hout (S*B,256) and the stashes, writes d(pre-activation gates) in place over
 *      `gates` (then dW_ih, dW_hh, db, dx are plain rlt_gemm / rlt_colsum calls on it).
 * w_hh_fwd, w_hh_rev: (512,128) each = weight_hh_l{k}, weight_hh_l{k}_reverse.
 */
int rlt_bilstm_rec_fwd(float* gates, const float* w_hh_fwd, const float* w_hh_rev, int S, int B,
                       float* h_out, float* c_out, int precision, void* stream);

/* The same recurrence with the input projection fused in, for narrow inputs (1 <= I <= 3: layer 0 of the
 * reference's encoders, input_size = 3, models/AttnCut.py:6,8): pre-activations x W_ih^T + b_ih + b_hh are formed
 * inside the kernel from x (S*B, I) and never touch HBM; `gates` (S*B, 1024) is output only (activated gates for
 * rlt_bilstm_rec_bwd).  RLT_E_SHAPE for I > 3: use rlt_gemm + rlt_bilstm_rec_fwd. */
int rlt_bilstm_rec_fwd_x(const float* x, int I, const float* w_ih_fwd, const float* b_ih_fwd, const float* b_hh_fwd,
                         const float* w_ih_rev, const float* b_ih_rev, const float* b_hh_rev,
                         const float* w_hh_fwd, const float* w_hh_rev, int S, int B,
                         float* gates, float* h_out, float* c_out, int precision, void* stream);
int rlt_bilstm_rec_bwd(float* gates, const float* c, const float* w_hh_fwd, const float* w_hh_rev,
                       const float* d_hout, int S, int B, int precision, void* stream);

/* ------------------------------------------------------------------ PATH-LEVEL ENTRY POINTS
 * One call = the forward or the backward of one module of the reference's models, composed inside the library from
 * the kernel-level entry points of this header (same launches, same order, same in-place accumulation): a host in any
 * language drives the hot path with these, the kernel-level entry points stay for tests and for hosts that fuse
 * differently.  Activations are position-major (S*B, E).  The caller owns two buffers per module:
 *   stash  written by the forward, read (the BiLSTM's: also overwritten) by the backward; layout documented at
 *          enc_stash() / lstm_stash() in csrc/path.hip;
 *   ws     scratch, dead after the call.
 * rlt_workspace_bytes(op, S, B, E, H, FF, train_dropout) reports their sizes; all regions are 256-byte aligned, pass
 * 256-byte aligned buffers.  For the RLT_OP_BILSTM_* queries E = the input feature count of layer 0, H and FF unused.
 */
#define RLT_OP_ENCODER_STASH   1   /* stash of rlt_encoder_layer_fwd/bwd (attention tile records only where the backward reads
                                    * them: rlt_list_attention_images_retained; independent of train_dropout)  */
#define RLT_OP_ENCODER_FWD_WS  2   /* ws of rlt_encoder_layer_fwd: split-K scratch + the attention images that are scratch of the
                                    * forward call (the pipelined bf16x6 forward kernels; train_dropout = the call's drop_p > 0) */
#define RLT_OP_ENCODER_BWD_WS  3   /* ws of rlt_encoder_layer_bwd (train_dropout != 0: + two (T,E) dropout grads) */
#define RLT_OP_BILSTM_STASH    4   /* stash of rlt_bilstm_fwd/bwd                                               */
#define RLT_OP_BILSTM_WS       5   /* ws of rlt_bilstm_fwd and rlt_bilstm_bwd                                   */
size_t rlt_workspace_bytes(int op, int S, int B, int E, int H, int FF, int train_dropout, int precision);

/* nn.TransformerEncoderLayer(d_model=E, nhead=H, dim_feedforward=FF, dropout) parameters, by state_dict name
 * (models/AttnCut.py:9: `attention_layer.layers.<i>.` + self_attn.in_proj_weight (3E,E), self_attn.in_proj_bias (3E),
 * self_attn.out_proj.weight (E,E), .bias (E), norm1.weight/.bias (E), linear1.weight (FF,E), .bias (FF),
 * linear2.weight (E,FF), .bias (E), norm2.weight/.bias (E)) */
typedef struct rlt_encoder_weights {
    const float *in_proj_weight, *in_proj_bias, *out_proj_weight, *out_proj_bias, *norm1_weight, *norm1_bias,
                *linear1_weight, *linear1_bias, *linear2_weight, *linear2_bias, *norm2_weight, *norm2_bias;
} rlt_encoder_weights;
typedef struct rlt_encoder_grads {     /* same shapes; every member is WRITTEN (=), not accumulated */
    float *in_proj_weight, *in_proj_bias, *out_proj_weight, *out_proj_bias, *norm1_weight, *norm1_bias,
          *linear1_weight, *linear1_bias, *linear2_weight, *linear2_bias, *norm2_weight, *norm2_bias;
} rlt_encoder_grads;
/* y = norm2(h1 + drop2(linear2(drop(relu(linear1(h1)))))),  h1 = norm1(x + drop1(out_proj(list_attention(in_proj(x)))))
 * (post-norm, ReLU, attention over the B lists at each of the S positions: F.multi_head_attention_forward on a
 * (L=B, N=S, E) input - models/AttnCut.py:9-10,18).  x, y: (S*B, E).  drop_p = 0 in eval();
 * seeds[4] = {attention probabilities, dropout1, FFN hidden, dropout2} (may be NULL when drop_p == 0).
 * bwd: dx (S*B, E) = d/dx, every member of g written; needs the same x, w, seeds and the forward's stash. */
int rlt_encoder_layer_fwd(const float* x, const rlt_encoder_weights* w, int S, int B, int E, int H, int FF, float eps,
                          float drop_p, const uint32_t* seeds, float* y, void* stash, size_t stash_bytes,
                          void* ws, size_t ws_bytes, int precision, void* stream);
int rlt_encoder_layer_bwd(const float* x, const rlt_encoder_weights* w, int S, int B, int E, int H, int FF, float eps,
                          float drop_p, const uint32_t* seeds, const float* dy, const void* stash, size_t stash_bytes,
                          float* dx, const rlt_encoder_grads* g, void* ws, size_t ws_bytes, int precision, void* stream);

/* One bidirectional layer of nn.LSTM(input, 128, num_layers=2, batch_first=True, bidirectional=True)
 * (models/AttnCut.py:8), index 0 = forward direction, 1 = reverse: weight_ih_l{k}[_reverse] (512, in),
 * weight_hh_l{k}[_reverse] (512,128), bias_ih_l{k}[_reverse] (512), bias_hh_l{k}[_reverse] (512). */
typedef struct rlt_lstm_layer_weights { const float *w_ih[2], *w_hh[2], *b_ih[2], *b_hh[2]; } rlt_lstm_layer_weights;
typedef struct rlt_lstm_layer_grads { float *w_ih[2], *w_hh[2], *b_ih[2], *b_hh[2]; } rlt_lstm_layer_grads;   /* written (=) */
/* The whole 2-layer stack: x (S*B, I) -> h_out (S*B, 256) = [forward | reverse] hidden states of layer 1; w[2], g[2] =
 * layers 0 and 1 (layer 1 has 256 inputs).  bwd: dh_out (S*B,256) -> g and, when dx != NULL, dx (S*B, I); it needs the
 * forward's h_out and stash, and overwrites the gate stashes (call it once per forward). */
int rlt_bilstm_fwd(const float* x, int I, const rlt_lstm_layer_weights* w, int S, int B, float* h_out,
                   void* stash, size_t stash_bytes, void* ws, size_t ws_bytes, int precision, void* stream);
int rlt_bilstm_bwd(const float* x, int I, const rlt_lstm_layer_weights* w, const float* h_out, const float* dh_out, int S, int B,
                   void* stash, size_t stash_bytes, float* dx, const rlt_lstm_layer_grads* g,
                   void* ws, size_t ws_bytes, int precision, void* stream);

/* The same 2-layer stack for ANY hidden size (`encoding_size` of models/MMOECut.py:57,63; every other model and every
 * BASELINE config uses 128, which runs on the persistent kernels above).  General form, built for coverage: one GEMM
 * per layer for the input projection, one small GEMM per direction and one cell kernel per time step, the backward
 * the same in reverse.  w[l].w_ih (4*hidden, in), w[l].w_hh (4*hidden, hidden), biases (4*hidden); layer 1 has
 * 2*hidden inputs; h_out (S*B, 2*hidden).  rlt_bilstm_generic_bytes(stash != 0, ...) / (0, ...) size the two buffers. */
size_t rlt_bilstm_generic_bytes(int stash, int S, int B, int I, int hidden);
int rlt_bilstm_generic_fwd(const float* x, int I, int hidden, const rlt_lstm_layer_weights* w, int S, int B, float* h_out,
                           void* stash, size_t stash_bytes, void* ws, size_t ws_bytes, int precision, void* stream);
int rlt_bilstm_generic_bwd(const float* x, int I, int hidden, const rlt_lstm_layer_weights* w, const float* h_out,
                           const float* dh_out, int S, int B, void* stash, size_t stash_bytes, float* dx,
                           const rlt_lstm_layer_grads* g, void* ws, size_t ws_bytes, int precision, void* stream);

/* ------------------------------------------------------------------ layout helpers
 * (B,S,F) user layout <-> (S*B,F) position-major */
int rlt_to_position_major(const float* x_bsf, int B, int S, int F, float* x_sbf, void* stream);
int rlt_from_position_major(const float* x_sbf, int B, int S, int F, float* x_bsf, void* stream);
/* Choopy input: out[(s*B+b), 0] = score[b,s]; out[.., 1+c] = pe[s,c] (models/Choopy.py:19-20).
 * E = 1 + pe columns (128). */
int rlt_choopy_embed(const float* score_bs, const float* pe, int B, int S, int E, float* out, void* stream);

/* ------------------------------------------------------------------ decision heads (M4, M6)
 * Up to 3 heads Linear(E,1) over the same position-major activations x (S*B,E), each followed
 * by softmax over the S positions of a list (kind 0), a sigmoid (1) or nothing (2):
 * models/AttnCut.py:11-14,19; models/MtAttnCut.py:11-19,24-26.
 *   w: (n_heads,E), b: (n_heads).  out: (n_heads,B,S) in the reference's (B,S) layout.
 * bwd: dout (n_heads,B,S) -> dx (S*B,E) (overwritten, or += with accumulate), dw (n_heads,E), db (n_heads).
 */
#define RLT_HEAD_SOFTMAX  0
#define RLT_HEAD_SIGMOID  1
#define RLT_HEAD_IDENTITY 2
int rlt_heads_fwd(const float* x, const float* w, const float* b, const int* kinds, int n_heads,
                  int S, int B, int E, float* out, void* stream);
size_t rlt_heads_bwd_workspace(int n_heads, int S, int B, int E);
int rlt_heads_bwd(const float* x, const float* w, const int* kinds, int n_heads,
                  const float* out, const float* dout, int S, int B, int E,
                  float* dx, int accumulate_dx, float* dw, float* db,
                  void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ MMOE gates / mixture (M7)
 * models/MMOECut.py:93-94: gate[t][b][:] = softmax_e( flatten_s(h[b]) @ w_gate[t] ), h the BiLSTM
 * output (position-major (S*B,C), C = 256), w_gate[t]: (S*C, n_e) row index s*C + c.
 * models/MMOECut.py:101-102: mixed[t][tok][:] = sum_e gate[t][b][e] * expert[e][tok][:].
 *   gates out: (n_tasks,B,n_e).  n_e <= 8, n_tasks <= 3.
 * bwd of the gate: from dgate (n_tasks,B,n_e): dh (S*B,C) = or += (accumulate_dh), dw_gate[t] (S*C,n_e) =.
 * mix fwd: experts[e] (S*B,E), e < n_e (host array of device pointers) -> mixed (n_tasks,S*B,E).
 * mix bwd: dmixed (n_tasks,S*B,E) -> dexperts[e] (S*B,E) = ; dgate (n_tasks,B,n_e) = .
 * w_gate / dw_gate / experts / dexperts are HOST arrays of device pointers.
 */
int rlt_mmoe_gate_fwd(const float* h, const float* const* w_gate, int n_tasks, int n_e,
                      int S, int B, int C, float* gates, void* stream);
size_t rlt_mmoe_gate_bwd_workspace(int n_tasks, int n_e, int S, int B, int C);
int rlt_mmoe_gate_bwd(const float* h, const float* const* w_gate, const float* gates, const float* dgates,
                      int n_tasks, int n_e, int S, int B, int C,
                      float* dh, int accumulate_dh, float* const* dw_gate,
                      void* ws, size_t ws_bytes, void* stream);
int rlt_mmoe_mix_fwd(const float* const* experts, const float* gates, int n_tasks, int n_e,
                     int S, int B, int E, float* mixed, void* stream);
int rlt_mmoe_mix_bwd(const float* const* experts, const float* gates, const float* dmixed,
                     int n_tasks, int n_e, int S, int B, int E,
                     float* const* dexperts, float* dgates, void* stream);

/* ------------------------------------------------------------------ BiCut (section 8f row N4)
 * Two-class head of models/Bicut.py:11-16: position-major logits z (S*B, 2) -> Dropout on the logits -> softmax over
 * the two classes {0: truncate, 1: continue}; out is (B, S, 2) in the reference's layout.  bwd: dz (S*B, 2). */
int rlt_pair_softmax_fwd(const float* z, int B, int S, float drop_p, uint32_t seed, float* out, void* stream);
int rlt_pair_softmax_bwd(const float* out, const float* dout, int B, int S, float drop_p, uint32_t seed, float* dz, void* stream);
/* BiCutLoss (utils/losses.py:11-45) and its gradient in one pass.  out (B,S,2), labels (B,S) in {0,1}.
 * mask = positions up to and including the last one whose argmax is class 0 (all positions when none is);
 * metric_nci != 0: reward (0, -1/log2(j+2)) for label 1, (0, (j+1)/alpha) for label 0; otherwise ((1-alpha)/r, 0) and
 * (0, alpha/(1-r)).  loss = sum(out * mask * reward) / B; dout = mask * reward / B; per_list (B) unnormalised. */
int rlt_bicut_loss(const float* out, const float* labels, int B, int S, int metric_nci, float alpha, float r,
                   float* per_list, float* loss, float* dout, void* stream);

/* ------------------------------------------------------------------ WassDistLoss (section 8f row N4)
 * utils/losses.py:236-311: entropic optimal transport between the B predicted distributions p (B,S) and the B label
 * vectors (B,S) of a batch: squared-distance cost, uniform marginals, at most max_iter log-domain Sinkhorn iterations
 * with regularisation eps, stopped after the iteration whose sum |u - u_prev| < thresh (reference: 0.1; decided on the
 * device, no host synchronisation), loss = sum(pi * C).  fwd records the iterations in ws; bwd replays them in reverse
 * and writes dp = gscale[0] * d(loss)/dp (gscale NULL = 1).  ws: rlt_wass_loss_workspace(B, max_iter) bytes, the
 * same buffer for fwd and bwd. */
size_t rlt_wass_loss_workspace(int B, int max_iter);
int rlt_wass_loss_fwd(const float* p, const float* labels, int B, int S, float eps, int max_iter, float thresh,
                      float* loss, void* ws, size_t ws_bytes, void* stream);
int rlt_wass_loss_bwd(const float* p, const float* labels, const float* gscale, int B, int S, float eps, int max_iter,
                      void* ws, size_t ws_bytes, float* dp, void* stream);

/* ------------------------------------------------------------------ optimizer (N2, run.py:104,129)
 * torch.optim.Adam with coupled L2 (grad += wd * p), bias correction, eps outside the sqrt,
 * on a flat fp32 bucket.  step: 1-based step count. */
int rlt_adam_step(float* p, const float* g, float* m, float* v, size_t n, int step,
                  float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RLT_HIP_H */

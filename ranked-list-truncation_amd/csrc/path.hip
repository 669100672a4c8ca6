// Path-level entry points: one call = the forward or the backward of one module of the
// reference's models - an nn.TransformerEncoderLayer with list-axis attention (models/AttnCut.py:9-10,18;
// models/Choopy.py:11-12,21; models/MMOECut.py:9-13) or the 2-layer bidirectional nn.LSTM(…,128) encoder
// (models/AttnCut.py:8,17) - composed HERE from the kernel launches of gemm.hip / attention*.hip / norm.hip / lstm.hip,
// so that a host in any language drives the hot path with two calls per module instead of re-implementing the
// launch sequence, the in-place accumulation order and the stash reuse.  Everything is stream-ordered; the caller owns
// the stash (forward -> backward) and the scratch workspace, whose sizes rlt_workspace_bytes() reports.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr size_t ALIGN = 256;
inline size_t rup(size_t b) { return (b + ALIGN - 1) / ALIGN * ALIGN; }

struct Cursor {              // carves 256-byte aligned regions out of a caller-owned buffer
    uint8_t* base; size_t off, cap;
    // cap = the caller's byte count when known.  The entry points compare the total against it before any launch; the
    // sanitizer / debug build (rlt_hip/build.py: build_sanitized, -DRLT_BOUNDS_CHECK) additionally aborts at the first region
    // that would leave the buffer, naming it - a layout bug then fails at its line, not as a GPU fault later.
    explicit Cursor(void* p, size_t cap_ = (size_t)-1) : base((uint8_t*)p), off(0), cap(cap_) {}
    template <typename T = float> T* take(size_t bytes) {
        T* p = (T*)(base ? base + off : nullptr);
#ifdef RLT_BOUNDS_CHECK
        if (base && cap != (size_t)-1 && off + bytes > cap) {
            fprintf(stderr, "rlt bounds check: region of %zu bytes at offset %zu leaves a %zu-byte buffer\n", bytes, off, cap);
            abort();
        }
#endif
        off += rup(bytes);
        return p;
    }
};

// ---------------------------------------------------------------------------------- encoder layer
struct EncDims { int S, B, E, H, FF; size_t T; int HD; bool bits; };

inline int enc_dims(int S, int B, int E, int H, int FF, EncDims& d) {
    if (S <= 0 || B <= 0 || E <= 0 || H <= 0 || FF <= 0 || E % H) return RLT_E_ARG;
    d = EncDims{S, B, E, H, FF, (size_t)S * B, E / H, FF % 32 == 0};
    if ((size_t)S * B > 0x7fffffffu) return RLT_E_SHAPE;
    return 0;
}

// stash written by the forward and read by the backward, in this order (all regions 256-byte aligned):
//   qkv (T,3E) | att (T,E) | lse (S,H,B) | proj (T,E) | st1 (T,2) | h1 (T,E) | hid (T,FF) | relu bits (ceil(T/32),FF) u32 |
//   ff (T,E) | st2 (T,2) | attention tile records (rlt_list_attention_fwd_workspace bytes; only where the backward reads them:
//   rlt_list_attention_images_retained - the split-bf16 mode)
struct EncStash {
    float *qkv, *att, *lse, *proj, *st1, *h1, *hid, *ff, *st2;
    uint32_t* bits;
    void* images; size_t images_bytes;
    size_t bytes;
};
inline EncStash enc_stash(const EncDims& d, void* base, size_t cap = (size_t)-1) {
    Cursor c(base, cap);
    EncStash s{};
    const size_t T = d.T, f = sizeof(float);
    s.qkv = c.take(T * 3 * d.E * f);
    s.att = c.take(T * d.E * f);
    s.lse = c.take((size_t)d.S * d.H * d.B * f);
    s.proj = c.take(T * d.E * f);
    s.st1 = c.take(T * 2 * f);
    s.h1 = c.take(T * d.E * f);
    s.hid = c.take(T * d.FF * f);
    s.bits = d.bits ? c.take<uint32_t>(rlt_gemm_bits_words((int)T, d.FF) * sizeof(uint32_t)) : nullptr;
    s.ff = c.take(T * d.E * f);
    s.st2 = c.take(T * 2 * f);
    // (only the records the BACKWARD reads live in the stash; the images of the pipelined bf16x6 forward kernels are scratch of the
    // forward call: enc_fwd_images_scratch below)
    const bool keep = rlt_list_attention_images_retained(d.S, d.B, d.H, d.HD, RLT_PRECISION_DEFAULT) != 0;
    s.images_bytes = keep ? rlt_list_attention_fwd_workspace(d.S, d.B, d.H, d.HD, 0.f, RLT_PRECISION_DEFAULT) : 0;
    s.images = s.images_bytes ? c.take<uint8_t>(s.images_bytes) : nullptr;
    s.bytes = c.off;
    return s;
}
// forward scratch: [ split-K workspace of the GEMMs | attention images that the backward does not read ]
inline size_t enc_fwd_images_scratch(const EncDims& d, float drop_p) {
    if (rlt_list_attention_images_retained(d.S, d.B, d.H, d.HD, RLT_PRECISION_DEFAULT)) return 0;
    return rlt_list_attention_fwd_workspace(d.S, d.B, d.H, d.HD, drop_p, RLT_PRECISION_DEFAULT);
}

// backward scratch.  Two phases share the big middle region:
//   dz2 (T,E) | dr2 (T,E; dropout only) | { FFN phase: dhid (T,FF) }  U  { attention phase: dr1 (T,E; dropout only) |
//   datt (T,E) | dqkv (T,3E) | attention ws } | LayerNorm ws | split-K ws
struct EncScratch {
    float *dz2, *dr2, *dhid, *dr1, *datt, *dqkv;
    void *attn_ws, *ln_ws, *gemm_ws;
    size_t attn_ws_bytes, ln_ws_bytes, gemm_ws_bytes, bytes;
};
inline size_t enc_gemm_ws(const EncDims& d) {
    const int T = (int)d.T;
    size_t m = 0;
    auto up = [&](size_t v) { if (v > m) m = v; };
    up(rlt_gemm_workspace(1, 0, d.E, d.FF, T));
    up(rlt_gemm_workspace(1, 0, d.FF, d.E, T));
    up(rlt_gemm_workspace(1, 0, d.E, d.E, T));
    up(rlt_gemm_workspace(1, 0, 3 * d.E, d.E, T));
    up(rlt_gemm_workspace(0, 1, T, 3 * d.E, d.E));
    up(rlt_gemm_workspace(0, 1, T, d.FF, d.E));
    up(rlt_gemm_workspace(0, 1, T, d.E, d.FF));
    up(rlt_gemm_workspace(0, 0, T, d.FF, d.E));
    up(rlt_gemm_workspace(0, 0, T, d.E, d.FF));
    up(rlt_gemm_workspace(0, 0, T, d.E, 3 * d.E));
    return m;
}
inline EncScratch enc_scratch(const EncDims& d, bool drop, void* base, size_t cap = (size_t)-1) {
    Cursor c(base, cap);
    EncScratch w{};
    const size_t T = d.T, f = sizeof(float);
    w.dz2 = c.take(T * d.E * f);
    w.dr2 = drop ? c.take(T * d.E * f) : nullptr;
    const size_t mid = c.off;
    w.dhid = c.take(T * d.FF * f);
    const size_t end_ffn = c.off;
    c.off = mid;
    w.dr1 = drop ? c.take(T * d.E * f) : nullptr;
    w.datt = c.take(T * d.E * f);
    w.dqkv = c.take(T * 3 * d.E * f);
    w.attn_ws_bytes = rlt_list_attention_bwd_workspace(d.S, d.B, d.H, d.HD, drop ? 0.5f : 0.f, RLT_PRECISION_DEFAULT);     // (any rate > 0: the size depends on train / eval only)
    w.attn_ws = c.take<uint8_t>(w.attn_ws_bytes);
    if (c.off < end_ffn) c.off = end_ffn;
    w.ln_ws_bytes = rlt_add_layernorm_bwd_workspace((int)T, d.E);
    w.ln_ws = c.take<uint8_t>(w.ln_ws_bytes);
    w.gemm_ws_bytes = enc_gemm_ws(d);
    w.gemm_ws = c.take<uint8_t>(w.gemm_ws_bytes);
    w.bytes = c.off;
    return w;
}

#define RLT_TRY(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

inline int gemm(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                const float* bias, int flags, float* colsum_a, void* ws, size_t ws_bytes, void* st) {
    return rlt_gemm_ex(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, nullptr, flags, nullptr, 0, 1.f, colsum_a, 0.f, 0u,
                       ws, ws_bytes, RLT_PRECISION_DEFAULT, st);
}

// ---------------------------------------------------------------------------------- BiLSTM stack
// stash of the 2-layer stack: per layer activated gates (T,1024) and cell states (T,256); the hidden states of layer 0
// (T,256) - the input of layer 1.  The backward overwrites the gate stashes in place with d(pre-activations).
struct LstmStash { float *gates[2], *c[2], *h0; size_t bytes; };
inline LstmStash lstm_stash(size_t T, void* base, size_t cap = (size_t)-1) {
    Cursor cur(base, cap);
    LstmStash s{};
    for (int l = 0; l < 2; ++l) {
        s.gates[l] = cur.take(T * 1024 * sizeof(float));
        s.c[l] = cur.take(T * 256 * sizeof(float));
    }
    s.h0 = cur.take(T * 256 * sizeof(float));
    s.bytes = cur.off;
    return s;
}
// scratch: packed [w_ih_f ; w_ih_r] (1024, I<=256) and packed biases 2 x 1024 | dh0 (T,256) | packed dW_ih (1024, 256) |
// packed db (1024) | narrow-dW / split-K workspace
struct LstmScratch { float *wcat, *bcat, *dh0, *dwcat, *dbcat; void* ws; size_t ws_bytes, bytes; };
inline LstmScratch lstm_scratch(size_t T, int I, void* base, size_t cap = (size_t)-1) {
    Cursor cur(base, cap);
    LstmScratch w{};
    const int Imax = I > 256 ? I : 256;
    w.wcat = cur.take((size_t)1024 * Imax * sizeof(float));
    w.bcat = cur.take(2 * 1024 * sizeof(float));
    w.dh0 = cur.take(T * 256 * sizeof(float));
    w.dwcat = cur.take((size_t)1024 * Imax * sizeof(float));
    w.dbcat = cur.take(1024 * sizeof(float));
    size_t m = rlt_narrow_dw_workspace((int)T, 1024);
    auto up = [&](size_t v) { if (v > m) m = v; };
    up(rlt_gemm_workspace(1, 0, 1024, Imax, (int)T));
    up(rlt_gemm_workspace(1, 0, 512, 128, (int)T));
    up(rlt_gemm_workspace(0, 1, (int)T, 1024, Imax));
    up(rlt_gemm_workspace(0, 0, (int)T, Imax, 1024));
    w.ws_bytes = m;
    w.ws = cur.take<uint8_t>(m);
    w.bytes = cur.off;
    return w;
}

__global__ __launch_bounds__(256) void pack2_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                    float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < 2 * n; i += (size_t)gridDim.x * 256)
        out[i] = i < n ? a[i] : b[i - n];
}
__global__ __launch_bounds__(256) void unpack2_kernel(const float* __restrict__ in, size_t n, float* __restrict__ a,
                                                      float* __restrict__ b) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < 2 * n; i += (size_t)gridDim.x * 256) {
        if (i < n) a[i] = in[i]; else b[i - n] = in[i];
    }
}
inline int pack2(const float* a, const float* b, size_t n, float* out, hipStream_t st) {
    const int grid = (int)((2 * n + 255) / 256 < 512 ? (2 * n + 255) / 256 : 512);
    hipLaunchKernelGGL(pack2_kernel, dim3(grid), dim3(256), 0, st, a, b, n, out);
    return RLT_LAUNCH_RESULT();
}
inline int unpack2(const float* in, size_t n, float* a, float* b, hipStream_t st) {
    const int grid = (int)((2 * n + 255) / 256 < 512 ? (2 * n + 255) / 256 : 512);
    hipLaunchKernelGGL(unpack2_kernel, dim3(grid), dim3(256), 0, st, in, n, a, b);
    return RLT_LAUNCH_RESULT();
}

int lstm_layer_fwd(const float* x, int I, const rlt_lstm_layer_weights& w, int S, int B, float* gates, float* c, float* h,
                   const LstmScratch& sc, void* stream) {
    const int T = S * B;
    if (I <= 3)            // narrow input (layer 0): the projection is formed inside the recurrence
        return rlt_bilstm_rec_fwd_x(x, I, w.w_ih[0], w.b_ih[0], w.b_hh[0], w.w_ih[1], w.b_ih[1], w.b_hh[1], w.w_hh[0], w.w_hh[1],
                                    S, B, gates, h, c, RLT_PRECISION_DEFAULT, stream);
    hipStream_t st = rlt_stream(stream);
    // input projections of both directions as one product (x is read once), biases b_ih + b_hh folded in
    RLT_TRY(pack2(w.w_ih[0], w.w_ih[1], (size_t)512 * I, sc.wcat, st));
    RLT_TRY(pack2(w.b_ih[0], w.b_ih[1], 512, sc.bcat, st));
    RLT_TRY(pack2(w.b_hh[0], w.b_hh[1], 512, sc.bcat + 1024, st));
    RLT_TRY(rlt_gemm_ex(0, 1, T, 1024, I, x, I, sc.wcat, I, gates, 1024, sc.bcat, sc.bcat + 1024, 0, nullptr, 0, 1.f, nullptr,
                        0.f, 0u, sc.ws, sc.ws_bytes, RLT_PRECISION_DEFAULT, stream));
    return rlt_bilstm_rec_fwd(gates, w.w_hh[0], w.w_hh[1], S, B, h, c, RLT_PRECISION_DEFAULT, stream);
}

// gates <- d(pre-activation gates) in place; weight gradients written (=); dx (T,I) written when not NULL
int lstm_layer_bwd(const float* x, int I, const rlt_lstm_layer_weights& w, const float* h, float* gates, const float* c,
                   const float* dh, int S, int B, float* dx, const rlt_lstm_layer_grads& g, const LstmScratch& sc, void* stream) {
    const int T = S * B;
    hipStream_t st = rlt_stream(stream);
    RLT_TRY(rlt_bilstm_rec_bwd(gates, c, w.w_hh[0], w.w_hh[1], dh, S, B, RLT_PRECISION_DEFAULT, stream));
    const float* dA = gates;
    if (I <= 3) {          // dW_ih of both directions and the bias gradients in ONE streaming pass over dA
        RLT_TRY(rlt_narrow_dw(dA, 1024, x, I, I, T, 1024, sc.dwcat, sc.dbcat, sc.ws, sc.ws_bytes, stream));
    } else {               // both directions in one product; the bias gradient (column sums of dA) rides on it
        RLT_TRY(gemm(1, 0, 1024, I, T, dA, 1024, x, I, sc.dwcat, I, nullptr, 0, sc.dbcat, sc.ws, sc.ws_bytes, stream));
    }
    RLT_TRY(unpack2(sc.dwcat, (size_t)512 * I, g.w_ih[0], g.w_ih[1], st));
    RLT_TRY(unpack2(sc.dbcat, 512, g.b_ih[0], g.b_ih[1], st));
    RLT_TRY(unpack2(sc.dbcat, 512, g.b_hh[0], g.b_hh[1], st));          // d b_hh = d b_ih
    if (S > 1) {
        const int K = T - B;
        // forward direction: h_{t-1} of position s is the row block of position s-1; reverse: of position s+1
        RLT_TRY(gemm(1, 0, 512, 128, K, dA + (size_t)B * 1024, 1024, h, 256, g.w_hh[0], 128, nullptr, 0, nullptr, sc.ws, sc.ws_bytes, stream));
        RLT_TRY(gemm(1, 0, 512, 128, K, dA + 512, 1024, h + (size_t)B * 256 + 128, 256, g.w_hh[1], 128, nullptr, 0, nullptr, sc.ws, sc.ws_bytes, stream));
    } else {
        RLT_TRY((int)hipMemsetAsync(g.w_hh[0], 0, 512 * 128 * sizeof(float), st));
        RLT_TRY((int)hipMemsetAsync(g.w_hh[1], 0, 512 * 128 * sizeof(float), st));
    }
    if (dx) {              // one product over both directions
        RLT_TRY(pack2(w.w_ih[0], w.w_ih[1], (size_t)512 * I, sc.wcat, st));
        RLT_TRY(gemm(0, 0, T, I, 1024, dA, 1024, sc.wcat, I, dx, I, nullptr, 0, nullptr, sc.ws, sc.ws_bytes, stream));
    }
    return 0;
}

bool lstm_weights_ok(const rlt_lstm_layer_weights& w) {
    for (int d = 0; d < 2; ++d)
        if (!w.w_ih[d] || !w.w_hh[d] || !w.b_ih[d] || !w.b_hh[d]) return false;
    return true;
}
bool lstm_grads_ok(const rlt_lstm_layer_grads& g) {
    for (int d = 0; d < 2; ++d)
        if (!g.w_ih[d] || !g.w_hh[d] || !g.b_ih[d] || !g.b_hh[d]) return false;
    return true;
}

}  // namespace

extern "C" {

size_t rlt_workspace_bytes(int op, int S, int B, int E, int H, int FF, int train_dropout, int precision) {
    RLT_PREC_SCOPE_SZ(precision);
    if (op == RLT_OP_ENCODER_STASH || op == RLT_OP_ENCODER_BWD_WS || op == RLT_OP_ENCODER_FWD_WS) {
        EncDims d;
        if (enc_dims(S, B, E, H, FF, d)) return 0;
        if (op == RLT_OP_ENCODER_STASH) return enc_stash(d, nullptr).bytes;
        if (op == RLT_OP_ENCODER_FWD_WS) return rup(enc_gemm_ws(d)) + rup(enc_fwd_images_scratch(d, train_dropout ? 0.5f : 0.f));
        return enc_scratch(d, train_dropout != 0, nullptr).bytes;
    }
    if (op == RLT_OP_BILSTM_STASH || op == RLT_OP_BILSTM_WS) {
        if (S <= 0 || B <= 0 || E <= 0) return 0;            // E = input features of layer 0
        const size_t T = (size_t)S * B;
        return op == RLT_OP_BILSTM_STASH ? lstm_stash(T, nullptr).bytes : lstm_scratch(T, E, nullptr).bytes;
    }
    return 0;
}

int rlt_encoder_layer_fwd(const float* x, const rlt_encoder_weights* w, int S, int B, int E, int H, int FF, float eps,
                          float drop_p, const uint32_t* seeds, float* y, void* stash, size_t stash_bytes,
                          void* ws, size_t ws_bytes, int precision, void* stream) {
    RLT_PREC_SCOPE(precision);
    RLT_CHECK_ARG(x && w && y && stash && drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seeds));
    RLT_CHECK_ARG(w->in_proj_weight && w->in_proj_bias && w->out_proj_weight && w->out_proj_bias && w->norm1_weight &&
                  w->norm1_bias && w->linear1_weight && w->linear1_bias && w->linear2_weight && w->linear2_bias &&
                  w->norm2_weight && w->norm2_bias);
    EncDims d;
    RLT_TRY(enc_dims(S, B, E, H, FF, d));
    const size_t gemm_ws = rup(enc_gemm_ws(d)), img_scratch = enc_fwd_images_scratch(d, drop_p);
    if (stash_bytes < enc_stash(d, nullptr).bytes || ws_bytes < gemm_ws + img_scratch || (gemm_ws + img_scratch && !ws)) return RLT_E_WORKSPACE;
    EncStash s = enc_stash(d, stash, stash_bytes);
    if (img_scratch) { s.images = static_cast<uint8_t*>(ws) + gemm_ws; s.images_bytes = img_scratch; }       // scratch of this call
    const int T = (int)d.T;
    const uint32_t s_attn = drop_p > 0.f ? seeds[0] : 0u, s_ln1 = drop_p > 0.f ? seeds[1] : 0u,
                   s_ffn = drop_p > 0.f ? seeds[2] : 0u, s_ln2 = drop_p > 0.f ? seeds[3] : 0u;
    // in_proj -> list-axis attention -> out_proj -> x + dropout1(.) -> norm1
    RLT_TRY(gemm(0, 1, T, 3 * E, E, x, E, w->in_proj_weight, E, s.qkv, 3 * E, w->in_proj_bias, 0, nullptr, ws, ws_bytes, stream));
    RLT_TRY(rlt_list_attention_fwd(s.qkv, S, B, H, d.HD, drop_p, s_attn, s.att, s.lse, s.images, s.images_bytes, RLT_PRECISION_DEFAULT, stream));
    RLT_TRY(gemm(0, 1, T, E, E, s.att, E, w->out_proj_weight, E, s.proj, E, w->out_proj_bias, 0, nullptr, ws, ws_bytes, stream));
    RLT_TRY(rlt_add_layernorm_fwd(x, s.proj, w->norm1_weight, w->norm1_bias, T, E, eps, drop_p, s_ln1, s.h1, s.st1, stream));
    // linear1 -> ReLU -> dropout -> linear2 -> h1 + dropout2(.) -> norm2
    if (d.bits) {
        // 1-bit mask (passed the ReLU and kept by the dropout) for the backward dH product, which then reads T*FF/8
        // bytes instead of the 4*T*FF of `hid`
        RLT_TRY(rlt_gemm_bits(0, 1, T, FF, E, s.h1, E, w->linear1_weight, E, s.hid, FF, w->linear1_bias, RLT_GEMM_RELU,
                              drop_p, s_ffn, s.bits, nullptr, 1.f, RLT_PRECISION_DEFAULT, stream));
    } else {
        RLT_TRY(rlt_gemm_ex(0, 1, T, FF, E, s.h1, E, w->linear1_weight, E, s.hid, FF, w->linear1_bias, nullptr, RLT_GEMM_RELU,
                            nullptr, 0, 1.f, nullptr, drop_p, s_ffn, ws, ws_bytes, RLT_PRECISION_DEFAULT, stream));
    }
    RLT_TRY(gemm(0, 1, T, E, FF, s.hid, FF, w->linear2_weight, FF, s.ff, E, w->linear2_bias, 0, nullptr, ws, ws_bytes, stream));
    return rlt_add_layernorm_fwd(s.h1, s.ff, w->norm2_weight, w->norm2_bias, T, E, eps, drop_p, s_ln2, y, s.st2, stream);
}

int rlt_encoder_layer_bwd(const float* x, const rlt_encoder_weights* w, int S, int B, int E, int H, int FF, float eps,
                          float drop_p, const uint32_t* seeds, const float* dy, const void* stash, size_t stash_bytes,
                          float* dx, const rlt_encoder_grads* g, void* ws, size_t ws_bytes, int precision, void* stream) {
    RLT_PREC_SCOPE(precision);
    (void)eps;
    RLT_CHECK_ARG(x && w && dy && stash && dx && g && ws && drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seeds));
    RLT_CHECK_ARG(g->in_proj_weight && g->in_proj_bias && g->out_proj_weight && g->out_proj_bias && g->norm1_weight &&
                  g->norm1_bias && g->linear1_weight && g->linear1_bias && g->linear2_weight && g->linear2_bias &&
                  g->norm2_weight && g->norm2_bias);
    EncDims d;
    RLT_TRY(enc_dims(S, B, E, H, FF, d));
    const bool drop = drop_p > 0.f;
    if (stash_bytes < enc_stash(d, nullptr).bytes || ws_bytes < enc_scratch(d, drop, nullptr).bytes) return RLT_E_WORKSPACE;
    const EncStash s = enc_stash(d, const_cast<void*>(stash), stash_bytes);
    const EncScratch k = enc_scratch(d, drop, ws, ws_bytes);
    const int T = (int)d.T;
    const uint32_t s_attn = drop ? seeds[0] : 0u, s_ln1 = drop ? seeds[1] : 0u, s_ln2 = drop ? seeds[3] : 0u;
    const float keep_scale = 1.f / (1.f - drop_p);
    // norm2: dz2 = gradient of h1 through the residual, dr2 = gradient of the FFN branch output
    RLT_TRY(rlt_add_layernorm_bwd(s.h1, s.ff, w->norm2_weight, s.st2, dy, T, E, drop_p, s_ln2, k.dz2, k.dr2,
                                  g->norm2_weight, g->norm2_bias, 0, k.ln_ws, k.ln_ws_bytes, stream));
    const float* dr2 = drop ? k.dr2 : k.dz2;
    RLT_TRY(gemm(1, 0, E, FF, T, dr2, E, s.hid, FF, g->linear2_weight, FF, nullptr, 0, g->linear2_bias, k.gemm_ws, k.gemm_ws_bytes, stream));
    // dH = (dY W2) * (H > 0) [/ (1-p)]: a dropped element has H == 0, so one mask covers ReLU and dropout
    if (d.bits) {
        RLT_TRY(rlt_gemm_bits(0, 0, T, FF, E, dr2, E, w->linear2_weight, FF, k.dhid, FF, nullptr, 0, 0.f, 0u, nullptr, s.bits,
                              keep_scale, RLT_PRECISION_DEFAULT, stream));
    } else {
        RLT_TRY(rlt_gemm_ex(0, 0, T, FF, E, dr2, E, w->linear2_weight, FF, k.dhid, FF, nullptr, nullptr, 0, s.hid, FF, keep_scale,
                            nullptr, 0.f, 0u, k.gemm_ws, k.gemm_ws_bytes, RLT_PRECISION_DEFAULT, stream));
    }
    RLT_TRY(gemm(1, 0, FF, E, T, k.dhid, FF, s.h1, E, g->linear1_weight, E, nullptr, 0, g->linear1_bias, k.gemm_ws, k.gemm_ws_bytes, stream));
    // dh1 = dz2 + dhid W1, accumulated in place
    RLT_TRY(gemm(0, 0, T, E, FF, k.dhid, FF, w->linear1_weight, E, k.dz2, E, nullptr, RLT_GEMM_ACCUMULATE, nullptr, k.gemm_ws, k.gemm_ws_bytes, stream));
    // norm1: dz1 goes straight into dx
    RLT_TRY(rlt_add_layernorm_bwd(x, s.proj, w->norm1_weight, s.st1, k.dz2, T, E, drop_p, s_ln1, dx, k.dr1,
                                  g->norm1_weight, g->norm1_bias, 0, k.ln_ws, k.ln_ws_bytes, stream));
    const float* dr1 = drop ? k.dr1 : dx;
    RLT_TRY(gemm(1, 0, E, E, T, dr1, E, s.att, E, g->out_proj_weight, E, nullptr, 0, g->out_proj_bias, k.gemm_ws, k.gemm_ws_bytes, stream));
    RLT_TRY(gemm(0, 0, T, E, E, dr1, E, w->out_proj_weight, E, k.datt, E, nullptr, 0, nullptr, k.gemm_ws, k.gemm_ws_bytes, stream));
    RLT_TRY(rlt_list_attention_bwd(s.qkv, s.att, k.datt, s.lse, S, B, H, d.HD, drop_p, s_attn, s.images, k.dqkv,
                                   k.attn_ws, k.attn_ws_bytes, RLT_PRECISION_DEFAULT, stream));
    RLT_TRY(gemm(1, 0, 3 * E, E, T, k.dqkv, 3 * E, x, E, g->in_proj_weight, E, nullptr, 0, g->in_proj_bias, k.gemm_ws, k.gemm_ws_bytes, stream));
    // dx = dz1 + dqkv W_in, accumulated in place
    return gemm(0, 0, T, E, 3 * E, k.dqkv, 3 * E, w->in_proj_weight, E, dx, E, nullptr, RLT_GEMM_ACCUMULATE, nullptr, k.gemm_ws, k.gemm_ws_bytes, stream);
}

int rlt_bilstm_fwd(const float* x, int I, const rlt_lstm_layer_weights* w, int S, int B, float* h_out,
                   void* stash, size_t stash_bytes, void* ws, size_t ws_bytes, int precision, void* stream) {
    RLT_PREC_SCOPE(precision);
    RLT_CHECK_ARG(x && w && h_out && stash && ws && I > 0 && S > 0 && B > 0 && lstm_weights_ok(w[0]) && lstm_weights_ok(w[1]));
    const size_t T = (size_t)S * B;
    RLT_CHECK_SHAPE(T <= 0x7fffffffu);
    if (stash_bytes < lstm_stash(T, nullptr).bytes || ws_bytes < lstm_scratch(T, I, nullptr).bytes) return RLT_E_WORKSPACE;
    const LstmStash s = lstm_stash(T, stash, stash_bytes);
    const LstmScratch k = lstm_scratch(T, I, ws, ws_bytes);
    RLT_TRY(lstm_layer_fwd(x, I, w[0], S, B, s.gates[0], s.c[0], s.h0, k, stream));
    return lstm_layer_fwd(s.h0, 256, w[1], S, B, s.gates[1], s.c[1], h_out, k, stream);
}

int rlt_bilstm_bwd(const float* x, int I, const rlt_lstm_layer_weights* w, const float* h_out, const float* dh_out, int S, int B,
                   void* stash, size_t stash_bytes, float* dx, const rlt_lstm_layer_grads* g,
                   void* ws, size_t ws_bytes, int precision, void* stream) {
    RLT_PREC_SCOPE(precision);
    RLT_CHECK_ARG(x && w && h_out && dh_out && stash && g && ws && I > 0 && S > 0 && B > 0);
    RLT_CHECK_ARG(lstm_weights_ok(w[0]) && lstm_weights_ok(w[1]) && lstm_grads_ok(g[0]) && lstm_grads_ok(g[1]));
    const size_t T = (size_t)S * B;
    RLT_CHECK_SHAPE(T <= 0x7fffffffu);
    if (stash_bytes < lstm_stash(T, nullptr).bytes || ws_bytes < lstm_scratch(T, I, nullptr).bytes) return RLT_E_WORKSPACE;
    const LstmStash s = lstm_stash(T, stash, stash_bytes);
    const LstmScratch k = lstm_scratch(T, I, ws, ws_bytes);
    RLT_TRY(lstm_layer_bwd(s.h0, 256, w[1], h_out, s.gates[1], s.c[1], dh_out, S, B, k.dh0, g[1], k, stream));
    return lstm_layer_bwd(x, I, w[0], s.h0, s.gates[0], s.c[0], k.dh0, S, B, dx, g[0], k, stream);
}

}  // extern "C"

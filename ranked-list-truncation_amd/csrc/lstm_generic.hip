// 2-layer bidirectional LSTM for ANY hidden size (nn.LSTM(input, hidden, num_layers=2, batch_first=True,
// bidirectional=True): models/MMOECut.py:57,63 takes `encoding_size` as a constructor argument).
//
// The reference hard-codes hidden 128 everywhere else (models/AttnCut.py:8) and every BASELINE config uses it: that
// size runs on the persistent recurrence kernels of lstm.hip (W_hh resident in registers).  This file is the general
// form, built for coverage, not speed: the input projection of a layer is one GEMM, every time step is one small GEMM
// per direction (h_{t-1} W_hh^T accumulated into the step's pre-activations) plus one cell kernel, and the backward
// walks the steps in reverse the same way; weight gradients are plain GEMMs over the whole sequence afterwards.
// Gate order i, f, g, o; h0 = c0 = 0; outputs [forward | reverse]; position-major rows t = s*B + b.
#include "common.h"

namespace {

constexpr size_t ALIGN = 256;
inline size_t rup(size_t b) { return (b + ALIGN - 1) / ALIGN * ALIGN; }
struct Cursor {
    uint8_t* base; size_t off;
    explicit Cursor(void* p) : base((uint8_t*)p), off(0) {}
    float* take(size_t bytes) { float* p = (float*)(base ? base + off : nullptr); off += rup(bytes); return p; }
};
#define RLT_TRY(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

// stash: per layer activated gates (T, 2, 4Hd) [dir][i|f|g|o][unit] and cell states (T, 2Hd); layer 0's output (T, 2Hd)
struct GStash { float *gates[2], *c[2], *h0; size_t bytes; };
GStash g_stash(size_t T, int Hd, void* base) {
    Cursor cur(base);
    GStash s{};
    for (int l = 0; l < 2; ++l) {
        s.gates[l] = cur.take(T * 8 * Hd * sizeof(float));
        s.c[l] = cur.take(T * 2 * Hd * sizeof(float));
    }
    s.h0 = cur.take(T * 2 * Hd * sizeof(float));
    s.bytes = cur.off;
    return s;
}
// scratch: packed input weights (8Hd, Imax), packed biases 2 x 8Hd, dh of layer 0 (T, 2Hd), packed dW_ih, packed db,
// recurrent dh and dc of the step in flight (2, B, Hd each), GEMM split-K workspace
struct GScratch { float *wcat, *bcat, *dh0, *dwcat, *dbcat, *dhr, *dc; void* ws; size_t ws_bytes, bytes; };
GScratch g_scratch(size_t T, int B, int I, int Hd, void* base) {
    Cursor cur(base);
    GScratch w{};
    const int Imax = I > 2 * Hd ? I : 2 * Hd;
    w.wcat = cur.take((size_t)8 * Hd * Imax * sizeof(float));
    w.bcat = cur.take((size_t)2 * 8 * Hd * sizeof(float));
    w.dh0 = cur.take(T * 2 * Hd * sizeof(float));
    w.dwcat = cur.take((size_t)8 * Hd * Imax * sizeof(float));
    w.dbcat = cur.take((size_t)8 * Hd * sizeof(float));
    w.dhr = cur.take((size_t)2 * B * Hd * sizeof(float));
    w.dc = cur.take((size_t)2 * B * Hd * sizeof(float));
    size_t m = 0;
    auto up = [&](size_t v) { if (v > m) m = v; };
    up(rlt_gemm_workspace(1, 0, 8 * Hd, Imax, (int)T));
    up(rlt_gemm_workspace(1, 0, 4 * Hd, Hd, (int)T));
    up(rlt_gemm_workspace(0, 1, (int)T, 8 * Hd, Imax));
    up(rlt_gemm_workspace(0, 0, (int)T, Imax, 8 * Hd));
    up(rlt_gemm_workspace(0, 1, B, 4 * Hd, Hd));
    up(rlt_gemm_workspace(0, 0, B, Hd, 4 * Hd));
    w.ws_bytes = m;
    w.ws = cur.take(m);
    w.bytes = cur.off;
    return w;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// one time step of both directions: position s_dir[d] of direction d.  gates (T, 2, 4, Hd) pre-activations in, activated out.
__global__ __launch_bounds__(256) void cell_fwd_kernel(float* __restrict__ gates, float* __restrict__ c, float* __restrict__ h,
                                                       int B, int Hd, int s_f, int s_r, int p_f, int p_r, int first) {
    const int n = 2 * B * Hd;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        const int u = idx % Hd, b = (idx / Hd) % B, d = idx / (Hd * B);
        const size_t t = (size_t)(d ? s_r : s_f) * B + b, tp = (size_t)(d ? p_r : p_f) * B + b;
        float* gp = gates + t * 8 * Hd + (size_t)d * 4 * Hd + u;
        const float gi = sigmoidf_(gp[0]), gf = sigmoidf_(gp[Hd]), gg = tanhf(gp[2 * Hd]), go = sigmoidf_(gp[3 * Hd]);
        const float cp = first ? 0.f : c[tp * 2 * Hd + d * Hd + u];
        const float cn = gf * cp + gi * gg;
        gp[0] = gi; gp[Hd] = gf; gp[2 * Hd] = gg; gp[3 * Hd] = go;
        c[t * 2 * Hd + d * Hd + u] = cn;
        h[t * 2 * Hd + d * Hd + u] = go * tanhf(cn);
    }
}

// reverse of one time step: gates (activated) -> d(pre-activations) in place; dc <- gradient of c_{t-1}
__global__ __launch_bounds__(256) void cell_bwd_kernel(float* __restrict__ gates, const float* __restrict__ c, const float* __restrict__ dh,
                                                       const float* __restrict__ dhr, float* __restrict__ dc,
                                                       int B, int Hd, int s_f, int s_r, int p_f, int p_r, int first, int last) {
    const int n = 2 * B * Hd;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        const int u = idx % Hd, b = (idx / Hd) % B, d = idx / (Hd * B);
        const size_t t = (size_t)(d ? s_r : s_f) * B + b, tp = (size_t)(d ? p_r : p_f) * B + b;
        float* gp = gates + t * 8 * Hd + (size_t)d * 4 * Hd + u;
        const float gi = gp[0], gf = gp[Hd], gg = gp[2 * Hd], go = gp[3 * Hd];
        const float ct = c[t * 2 * Hd + d * Hd + u];
        const float cp = first ? 0.f : c[tp * 2 * Hd + d * Hd + u];
        const float th = tanhf(ct);
        const float dht = dh[t * 2 * Hd + d * Hd + u] + (last ? 0.f : dhr[((size_t)d * B + b) * Hd + u]);
        const float dct = dht * go * (1.f - th * th) + (last ? 0.f : dc[((size_t)d * B + b) * Hd + u]);
        gp[0] = dct * gg * gi * (1.f - gi);
        gp[Hd] = dct * cp * gf * (1.f - gf);
        gp[2 * Hd] = dct * gi * (1.f - gg * gg);
        gp[3 * Hd] = dht * th * go * (1.f - go);
        dc[((size_t)d * B + b) * Hd + u] = dct * gf;
    }
}

__global__ __launch_bounds__(256) void pack2g_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < 2 * n; i += (size_t)gridDim.x * 256) out[i] = i < n ? a[i] : b[i - n];
}
__global__ __launch_bounds__(256) void unpack2g_kernel(const float* __restrict__ in, size_t n, float* __restrict__ a, float* __restrict__ b) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < 2 * n; i += (size_t)gridDim.x * 256) {
        if (i < n) a[i] = in[i]; else b[i - n] = in[i];
    }
}
int pack2g(const float* a, const float* b, size_t n, float* out, hipStream_t st) {
    hipLaunchKernelGGL(pack2g_kernel, dim3((int)((2 * n + 255) / 256 < 512 ? (2 * n + 255) / 256 : 512)), dim3(256), 0, st, a, b, n, out);
    return RLT_LAUNCH_RESULT();
}
int unpack2g(const float* in, size_t n, float* a, float* b, hipStream_t st) {
    hipLaunchKernelGGL(unpack2g_kernel, dim3((int)((2 * n + 255) / 256 < 512 ? (2 * n + 255) / 256 : 512)), dim3(256), 0, st, in, n, a, b);
    return RLT_LAUNCH_RESULT();
}
int gemm(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* Bm, int ldb, float* C, int ldc,
         int flags, float* colsum_a, const GScratch& k, void* stream) {
    return rlt_gemm_ex(ta, tb, M, N, K, A, lda, Bm, ldb, C, ldc, nullptr, nullptr, flags, nullptr, 0, 1.f, colsum_a, 0.f, 0u,
                       k.ws, k.ws_bytes, RLT_PRECISION_DEFAULT, stream);
}
inline int cell_grid(int B, int Hd) { const int g = (2 * B * Hd + 255) / 256; return g < 2048 ? g : 2048; }

int layer_fwd(const float* x, int I, int Hd, const rlt_lstm_layer_weights& w, int S, int B, float* gates, float* c, float* h,
              const GScratch& k, void* stream) {
    hipStream_t st = rlt_stream(stream);
    const int T = S * B, G4 = 4 * Hd, G8 = 8 * Hd;
    RLT_TRY(pack2g(w.w_ih[0], w.w_ih[1], (size_t)G4 * I, k.wcat, st));
    RLT_TRY(pack2g(w.b_ih[0], w.b_ih[1], G4, k.bcat, st));
    RLT_TRY(pack2g(w.b_hh[0], w.b_hh[1], G4, k.bcat + G8, st));
    RLT_TRY(rlt_gemm_ex(0, 1, T, G8, I, x, I, k.wcat, I, gates, G8, k.bcat, k.bcat + G8, 0, nullptr, 0, 1.f, nullptr, 0.f, 0u,
                        k.ws, k.ws_bytes, RLT_PRECISION_DEFAULT, stream));
    for (int step = 0; step < S; ++step) {
        const int s_f = step, s_r = S - 1 - step;
        if (step > 0) {         // pre-activations += h_{t-1} W_hh^T, per direction
            RLT_TRY(gemm(0, 1, B, G4, Hd, h + (size_t)(s_f - 1) * B * 2 * Hd, 2 * Hd, w.w_hh[0], Hd,
                         gates + (size_t)s_f * B * G8, G8, RLT_GEMM_ACCUMULATE, nullptr, k, stream));
            RLT_TRY(gemm(0, 1, B, G4, Hd, h + (size_t)(s_r + 1) * B * 2 * Hd + Hd, 2 * Hd, w.w_hh[1], Hd,
                         gates + (size_t)s_r * B * G8 + G4, G8, RLT_GEMM_ACCUMULATE, nullptr, k, stream));
        }
        hipLaunchKernelGGL(cell_fwd_kernel, dim3(cell_grid(B, Hd)), dim3(256), 0, st, gates, c, h, B, Hd, s_f, s_r, s_f - 1, s_r + 1,
                           step == 0 ? 1 : 0);
    }
    return RLT_LAUNCH_RESULT();
}

int layer_bwd(const float* x, int I, int Hd, const rlt_lstm_layer_weights& w, const float* h, float* gates, const float* c,
              const float* dh, int S, int B, float* dx, const rlt_lstm_layer_grads& g, const GScratch& k, void* stream) {
    hipStream_t st = rlt_stream(stream);
    const int T = S * B, G4 = 4 * Hd, G8 = 8 * Hd;
    for (int step = S - 1; step >= 0; --step) {
        const int s_f = step, s_r = S - 1 - step;
        hipLaunchKernelGGL(cell_bwd_kernel, dim3(cell_grid(B, Hd)), dim3(256), 0, st, gates, c, dh, k.dhr, k.dc, B, Hd, s_f, s_r,
                           s_f - 1, s_r + 1, step == 0 ? 1 : 0, step == S - 1 ? 1 : 0);
        if (step > 0) {         // dh_{t-1} (recurrent part) = dA_t W_hh, per direction
            RLT_TRY(gemm(0, 0, B, Hd, G4, gates + (size_t)s_f * B * G8, G8, w.w_hh[0], Hd, k.dhr, Hd, 0, nullptr, k, stream));
            RLT_TRY(gemm(0, 0, B, Hd, G4, gates + (size_t)s_r * B * G8 + G4, G8, w.w_hh[1], Hd, k.dhr + (size_t)B * Hd, Hd, 0, nullptr, k, stream));
        }
    }
    const float* dA = gates;
    RLT_TRY(gemm(1, 0, G8, I, T, dA, G8, x, I, k.dwcat, I, 0, k.dbcat, k, stream));
    RLT_TRY(unpack2g(k.dwcat, (size_t)G4 * I, g.w_ih[0], g.w_ih[1], st));
    RLT_TRY(unpack2g(k.dbcat, G4, g.b_ih[0], g.b_ih[1], st));
    RLT_TRY(unpack2g(k.dbcat, G4, g.b_hh[0], g.b_hh[1], st));
    if (S > 1) {
        const int K = T - B;
        RLT_TRY(gemm(1, 0, G4, Hd, K, dA + (size_t)B * G8, G8, h, 2 * Hd, g.w_hh[0], Hd, 0, nullptr, k, stream));
        RLT_TRY(gemm(1, 0, G4, Hd, K, dA + G4, G8, h + (size_t)B * 2 * Hd + Hd, 2 * Hd, g.w_hh[1], Hd, 0, nullptr, k, stream));
    } else {
        RLT_TRY((int)hipMemsetAsync(g.w_hh[0], 0, (size_t)G4 * Hd * sizeof(float), st));
        RLT_TRY((int)hipMemsetAsync(g.w_hh[1], 0, (size_t)G4 * Hd * sizeof(float), st));
    }
    if (dx) {
        RLT_TRY(pack2g(w.w_ih[0], w.w_ih[1], (size_t)G4 * I, k.wcat, st));
        RLT_TRY(gemm(0, 0, T, I, G8, dA, G8, k.wcat, I, dx, I, 0, nullptr, k, stream));
    }
    return RLT_LAUNCH_RESULT();
}

bool wok(const rlt_lstm_layer_weights& w) {
    for (int d = 0; d < 2; ++d) if (!w.w_ih[d] || !w.w_hh[d] || !w.b_ih[d] || !w.b_hh[d]) return false;
    return true;
}
bool gok(const rlt_lstm_layer_grads& g) {
    for (int d = 0; d < 2; ++d) if (!g.w_ih[d] || !g.w_hh[d] || !g.b_ih[d] || !g.b_hh[d]) return false;
    return true;
}

}  // namespace

extern "C" {

size_t rlt_bilstm_generic_bytes(int stash, int S, int B, int I, int hidden) {
    if (S <= 0 || B <= 0 || I <= 0 || hidden <= 0) return 0;
    const size_t T = (size_t)S * B;
    return stash ? g_stash(T, hidden, nullptr).bytes : g_scratch(T, B, I, hidden, nullptr).bytes;
}

int rlt_bilstm_generic_fwd(const float* x, int I, int hidden, const rlt_lstm_layer_weights* w, int S, int B, float* h_out,
                           void* stash, size_t stash_bytes, void* ws, size_t ws_bytes, int precision, void* stream) {
    RLT_PREC_SCOPE(precision);
    RLT_CHECK_ARG(x && w && h_out && stash && ws && I > 0 && hidden > 0 && S > 0 && B > 0 && wok(w[0]) && wok(w[1]));
    const size_t T = (size_t)S * B;
    RLT_CHECK_SHAPE(T * 8 * hidden <= 0x7fffffffu);
    const GStash s = g_stash(T, hidden, stash);
    const GScratch k = g_scratch(T, B, I, hidden, ws);
    if (stash_bytes < s.bytes || ws_bytes < k.bytes) return RLT_E_WORKSPACE;
    RLT_TRY(layer_fwd(x, I, hidden, w[0], S, B, s.gates[0], s.c[0], s.h0, k, stream));
    return layer_fwd(s.h0, 2 * hidden, hidden, w[1], S, B, s.gates[1], s.c[1], h_out, k, stream);
}

int rlt_bilstm_generic_bwd(const float* x, int I, int hidden, const rlt_lstm_layer_weights* w, const float* h_out,
                           const float* dh_out, int S, int B, void* stash, size_t stash_bytes, float* dx,
                           const rlt_lstm_layer_grads* g, void* ws, size_t ws_bytes, int precision, void* stream) {
    RLT_PREC_SCOPE(precision);
    RLT_CHECK_ARG(x && w && h_out && dh_out && stash && g && ws && I > 0 && hidden > 0 && S > 0 && B > 0);
    RLT_CHECK_ARG(wok(w[0]) && wok(w[1]) && gok(g[0]) && gok(g[1]));
    const size_t T = (size_t)S * B;
    RLT_CHECK_SHAPE(T * 8 * hidden <= 0x7fffffffu);
    const GStash s = g_stash(T, hidden, stash);
    const GScratch k = g_scratch(T, B, I, hidden, ws);
    if (stash_bytes < s.bytes || ws_bytes < k.bytes) return RLT_E_WORKSPACE;
    RLT_TRY(layer_bwd(s.h0, 2 * hidden, hidden, w[1], h_out, s.gates[1], s.c[1], dh_out, S, B, k.dh0, g[1], k, stream));
    return layer_bwd(x, I, hidden, w[0], s.h0, s.gates[0], s.c[0], k.dh0, S, B, dx, g[0], k, stream);
}

}  // extern "C"

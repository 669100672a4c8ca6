// Residual add + LayerNorm, forward and backward: the two post-norm blocks of
// nn.TransformerEncoderLayer (models/AttnCut.py:9; x -> LN(x + sublayer(x)), eps 1e-5, biased
// variance).  HBM-bound: one token row per wavefront, 16-byte loads, statistics by wavefront
// shuffles, the residual add fused in.  Backward emits the single gradient dz that both the
// residual branch and the sublayer output receive, and per-workgroup partial sums for
// dgamma/dbeta that a second tiny kernel adds in a fixed order (deterministic, no atomics).
#include "common.h"

namespace {

constexpr int MAXCH = 4;

template <int V> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<2> { typedef float2 T; };
template <> struct VecT<4> { typedef float4 T; };

template <int V>
__device__ __forceinline__ void ldv(const float* p, float (&d)[V]) {
    typename VecT<V>::T v = *reinterpret_cast<const typename VecT<V>::T*>(p);
    const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
    for (int i = 0; i < V; ++i) d[i] = f[i];
}
template <int V>
__device__ __forceinline__ void stv(float* p, const float (&d)[V]) {
    typename VecT<V>::T v;
    float* f = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int i = 0; i < V; ++i) f[i] = d[i];
    *reinterpret_cast<typename VecT<V>::T*>(p) = v;
}

template <int V>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int T, int E, float eps, float* __restrict__ y,
                                                         float* __restrict__ stats, float drop_p, uint32_t seed) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nch = E / (64 * V);
    float gm[MAXCH][V], bt[MAXCH][V];
#pragma unroll
    for (int c = 0; c < MAXCH; ++c)
        if (c < nch) { ldv<V>(gamma + c * 64 * V + lane * V, gm[c]); ldv<V>(beta + c * 64 * V + lane * V, bt[c]); }
    const float inv_e = 1.f / (float)E;
    const bool drop = drop_p > 0.f;
    const uint32_t thr = rlt_drop_threshold(drop_p);
    const float keep_scale = drop ? 1.f / (1.f - drop_p) : 1.f;
    uint32_t hc[MAXCH][V];                   // dropout: this lane's column hashes (the token's row hash is wave-uniform)
#pragma unroll
    for (int c = 0; c < MAXCH; ++c)
#pragma unroll
        for (int i = 0; i < V; ++i) hc[c][i] = (drop && c < nch) ? rlt_col_hash(seed, (uint32_t)(c * 64 * V + lane * V + i)) : 1u;
    for (int t = blockIdx.x * 4 + wv; t < T; t += gridDim.x * 4) {
        const size_t row = (size_t)t * E;
        const uint32_t hr = rlt_row_hash(seed, (uint32_t)t);
        float z[MAXCH][V];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c)
            if (c < nch) {
                const int off = c * 64 * V + lane * V;
                ldv<V>(x + row + off, z[c]);
                if (r) {
                    float rr[V];
                    ldv<V>(r + row + off, rr);
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        float rv = rr[i];
                        if (drop) rv = rlt_keep_rc(hr, hc[c][i], thr) ? rv * keep_scale : 0.f;
                        z[c][i] += rv;
                    }
                }
#pragma unroll
                for (int i = 0; i < V; ++i) s += z[c][i];
            }
        const float mean = wave_sum(s) * inv_e;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c)
            if (c < nch) {
#pragma unroll
                for (int i = 0; i < V; ++i) { const float d = z[c][i] - mean; q += d * d; }
            }
        const float var = wave_sum(q) * inv_e;
        const float rstd = 1.f / sqrtf(var + eps);
#pragma unroll
        for (int c = 0; c < MAXCH; ++c)
            if (c < nch) {
                float o[V];
#pragma unroll
                for (int i = 0; i < V; ++i) o[i] = (z[c][i] - mean) * rstd * gm[c][i] + bt[c][i];
                stv<V>(y + row + c * 64 * V + lane * V, o);
            }
        if (lane == 0 && stats) { stats[2 * (size_t)t] = mean; stats[2 * (size_t)t + 1] = rstd; }
    }
}

template <int V>
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                         const float* __restrict__ gamma, const float* __restrict__ stats,
                                                         const float* __restrict__ dy, int T, int E,
                                                         float* __restrict__ dz, float* __restrict__ partial,
                                                         float* __restrict__ dr, float drop_p, uint32_t seed) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][2E]
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nch = E / (64 * V);
    float gm[MAXCH][V], dg[MAXCH][V], db[MAXCH][V];
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        if (c < nch) ldv<V>(gamma + c * 64 * V + lane * V, gm[c]);
#pragma unroll
        for (int i = 0; i < V; ++i) { dg[c][i] = 0.f; db[c][i] = 0.f; }
    }
    const float inv_e = 1.f / (float)E;
    const bool drop = drop_p > 0.f;
    const uint32_t thr = rlt_drop_threshold(drop_p);
    const float keep_scale = drop ? 1.f / (1.f - drop_p) : 1.f;
    uint32_t hc[MAXCH][V];                   // dropout: this lane's column hashes (the token's row hash is wave-uniform)
#pragma unroll
    for (int c = 0; c < MAXCH; ++c)
#pragma unroll
        for (int i = 0; i < V; ++i) hc[c][i] = (drop && c < nch) ? rlt_col_hash(seed, (uint32_t)(c * 64 * V + lane * V + i)) : 1u;
    for (int t = blockIdx.x * 4 + wv; t < T; t += gridDim.x * 4) {
        const size_t row = (size_t)t * E;
        const uint32_t hr = rlt_row_hash(seed, (uint32_t)t);
        const float mean = stats[2 * (size_t)t], rstd = stats[2 * (size_t)t + 1];
        float xh[MAXCH][V], gy[MAXCH][V];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c)
            if (c < nch) {
                const int off = c * 64 * V + lane * V;
                float z[V], d[V];
                ldv<V>(x + row + off, z);
                if (r) {
                    float rr[V];
                    ldv<V>(r + row + off, rr);
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        float rv = rr[i];
                        if (drop) rv = rlt_keep_rc(hr, hc[c][i], thr) ? rv * keep_scale : 0.f;
                        z[i] += rv;
                    }
                }
                ldv<V>(dy + row + off, d);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    xh[c][i] = (z[i] - mean) * rstd;
                    gy[c][i] = d[i] * gm[c][i];
                    s1 += gy[c][i];
                    s2 += gy[c][i] * xh[c][i];
                    dg[c][i] += d[i] * xh[c][i];
                    db[c][i] += d[i];
                }
            }
        const float m1 = wave_sum(s1) * inv_e, m2 = wave_sum(s2) * inv_e;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c)
            if (c < nch) {
                float o[V];
#pragma unroll
                for (int i = 0; i < V; ++i) o[i] = rstd * (gy[c][i] - m1 - xh[c][i] * m2);
                stv<V>(dz + row + c * 64 * V + lane * V, o);
                if (dr) {       // gradient of the dropped branch: dz * keep / (1-p)
                    const int off = c * 64 * V + lane * V;
#pragma unroll
                    for (int i = 0; i < V; ++i)
                        o[i] = rlt_keep_rc(hr, hc[c][i], thr) ? o[i] * keep_scale : 0.f;
                    stv<V>(dr + row + off, o);
                }
            }
    }
    // per-workgroup partial sums of dgamma | dbeta
#pragma unroll
    for (int c = 0; c < MAXCH; ++c)
        if (c < nch) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int col = c * 64 * V + lane * V + i;
                sm[wv * 2 * E + col] = dg[c][i];
                sm[wv * 2 * E + E + col] = db[c][i];
            }
        }
    __syncthreads();
    for (int col = threadIdx.x; col < 2 * E; col += 256)
        partial[(size_t)blockIdx.x * 2 * E + col] = sm[col] + sm[2 * E + col] + sm[4 * E + col] + sm[6 * E + col];
}

int ln_grid(int T) { const int g = rlt_cdiv(T, 4); return g > 2048 ? 2048 : g; }
int pick_v(int E) { return (E % 256 == 0) ? 4 : ((E % 128 == 0) ? 2 : 1); }

}  // namespace

extern "C" {

int rlt_add_layernorm_fwd(const float* x, const float* r, const float* gamma, const float* beta,
                          int T, int E, float eps, float drop_p, uint32_t seed,
                          float* y, float* stats, void* stream) {
    RLT_CHECK_ARG(x && gamma && beta && y && T > 0 && E > 0);
    RLT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
    const int V = pick_v(E);
    RLT_CHECK_SHAPE(E % 64 == 0 && E / (64 * V) <= MAXCH);
    if (!(rlt_aligned16(x) && rlt_aligned16(y) && (!r || rlt_aligned16(r)) && rlt_aligned16(gamma) && rlt_aligned16(beta)))
        return RLT_E_ALIGN;
    hipStream_t st = rlt_stream(stream);
    dim3 grid(ln_grid(T)), block(256);
    if (V == 4) hipLaunchKernelGGL(add_ln_fwd_kernel<4>, grid, block, 0, st, x, r, gamma, beta, T, E, eps, y, stats, drop_p, seed);
    else if (V == 2) hipLaunchKernelGGL(add_ln_fwd_kernel<2>, grid, block, 0, st, x, r, gamma, beta, T, E, eps, y, stats, drop_p, seed);
    else hipLaunchKernelGGL(add_ln_fwd_kernel<1>, grid, block, 0, st, x, r, gamma, beta, T, E, eps, y, stats, drop_p, seed);
    return RLT_LAUNCH_RESULT();
}

size_t rlt_add_layernorm_bwd_workspace(int T, int E) {
    if (T <= 0 || E <= 0) return 0;
    return (size_t)ln_grid(T) * 2 * E * sizeof(float);
}

int rlt_add_layernorm_bwd(const float* x, const float* r, const float* gamma, const float* stats,
                          const float* dy, int T, int E, float drop_p, uint32_t seed,
                          float* dz, float* dr, float* dgamma, float* dbeta,
                          int accumulate, void* ws, size_t ws_bytes, void* stream) {
    RLT_CHECK_ARG(x && gamma && stats && dy && dz && dgamma && dbeta && ws && T > 0 && E > 0);
    RLT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || (r && dr)));
    if (drop_p == 0.f) dr = nullptr;
    const int V = pick_v(E);
    RLT_CHECK_SHAPE(E % 64 == 0 && E / (64 * V) <= MAXCH);
    if (ws_bytes < rlt_add_layernorm_bwd_workspace(T, E)) return RLT_E_WORKSPACE;
    if (!(rlt_aligned16(x) && rlt_aligned16(dy) && rlt_aligned16(dz) && (!r || rlt_aligned16(r)) && rlt_aligned16(gamma)))
        return RLT_E_ALIGN;
    hipStream_t st = rlt_stream(stream);
    const int nblk = ln_grid(T);
    dim3 grid(nblk), block(256);
    const size_t shm = (size_t)8 * E * sizeof(float);
    float* part = (float*)ws;
    if (V == 4) hipLaunchKernelGGL(add_ln_bwd_kernel<4>, grid, block, shm, st, x, r, gamma, stats, dy, T, E, dz, part, dr, drop_p, seed);
    else if (V == 2) hipLaunchKernelGGL(add_ln_bwd_kernel<2>, grid, block, shm, st, x, r, gamma, stats, dy, T, E, dz, part, dr, drop_p, seed);
    else hipLaunchKernelGGL(add_ln_bwd_kernel<1>, grid, block, shm, st, x, r, gamma, stats, dy, T, E, dz, part, dr, drop_p, seed);
    hipLaunchKernelGGL(rlt_rows_reduce_kernel, dim3(rlt_cdiv(2 * E, 16)), dim3(256), 0, st, (const float*)part, nblk, 2 * E,
                       2 * E, E, dgamma, dbeta, accumulate);
    return RLT_LAUNCH_RESULT();
}

}  // extern "C"

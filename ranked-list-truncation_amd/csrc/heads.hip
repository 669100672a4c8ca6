// Decision heads and layout helpers (rows M4, M6 of SURVEY.md section 8a).
//
// Up to three Linear(E,1) heads over the same encoder output, each followed by a softmax over
// the S positions of one list, a sigmoid, or nothing (models/AttnCut.py:11-14,19,
// models/MtAttnCut.py:11-19,24-26, models/MMOECut.py:17-53).  One workgroup per ranked list:
// its S token rows (E contiguous floats each, position-major) are streamed once with 16-byte
// loads, the E-wide dot products are wavefront shuffle reductions, the S logits of every head
// live in LDS where the softmax over positions is taken, and the (B,S) outputs are written
// coalesced.  x is read exactly once in forward and once in backward (where dx is written once):
// HBM-bound, 4*E bytes per token per pass.
#include "common.h"

namespace {

constexpr int MAXH = 3, MAXCH = 4, MAXS = 1024;

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

struct HeadKinds { int k[MAXH]; };

template <int V>
__global__ __launch_bounds__(256) void heads_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, HeadKinds kinds, int nh,
                                                        int S, int B, int E, float* __restrict__ out) {
    __shared__ float z[MAXH][MAXS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const int nch = E / (64 * V);
    float wr[MAXH][MAXCH][V];
#pragma unroll
    for (int h = 0; h < MAXH; ++h)
#pragma unroll
        for (int c = 0; c < MAXCH; ++c)
            if (h < nh && c < nch) ldv<V>(w + (size_t)h * E + c * 64 * V + lane * V, wr[h][c]);
    for (int s = wv; s < S; s += 4) {
        const float* row = x + ((size_t)s * B + b) * E;
        float acc[MAXH] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < MAXCH; ++c)
            if (c < nch) {
                float xv[V];
                ldv<V>(row + c * 64 * V + lane * V, xv);
#pragma unroll
                for (int h = 0; h < MAXH; ++h)
                    if (h < nh) {
#pragma unroll
                        for (int i = 0; i < V; ++i) acc[h] += xv[i] * wr[h][c][i];
                    }
            }
#pragma unroll
        for (int h = 0; h < MAXH; ++h)
            if (h < nh) {
                const float t = wave_sum(acc[h]);
                if (lane == 0) z[h][s] = t + bias[h];
            }
    }
    __syncthreads();
    // wavefront h finishes head h
    if (wv < nh) {
        const int h = wv, kind = kinds.k[h];
        float* dst = out + ((size_t)h * B + b) * S;
        if (kind == RLT_HEAD_SOFTMAX) {
            float m = -INFINITY;
            for (int s = lane; s < S; s += 64) m = fmaxf(m, z[h][s]);
            m = wave_max(m);
            float sum = 0.f;
            for (int s = lane; s < S; s += 64) { const float e = expf(z[h][s] - m); z[h][s] = e; sum += e; }
            sum = wave_sum(sum);
            for (int s = lane; s < S; s += 64) dst[s] = z[h][s] / sum;
        } else if (kind == RLT_HEAD_SIGMOID) {
            for (int s = lane; s < S; s += 64) dst[s] = 1.f / (1.f + expf(-z[h][s]));
        } else {
            for (int s = lane; s < S; s += 64) dst[s] = z[h][s];
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void heads_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        HeadKinds kinds, int nh, const float* __restrict__ out,
                                                        const float* __restrict__ dout, int S, int B, int E,
                                                        float* __restrict__ dx, int accumulate_dx,
                                                        float* __restrict__ partial, int pw) {
    __shared__ float dz[MAXH][MAXS];
    extern __shared__ __attribute__((aligned(16))) float red[];   // [4][nh*E]
    __shared__ float dbs[MAXH];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const int nch = E / (64 * V);
    // ---- d(logit) per head ----------------------------------------------------------------------
    if (wv < nh) {
        const int h = wv, kind = kinds.k[h];
        const float* y = out + ((size_t)h * B + b) * S;
        const float* g = dout + ((size_t)h * B + b) * S;
        float dot = 0.f;
        if (kind == RLT_HEAD_SOFTMAX) {
            for (int s = lane; s < S; s += 64) dot += g[s] * y[s];
            dot = wave_sum(dot);
        }
        float bsum = 0.f;
        for (int s = lane; s < S; s += 64) {
            const float yv = y[s], gv = g[s];
            float d;
            if (kind == RLT_HEAD_SOFTMAX) d = yv * (gv - dot);
            else if (kind == RLT_HEAD_SIGMOID) d = gv * yv * (1.f - yv);
            else d = gv;
            dz[h][s] = d;
            bsum += d;
        }
        bsum = wave_sum(bsum);
        if (lane == 0) dbs[h] = bsum;
    }
    __syncthreads();
    float wr[MAXH][MAXCH][V], dw[MAXH][MAXCH][V];
#pragma unroll
    for (int h = 0; h < MAXH; ++h)
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            if (h < nh && c < nch) ldv<V>(w + (size_t)h * E + c * 64 * V + lane * V, wr[h][c]);
#pragma unroll
            for (int i = 0; i < V; ++i) dw[h][c][i] = 0.f;
        }
    for (int s = wv; s < S; s += 4) {
        const size_t roff = ((size_t)s * B + b) * E;
        float d[MAXH];
#pragma unroll
        for (int h = 0; h < MAXH; ++h) d[h] = h < nh ? dz[h][s] : 0.f;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c)
            if (c < nch) {
                const int off = c * 64 * V + lane * V;
                float xv[V], o[V];
                ldv<V>(x + roff + off, xv);
                if (accumulate_dx) ldv<V>(dx + roff + off, o);
                else {
#pragma unroll
                    for (int i = 0; i < V; ++i) o[i] = 0.f;
                }
#pragma unroll
                for (int h = 0; h < MAXH; ++h)
                    if (h < nh) {
#pragma unroll
                        for (int i = 0; i < V; ++i) {
                            o[i] += d[h] * wr[h][c][i];
                            dw[h][c][i] += d[h] * xv[i];
                        }
                    }
                stv<V>(dx + roff + off, o);
            }
    }
#pragma unroll
    for (int h = 0; h < MAXH; ++h)
#pragma unroll
        for (int c = 0; c < MAXCH; ++c)
            if (h < nh && c < nch) {
#pragma unroll
                for (int i = 0; i < V; ++i) red[(size_t)wv * nh * E + h * E + c * 64 * V + lane * V + i] = dw[h][c][i];
            }
    __syncthreads();
    const int nhe = nh * E;
    float* prow = partial + (size_t)b * pw;
    for (int col = threadIdx.x; col < nhe; col += 256)
        prow[col] = red[col] + red[nhe + col] + red[2 * nhe + col] + red[3 * nhe + col];
    if (threadIdx.x < nh) prow[nhe + threadIdx.x] = dbs[threadIdx.x];
}

__global__ __launch_bounds__(256) void to_pm_kernel(const float* __restrict__ in, int B, int S, int F,
                                                    float* __restrict__ out, int reverse) {
    const size_t n = (size_t)B * S * F;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        // i indexes the position-major tensor (s,b,f)
        const int f = (int)(i % F);
        const size_t tok = i / F;
        const int b = (int)(tok % B), s = (int)(tok / B);
        const size_t j = ((size_t)b * S + s) * F + f;
        if (reverse) out[j] = in[i]; else out[i] = in[j];
    }
}

__global__ __launch_bounds__(256) void choopy_embed_kernel(const float* __restrict__ score, const float* __restrict__ pe,
                                                           int B, int S, int E, float* __restrict__ out) {
    const size_t n = (size_t)B * S * E;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % E);
        const size_t tok = i / E;
        const int b = (int)(tok % B), s = (int)(tok / B);
        out[i] = (c == 0) ? score[(size_t)b * S + s] : pe[(size_t)s * (E - 1) + (c - 1)];
    }
}

int ew_grid(size_t n) { size_t g = (n + 1023) / 1024; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }
int pick_v(int E) { return (E % 256 == 0) ? 4 : ((E % 128 == 0) ? 2 : 1); }

}  // namespace

extern "C" {

int rlt_heads_fwd(const float* x, const float* w, const float* b, const int* kinds, int n_heads,
                  int S, int B, int E, float* out, void* stream) {
    RLT_CHECK_ARG(x && w && b && kinds && out && S > 0 && B > 0 && E > 0);
    RLT_CHECK_SHAPE(n_heads >= 1 && n_heads <= MAXH && S <= MAXS);
    const int V = pick_v(E);
    RLT_CHECK_SHAPE(E % 64 == 0 && E / (64 * V) <= MAXCH);
    if (!(rlt_aligned16(x) && rlt_aligned16(w))) return RLT_E_ALIGN;
    HeadKinds hk;
    for (int i = 0; i < MAXH; ++i) hk.k[i] = i < n_heads ? kinds[i] : RLT_HEAD_IDENTITY;
    hipStream_t st = rlt_stream(stream);
    dim3 grid(B), block(256);
    if (V == 4) hipLaunchKernelGGL(heads_fwd_kernel<4>, grid, block, 0, st, x, w, b, hk, n_heads, S, B, E, out);
    else if (V == 2) hipLaunchKernelGGL(heads_fwd_kernel<2>, grid, block, 0, st, x, w, b, hk, n_heads, S, B, E, out);
    else hipLaunchKernelGGL(heads_fwd_kernel<1>, grid, block, 0, st, x, w, b, hk, n_heads, S, B, E, out);
    return RLT_LAUNCH_RESULT();
}

size_t rlt_heads_bwd_workspace(int n_heads, int S, int B, int E) {
    (void)S;
    if (n_heads <= 0 || B <= 0 || E <= 0) return 0;
    return (size_t)B * (n_heads * E + 4) * sizeof(float);
}

int rlt_heads_bwd(const float* x, const float* w, const int* kinds, int n_heads,
                  const float* out, const float* dout, int S, int B, int E,
                  float* dx, int accumulate_dx, float* dw, float* db,
                  void* ws, size_t ws_bytes, void* stream) {
    RLT_CHECK_ARG(x && w && kinds && out && dout && dx && dw && db && ws && S > 0 && B > 0 && E > 0);
    RLT_CHECK_SHAPE(n_heads >= 1 && n_heads <= MAXH && S <= MAXS);
    const int V = pick_v(E);
    RLT_CHECK_SHAPE(E % 64 == 0 && E / (64 * V) <= MAXCH);
    if (ws_bytes < rlt_heads_bwd_workspace(n_heads, S, B, E)) return RLT_E_WORKSPACE;
    if (!(rlt_aligned16(x) && rlt_aligned16(w) && rlt_aligned16(dx))) return RLT_E_ALIGN;
    HeadKinds hk;
    for (int i = 0; i < MAXH; ++i) hk.k[i] = i < n_heads ? kinds[i] : RLT_HEAD_IDENTITY;
    hipStream_t st = rlt_stream(stream);
    const int pw = n_heads * E + 4;
    const size_t shm = (size_t)4 * n_heads * E * sizeof(float);
    dim3 grid(B), block(256);
    float* part = (float*)ws;
    if (V == 4) hipLaunchKernelGGL(heads_bwd_kernel<4>, grid, block, shm, st, x, w, hk, n_heads, out, dout, S, B, E, dx, accumulate_dx, part, pw);
    else if (V == 2) hipLaunchKernelGGL(heads_bwd_kernel<2>, grid, block, shm, st, x, w, hk, n_heads, out, dout, S, B, E, dx, accumulate_dx, part, pw);
    else hipLaunchKernelGGL(heads_bwd_kernel<1>, grid, block, shm, st, x, w, hk, n_heads, out, dout, S, B, E, dx, accumulate_dx, part, pw);
    hipLaunchKernelGGL(rlt_rows_reduce_kernel, dim3(rlt_cdiv(n_heads * E + n_heads, 16)), dim3(256), 0, st, (const float*)part, B, pw,
                       n_heads * E + n_heads, n_heads * E, dw, db, 0);
    return RLT_LAUNCH_RESULT();
}

int rlt_to_position_major(const float* x_bsf, int B, int S, int F, float* x_sbf, void* stream) {
    RLT_CHECK_ARG(x_bsf && x_sbf && B > 0 && S > 0 && F > 0);
    hipLaunchKernelGGL(to_pm_kernel, dim3(ew_grid((size_t)B * S * F)), dim3(256), 0, rlt_stream(stream), x_bsf, B, S, F, x_sbf, 0);
    return RLT_LAUNCH_RESULT();
}

int rlt_from_position_major(const float* x_sbf, int B, int S, int F, float* x_bsf, void* stream) {
    RLT_CHECK_ARG(x_bsf && x_sbf && B > 0 && S > 0 && F > 0);
    hipLaunchKernelGGL(to_pm_kernel, dim3(ew_grid((size_t)B * S * F)), dim3(256), 0, rlt_stream(stream), x_sbf, B, S, F, x_bsf, 1);
    return RLT_LAUNCH_RESULT();
}

int rlt_choopy_embed(const float* score_bs, const float* pe, int B, int S, int E, float* out, void* stream) {
    RLT_CHECK_ARG(score_bs && pe && out && B > 0 && S > 0 && E > 1);
    hipLaunchKernelGGL(choopy_embed_kernel, dim3(ew_grid((size_t)B * S * E)), dim3(256), 0, rlt_stream(stream), score_bs, pe, B, S, E, out);
    return RLT_LAUNCH_RESULT();
}

}  // extern "C"

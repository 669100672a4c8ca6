// MMOE gates and expert mixture (row M7 of SURVEY.md section 8a; models/MMOECut.py:86-110) and
// the fused Adam step (row N2; run.py:104,129).
//
//   gate[t][b][:]  = softmax_e( flatten_s(h[b]) @ w_gate[t] )          models/MMOECut.py:93-94
//   mixed[t][tok]  = sum_e gate[t][b][e] * expert[e][tok]               models/MMOECut.py:101-102
//
// All HBM-bound streaming work.  The gate product is a (B x S*C) x (S*C x n_e) contraction with
// n_e <= 8 output columns: far too skinny for MFMA tiles, so it is done as one workgroup per
// list streaming that list's S rows once (16-byte loads) against the L2-resident gate matrix,
// with wavefront shuffle reductions; dW_gate is the mirror-image (one workgroup per position,
// threads over the C columns, lists streamed).
#include "common.h"

namespace {

constexpr int MAXT = 3, MAXE = 8;

struct GatePtrs { const float* w[MAXT]; float* dw[MAXT]; };
struct ExpertPtrs { const float* x[MAXE]; float* dx[MAXE]; };

// ---------------------------------------------------------------- gate forward: one workgroup per list
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ h, GatePtrs gp, int nt, int ne,
                                                       int S, int B, int C, float* __restrict__ gates) {
    __shared__ float red[4][MAXT * MAXE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x;
    float acc[MAXT][MAXE];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int e = 0; e < MAXE; ++e) acc[t][e] = 0.f;
    for (int s = wv; s < S; s += 4) {
        const float* row = h + ((size_t)s * B + b) * C;
        for (int c = lane; c < C; c += 64) {
            const float x = row[c];
            const size_t wrow = ((size_t)s * C + c) * ne;
#pragma unroll
            for (int t = 0; t < MAXT; ++t)
                if (t < nt) {
#pragma unroll
                    for (int e = 0; e < MAXE; ++e)
                        if (e < ne) acc[t][e] += x * gp.w[t][wrow + e];
                }
        }
    }
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int e = 0; e < MAXE; ++e) {
            const float v = wave_sum(acc[t][e]);
            if (lane == 0) red[wv][t * MAXE + e] = v;
        }
    __syncthreads();
    if (threadIdx.x < nt) {
        const int t = threadIdx.x;
        float z[MAXE];
        float m = -INFINITY;
        for (int e = 0; e < ne; ++e) {
            z[e] = red[0][t * MAXE + e] + red[1][t * MAXE + e] + red[2][t * MAXE + e] + red[3][t * MAXE + e];
            m = fmaxf(m, z[e]);
        }
        float sum = 0.f;
        for (int e = 0; e < ne; ++e) { z[e] = expf(z[e] - m); sum += z[e]; }
        for (int e = 0; e < ne; ++e) gates[((size_t)t * B + b) * ne + e] = z[e] / sum;
    }
}

// dlogit[t][b][e] = g * (dg - sum_e' dg g)
__global__ __launch_bounds__(256) void gate_dlogit_kernel(const float* __restrict__ gates, const float* __restrict__ dgates,
                                                          int n, int ne, float* __restrict__ dlogit) {
    const int i = blockIdx.x * 256 + threadIdx.x;     // (t,b) pair
    if (i >= n) return;
    float dot = 0.f;
    for (int e = 0; e < ne; ++e) dot += gates[(size_t)i * ne + e] * dgates[(size_t)i * ne + e];
    for (int e = 0; e < ne; ++e) dlogit[(size_t)i * ne + e] = gates[(size_t)i * ne + e] * (dgates[(size_t)i * ne + e] - dot);
}

// dh[tok][c] += sum_t sum_e dlogit[t][b][e] * w[t][(s*C+c)*ne + e]; one workgroup per list
__global__ __launch_bounds__(256) void gate_dh_kernel(const float* __restrict__ dlogit, GatePtrs gp, int nt, int ne,
                                                      int S, int B, int C, float* __restrict__ dh, int accumulate) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x;
    float dl[MAXT][MAXE];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int e = 0; e < MAXE; ++e) dl[t][e] = (t < nt && e < ne) ? dlogit[((size_t)t * B + b) * ne + e] : 0.f;
    for (int s = wv; s < S; s += 4) {
        float* row = dh + ((size_t)s * B + b) * C;
        for (int c = lane; c < C; c += 64) {
            const size_t wrow = ((size_t)s * C + c) * ne;
            float v = accumulate ? row[c] : 0.f;
#pragma unroll
            for (int t = 0; t < MAXT; ++t)
                if (t < nt) {
#pragma unroll
                    for (int e = 0; e < MAXE; ++e)
                        if (e < ne) v += dl[t][e] * gp.w[t][wrow + e];
                }
            row[c] = v;
        }
    }
}

// dw[t][(s*C+c)*ne + e] = sum_b h[(s*B+b)*C + c] * dlogit[t][b][e]; workgroup = (position s, 256 columns)
__global__ __launch_bounds__(256) void gate_dw_kernel(const float* __restrict__ h, const float* __restrict__ dlogit,
                                                      GatePtrs gp, int nt, int ne, int S, int B, int C) {
    __shared__ float dls[64][MAXT * MAXE];
    const int s = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    float acc[MAXT][MAXE];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int e = 0; e < MAXE; ++e) acc[t][e] = 0.f;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int nb = min(64, B - b0);
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * MAXT * MAXE; i += 256) {
            const int bb = i / (MAXT * MAXE), te = i % (MAXT * MAXE), t = te / MAXE, e = te % MAXE;
            dls[bb][te] = (bb < nb && t < nt && e < ne) ? dlogit[((size_t)t * B + b0 + bb) * ne + e] : 0.f;
        }
        __syncthreads();
        if (c < C) {
            for (int bb = 0; bb < nb; ++bb) {
                const float x = h[((size_t)s * B + b0 + bb) * C + c];
#pragma unroll
                for (int t = 0; t < MAXT; ++t)
                    if (t < nt) {
#pragma unroll
                        for (int e = 0; e < MAXE; ++e)
                            if (e < ne) acc[t][e] += x * dls[bb][t * MAXE + e];
                    }
            }
        }
    }
    if (c < C) {
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (t < nt) {
#pragma unroll
                for (int e = 0; e < MAXE; ++e)
                    if (e < ne) gp.dw[t][((size_t)s * C + c) * ne + e] = acc[t][e];
            }
    }
}

// ---------------------------------------------------------------- mixture
__global__ __launch_bounds__(256) void mix_fwd_kernel(ExpertPtrs ep, const float* __restrict__ gates,
                                                      int nt, int ne, size_t T, int B, int E, float* __restrict__ mixed) {
    const size_t n4 = T * E / 4;
    const size_t TE = T * E;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const size_t tok = (i * 4) / E;
        const int b = (int)(tok % B);
        float4 x[MAXE];
#pragma unroll
        for (int e = 0; e < MAXE; ++e)
            if (e < ne) x[e] = *reinterpret_cast<const float4*>(ep.x[e] + i * 4);
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (t < nt) {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int e = 0; e < MAXE; ++e)
                    if (e < ne) {
                        const float g = gates[((size_t)t * B + b) * ne + e];
                        o.x += g * x[e].x; o.y += g * x[e].y; o.z += g * x[e].z; o.w += g * x[e].w;
                    }
                *reinterpret_cast<float4*>(mixed + (size_t)t * TE + i * 4) = o;
            }
    }
}

// one workgroup per list: dexperts rows written, dgates accumulated over the list's S rows
__global__ __launch_bounds__(256) void mix_bwd_kernel(ExpertPtrs ep, const float* __restrict__ gates,
                                                      const float* __restrict__ dmixed, int nt, int ne, int S, int B, int E,
                                                      float* __restrict__ dgates) {
    __shared__ float red[4][MAXT * MAXE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const size_t TE = (size_t)S * B * E;
    float g[MAXT][MAXE], dg[MAXT][MAXE];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int e = 0; e < MAXE; ++e) {
            g[t][e] = (t < nt && e < ne) ? gates[((size_t)t * B + b) * ne + e] : 0.f;
            dg[t][e] = 0.f;
        }
    for (int s = wv; s < S; s += 4) {
        const size_t roff = ((size_t)s * B + b) * E;
        for (int c = lane * 4; c < E; c += 256) {
            float4 dm[MAXT];
#pragma unroll
            for (int t = 0; t < MAXT; ++t)
                if (t < nt) dm[t] = *reinterpret_cast<const float4*>(dmixed + (size_t)t * TE + roff + c);
#pragma unroll
            for (int e = 0; e < MAXE; ++e)
                if (e < ne) {
                    const float4 x = *reinterpret_cast<const float4*>(ep.x[e] + roff + c);
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int t = 0; t < MAXT; ++t)
                        if (t < nt) {
                            o.x += g[t][e] * dm[t].x; o.y += g[t][e] * dm[t].y;
                            o.z += g[t][e] * dm[t].z; o.w += g[t][e] * dm[t].w;
                            dg[t][e] += dm[t].x * x.x + dm[t].y * x.y + dm[t].z * x.z + dm[t].w * x.w;
                        }
                    *reinterpret_cast<float4*>(ep.dx[e] + roff + c) = o;
                }
        }
    }
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int e = 0; e < MAXE; ++e) {
            const float v = wave_sum(dg[t][e]);
            if (lane == 0) red[wv][t * MAXE + e] = v;
        }
    __syncthreads();
    if (threadIdx.x < MAXT * MAXE) {
        const int t = threadIdx.x / MAXE, e = threadIdx.x % MAXE;
        if (t < nt && e < ne)
            dgates[((size_t)t * B + b) * ne + e] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

// ---------------------------------------------------------------- Adam (torch.optim.Adam, coupled L2)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2_sqrt) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float gr = g[i];
        const float pv = p[i];
        if (wd != 0.f) gr += wd * pv;
        const float mi = b1 * m[i] + (1.f - b1) * gr;
        const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pv - (lr / bc1) * (mi / denom);
    }
}

int ew_grid(size_t n) { size_t g = (n + 1023) / 1024; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

int fill_ptrs(GatePtrs& gp, const float* const* w, float* const* dw, int nt) {
    for (int t = 0; t < MAXT; ++t) {
        gp.w[t] = (w && t < nt) ? w[t] : nullptr;
        gp.dw[t] = (dw && t < nt) ? dw[t] : nullptr;
        if (t < nt && w && !w[t]) return RLT_E_ARG;
        if (t < nt && dw && !dw[t]) return RLT_E_ARG;
    }
    return 0;
}

}  // namespace

extern "C" {

int rlt_mmoe_gate_fwd(const float* h, const float* const* w_gate, int n_tasks, int n_e,
                      int S, int B, int C, float* gates, void* stream) {
    RLT_CHECK_ARG(h && w_gate && gates && S > 0 && B > 0 && C > 0);
    RLT_CHECK_SHAPE(n_tasks >= 1 && n_tasks <= MAXT && n_e >= 1 && n_e <= MAXE);
    GatePtrs gp;
    int rc = fill_ptrs(gp, w_gate, nullptr, n_tasks);
    if (rc) return rc;
    hipLaunchKernelGGL(gate_fwd_kernel, dim3(B), dim3(256), 0, rlt_stream(stream), h, gp, n_tasks, n_e, S, B, C, gates);
    return RLT_LAUNCH_RESULT();
}

size_t rlt_mmoe_gate_bwd_workspace(int n_tasks, int n_e, int S, int B, int C) {
    (void)S; (void)C;
    if (n_tasks <= 0 || n_e <= 0 || B <= 0) return 0;
    return (size_t)n_tasks * B * n_e * sizeof(float);
}

int rlt_mmoe_gate_bwd(const float* h, const float* const* w_gate, const float* gates, const float* dgates,
                      int n_tasks, int n_e, int S, int B, int C,
                      float* dh, int accumulate_dh, float* const* dw_gate,
                      void* ws, size_t ws_bytes, void* stream) {
    RLT_CHECK_ARG(h && w_gate && gates && dgates && dh && dw_gate && ws && S > 0 && B > 0 && C > 0);
    RLT_CHECK_SHAPE(n_tasks >= 1 && n_tasks <= MAXT && n_e >= 1 && n_e <= MAXE);
    if (ws_bytes < rlt_mmoe_gate_bwd_workspace(n_tasks, n_e, S, B, C)) return RLT_E_WORKSPACE;
    GatePtrs gp;
    int rc = fill_ptrs(gp, w_gate, dw_gate, n_tasks);
    if (rc) return rc;
    hipStream_t st = rlt_stream(stream);
    float* dlogit = (float*)ws;
    const int n = n_tasks * B;
    hipLaunchKernelGGL(gate_dlogit_kernel, dim3(rlt_cdiv(n, 256)), dim3(256), 0, st, gates, dgates, n, n_e, dlogit);
    hipLaunchKernelGGL(gate_dh_kernel, dim3(B), dim3(256), 0, st, (const float*)dlogit, gp, n_tasks, n_e, S, B, C, dh, accumulate_dh);
    hipLaunchKernelGGL(gate_dw_kernel, dim3(S, rlt_cdiv(C, 256)), dim3(256), 0, st, h, (const float*)dlogit, gp, n_tasks, n_e, S, B, C);
    return RLT_LAUNCH_RESULT();
}

int rlt_mmoe_mix_fwd(const float* const* experts, const float* gates, int n_tasks, int n_e,
                     int S, int B, int E, float* mixed, void* stream) {
    RLT_CHECK_ARG(experts && gates && mixed && S > 0 && B > 0 && E > 0);
    RLT_CHECK_SHAPE(n_tasks >= 1 && n_tasks <= MAXT && n_e >= 1 && n_e <= MAXE && E % 4 == 0);
    ExpertPtrs ep{};
    for (int e = 0; e < n_e; ++e) {
        RLT_CHECK_ARG(experts[e]);
        if (!rlt_aligned16(experts[e])) return RLT_E_ALIGN;
        ep.x[e] = experts[e];
    }
    if (!rlt_aligned16(mixed)) return RLT_E_ALIGN;
    const size_t T = (size_t)S * B;
    hipLaunchKernelGGL(mix_fwd_kernel, dim3(ew_grid(T * E / 4)), dim3(256), 0, rlt_stream(stream), ep, gates,
                       n_tasks, n_e, T, B, E, mixed);
    return RLT_LAUNCH_RESULT();
}

int rlt_mmoe_mix_bwd(const float* const* experts, const float* gates, const float* dmixed,
                     int n_tasks, int n_e, int S, int B, int E,
                     float* const* dexperts, float* dgates, void* stream) {
    RLT_CHECK_ARG(experts && gates && dmixed && dexperts && dgates && S > 0 && B > 0 && E > 0);
    RLT_CHECK_SHAPE(n_tasks >= 1 && n_tasks <= MAXT && n_e >= 1 && n_e <= MAXE && E % 4 == 0);
    ExpertPtrs ep{};
    for (int e = 0; e < n_e; ++e) {
        RLT_CHECK_ARG(experts[e] && dexperts[e]);
        if (!(rlt_aligned16(experts[e]) && rlt_aligned16(dexperts[e]))) return RLT_E_ALIGN;
        ep.x[e] = experts[e];
        ep.dx[e] = dexperts[e];
    }
    if (!rlt_aligned16(dmixed)) return RLT_E_ALIGN;
    hipLaunchKernelGGL(mix_bwd_kernel, dim3(B), dim3(256), 0, rlt_stream(stream), ep, gates, dmixed, n_tasks, n_e,
                       S, B, E, dgates);
    return RLT_LAUNCH_RESULT();
}

int rlt_adam_step(float* p, const float* g, float* m, float* v, size_t n, int step,
                  float lr, float beta1, float beta2, float eps, float weight_decay, void* stream) {
    RLT_CHECK_ARG(p && g && m && v && n > 0 && step >= 1);
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, rlt_stream(stream), p, g, m, v, n, lr, beta1, beta2,
                       eps, weight_decay, bc1, bc2s);
    return RLT_LAUNCH_RESULT();
}

}  // extern "C"

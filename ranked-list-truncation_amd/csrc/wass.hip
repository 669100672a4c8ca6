// WassDistLoss (utils/losses.py:236-311) on the GPU: entropic optimal transport between
// the B predicted cut distributions and the B label vectors of a batch, log-domain Sinkhorn, and its gradient by an
// explicit reverse sweep over the recorded iterations (the reference differentiates through the loop with autograd).
//
//   C_ij  = sum_s (p_is - y_js)^2                                         cost, (B,B) row-major
//   u_k,i = u_{k-1},i + eps (log m - LSE_j((-C_ij + u_{k-1},i + v_{k-1},j)/eps))          row pass
//   v_k,j = v_{k-1},j + eps (log m - LSE_i((-C_ij + u_k,i     + v_{k-1},j)/eps))          column pass
//   stop after the iteration whose sum_i |u_k,i - u_{k-1},i| < thresh (device flag: no host round trip; the launches of
//   the remaining iterations become no-ops)
//   loss  = sum_ij pi_ij C_ij,  pi_ij = exp((-C_ij + u_K,i + v_K,j)/eps)
//
// Reverse sweep (P = row softmax of the row pass, Q = column softmax of the column pass; v_k does not depend on
// v_{k-1}, nor u_k on u_{k-1}: the terms cancel):
//   gC = pi (1 - C/eps);  gu_i = sum_j pi C / eps;  gv_j = sum_i pi C / eps
//   for k = K..1:   gC_ij += gv_j Q_ij;  gu_i -= sum_j gv_j Q_ij;        (column pass of iteration k)
//                   gC_ij += gu_i P_ij;  gv_j  = -sum_i gu_i P_ij;  gu = 0 (row pass of iteration k)
//   dp_is = 2 sum_j gC_ij (p_is - y_js)
//
// Workspace: C | gC | history of u, v, row LSE, column LSE per iteration | scalars.  Everything fp32 like the reference;
// all reductions in a fixed order.
#include "common.h"

namespace {

struct WassWs {
    float* C; float* gC;
    float* hu; float* hv;          // [max_iter+1][B]
    float* lrow; float* lcol;      // [max_iter+1][B]  LSE of the row / column pass of iteration k (index k)
    float* du;                     // [B] |u_k - u_{k-1}|
    float* gu; float* gv;          // [B]
    float* rowpart;                // [B] per-row partial of the final cost
    int* state;                    // [0] = done flag, [1] = iterations performed
};
size_t wass_bytes(int B, int max_iter) {
    const size_t bb = (size_t)B * B, hb = (size_t)(max_iter + 1) * B;
    return (2 * bb + 4 * hb + 4 * (size_t)B) * sizeof(float) + 64;
}
WassWs wass_carve(void* ws, int B, int max_iter) {
    const size_t bb = (size_t)B * B, hb = (size_t)(max_iter + 1) * B;
    WassWs w;
    float* p = (float*)ws;
    w.C = p; p += bb; w.gC = p; p += bb;
    w.hu = p; p += hb; w.hv = p; p += hb; w.lrow = p; p += hb; w.lcol = p; p += hb;
    w.du = p; p += B; w.gu = p; p += B; w.gv = p; p += B; w.rowpart = p; p += B;
    w.state = (int*)p;
    return w;
}

// block sum / max over 256 threads, result broadcast
__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__device__ __forceinline__ float block_max(float v, float* sm) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// C_ij: one workgroup per (i, block of 256 columns j); p row staged in LDS
__global__ __launch_bounds__(256) void wass_cost_kernel(const float* __restrict__ p, const float* __restrict__ y, int B, int S,
                                                        float* __restrict__ C) {
    extern __shared__ float prow[];
    const int i = blockIdx.x, j = blockIdx.y * 256 + threadIdx.x;
    for (int s = threadIdx.x; s < S; s += 256) prow[s] = p[(size_t)i * S + s];
    __syncthreads();
    if (j >= B) return;
    const float* yr = y + (size_t)j * S;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) { const float d = fabsf(prow[s] - yr[s]); acc += d * d; }
    C[(size_t)i * B + j] = acc;
}

__global__ void wass_init_kernel(WassWs w, int B) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B) { w.hu[i] = 0.f; w.hv[i] = 0.f; }
    if (i == 0) { w.state[0] = 0; w.state[1] = 0; }
}

// row pass of iteration k (one workgroup per row i)
__global__ __launch_bounds__(256) void wass_row_kernel(WassWs w, int B, int k, float eps, float logm) {
    __shared__ float sm[4];
    if (w.state[0]) return;
    const int i = blockIdx.x;
    const float* up = w.hu + (size_t)(k - 1) * B;
    const float* vp = w.hv + (size_t)(k - 1) * B;
    const float ui = up[i];
    const float* Cr = w.C + (size_t)i * B;
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < B; j += 256) mx = fmaxf(mx, ((-Cr[j] + ui) + vp[j]) / eps);
    mx = block_max(mx, sm);
    float se = 0.f;
    for (int j = threadIdx.x; j < B; j += 256) se += expf(((-Cr[j] + ui) + vp[j]) / eps - mx);
    se = block_sum(se, sm);
    if (threadIdx.x == 0) {
        const float lse = mx + logf(se);
        const float un = (logm - lse) * eps + ui;
        w.hu[(size_t)k * B + i] = un;
        w.lrow[(size_t)k * B + i] = lse;
        w.du[i] = fabsf(un - ui);
    }
}
// column pass of iteration k: 64 columns per workgroup, 4 row lanes, online (max, sum) pairs merged in a fixed order
__global__ __launch_bounds__(256) void wass_col_kernel(WassWs w, int B, int k, float eps, float logm) {
    __shared__ float smx[4][64], sse[4][64];
    if (w.state[0]) return;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const float* uk = w.hu + (size_t)k * B;
    const float* vp = w.hv + (size_t)(k - 1) * B;
    float mx = -INFINITY, se = 0.f;
    if (j < B) {
        const float vj = vp[j];
        for (int i = rl; i < B; i += 4) {
            const float m = ((-w.C[(size_t)i * B + j] + uk[i]) + vj) / eps;
            if (m > mx) { se = se * expf(mx - m) + 1.f; mx = m; } else se += expf(m - mx);
        }
    }
    smx[rl][threadIdx.x & 63] = mx; sse[rl][threadIdx.x & 63] = se;
    __syncthreads();
    if (rl == 0 && j < B) {
        float M = fmaxf(fmaxf(smx[0][threadIdx.x], smx[1][threadIdx.x]), fmaxf(smx[2][threadIdx.x], smx[3][threadIdx.x]));
        float Ssum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) Ssum += sse[r][threadIdx.x] * expf(smx[r][threadIdx.x] - M);
        const float lse = M + logf(Ssum);
        w.hv[(size_t)k * B + j] = (logm - lse) * eps + vp[j];
        w.lcol[(size_t)k * B + j] = lse;
    }
}
// after iteration k: err = sum_i |du_i|; stop when below thresh (the iteration just done is kept)
__global__ __launch_bounds__(256) void wass_check_kernel(WassWs w, int B, int k, float thresh) {
    __shared__ float sm[4];
    if (w.state[0]) return;
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) s += w.du[i];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) {
        w.state[1] = k;
        if (s < thresh) w.state[0] = 1;
    }
}
// final cost and the seeds of the reverse sweep; one workgroup per row
__global__ __launch_bounds__(256) void wass_final_row_kernel(WassWs w, int B, float eps) {
    __shared__ float sm[4];
    const int K = w.state[1], i = blockIdx.x;
    const float* uK = w.hu + (size_t)K * B;
    const float* vK = w.hv + (size_t)K * B;
    const float ui = uK[i];
    float cost = 0.f;
    for (int j = threadIdx.x; j < B; j += 256) {
        const float c = w.C[(size_t)i * B + j];
        const float pi = expf(((-c + ui) + vK[j]) / eps);
        cost += pi * c;
        w.gC[(size_t)i * B + j] = pi * (1.f - c / eps);
    }
    cost = block_sum(cost, sm);
    if (threadIdx.x == 0) { w.rowpart[i] = cost; w.gu[i] = cost / eps; }
}
__global__ __launch_bounds__(256) void wass_final_col_kernel(WassWs w, int B, float eps, float* __restrict__ loss) {
    __shared__ float part[4][64];
    __shared__ float sm[4];
    const int K = w.state[1];
    const float* uK = w.hu + (size_t)K * B;
    const float* vK = w.hv + (size_t)K * B;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    float acc = 0.f;
    if (j < B)
        for (int i = rl; i < B; i += 4) {
            const float c = w.C[(size_t)i * B + j];
            acc += expf(((-c + uK[i]) + vK[j]) / eps) * c;
        }
    part[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && j < B) w.gv[j] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x])) / eps;
    if (blockIdx.x == 0) {                      // block 0 also sums the row partials into the loss
        float s = 0.f;
        for (int i = threadIdx.x; i < B; i += 256) s += w.rowpart[i];
        s = block_sum(s, sm);
        if (threadIdx.x == 0) *loss = s;
    }
}
// reverse of the column pass of iteration k (row-oriented): gC_ij += gv_j Q_ij; gu_i -= sum_j gv_j Q_ij
__global__ __launch_bounds__(256) void wass_bwd_col_kernel(WassWs w, int B, int k, float eps) {
    __shared__ float sm[4];
    if (k > w.state[1]) return;
    const int i = blockIdx.x;
    const float ui = w.hu[(size_t)k * B + i];
    const float* vp = w.hv + (size_t)(k - 1) * B;
    const float* lc = w.lcol + (size_t)k * B;
    float acc = 0.f;
    for (int j = threadIdx.x; j < B; j += 256) {
        const float q = expf(((-w.C[(size_t)i * B + j] + ui) + vp[j]) / eps - lc[j]);
        const float t = w.gv[j] * q;
        w.gC[(size_t)i * B + j] += t;
        acc += t;
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) w.gu[i] -= acc;
}
// reverse of the row pass of iteration k (column-oriented): gC_ij += gu_i P_ij; gv_j = -sum_i gu_i P_ij
__global__ __launch_bounds__(256) void wass_bwd_row_kernel(WassWs w, int B, int k, float eps) {
    __shared__ float part[4][64];
    if (k > w.state[1]) return;
    const float* up = w.hu + (size_t)(k - 1) * B;
    const float* vp = w.hv + (size_t)(k - 1) * B;
    const float* lr = w.lrow + (size_t)k * B;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    float acc = 0.f;
    if (j < B) {
        const float vj = vp[j];
        for (int i = rl; i < B; i += 4) {
            const float pr = expf(((-w.C[(size_t)i * B + j] + up[i]) + vj) / eps - lr[i]);
            const float t = w.gu[i] * pr;
            w.gC[(size_t)i * B + j] += t;
            acc += t;
        }
    }
    part[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && j < B) w.gv[j] = -((part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
}
__global__ void wass_zero_gu_kernel(WassWs w, int B, int k) {
    if (k > w.state[1]) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B) w.gu[i] = 0.f;
}
// dp_is = go * 2 sum_j gC_ij (p_is - y_js); one workgroup per row i, threads over s
__global__ __launch_bounds__(256) void wass_dp_kernel(WassWs w, const float* __restrict__ p, const float* __restrict__ y,
                                                      const float* __restrict__ go, int B, int S, float* __restrict__ dp) {
    const int i = blockIdx.x;
    const float g0 = go ? *go : 1.f;
    for (int s = threadIdx.x; s < S; s += 256) {
        const float pis = p[(size_t)i * S + s];
        float acc = 0.f;
        for (int j = 0; j < B; ++j) acc += w.gC[(size_t)i * B + j] * (pis - y[(size_t)j * S + s]);
        dp[(size_t)i * S + s] = 2.f * g0 * acc;
    }
}

}  // namespace

extern "C" {

size_t rlt_wass_loss_workspace(int B, int max_iter) {
    if (B <= 0 || max_iter <= 0) return 0;
    return wass_bytes(B, max_iter);
}

int rlt_wass_loss_fwd(const float* p, const float* labels, int B, int S, float eps, int max_iter, float thresh,
                      float* loss, void* ws, size_t ws_bytes, void* stream) {
    RLT_CHECK_ARG(p && labels && loss && ws && B > 0 && S > 0 && eps > 0.f && max_iter > 0);
    if (ws_bytes < wass_bytes(B, max_iter)) return RLT_E_WORKSPACE;
    const WassWs w = wass_carve(ws, B, max_iter);
    hipStream_t st = rlt_stream(stream);
    const float logm = logf(1.0f / (float)B + 1e-8f);
    const int cb = rlt_cdiv(B, 64), eb = rlt_cdiv(B, 256);
    hipLaunchKernelGGL(wass_cost_kernel, dim3(B, eb), dim3(256), (size_t)S * sizeof(float), st, p, labels, B, S, w.C);
    hipLaunchKernelGGL(wass_init_kernel, dim3(eb), dim3(256), 0, st, w, B);
    for (int k = 1; k <= max_iter; ++k) {
        hipLaunchKernelGGL(wass_row_kernel, dim3(B), dim3(256), 0, st, w, B, k, eps, logm);
        hipLaunchKernelGGL(wass_col_kernel, dim3(cb), dim3(256), 0, st, w, B, k, eps, logm);
        hipLaunchKernelGGL(wass_check_kernel, dim3(1), dim3(256), 0, st, w, B, k, thresh);
    }
    hipLaunchKernelGGL(wass_final_row_kernel, dim3(B), dim3(256), 0, st, w, B, eps);
    hipLaunchKernelGGL(wass_final_col_kernel, dim3(cb), dim3(256), 0, st, w, B, eps, loss);
    return RLT_LAUNCH_RESULT();
}

int rlt_wass_loss_bwd(const float* p, const float* labels, const float* gscale, int B, int S, float eps, int max_iter,
                      void* ws, size_t ws_bytes, float* dp, void* stream) {
    RLT_CHECK_ARG(p && labels && dp && ws && B > 0 && S > 0 && eps > 0.f && max_iter > 0);
    if (ws_bytes < wass_bytes(B, max_iter)) return RLT_E_WORKSPACE;
    const WassWs w = wass_carve(ws, B, max_iter);
    hipStream_t st = rlt_stream(stream);
    const int cb = rlt_cdiv(B, 64), eb = rlt_cdiv(B, 256);
    for (int k = max_iter; k >= 1; --k) {       // iterations beyond the recorded count are no-ops on the device
        hipLaunchKernelGGL(wass_bwd_col_kernel, dim3(B), dim3(256), 0, st, w, B, k, eps);
        hipLaunchKernelGGL(wass_bwd_row_kernel, dim3(cb), dim3(256), 0, st, w, B, k, eps);
        hipLaunchKernelGGL(wass_zero_gu_kernel, dim3(eb), dim3(256), 0, st, w, B, k);
    }
    hipLaunchKernelGGL(wass_dp_kernel, dim3(B), dim3(256), 0, st, w, p, labels, gscale, B, S, dp);
    return RLT_LAUNCH_RESULT();
}

}  // extern "C"

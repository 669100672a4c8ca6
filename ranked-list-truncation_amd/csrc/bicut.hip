// BiCut (models/Bicut.py): the per-position two-class head and BiCutLoss.
//  * pair_softmax: position-major logits (S*B, 2) -> dropout on the logits (models/Bicut.py:13) -> softmax over the two
//    classes -> (B, S, 2) in the reference's layout; and its backward.  Streaming, one thread per token.
//  * bicut_loss: utils/losses.py:11-45 fused with its gradient: one wavefront per ranked list finds the last position
//    whose argmax is class 0 (truncate), masks what follows, applies the label-dependent reward pair and reduces.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void pair_softmax_fwd_kernel(const float* __restrict__ z, int B, int S, float drop_p,
                                                               uint32_t thr, uint32_t seed, float* __restrict__ out) {
    const size_t T = (size_t)S * B;
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < T; t += (size_t)gridDim.x * 256) {
        const float2 v = *reinterpret_cast<const float2*>(z + 2 * t);
        float a = v.x, b = v.y;
        if (drop_p > 0.f) {
            a = rlt_keep(seed, (uint32_t)t, 0u, thr) ? a * inv_keep : 0.f;
            b = rlt_keep(seed, (uint32_t)t, 1u, thr) ? b * inv_keep : 0.f;
        }
        const float m = fmaxf(a, b);
        const float ea = expf(a - m), eb = expf(b - m);
        const float inv = 1.f / (ea + eb);
        const int s = (int)(t / B), bb = (int)(t % B);
        *reinterpret_cast<float2*>(out + 2 * ((size_t)bb * S + s)) = make_float2(ea * inv, eb * inv);
    }
}

__global__ __launch_bounds__(256) void pair_softmax_bwd_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                                               int B, int S, float drop_p, uint32_t thr, uint32_t seed,
                                                               float* __restrict__ dz) {
    const size_t T = (size_t)S * B;
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < T; t += (size_t)gridDim.x * 256) {
        const int s = (int)(t / B), bb = (int)(t % B);
        const size_t o = 2 * ((size_t)bb * S + s);
        const float2 y = *reinterpret_cast<const float2*>(out + o);
        const float2 g = *reinterpret_cast<const float2*>(dout + o);
        const float dot = y.x * g.x + y.y * g.y;
        float da = y.x * (g.x - dot), db = y.y * (g.y - dot);
        if (drop_p > 0.f) {
            da = rlt_keep(seed, (uint32_t)t, 0u, thr) ? da * inv_keep : 0.f;
            db = rlt_keep(seed, (uint32_t)t, 1u, thr) ? db * inv_keep : 0.f;
        }
        *reinterpret_cast<float2*>(dz + 2 * t) = make_float2(da, db);
    }
}

// one wavefront per list, 4 lists per workgroup
__global__ __launch_bounds__(256) void bicut_loss_kernel(const float* __restrict__ out, const float* __restrict__ labels,
                                                         int B, int S, int nci, float alpha, float r,
                                                         float* __restrict__ per_list, float* __restrict__ dout) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* o = out + (size_t)b * S * 2;
    const float* y = labels + (size_t)b * S;
    // last position whose argmax over (p0, p1) is class 0; ties go to class 0 (torch.argmax returns the first index)
    int last0 = -1;
    for (int j = lane; j < S; j += 64) {
        const float2 p = *reinterpret_cast<const float2*>(o + 2 * j);
        if (!(p.y > p.x)) last0 = j;
    }
    last0 = (int)wave_max((float)last0);           // S <= 2^24: exact in fp32
    const int idx = last0 < 0 ? S : last0;                     // every position says "continue": nothing is masked
    const float r_pos0 = (1.f - alpha) / r, r_neg1 = alpha / (1.f - r);
    const float invB = 1.f / (float)B;
    float acc = 0.f;
    for (int j = lane; j < S; j += 64) {
        const float2 p = *reinterpret_cast<const float2*>(o + 2 * j);
        const bool pos = y[j] == 1.f;
        float r0, r1;
        if (nci) { r0 = 0.f; r1 = pos ? (float)(-1.0 / log2((double)j + 2.0)) : (float)(((double)j + 1.0) / (double)alpha); }
        else { r0 = pos ? r_pos0 : 0.f; r1 = pos ? 0.f : r_neg1; }
        const float m = j <= idx ? 1.f : 0.f;
        acc += m * (p.x * r0 + p.y * r1);
        *reinterpret_cast<float2*>(dout + ((size_t)b * S + j) * 2) = make_float2(m * r0 * invB, m * r1 * invB);
    }
    acc = wave_sum(acc);
    if (lane == 0) per_list[b] = acc;
}

__global__ __launch_bounds__(256) void bicut_sum_kernel(const float* __restrict__ per_list, int B, float* __restrict__ loss) {
    __shared__ float sm[4];
    float v = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) v += per_list[i];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) *loss = ((sm[0] + sm[1]) + (sm[2] + sm[3])) / (float)B;
}

int grid_for(size_t n) { size_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

}  // namespace

extern "C" {

int rlt_pair_softmax_fwd(const float* z, int B, int S, float drop_p, uint32_t seed, float* out, void* stream) {
    RLT_CHECK_ARG(z && out && B > 0 && S > 0 && drop_p >= 0.f && drop_p < 1.f);
    hipLaunchKernelGGL(pair_softmax_fwd_kernel, dim3(grid_for((size_t)B * S)), dim3(256), 0, rlt_stream(stream),
                       z, B, S, drop_p, rlt_drop_threshold(drop_p), seed, out);
    return RLT_LAUNCH_RESULT();
}

int rlt_pair_softmax_bwd(const float* out, const float* dout, int B, int S, float drop_p, uint32_t seed, float* dz, void* stream) {
    RLT_CHECK_ARG(out && dout && dz && B > 0 && S > 0 && drop_p >= 0.f && drop_p < 1.f);
    hipLaunchKernelGGL(pair_softmax_bwd_kernel, dim3(grid_for((size_t)B * S)), dim3(256), 0, rlt_stream(stream),
                       out, dout, B, S, drop_p, rlt_drop_threshold(drop_p), seed, dz);
    return RLT_LAUNCH_RESULT();
}

int rlt_bicut_loss(const float* out, const float* labels, int B, int S, int metric_nci, float alpha, float r,
                   float* per_list, float* loss, float* dout, void* stream) {
    RLT_CHECK_ARG(out && labels && per_list && loss && dout && B > 0 && S > 0);
    RLT_CHECK_ARG(alpha > 0.f && r > 0.f && r < 1.f);
    hipStream_t st = rlt_stream(stream);
    hipLaunchKernelGGL(bicut_loss_kernel, dim3(rlt_cdiv(B, 4)), dim3(256), 0, st, out, labels, B, S, metric_nci, alpha, r, per_list, dout);
    hipLaunchKernelGGL(bicut_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)per_list, B, loss);
    return RLT_LAUNCH_RESULT();
}

}  // extern "C"

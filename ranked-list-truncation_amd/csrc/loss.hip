// Reward losses, multi-task terms and cut metrics: rows L1-L8, E1-E3 of SURVEY.md section 8(a).
//
// HBM-bound scan/reduction kernels.  One ranked list per wavefront: the rows of a workgroup's 4
// lists are loaded with coalesced (float4) reads and staged in LDS, every lane then owns a
// contiguous chunk of C = ceil(S/64) positions, so the prefix sums c_k = sum_{j<k} y_j and the
// DCG prefix are a short serial scan per lane plus one 64-lane shuffle scan; the softmax
// normaliser and the loss terms are wavefront shuffle reductions.  The reference builds the
// same reward matrix with B*S python calls of O(S) tensor ops each (utils/losses.py:217-225 ->
// utils/metrics.py:85-101).
//
// Algorithmic bytes per list (fp32): read p 4S + labels 4S, write dL/dp 4S (+ 4 B loss) = 3.6 KB at
// S = 300.
#include "common.h"
#include <cmath>
#include <cstdlib>
#include <mutex>

namespace {

constexpr int LISTS_PER_WG = 4;

// The pass is meant to be bound by its 3.6 KB of HBM traffic per list, which it is only if the arithmetic per position
// stays around forty instructions: IEEE-exact fp32 division (~10 instructions each, 5-6 per position) and libm's
// logf / expf (~20-30 each, 4 per position) made it VALU-bound at 1.5 TB/s.  Hardware reciprocal, log2 and exp2
// (v_rcp_f32, v_log_f32, v_exp_f32: 1 ulp) are ~1e-7 relative per value - two orders below the 1e-5 the parity tests hold
// on losses and the 1e-4 on gradients.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }   // ln x
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); } // e^x
// v_log_f32 / v_rcp_f32 read a denormal input as zero (log -> -inf, rcp -> inf) where the reference's logf / division give
// finite values (ln 1e-40 = -92.1).  The cut distribution p can underflow into that range (a softmax over 300 logits), so
// the terms in p scale a denormal argument into the normal range first; an exact 0 still gives -inf / inf as in torch.
__device__ __forceinline__ float safe_log(float x) {
    const bool den = x < 1.17549435e-38f;
    return (__builtin_amdgcn_logf(den ? x * 4294967296.f : x) - (den ? 32.f : 0.f)) * 0.6931471805599453f;
}
// num / x: 1 / x alone overflows for x < 2.9e-39 where the quotient the reference forms, (-q / B) / p, is still finite
__device__ __forceinline__ float safe_div(float num, float x) {
    const bool den = x < 1.17549435e-38f;
    const float t = num * __builtin_amdgcn_rcpf(den ? x * 4294967296.f : x);
    return den ? t * 4294967296.f : t;
}

// (value, index) argmax over the wavefront, first maximum: the wave maximum by DPP, then the smallest index among the
// lanes that hold it (each lane brings the first maximum of its own positions)
__device__ __forceinline__ int wave_first_index_of_max(float best, int bi) {
    const float m = wave_max(best);
    const int cand = (best == m) ? bi : 0x7fffffff;
    return rlt_readlane(wave_scan_op(cand, 0x7fffffff, [](int a, int b) { return a < b ? a : b; }), 63);
}

struct RewardArgs {
    const float* p;        // (B,S) or null
    const float* y;        // (B,S)
    const float* coef;     // (S) log2(j+2) or null
    float* loss_per_list;  // (B) or null
    float* dp;             // (B,S) or null
    float* r_out;          // (B,S) or null
    float* q_out;          // (B,S) or null
    int B, S, metric, kind;
    float tau, gscale;     // gscale = 1/B
    float penalty;         // DCG gain of a non-relevant document (utils/metrics.py:94; the reference's default is -1)
    // fused cut metrics (E1-E3, run.py:141-145), only read by the METRICS instantiation
    int32_t* k_out;        // (B) argmax_j p + 1
    double* f1_out;        // (B) F1@k, float64
    double* dcg_out;       // (B) DCG@k, float64 (metric penalty = mpenalty)
    double mpenalty;
    double* partials;      // (grid * 4, 3): per-wavefront sums of loss, F1, DCG
    const double* icoef_tab;   // the caller's DCG coefficient table (rlt_dcg_table_init), METRICS instantiation of the two-list kernel
};

// element j of a list lives at lds[(j / C) * STRIDE + j % C], STRIDE = C|1 (odd => conflict-free)
template <int C>
__device__ __forceinline__ int lds_slot(int j) { return (j / C) * (C | 1) + (j % C); }

// METRICS: the same pass also emits the cut position k = argmax_j p + 1 (first maximum) and F1@k / DCG@k in float64
// (run.py:141-145 -> utils/metrics.py:15-38) - labels and p are already in LDS / registers, so the metric costs no
// HBM traffic beyond its 20 B of results per list - and per-wavefront partial sums of the loss and the two metrics
// (a.partials, one record per wavefront of the grid) for the final reduction.
// A wavefront owns a list from the first load to the last store: rows are read with coalesced 16-byte loads (lane =
// 4 consecutive positions), turned through the wavefront's own LDS region so that a lane owns C CONSECUTIVE positions
// for the scans, and written back the same way - no workgroup barrier, no index division.  The grid is sized to the
// chip (launch_reward) and the wavefronts stride over the lists.
template <int C, bool METRICS>
__global__ __launch_bounds__(256) void reward_loss_kernel(RewardArgs a) {
    constexpr int STRIDE = C | 1;
    constexpr int ROW = 64 * STRIDE;
    __shared__ float sp[LISTS_PER_WG * ROW];
    __shared__ float sy[LISTS_PER_WG * ROW];
    __shared__ double icoef[METRICS ? 64 * C : 1];      // 1 / log2(j + 2), float64 (utils/metrics.py:7)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = a.S;
    if (METRICS) {
        for (int j = tid; j < S; j += 256) icoef[j] = 1.0 / log2((double)(j + 2));
        __syncthreads();
    }
    float* const wp = sp + wv * ROW;          // this wavefront's LDS rows
    float* const wy = sy + wv * ROW;
    const bool vec = (S & 3) == 0;
    double part_loss = 0.0, part_f1 = 0.0, part_dcg = 0.0;
    for (int b = blockIdx.x * LISTS_PER_WG + wv; b < a.B; b += gridDim.x * LISTS_PER_WG) {
    const size_t base = (size_t)b * S;
    // ---- stage the rows in LDS (coalesced global reads) ------------------------------------
    if (vec) {
        for (int j = 4 * lane; j < S; j += 256) {
            const float4 vy = *reinterpret_cast<const float4*>(a.y + base + j);
            float4 vp = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.p) vp = *reinterpret_cast<const float4*>(a.p + base + j);
            const float ys[4] = {vy.x, vy.y, vy.z, vy.w};
            const float ps[4] = {vp.x, vp.y, vp.z, vp.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int s = lds_slot<C>(j + i);
                wy[s] = ys[i];
                wp[s] = ps[i];
            }
        }
    } else {
        for (int j = lane; j < S; j += 64) {
            const int s = lds_slot<C>(j);
            wy[s] = a.y[base + j];
            wp[s] = a.p ? a.p[base + j] : 0.f;
        }
    }
    __builtin_amdgcn_wave_barrier();          // LDS is in-order per wavefront: the reads below see the stores above

    float dpv[C];
    float rv[C], qv[C];
#pragma unroll
    for (int i = 0; i < C; ++i) { dpv[i] = 0.f; rv[i] = 0.f; qv[i] = 0.f; }

    float yv[C], pv[C];                        // this lane's C consecutive positions (also used by the metrics below)
    {
        const float* ly = wy + lane * STRIDE;
        const float* lp = wp + lane * STRIDE;
        bool ok[C];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const int j = lane * C + i;
            ok[i] = j < S;
            yv[i] = ok[i] ? ly[i] : 0.f;
            pv[i] = ok[i] ? lp[i] : 1.f;
        }
        // ---- reward r_k, k = j+1 -------------------------------------------------------------
        if (a.metric == RLT_METRIC_F1) {
            // utils/metrics.py:85-91 on prefix sums: hits c_k (exact small integers in fp32)
            float run = 0.f;
            float pre[C];
#pragma unroll
            for (int i = 0; i < C; ++i) { run += yv[i]; pre[i] = run; }
            const float incl = wave_scan_incl(run, lane);
            const float excl = incl - run;
            const float n_rel = rlt_readlane(incl, 63);
            // p = c/k, r = c/N, F1 = 2pr/(p+r) (0 when c = 0 or N = 0) = 2c/(k+N) for c > 0
#pragma unroll
            for (int i = 0; i < C; ++i) {
                const float hits = excl + pre[i];
                const float k = (float)(lane * C + i + 1);
                rv[i] = (hits > 0.f) ? (2.f * hits) * fast_rcp(k + n_rel) : 0.f;
            }
        } else {
            // utils/metrics.py:93-101: prefix sum of (+1 | -1) / log2(j+2)
            float run = 0.f;
            float pre[C];
#pragma unroll
            for (int i = 0; i < C; ++i) {
                const int j = lane * C + i;
                float g = 0.f;
                if (ok[i]) {
                    const float icf = fast_rcp(a.coef[j]);
                    g = (yv[i] == 1.f) ? icf : icf * a.penalty;
                }
                run += g;
                pre[i] = run;
            }
            const float incl = wave_scan_incl(run, lane);
            const float excl = incl - run;
#pragma unroll
            for (int i = 0; i < C; ++i) rv[i] = excl + pre[i];
        }
        // ---- q = exp(r/tau) / sum (utils/losses.py:226-228; no max subtraction, like the reference)
        float part = 0.f;
        if (a.kind != RLT_LOSS_EXPECT || a.q_out) {
            float zs = 0.f;
            const float itau = 1.f / a.tau;
#pragma unroll
            for (int i = 0; i < C; ++i) {
                qv[i] = ok[i] ? fast_exp(rv[i] * itau) : 0.f;
                zs += qv[i];
            }
            const float iz = fast_rcp(wave_sum(zs));
#pragma unroll
            for (int i = 0; i < C; ++i) qv[i] = qv[i] * iz;
        }
        // ---- loss terms and d/dp ---------------------------------------------------------------
        if (a.p) {
#pragma unroll
            for (int i = 0; i < C; ++i) {
                if (!ok[i]) continue;
                const float p = pv[i], q = qv[i], r = rv[i];
                float term, g;
                if (a.kind == RLT_LOSS_EXPECT) {            // utils/losses.py:67-68
                    term = -(p * r);
                    g = -r * a.gscale;
                } else if (a.kind == RLT_LOSS_CE) {          // utils/losses.py:94-96
                    term = -(safe_log(p) * q);
                    g = safe_div(-q * a.gscale, p);
                } else if (a.kind == RLT_LOSS_KL) {          // utils/losses.py:230, kl_div(log p, q)
                    const float qlq = (q > 0.f) ? q * fast_log(q) : 0.f;
                    term = qlq - q * safe_log(p);
                    g = safe_div(-q * a.gscale, p);
                } else {                                     // utils/losses.py:232-233, JS
                    const float lm = fast_log((p + q) * 0.5f);
                    const float lp_ = safe_log(p);
                    const float qlq = (q > 0.f) ? q * fast_log(q) : 0.f;
                    const float plp = (p > 0.f) ? p * lp_ : 0.f;
                    term = 0.5f * ((qlq - q * lm) + (plp - p * lm));
                    g = 0.5f * (lp_ - lm) * a.gscale;        // gradient flows through log m AND the target p
                }
                part += term;
                dpv[i] = g;
            }
            const float tot = wave_sum(part);
            if (lane == 0 && a.loss_per_list) a.loss_per_list[b] = tot;
            part_loss += (double)tot;
        }
    }
    // ---- cut metrics of the same lists (METRICS) ---------------------------------------------------
    if (METRICS) {
        const float* ly = yv;                  // (registers: the rows were read from LDS once, above)
        const float* lp = pv;
        // first maximum, like np.argmax (run.py:141-142): per-lane scan in index order, then (value, index) reduction
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const int j = lane * C + i;
            if (j < S && lp[i] > best) { best = lp[i]; bi = j; }
        }
        bi = wave_first_index_of_max(best, bi);
        const int k = (bi == 0x7fffffff ? 0 : bi) + 1;
        // utils/metrics.py:15-38: the counts are small integers (exact in fp32), only the DCG sum needs float64
        float hits_f = 0.f, nrel_f = 0.f;
        double dcg = 0.0;
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const int j = lane * C + i;
            if (j < S) {
                nrel_f += ly[i];
                if (j < k) {
                    hits_f += ly[i];
                    dcg += ((ly[i] == 1.f) ? 1.0 : a.mpenalty) * icoef[j];
                }
            }
        }
        hits_f = wave_sum(hits_f);
        nrel_f = wave_sum(nrel_f);
        dcg = wave_sum(dcg);
        {
            // p = c/k, r = c/N, F1 = 2pr/(p+r) = 2c/(k+N) for c > 0 (0 when c = 0, which covers N = 0), float64
            const double f1 = hits_f > 0.f ? 2.0 * (double)hits_f / ((double)k + (double)nrel_f) : 0.0;
            if (lane == 0) {
                a.k_out[b] = k;
                a.f1_out[b] = f1;
                a.dcg_out[b] = dcg;
            }
            part_f1 += f1;
            part_dcg += dcg;
        }
    }
    // ---- coalesced write-back through the wavefront's LDS rows ---------------------------------------
    auto write_back = [&](float* dst, const float (&vals)[C]) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < C; ++i) wp[lane * STRIDE + i] = vals[i];
        __builtin_amdgcn_wave_barrier();
        if (vec) {
            for (int j = 4 * lane; j < S; j += 256)
                *reinterpret_cast<float4*>(dst + base + j) =
                    make_float4(wp[lds_slot<C>(j)], wp[lds_slot<C>(j + 1)], wp[lds_slot<C>(j + 2)], wp[lds_slot<C>(j + 3)]);
        } else {
            for (int j = lane; j < S; j += 64) dst[base + j] = wp[lds_slot<C>(j)];
        }
    };
    if (a.dp) write_back(a.dp, dpv);
    if (a.r_out) write_back(a.r_out, rv);
    if (a.q_out) write_back(a.q_out, qv);
    __builtin_amdgcn_wave_barrier();          // the next list's staging overwrites the rows
    }
    if (METRICS && lane == 0) {               // per-wavefront partial sums, fixed order (this wavefront's lists in turn)
        double* rec = a.partials + 3 * ((size_t)blockIdx.x * LISTS_PER_WG + wv);
        rec[0] = part_loss; rec[1] = part_f1; rec[2] = part_dcg;
    }
}

// ---- the same pass, two or FOUR lists per wavefront (the form the training step runs) ---------------------------
// A group of LL = 32 (16) lanes owns a list and every lane keeps the 16-byte pieces it loaded: in round r lane l of the group holds
// positions 4 LL r + 4 l .. + 3, so loads and stores are whole coalesced rows and nothing is turned through LDS.  The
// prefix sums run round by round (3 serial adds, a DPP scan over the group - the row_shr steps, for 32 lanes plus one
// row_bcast:15 - and the running total of the rounds before); the reductions are the same scan read at the last lane of the group.
// Per-position constants (position numbers, the float64 DCG coefficients) are fetched once per wavefront, outside the list loop.
// Requires S % 4 == 0, S <= 4 LL R, 16-byte aligned rows, p and dp present (the general kernel above takes the rest).
// LL = 16 (four lists per wavefront, rounds of 64 positions) where it wastes fewer lane slots: the pass is bound by vector issue, and
// S = 300 fills 300 of 384 slots in three rounds of 128 but 300 of 320 in five rounds of 64 - a sixth less vector work per list, and
// the 16-lane scans and "last lane of the group" reads are one DPP step shorter / a single row broadcast.
template <int LL, typename T, typename Op>
__device__ __forceinline__ T half_scan_op(T v, T id, Op op) {
    v = op(v, rlt_dpp<0x111, 0xf>(id, v));
    v = op(v, rlt_dpp<0x112, 0xf>(id, v));
    v = op(v, rlt_dpp<0x114, 0xf>(id, v));
    v = op(v, rlt_dpp<0x118, 0xf>(id, v));
    if (LL == 32) v = op(v, rlt_dpp<0x142, 0xa>(id, v));      // row 0 -> row 1, row 2 -> row 3: inclusive inside each half
    return v;
}
template <int LL, typename T>
__device__ __forceinline__ T half_last(T v, bool upper) {     // the value of the last lane of the caller's group
    if (LL == 16) return rlt_dpp<0x15F, 0xf>(v, v);          // row_newbcast:15: lane 15 of the row to every lane of the row
    const T lo = rlt_readlane(v, 31), hi = rlt_readlane(v, 63);
    return upper ? hi : lo;
}
template <int LL, typename T>
__device__ __forceinline__ T half_sum(T v, bool upper) {
    return half_last<LL>(half_scan_op<LL>(v, T(0), [](T x, T y) { return x + y; }), upper);
}

template <int R, bool METRICS, bool F1, int LL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(F1 && R <= 3 ? 4 : 3)))   // <= 128 (168) registers, no spills
void reward_loss_h_kernel(RewardArgs a, const double* __restrict__ icoef_tab) {
    constexpr int N = 4 * R, PR = 4 * LL, LPW = 64 / LL;      // positions per lane, per round; lists per wavefront
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool upper = lane >= 32;
    const int hl = lane & (LL - 1), grp = lane / LL;
    const int S = a.S, B = a.B;
    const bool last_ok = (R - 1) * PR + 4 * hl < S;           // rounds before the last lie inside the list by construction
    auto ok = [&](int r) { return r < R - 1 || last_ok; };
    // ---- per-position constants of this lane -------------------------------------------------------------------
    float icf[F1 ? 1 : N];                                    // DCG reward: 1 / log2(j + 2), 0 beyond the list
    if (!F1) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) icf[4 * r + i] = ok(r) ? fast_rcp(a.coef[r * PR + 4 * hl + i]) : 0.f;
    }
    const float c_exp = (1.f / a.tau) * 1.4426950408889634f;
    // F1 reward 2c / (k + N): k + N <= 2 S <= 768 is a small integer, so 2 / (k + N) comes from a table in LDS (filled with the same
    // v_rcp_f32 the per-position form executed, doubled - exact: the products are bit-identical) instead of a quarter-rate
    // reciprocal per position.  The pass is bound by vector issue (profiles/r04_notes.md), not by loads.
    __shared__ float rtab[F1 ? 1024 : 1];
    if (F1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rtab[tid + 256 * i] = 2.f * fast_rcp((float)(tid + 256 * i));
        __syncthreads();
    }
    double part_loss = 0.0, part_f1 = 0.0, part_dcg = 0.0;
    const int nwaves = gridDim.x * 4;
    for (int pb = LPW * (blockIdx.x * 4 + wv); pb < B; pb += LPW * nwaves) {
        const bool live = pb + grp < B;                       // lists beyond B: the idle group shadows the first one, stores masked
        const int b = live ? pb + grp : pb;
        const size_t base = (size_t)b * S;
        float y[N], p[N];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const size_t at = base + (ok(r) ? r * PR + 4 * hl : 0);
            float4 vy = *reinterpret_cast<const float4*>(a.y + at);
            float4 vp = *reinterpret_cast<const float4*>(a.p + at);
            if (r == R - 1 && !last_ok) { vy = make_float4(0.f, 0.f, 0.f, 0.f); vp = make_float4(1.f, 1.f, 1.f, 1.f); }
            y[4 * r] = vy.x; y[4 * r + 1] = vy.y; y[4 * r + 2] = vy.z; y[4 * r + 3] = vy.w;
            p[4 * r] = vp.x; p[4 * r + 1] = vp.y; p[4 * r + 2] = vp.z; p[4 * r + 3] = vp.w;
        }
        // ---- reward r_k (k = j + 1) by prefix sums over the rounds --------------------------------------------
        float rv[N];
        float n_rel = 0.f;
        if (F1) {                                             // utils/metrics.py:85-91: F1 = 2c / (k + N), 0 when c = 0
            float off = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float s0 = y[4 * r], s1 = s0 + y[4 * r + 1], s2 = s1 + y[4 * r + 2], s3 = s2 + y[4 * r + 3];
                const float incl = half_scan_op<LL>(s3, 0.f, [](float x, float z) { return x + z; });
                const float ex = (incl - s3) + off;
                rv[4 * r] = ex + s0; rv[4 * r + 1] = ex + s1; rv[4 * r + 2] = ex + s2; rv[4 * r + 3] = ex + s3;
                off += half_last<LL>(incl, upper);
            }
            n_rel = off;
            const float* rt = rtab + (4 * hl + 1 + (int)n_rel);       // 2 / (k + N), k = position + 1
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rv[4 * r + i] *= rt[r * PR + i];              // c = 0 gives 0 (k + N >= 1)
            }
        } else {                                              // utils/metrics.py:93-101: prefix of (+1 | penalty) / log2(j+2)
            float off = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float g[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) g[i] = (y[4 * r + i] == 1.f) ? icf[4 * r + i] : icf[4 * r + i] * a.penalty;
                const float s0 = g[0], s1 = s0 + g[1], s2 = s1 + g[2], s3 = s2 + g[3];
                const float incl = half_scan_op<LL>(s3, 0.f, [](float x, float z) { return x + z; });
                const float ex = (incl - s3) + off;
                rv[4 * r] = ex + s0; rv[4 * r + 1] = ex + s1; rv[4 * r + 2] = ex + s2; rv[4 * r + 3] = ex + s3;
                off += half_last<LL>(incl, upper);
            }
            if (METRICS) {
                float cnt = 0.f;
#pragma unroll
                for (int n = 0; n < N; ++n) cnt += y[n];
                n_rel = half_sum<LL>(cnt, upper);
            }
        }
        // ---- q = exp(r / tau) / sum (utils/losses.py:226-228; no maximum subtracted, like the reference) ----------
        float q[N];
        float l2z = 0.f;
        if (a.kind != RLT_LOSS_EXPECT) {
            float zs = 0.f;
#pragma unroll
            for (int n = 0; n < N; ++n) {
                q[n] = rlt_exp2(rv[n] * c_exp);
                if (n >= N - 4 && !last_ok) q[n] = 0.f;
                zs += q[n];
            }
            const float zsum = half_sum<LL>(zs, upper);
            const float iz = fast_rcp(zsum);
            l2z = __builtin_amdgcn_logf(zsum);                // log2 q_n = r_n c_exp - log2 Z: no v_log_f32 per position for q
#pragma unroll
            for (int n = 0; n < N; ++n) q[n] *= iz;
        }
        // ---- loss terms and d/dp (positions beyond the list carry p = 1, q = 0) -----------------------------------
        float part = 0.f;
        float dpv[N];
        if (a.kind == RLT_LOSS_EXPECT) {                      // utils/losses.py:67-68
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const float r_ = (n >= N - 4 && !last_ok) ? 0.f : rv[n];
                part -= p[n] * r_;
                dpv[n] = -r_ * a.gscale;
            }
        } else if (a.kind == RLT_LOSS_CE || a.kind == RLT_LOSS_KL) {   // utils/losses.py:94-96 / :230, kl_div(log p, q)
            const bool kl = a.kind == RLT_LOSS_KL;
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const float qlq = (kl && q[n] > 0.f) ? q[n] * fast_log(q[n]) : 0.f;
                part += qlq - q[n] * safe_log(p[n]);
                dpv[n] = safe_div(-q[n] * a.gscale, p[n]);
            }
        } else {
            // utils/losses.py:232-233, JS: (q ln(q/m) + p ln(p/m)) / 2 with m = (p + q) / 2.  In log2 units, with
            // s = p + q: q (log2 q - log2 s) + p (log2 p - log2 s) + s per position, scaled by ln 2 / 2 once per list;
            // log2 of max(x, 2^-126): x = 0 contributes 0 (0 * finite), as the reference's 0 log 0 = 0.  v_log_f32 reads a DENORMAL
            // p as zero, where the reference's gradient (ln p - ln m) / 2 is finite down to -103: a wavefront that holds one
            // (a vote on the lane minima; a softmax over 300 logits can underflow that far, it is rare) takes the second loop,
            // which scales such a p into the normal range first - the form of safe_log, as in the general pass above.
            // d/dp = (ln p - ln m) / 2: the gradient flows through log m AND the target p
            const float cg = 0.5f * 0.6931471805599453f * a.gscale;
            float pmin = p[0];
#pragma unroll
            for (int n = 1; n < N; ++n) pmin = fminf(pmin, p[n]);
            if (__builtin_amdgcn_ballot_w64(pmin < 1.17549435e-38f) == 0) {      // (an exactly-zero p too: the fast loop takes log2 p as it is)
#pragma unroll
                for (int n = 0; n < N; ++n) {
                    const float sm = p[n] + q[n];
                    const float l2s = __builtin_amdgcn_logf(sm);
                    // log2 q from the exponent it was formed with (exact where v_log_f32 of the rounded q has 1 ulp; a q that
                    // underflowed to 0 - or a position beyond the list - multiplies it by 0); clamped like the log form was
                    // (no clamp at -126: below it q = exp2(.) has flushed to 0 and multiplies a finite number)
                    const float dq_ = __builtin_fmaf(rv[n], c_exp, -l2z) - l2s;
                    const float dp_ = __builtin_amdgcn_logf(p[n]) - l2s;
                    part = __builtin_fmaf(q[n], dq_, part);
                    part = __builtin_fmaf(p[n], dp_, part);
                    part += sm;
                    dpv[n] = __builtin_fmaf(dp_, cg, cg);
                }
            } else {
#pragma unroll
                for (int n = 0; n < N; ++n) {
                    const float sm = p[n] + q[n];
                    const float l2s = __builtin_amdgcn_logf(sm);
                    const float dq_ = __builtin_amdgcn_logf(fmaxf(q[n], 1.17549435e-38f)) - l2s;
                    const bool den = p[n] < 1.17549435e-38f && p[n] > 0.f;
                    const float l2p = den ? __builtin_amdgcn_logf(p[n] * 4294967296.f) - 32.f
                                          : __builtin_amdgcn_logf(fmaxf(p[n], 1.17549435e-38f));
                    const float dp_ = l2p - l2s;
                    part = __builtin_fmaf(q[n], dq_, part);
                    part = __builtin_fmaf(p[n], dp_, part);
                    part += sm;
                    dpv[n] = __builtin_fmaf(dp_, cg, cg);
                }
            }
            if (!last_ok) part -= 4.f;                        // the 4 positions beyond the list carry p = 1, q = 0: s = 1 each
            part *= 0.5f * 0.6931471805599453f;
        }
        const float tot = half_sum<LL>(part, upper);
        if (hl == 0 && live && a.loss_per_list) a.loss_per_list[b] = tot;
        if (live) part_loss += (double)tot;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (live && ok(r))
                *reinterpret_cast<float4*>(a.dp + base + r * PR + 4 * hl) =
                    make_float4(dpv[4 * r], dpv[4 * r + 1], dpv[4 * r + 2], dpv[4 * r + 3]);
        // ---- cut metrics of the same lists (run.py:141-145 -> utils/metrics.py:15-38) ----------------------------
        if (METRICS) {
            float best = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = r * PR + 4 * hl + i;
                    if (ok(r) && p[4 * r + i] > best) { best = p[4 * r + i]; bi = j; }
                }
            const float m = half_last<LL>(half_scan_op<LL>(best, -INFINITY, [](float x, float z) { return fmaxf(x, z); }), upper);
            const int cand = (best == m) ? bi : 0x7fffffff;
            const int kmin = half_last<LL>(half_scan_op<LL>(cand, 0x7fffffff, [](int x, int z) { return x < z ? x : z; }), upper);
            const int k = (kmin == 0x7fffffff ? 0 : kmin) + 1;
            // DCG@k = sum_{j<k} (y_j == 1 ? 1 : pen) / log2(j + 2) = pen * T[k] + (1 - pen) * sum_{j<k, y_j == 1} 1 / log2(j + 2),
            // T[k] = sum_{j<k} 1 / log2(j + 2) from the per-device table (float64; 1e-16 relative apart from the serial sum)
            float hits = 0.f;
            double rel = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int kmax = max(rlt_readlane(k, 0), rlt_readlane(k, 32));
                if (LL == 16) kmax = max(kmax, max(rlt_readlane(k, 16), rlt_readlane(k, 48)));
                if (r * PR >= __builtin_amdgcn_readfirstlane(kmax)) break;   // every group's cut lies before this round
                // 1 / log2(j + 2) in float64 (utils/metrics.py:7): the same 32 bytes per lane for every list, served by the cache
                double ic[4];
                {
                    const double* src = icoef_tab + (ok(r) ? r * PR + 4 * hl : 0);
                    asm volatile("" : "+v"(src));             // keep the loads here: hoisted out of the list loop they cost 8 R registers
                    const double2 lo = *reinterpret_cast<const double2*>(src), hi = *reinterpret_cast<const double2*>(src + 2);
                    ic[0] = lo.x; ic[1] = lo.y; ic[2] = hi.x; ic[3] = hi.y;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = r * PR + 4 * hl + i;
                    const bool in = j < k && ok(r);
                    hits += in ? y[4 * r + i] : 0.f;
                    rel += (in && y[4 * r + i] == 1.f) ? ic[i] : 0.0;
                }
            }
            hits = half_sum<LL>(hits, upper);
            rel = half_sum<LL>(rel, upper);
            const double dcg = a.mpenalty * icoef_tab[1024 + k] + (1.0 - a.mpenalty) * rel;
            // F1 = 2pr / (p + r) = 2c / (k + N) (0 when c = 0, which covers N = 0): small integers, reciprocal by two
            // Newton steps from v_rcp_f64 (~1e-16 relative; the parity bound on the mean is 1e-12)
            const double den = (double)k + (double)n_rel;
            double rc = __builtin_amdgcn_rcp(den);
            rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
            rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
            const double f1 = 2.0 * (double)hits * rc;
            if (hl == 0 && live) {
                a.k_out[b] = k;
                a.f1_out[b] = f1;
                a.dcg_out[b] = dcg;
            }
            if (live) { part_f1 += f1; part_dcg += dcg; }
        }
    }
    if (METRICS) {                                            // one record per wavefront: its groups in lane order
        auto groups = [&](double v) {
            double t = rlt_readlane(v, 0);
            if (LL == 16) t += rlt_readlane(v, 16);
            t += rlt_readlane(v, 32);
            if (LL == 16) t += rlt_readlane(v, 48);
            return t;
        };
        const double l = groups(part_loss), f = groups(part_f1), d = groups(part_dcg);
        if (lane == 0) {
            double* rec = a.partials + 3 * ((size_t)blockIdx.x * 4 + wv);
            rec[0] = l; rec[1] = f; rec[2] = d;
        }
    }
}

// out[0] = scale * sum_i x[i], single workgroup, fixed summation order (deterministic)
__global__ __launch_bounds__(256) void sum_scale_kernel(const float* x, int n, float scale, float* out) {
    __shared__ double sm[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)x[i];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(sm[0] * (double)scale);
}

// loss = sum(per-wavefront loss sums) / B in float32 from a float64 sum, sums = {sum F1, sum DCG} in float64: one workgroup
// over the <= 8192 partial records of the pass, fixed summation order (deterministic)
__global__ __launch_bounds__(256) void loss_metrics_final_kernel(const double* partials, int n, float scale, float* loss_out,
                                                                 double* sums) {
    __shared__ double sm[3][256];
    double l = 0.0, x = 0.0, z = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { l += partials[3 * i]; x += partials[3 * i + 1]; z += partials[3 * i + 2]; }
    sm[0][threadIdx.x] = l; sm[1][threadIdx.x] = x; sm[2][threadIdx.x] = z;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int k = 0; k < 3; ++k) sm[k][threadIdx.x] += sm[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        loss_out[0] = (float)(sm[0][0] * (double)scale);
        sums[0] = sm[1][0];
        sums[1] = sm[2][0];
    }
}

// grid: one workgroup per group of 4 lists up to 8 workgroups per CU (256 CUs), beyond that the workgroups stride
constexpr int REWARD_MAX_GRID = 256 * 8;

int reward_grid(int B) {
    const int groups = rlt_cdiv(B, LISTS_PER_WG);
    return groups < REWARD_MAX_GRID ? groups : REWARD_MAX_GRID;
}
template <int C, bool METRICS>
int launch_reward(const RewardArgs& a, hipStream_t st) {
    const int grid = reward_grid(a.B);
    hipLaunchKernelGGL((reward_loss_kernel<C, METRICS>), dim3(grid), dim3(256), 0, st, a);
    return RLT_LAUNCH_RESULT();
}

// 1 / log2(j + 2) for j < 1024 in float64 and their prefix sums: [j] = 1 / log2(j + 2); [1024 + k] = sum_{j<k} of them.  The table
// lives in CALLER memory (rlt_dcg_table_bytes / rlt_dcg_table_init, once per device buffer): the library allocates nothing and
// never synchronises with the host.
constexpr int DCG_TAB_N = 1024;
__global__ __launch_bounds__(1024) void dcg_table_kernel(double* __restrict__ tab) {
    __shared__ double c[DCG_TAB_N];
    const int j = threadIdx.x;
    c[j] = 1.0 / log2((double)(j + 2));
    tab[j] = c[j];
    __syncthreads();
    if (j == 0) {                                   // fixed order: the sums a sequential float64 loop produces
        double acc = 0.0;
        tab[DCG_TAB_N] = 0.0;
        for (int k = 0; k < DCG_TAB_N; ++k) { acc += c[k]; tab[DCG_TAB_N + k + 1] = acc; }
    }
}

// two (four) lists per wavefront: one workgroup per 8 (16) lists up to the same number of workgroups
int reward_h_grid(int B, int lpw = 2) {
    const int groups = rlt_cdiv(B, lpw * LISTS_PER_WG);
    return groups < REWARD_MAX_GRID ? groups : REWARD_MAX_GRID;
}
template <int R, bool METRICS, int LL = 32>
int launch_reward_h(const RewardArgs& a, hipStream_t st) {
    const double* tab = a.icoef_tab;
    if (METRICS && !tab) return RLT_E_ARG;
    const dim3 grid(reward_h_grid(a.B, 64 / LL));
    if (a.metric == RLT_METRIC_F1)
        hipLaunchKernelGGL((reward_loss_h_kernel<R, METRICS, true, LL>), grid, dim3(256), 0, st, a, tab);
    else
        hipLaunchKernelGGL((reward_loss_h_kernel<R, METRICS, false, LL>), grid, dim3(256), 0, st, a, tab);
    return RLT_LAUNCH_RESULT();
}

// *records: partial-sum records the pass writes (METRICS), one per wavefront of its grid
template <bool METRICS>
int dispatch_reward_m(const RewardArgs& a, hipStream_t st, int* records = nullptr) {
    static const bool general_only = getenv("RLT_LOSS_GENERAL") != nullptr;      // developer switch: A/B against the general kernel
    const bool halves = a.p && a.dp && !a.r_out && !a.q_out && (a.S & 3) == 0 && a.S <= 384 && rlt_aligned16(a.dp) &&
                        rlt_aligned16(a.p) && rlt_aligned16(a.y) && !general_only;
    if (halves) {
        // four lists per wavefront where rounds of 64 positions waste fewer lane slots than rounds of 128 (an odd number of them:
        // S in 1..64, 129..192, 257..320 - the reference's 300); RLT_LOSS_QUARTERS=0: two lists per wavefront everywhere (A/B runs)
        static const bool quarters = [] { const char* e = getenv("RLT_LOSS_QUARTERS"); return !e || atoi(e) != 0; }();
        const int r16 = rlt_cdiv(a.S, 64);
        // (the DCG reward keeps a coefficient per position in registers: its five-round form would spill)
        if (quarters && (r16 & 1) && r16 <= (a.metric == RLT_METRIC_F1 ? 5 : 3)) {
            if (records) *records = reward_h_grid(a.B, 4) * LISTS_PER_WG;
            if (r16 == 1) return launch_reward_h<1, METRICS, 16>(a, st);
            if (r16 == 3) return launch_reward_h<3, METRICS, 16>(a, st);
            return launch_reward_h<5, METRICS, 16>(a, st);
        }
        if (records) *records = reward_h_grid(a.B) * LISTS_PER_WG;
        const int r = rlt_cdiv(a.S, 128);
        if (r <= 1) return launch_reward_h<1, METRICS>(a, st);
        if (r <= 2) return launch_reward_h<2, METRICS>(a, st);
        return launch_reward_h<3, METRICS>(a, st);
    }
    if (records) *records = reward_grid(a.B) * LISTS_PER_WG;
    const int c = rlt_cdiv(a.S, 64);
    if (c <= 1) return launch_reward<1, METRICS>(a, st);
    if (c <= 2) return launch_reward<2, METRICS>(a, st);
    if (c <= 3) return launch_reward<3, METRICS>(a, st);
    if (c <= 4) return launch_reward<4, METRICS>(a, st);
    if (c <= 5) return launch_reward<5, METRICS>(a, st);
    if (c <= 6) return launch_reward<6, METRICS>(a, st);
    if (c <= 8) return launch_reward<8, METRICS>(a, st);
    if (c <= 12) return launch_reward<12, METRICS>(a, st);
    if (c <= 16) return launch_reward<16, METRICS>(a, st);
    return RLT_E_SHAPE;
}
int dispatch_reward(const RewardArgs& a, hipStream_t st) { return dispatch_reward_m<false>(a, st); }

// ------------------------------------------------------------------------------ multi-task terms
constexpr int MT_PART = 8;   // floats per partial record

__global__ __launch_bounds__(256) void mt_partial_kernel(const float* rerank, const float* cls, const float* y,
                                                         size_t n, float* partial) {
    float s_pos = 0.f, n_pos = 0.f, s_neg = 0.f, n_neg = 0.f, bce = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float t = y[i];
        if (rerank) {
            const float s = rerank[i];
            if (t == 1.f) { s_pos += s; n_pos += 1.f; }
            if (t == 0.f) { s_neg += s; n_neg += 1.f; }
        }
        if (cls) {   // torch binary_cross_entropy: logs clamped at -100
            const float c = cls[i];
            const float l1 = fmaxf(logf(c), -100.f);
            const float l0 = fmaxf(log1pf(-c), -100.f);
            bce += (t - 1.f) * l0 - t * l1;
        }
    }
    float v[5] = {s_pos, n_pos, s_neg, n_neg, bce};
    __shared__ float sm[4][5];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float t = wave_sum(v[k]);
        if (lane == 0) sm[wv][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < 5)
        partial[blockIdx.x * MT_PART + threadIdx.x] =
            sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

__global__ __launch_bounds__(64) void mt_final_kernel(const float* partial, int nparts, double n_elem, float margin,
                                                      int has_rerank, int has_cls, float* terms) {
    const int lane = threadIdx.x;
    double v[5] = {0, 0, 0, 0, 0};
    for (int i = lane; i < nparts; i += 64)
        for (int k = 0; k < 5; ++k) v[k] += (double)partial[i * MT_PART + k];
    for (int k = 0; k < 5; ++k) v[k] = wave_sum(v[k]);
    if (lane == 0) {
        float hinge = 0.f, gpos = 0.f, gneg = 0.f;
        if (has_rerank && v[1] > 0 && v[3] > 0) {            // utils/losses.py:136-141
            const float gap = (float)(v[2] / v[3]) - (float)(v[0] / v[1]) + margin;
            if (gap > 0.f) { hinge = gap; gpos = (float)(-1.0 / v[1]); gneg = (float)(1.0 / v[3]); }
        }
        terms[0] = hinge;
        terms[1] = has_cls ? (float)(v[4] / n_elem) : 0.f;
        terms[2] = gpos;
        terms[3] = gneg;
    }
}

__global__ __launch_bounds__(256) void mt_bwd_kernel(const float* cls, const float* y, const float* terms, size_t n,
                                                     float w_r, float w_c, const float* gscale,
                                                     float* d_rerank, float* d_class) {
    const float gs = gscale ? gscale[0] : 1.f;
    const float gpos = terms[2] * w_r * gs, gneg = terms[3] * w_r * gs;
    const float cscale = w_c * gs / (float)n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float t = y[i];
        if (d_rerank) d_rerank[i] = (t == 1.f) ? gpos : ((t == 0.f) ? gneg : 0.f);
        if (d_class) {   // torch binary_cross_entropy_backward: (x - t) / max((1-x)x, 1e-12)
            const float c = cls[i];
            d_class[i] = (c - t) / fmaxf((1.f - c) * c, 1e-12f) * cscale;
        }
    }
}

struct WSumArgs { const float* x[8]; float w[8]; int n; };
__global__ void weighted_sum_kernel(WSumArgs a, float* out) {
    float acc = 0.f;
    for (int i = 0; i < a.n; ++i) acc += a.w[i] * a.x[i][0];
    out[0] = acc;
}

// ------------------------------------------------------------------------------ cut metrics
__global__ __launch_bounds__(256) void cut_metrics_kernel(const float* p, const float* y, const int32_t* k_in,
                                                          int B, int S, double penalty, int32_t* k_out, double* f1_out, double* dcg_out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wv;
    if (b >= B) return;
    const float* yr = y + (size_t)b * S;
    int k;
    if (k_in) {
        k = k_in[b];
    } else {
        // first maximum, like np.argmax (run.py:141-142)
        const float* pr = p + (size_t)b * S;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = lane; j < S; j += 64) {
            const float v = pr[j];
            if (v > best) { best = v; bi = j; }
        }
        bi = wave_first_index_of_max(best, bi);
        k = (bi == 0x7fffffff ? 0 : bi) + 1;
    }
    // utils/metrics.py:15-38 in float64
    double hits = 0.0, n_rel = 0.0, dcg = 0.0;
    for (int j = lane; j < S; j += 64) {
        const double t = (double)yr[j];
        n_rel += t;
        if (j < k) {
            hits += t;
            dcg += ((yr[j] == 1.f) ? 1.0 : penalty) / log2((double)(j + 2));
        }
    }
    hits = wave_sum(hits);
    n_rel = wave_sum(n_rel);
    dcg = wave_sum(dcg);
    if (lane == 0) {
        const double prec = hits / (double)k;
        const double rec = (n_rel != 0.0) ? hits / n_rel : 0.0;
        const double f1 = (prec + rec != 0.0) ? 2.0 * prec * rec / (prec + rec) : 0.0;
        if (k_out) k_out[b] = k;
        if (f1_out) f1_out[b] = f1;
        if (dcg_out) dcg_out[b] = dcg;
    }
}

__global__ __launch_bounds__(256) void sum2_f64_kernel(const double* a, const double* b, int n, double* out) {
    __shared__ double sa[256], sb[256];
    double x = 0.0, z = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { x += a[i]; z += b[i]; }
    sa[threadIdx.x] = x;
    sb[threadIdx.x] = z;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { sa[threadIdx.x] += sa[threadIdx.x + s]; sb[threadIdx.x] += sb[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = sa[0]; out[1] = sb[0]; }
}

// ---- task metrics (utils/metrics.py:40-76) -----------------------------------------
// taskr_metric: DCG of the whole list re-ordered by descending prediction (+1/log2(i+2) for a relevant document at
// sorted position i, -1/log2(i+2) otherwise).  taskc_metric: ROC AUC of the predictions against the labels per list
// (sklearn's roc_auc_score: ties count 1/2), lists with only one class skipped.  One wavefront per list, predictions
// and labels in LDS; rank of element j = #(pred > pred_j) + #(pred == pred_j at an earlier index) by an O(S^2) sweep
// (S <= 1024), float64 accumulation.
constexpr int TM_MAXS = 1024;
__global__ __launch_bounds__(256) void task_metrics_kernel(const float* __restrict__ labels, const float* __restrict__ pred,
                                                           int B, int S, double* __restrict__ dcg_out,
                                                           double* __restrict__ auc_out) {
    __shared__ float sp[4][TM_MAXS];
    __shared__ float sy[4][TM_MAXS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wv;
    if (b >= B) return;
    for (int j = lane; j < S; j += 64) { sp[wv][j] = pred[(size_t)b * S + j]; sy[wv][j] = labels[(size_t)b * S + j]; }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    double dcg = 0.0, pairs = 0.0, npos = 0.0;
    for (int j = lane; j < S; j += 64) {
        const float pj = sp[wv][j];
        const bool rel = sy[wv][j] != 0.f;
        int rank = 0;
        double wins = 0.0;
        for (int i = 0; i < S; ++i) {
            const float pi = sp[wv][i];
            rank += (pi > pj) || (pi == pj && i < j);
            if (rel && sy[wv][i] == 0.f) wins += pj > pi ? 1.0 : (pj == pi ? 0.5 : 0.0);
        }
        dcg += (rel ? 1.0 : -1.0) / log2((double)rank + 2.0);
        if (rel) { pairs += wins; npos += 1.0; }
    }
    dcg = wave_sum(dcg);
    pairs = wave_sum(pairs);
    npos = wave_sum(npos);
    if (lane == 0) {
        if (dcg_out) dcg_out[b] = dcg;
        if (auc_out) {
            const double nneg = (double)S - npos;
            auc_out[b] = (npos == 0.0 || nneg == 0.0) ? -1.0 : pairs / (npos * nneg);      // -1: skipped list
        }
    }
}
__global__ __launch_bounds__(256) void task_metrics_sum_kernel(const double* __restrict__ dcg, const double* __restrict__ auc,
                                                               int B, double* __restrict__ sums) {
    __shared__ double sm[3][4];
    double a = 0.0, c = 0.0, n = 0.0;
    for (int i = threadIdx.x; i < B; i += 256) {
        a += dcg[i];
        if (auc[i] >= 0.0) { c += auc[i]; n += 1.0; }
    }
    a = wave_sum(a); c = wave_sum(c); n = wave_sum(n);
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = a; sm[1][threadIdx.x >> 6] = c; sm[2][threadIdx.x >> 6] = n; }
    __syncthreads();
    if (threadIdx.x < 3) sums[threadIdx.x] = (sm[threadIdx.x][0] + sm[threadIdx.x][1]) + (sm[threadIdx.x][2] + sm[threadIdx.x][3]);
}

}  // namespace

extern "C" {

static int reward_args_ok(const float* p, const float* labels, const float* dcg_coef, int B, int S, int metric, int kind) {
    RLT_CHECK_ARG(labels && B > 0 && S > 0);
    RLT_CHECK_ARG(metric == RLT_METRIC_F1 || (metric == RLT_METRIC_DCG && dcg_coef));
    RLT_CHECK_ARG(kind >= RLT_LOSS_EXPECT && kind <= RLT_LOSS_JS);
    RLT_CHECK_SHAPE(S <= 1024);
    if ((S & 3) == 0 && !((!p || rlt_aligned16(p)) && rlt_aligned16(labels))) return RLT_E_ALIGN;
    return 0;
}

int rlt_reward_loss_ex(const float* p, const float* labels, const float* dcg_coef, int B, int S,
                       int metric, float penalty, int kind, float tau,
                       float* loss_per_list, float* loss_out, float* dp, void* stream) {
    RLT_CHECK_ARG(p && loss_per_list);
    int rc = reward_args_ok(p, labels, dcg_coef, B, S, metric, kind);
    if (rc) return rc;
    RewardArgs a{p, labels, dcg_coef, loss_per_list, dp, nullptr, nullptr, B, S, metric, kind, tau, 1.0f / (float)B, penalty,
                 nullptr, nullptr, nullptr, -1.0, nullptr, nullptr};
    hipStream_t st = rlt_stream(stream);
    rc = dispatch_reward(a, st);
    if (rc) return rc;
    if (loss_out) {
        hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, st, loss_per_list, B, 1.0f / (float)B, loss_out);
        rc = RLT_LAUNCH_RESULT();
    }
    return rc;
}

int rlt_reward_loss(const float* p, const float* labels, const float* dcg_coef, int B, int S,
                    int metric, int kind, float tau,
                    float* loss_per_list, float* loss_out, float* dp, void* stream) {
    return rlt_reward_loss_ex(p, labels, dcg_coef, B, S, metric, -1.f, kind, tau, loss_per_list, loss_out, dp, stream);
}

size_t rlt_loss_metrics_workspace(int B) {
    return B > 0 ? (size_t)reward_grid(B) * LISTS_PER_WG * 3 * sizeof(double) : 0;      // one record per wavefront of the pass
}

size_t rlt_dcg_table_bytes(void) { return ((size_t)(2 * DCG_TAB_N + 1) * sizeof(double) + 15) / 16 * 16; }

int rlt_dcg_table_init(void* table, size_t table_bytes, void* stream) {
    RLT_CHECK_ARG(table);
    if (table_bytes < rlt_dcg_table_bytes()) return RLT_E_WORKSPACE;
    if (((uintptr_t)table & 7u) != 0) return RLT_E_ALIGN;
    hipLaunchKernelGGL(dcg_table_kernel, dim3(1), dim3(DCG_TAB_N), 0, rlt_stream(stream), (double*)table);
    return RLT_LAUNCH_RESULT();
}

int rlt_loss_metrics(const float* p, const float* labels, const float* dcg_coef, int B, int S,
                     int metric, float penalty, int kind, float tau, double metric_penalty,
                     float* loss_per_list, float* loss_out, float* dp,
                     int32_t* k_out, double* f1_out, double* dcg_out, double* sums,
                     const void* dcg_table, void* ws, size_t ws_bytes, void* stream) {
    RLT_CHECK_ARG(p && loss_per_list && loss_out && k_out && f1_out && dcg_out && sums && ws && dcg_table);
    if (((uintptr_t)dcg_table & 7u) != 0) return RLT_E_ALIGN;
    int rc = reward_args_ok(p, labels, dcg_coef, B, S, metric, kind);
    if (rc) return rc;
    if (ws_bytes < rlt_loss_metrics_workspace(B)) return RLT_E_WORKSPACE;
    RewardArgs a{p, labels, dcg_coef, loss_per_list, dp, nullptr, nullptr, B, S, metric, kind, tau, 1.0f / (float)B, penalty,
                 k_out, f1_out, dcg_out, metric_penalty, (double*)ws, (const double*)dcg_table};
    hipStream_t st = rlt_stream(stream);
    int records = 0;
    rc = dispatch_reward_m<true>(a, st, &records);
    if (rc) return rc;
    hipLaunchKernelGGL(loss_metrics_final_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, records,
                       1.0f / (float)B, loss_out, sums);
    return RLT_LAUNCH_RESULT();
}

int rlt_reward_matrix_ex(const float* labels, const float* dcg_coef, int B, int S, int metric, float penalty, float tau,
                         float* r_out, float* q_out, void* stream) {
    RLT_CHECK_ARG(r_out || q_out);
    int rc = reward_args_ok(nullptr, labels, dcg_coef, B, S, metric, RLT_LOSS_KL);
    if (rc) return rc;
    RewardArgs a{nullptr, labels, dcg_coef, nullptr, nullptr, r_out, q_out, B, S, metric, RLT_LOSS_KL, tau, 1.0f, penalty,
                 nullptr, nullptr, nullptr, -1.0, nullptr, nullptr};
    return dispatch_reward(a, rlt_stream(stream));
}

int rlt_reward_matrix(const float* labels, const float* dcg_coef, int B, int S, int metric, float tau,
                      float* r_out, float* q_out, void* stream) {
    return rlt_reward_matrix_ex(labels, dcg_coef, B, S, metric, -1.f, tau, r_out, q_out, stream);
}

static int mt_grid(size_t n) { return (int)((n + 256 * 8 - 1) / (256 * 8) < 1024 ? (n + 256 * 8 - 1) / (256 * 8) : 1024); }

size_t rlt_mt_terms_workspace(int B, int S) {
    return (size_t)mt_grid((size_t)B * S) * MT_PART * sizeof(float);
}

int rlt_mt_terms(const float* rerank, const float* cls, const float* labels, int B, int S, float margin,
                 float* terms, void* ws, size_t ws_bytes, void* stream) {
    RLT_CHECK_ARG(labels && terms && ws && B > 0 && S > 0 && (rerank || cls));
    if (ws_bytes < rlt_mt_terms_workspace(B, S)) return RLT_E_WORKSPACE;
    const size_t n = (size_t)B * S;
    const int grid = mt_grid(n);
    hipStream_t st = rlt_stream(stream);
    hipLaunchKernelGGL(mt_partial_kernel, dim3(grid), dim3(256), 0, st, rerank, cls, labels, n, (float*)ws);
    hipLaunchKernelGGL(mt_final_kernel, dim3(1), dim3(64), 0, st, (const float*)ws, grid, (double)n, margin,
                       rerank ? 1 : 0, cls ? 1 : 0, terms);
    return RLT_LAUNCH_RESULT();
}

int rlt_mt_terms_bwd(const float* cls, const float* labels, const float* terms, int B, int S,
                     float w_rerank, float w_class, const float* gscale,
                     float* d_rerank, float* d_class, void* stream) {
    RLT_CHECK_ARG(labels && terms && B > 0 && S > 0 && (d_rerank || d_class));
    RLT_CHECK_ARG(!d_class || cls);
    const size_t n = (size_t)B * S;
    hipLaunchKernelGGL(mt_bwd_kernel, dim3(mt_grid(n)), dim3(256), 0, rlt_stream(stream), cls, labels, terms, n,
                       w_rerank, w_class, gscale, d_rerank, d_class);
    return RLT_LAUNCH_RESULT();
}

int rlt_weighted_sum(const float* const* x, const float* w, int n, float* out, void* stream) {
    RLT_CHECK_ARG(x && w && out && n > 0);
    RLT_CHECK_SHAPE(n <= 8);
    WSumArgs a;
    for (int i = 0; i < 8; ++i) { a.x[i] = i < n ? x[i] : nullptr; a.w[i] = i < n ? w[i] : 0.f; }
    a.n = n;
    for (int i = 0; i < n; ++i) RLT_CHECK_ARG(x[i]);
    hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(1), 0, rlt_stream(stream), a, out);
    return RLT_LAUNCH_RESULT();
}

int rlt_cut_metrics_ex(const float* p, const float* labels, const int32_t* k_in, int B, int S, double penalty,
                       int32_t* k_out, double* f1_out, double* dcg_out, double* sums, void* stream) {
    RLT_CHECK_ARG(labels && (p || k_in) && B > 0 && S > 0);
    RLT_CHECK_ARG(!sums || (f1_out && dcg_out));
    hipStream_t st = rlt_stream(stream);
    hipLaunchKernelGGL(cut_metrics_kernel, dim3(rlt_cdiv(B, 4)), dim3(256), 0, st, p, labels, k_in, B, S, penalty,
                       k_out, f1_out, dcg_out);
    if (sums) hipLaunchKernelGGL(sum2_f64_kernel, dim3(1), dim3(256), 0, st, f1_out, dcg_out, B, sums);
    return RLT_LAUNCH_RESULT();
}

int rlt_cut_metrics(const float* p, const float* labels, const int32_t* k_in, int B, int S,
                    int32_t* k_out, double* f1_out, double* dcg_out, double* sums, void* stream) {
    return rlt_cut_metrics_ex(p, labels, k_in, B, S, -1.0, k_out, f1_out, dcg_out, sums, stream);
}

int rlt_task_metrics(const float* labels, const float* pred, int B, int S,
                     double* dcg_out, double* auc_out, double* sums, void* stream) {
    RLT_CHECK_ARG(labels && pred && dcg_out && auc_out && B > 0 && S > 0);
    RLT_CHECK_SHAPE(S <= TM_MAXS);
    hipStream_t st = rlt_stream(stream);
    hipLaunchKernelGGL(task_metrics_kernel, dim3(rlt_cdiv(B, 4)), dim3(256), 0, st, labels, pred, B, S, dcg_out, auc_out);
    if (sums) hipLaunchKernelGGL(task_metrics_sum_kernel, dim3(1), dim3(256), 0, st, (const double*)dcg_out, (const double*)auc_out, B, sums);
    return RLT_LAUNCH_RESULT();
}

}  // extern "C"

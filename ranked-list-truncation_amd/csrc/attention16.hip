// List-axis attention at head dim 16 (Choopy / MtChoopy: d_model 128, 8 heads - models/Choopy.py:7,11-12), exact fp32, on
// v_mfma_f32_16x16x4_f32.
//
// The 32x32x2 kernels of attention.hip put the head dim on a 32-wide MFMA axis in every product whose OUTPUT is d-indexed
// (P.V, dV, dK, dQ): at 16 that axis is half padding, i.e. forward executes 3 products' worth of matrix-pipe time for 2,
// dQ 4 for 3 and dK+dV 6 for 4.  The 16x16x4 instruction has the same FLOP rate (256 FLOP/clk/CU-SIMD... 2048 FLOP in 8
// passes) and a 16-wide output, so nothing is padded:
//      lane l supplies A[m = l&15][k = l>>4], B[k = l>>4][n = l&15];   D[row = 4*(l>>4) + r][col = l&15] in register r of 4.
// Same structure as attention.hip otherwise: scores are produced with the wavefront's own rows (queries in forward / dQ, keys
// in dK+dV) as the lane-indexed column, so softmax statistics are per-lane values, and an accumulator register r of a 16x16
// score tile is directly the B operand (k = l>>4 <-> tile row 4*(l>>4) + r) of the product that consumes it.  That consumer
// reads its A operand (V^T, K^T, Q^T, dO^T: m = d) from a TRANSPOSED LDS copy of the tile, so that the four rows
// 4*(l>>4) + 0..3 are one ds_read_b128; the score products read the row-major copy ([row][16 + 4 pad], k-step c <-> d =
// 4*(l>>4) + c, one ds_read_b128 per 16 rows as well).  Tiles are 4 KB, so both copies of both matrices, double-buffered,
// are 20-40 KB of LDS per workgroup.
//
// Per wavefront and 64-row tile: forward 64 MFMAs (2048 matrix-pipe cycles; attention.hip at head dim 16: 3072), dQ 96
// (3072; 4096), dK+dV 128 (4096; 6144).  Deterministic, no atomics, same dropout stream as every other attention kernel.
#include "attention_common.h"

namespace {

#ifndef RLT_A16_T
#define RLT_A16_T 64
#endif
#ifndef RLT_A16_NS
#define RLT_A16_NS 2
#endif
#ifndef RLT_A16_OCC
#define RLT_A16_OCC 4
#endif
#ifndef RLT_A16_PRIO
#define RLT_A16_PRIO 1
#endif
// matrix bursts run at raised wave priority (see the forward kernel)
#define PRIO_MFMA() do { if (RLT_A16_PRIO) __builtin_amdgcn_s_setprio(RLT_A16_PRIO); } while (0)
#define PRIO_VALU() do { if (RLT_A16_PRIO) __builtin_amdgcn_s_setprio(0); } while (0)
constexpr int T16 = RLT_A16_T;     // rows per LDS tile
constexpr int LDR = 20;            // row-major copy: floats per row (16 + 4: conflict-free b128 reads at row stride 20)
constexpr int LDT = T16 + 4;       // transposed copy: floats per d row
constexpr int NS = RLT_A16_NS;     // 16-row sub-tiles owned by a wavefront
constexpr int NT = T16 / 16;       // 16-row sub-tiles per LDS tile
constexpr int WROWS = NS * 16;     // rows owned by a wavefront
constexpr int GROWS = 4 * WROWS;   // ... by a workgroup
constexpr int NI = T16 * 4 / 256;  // float4 per thread to stage one [T16][16] tile
static_assert(T16 % 64 == 0 && T16 <= 256, "whole staging passes; the per-row tables are filled by the first T16 threads");

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16x4(const float4& a, const float4& b, f32x4 c) {
    c = mfma16(a.x, b.x, c);
    c = mfma16(a.y, b.y, c);
    c = mfma16(a.z, b.z, c);
    return mfma16(a.w, b.w, c);
}
__device__ __forceinline__ f32x4 mfma16x4(const float4& a, const f32x4& b, f32x4 c) {
    c = mfma16(a.x, b[0], c);
    c = mfma16(a.y, b[1], c);
    c = mfma16(a.z, b[2], c);
    return mfma16(a.w, b[3], c);
}

// one float4 per thread of a [T16][16] tile (row stride ld floats in global); rows beyond nrows read as zero (clamped
// address + select: a guarded load would make hipcc branch around it)
struct Stage { float4 v[NI]; };
__device__ __forceinline__ void stage_load(const float* __restrict__ base, size_t ld, int row0, int nrows, int tid, Stage& r) {
    if (row0 + T16 <= nrows) {           // whole tile in range (uniform): no selects
#pragma unroll
        for (int i = 0; i < NI; ++i)
            r.v[i] = *reinterpret_cast<const float4*>(base + (size_t)(row0 + 64 * i + (tid >> 2)) * ld + 4 * (tid & 3));
        return;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = row0 + 64 * i + (tid >> 2);
        const float4 t = *reinterpret_cast<const float4*>(base + (size_t)min(row, nrows - 1) * ld + 4 * (tid & 3));
        const bool ok = row < nrows;
        r.v[i] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    }
}
__device__ __forceinline__ void stage_rows(float* __restrict__ lds, int tid, const Stage& r) {
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<float4*>(lds + (64 * i + (tid >> 2)) * LDR + 4 * (tid & 3)) = r.v[i];
}
__device__ __forceinline__ void stage_cols(float* __restrict__ lds, int tid, const Stage& r) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        float* p = lds + (4 * (tid & 3)) * LDT + 64 * i + (tid >> 2);
        p[0] = r.v[i].x; p[LDT] = r.v[i].y; p[2 * LDT] = r.v[i].z; p[3 * LDT] = r.v[i].w;
    }
}
// A operand from the row-major copy: m = row sub*16 + (lane&15); component c is k-step c (d = 4*(lane>>4) + c)
__device__ __forceinline__ float4 a_rows(const float* __restrict__ tile, int sub, int l15, int g) {
    return *reinterpret_cast<const float4*>(tile + (sub * 16 + l15) * LDR + 4 * g);
}
// A operand from the transposed copy: m = d = lane&15; component r pairs with register r of a score tile (row sub*16 + 4*(lane>>4) + r)
__device__ __forceinline__ float4 a_cols(const float* __restrict__ tileT, int sub, int l15, int g) {
    return *reinterpret_cast<const float4*>(tileT + l15 * LDT + sub * 16 + 4 * g);
}
// this lane's B operand of a score product: own row (clamped), d = 4*(lane>>4) + c
__device__ __forceinline__ float4 own_row(const float* __restrict__ rowp, int g, float mul) {
    const float4 v = *reinterpret_cast<const float4*>(rowp + 4 * g);
    return make_float4(v.x * mul, v.y * mul, v.z * mul, v.w * mul);
}
// max / sum over the four lanes (l&15) + 16*{0,1,2,3} that share a column
__device__ __forceinline__ float col_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float col_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ------------------------------------------------------------------------------------------ forward
template <bool DROP>
__global__ __launch_bounds__(256, RLT_A16_OCC) void attn16_fwd_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[2][T16 * LDR];      // K, row-major
    __shared__ __attribute__((aligned(16))) float Vt[2][16 * LDT];       // V, transposed
    __shared__ __attribute__((aligned(16))) uint32_t Hc[2][T16];         // dropout: column (key) hashes of the tile
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int B = a.B, H = a.H, E = H * 16;
    const size_t ld = (size_t)3 * E;
    int pair, qt;
    map_block(blockIdx.x, a.S * H, rlt_cdiv_dev(B, GROWS), pair, qt);
    const int s = pair / H, h = pair % H;
    const float* base = a.qkv + (size_t)s * B * ld + h * 16;
    const int row0 = qt * GROWS + wv * WROWS;
    const bool wave_live = row0 < B;
    const uint32_t ps = pair_seed(a.seed, pair);

    float4 qb[NS];
    uint32_t hq[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        const int q = row0 + n * 16 + l15;
        qb[n] = own_row(base + (size_t)min(q, B - 1) * ld, g, a.scale * LOG2E);
        hq[n] = DROP ? rlt_row_hash(ps, (uint32_t)q) : 0u;
    }
    f32x4 o[NS];
    float m_run[NS], l_run[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        m_run[n] = 0.f; l_run[n] = 0.f;          // m_run: the reference of the weights, set by the first tile
    }
    const float inv_keep = DROP ? 1.f / (1.f - a.drop_p) : 1.f;

    const int nt = rlt_cdiv_dev(B, T16);
    Stage rk, rv;
    stage_load(base + E, ld, 0, B, tid, rk);
    stage_load(base + 2 * E, ld, 0, B, tid, rv);
    stage_rows(Ks[0], tid, rk);
    stage_cols(Vt[0], tid, rv);
    if (DROP && tid < T16) Hc[0][tid] = rlt_col_hash(ps, (uint32_t)tid);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) {
            stage_load(base + E, ld, (t + 1) * T16, B, tid, rk);
            stage_load(base + 2 * E, ld, (t + 1) * T16, B, tid, rv);
        }
        if (wave_live) {
            // scores are accumulated ON TOP of -m_run (the accumulator's initial value; a query is a lane, so it is a per-lane
            // constant): the tile's weights are exp2(sc) with no subtraction, and the reference max m_run moves only when a
            // weight would leave the comfortable fp32 range (lazy rescaling - the quotient O / l does not depend on the
            // reference, rounding aside).  The common tile therefore has no max search, no cross-lane step, no rescale of O:
            // on gfx950 the fp32 MFMA shares the vector ALU (tools/micro/mfma_valu_overlap.hip: MFMA and VALU time of one SIMD
            // ADD), so every vector instruction removed from the tile is time won.
            f32x4 sc[NT][NS];                                   // S^T[key][q] - m_run[q], log2 domain
            PRIO_MFMA();
#pragma unroll
            for (int ks = 0; ks < NT; ++ks) {
                const float4 ak = a_rows(Ks[buf], ks, l15, g);
#pragma unroll
                for (int n = 0; n < NS; ++n) {
                    const float c0 = -m_run[n];
                    sc[ks][n] = mfma16x4(ak, qb[n], f32x4{c0, c0, c0, c0});
                }
            }
            PRIO_VALU();
            if ((t + 1) * T16 > B) {                             // last tile only: keys beyond B
#pragma unroll
                for (int ks = 0; ks < NT; ++ks)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (t * T16 + ks * 16 + 4 * g + r >= B) {
#pragma unroll
                            for (int n = 0; n < NS; ++n) sc[ks][n][r] = -INFINITY;
                        }
            }
            f32x4 pe[NT][NS];
            float psum[NS];
            bool redo = t == 0;                                  // the first tile sets the reference
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                psum[n] = 0.f;
#pragma unroll
                for (int ks = 0; ks < NT; ++ks)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pe[ks][n][r] = rlt_exp2(sc[ks][n][r]);
                        psum[n] += pe[ks][n][r];
                    }
                redo |= !(psum[n] <= 4096.f);                    // some weight above ~2^8..2^12 (or not finite): move the reference
            }
            if (__any(redo)) {                                   // wave-uniform; rare after the first tile
#pragma unroll
                for (int n = 0; n < NS; ++n) {
                    float tmax = -INFINITY;
#pragma unroll
                    for (int ks = 0; ks < NT; ++ks)
#pragma unroll
                        for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, sc[ks][n][r]);
                    tmax = col_max(tmax);                        // relative to the old reference
                    const float d = t == 0 ? tmax : fmaxf(tmax, 0.f);
                    const float alpha = t == 0 ? 0.f : rlt_exp2(-d);
                    psum[n] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < NT; ++ks)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            pe[ks][n][r] = rlt_exp2(sc[ks][n][r] - d);
                            psum[n] += pe[ks][n][r];
                        }
                    l_run[n] *= alpha;
                    m_run[n] += d;
                    o[n] *= alpha;
                }
            }
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                l_run[n] += psum[n];
#pragma unroll
                for (int ks = 0; ks < NT; ++ks) sc[ks][n] = pe[ks][n];
                if (DROP) {                 // on the normalised probabilities: the normaliser keeps every key
#pragma unroll
                    for (int ks = 0; ks < NT; ++ks) {
                        const uint4 hc = *reinterpret_cast<const uint4*>(&Hc[buf][ks * 16 + 4 * g]);
                        sc[ks][n][0] = rlt_keep_rc(hq[n], hc.x, a.drop_thr) ? sc[ks][n][0] * inv_keep : 0.f;
                        sc[ks][n][1] = rlt_keep_rc(hq[n], hc.y, a.drop_thr) ? sc[ks][n][1] * inv_keep : 0.f;
                        sc[ks][n][2] = rlt_keep_rc(hq[n], hc.z, a.drop_thr) ? sc[ks][n][2] * inv_keep : 0.f;
                        sc[ks][n][3] = rlt_keep_rc(hq[n], hc.w, a.drop_thr) ? sc[ks][n][3] * inv_keep : 0.f;
                    }
                }
            }
            PRIO_MFMA();
#pragma unroll
            for (int ks = 0; ks < NT; ++ks) {
                const float4 av = a_cols(Vt[buf], ks, l15, g);
#pragma unroll
                for (int n = 0; n < NS; ++n) o[n] = mfma16x4(av, sc[ks][n], o[n]);       // O^T[d][q]
            }
            PRIO_VALU();
        }
        if (t + 1 < nt) {
            stage_rows(Ks[buf ^ 1], tid, rk);
            stage_cols(Vt[buf ^ 1], tid, rv);
            if (DROP && tid < T16) Hc[buf ^ 1][tid] = rlt_col_hash(ps, (uint32_t)((t + 1) * T16 + tid));
        }
        __syncthreads();
    }
    if (!wave_live) return;
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        const float l_tot = col_sum(l_run[n]);
        const int q = row0 + n * 16 + l15;
        if (q < B) {
            const float inv = 1.f / l_tot;
            *reinterpret_cast<float4*>(a.o + ((size_t)s * B + q) * E + h * 16 + 4 * g) =
                make_float4(o[n][0] * inv, o[n][1] * inv, o[n][2] * inv, o[n][3] * inv);
            if (g == 0) a.lse_o[((size_t)s * H + h) * B + q] = (m_run[n] + log2f(l_tot)) * LN2;
        }
    }
}

// ------------------------------------------------------------------------------------------ dQ
template <bool DROP>
__global__ __launch_bounds__(256, RLT_A16_OCC) void attn16_bwd_dq_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[2][T16 * LDR];      // K, row-major (scores)
    __shared__ __attribute__((aligned(16))) float Kt[2][16 * LDT];       // K, transposed (dQ)
    __shared__ __attribute__((aligned(16))) float Vs[2][T16 * LDR];      // V, row-major (dP)
    __shared__ __attribute__((aligned(16))) uint32_t Hc[2][T16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int B = a.B, H = a.H, E = H * 16;
    const size_t ld = (size_t)3 * E;
    int pair, qt;
    map_block(blockIdx.x, a.S * H, rlt_cdiv_dev(B, GROWS), pair, qt);
    const int s = pair / H, h = pair % H;
    const float* base = a.qkv + (size_t)s * B * ld + h * 16;
    const int row0 = qt * GROWS + wv * WROWS;
    const bool wave_live = row0 < B;
    const uint32_t ps = pair_seed(a.seed, pair);

    float4 qb[NS], dob[NS];
    float lse2[NS], del[NS];
    uint32_t hq[NS];
    f32x4 dq[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        const int q = row0 + n * 16 + l15, qc = min(q, B - 1);
        qb[n] = own_row(base + (size_t)qc * ld, g, a.scale * LOG2E);
        dob[n] = own_row(a.dout + ((size_t)s * B + qc) * E + h * 16, g, 1.f);
        lse2[n] = a.lse[((size_t)s * H + h) * B + qc] * LOG2E;
        del[n] = a.delta[((size_t)s * H + h) * B + qc];
        hq[n] = DROP ? rlt_row_hash(ps, (uint32_t)q) : 0u;
        dq[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float inv_keep = DROP ? 1.f / (1.f - a.drop_p) : 1.f;

    const int nt = rlt_cdiv_dev(B, T16);
    Stage rk, rv;
    stage_load(base + E, ld, 0, B, tid, rk);
    stage_load(base + 2 * E, ld, 0, B, tid, rv);
    stage_rows(Ks[0], tid, rk);
    stage_cols(Kt[0], tid, rk);
    stage_rows(Vs[0], tid, rv);
    if (DROP && tid < T16) Hc[0][tid] = rlt_col_hash(ps, (uint32_t)tid);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) {
            stage_load(base + E, ld, (t + 1) * T16, B, tid, rk);
            stage_load(base + 2 * E, ld, (t + 1) * T16, B, tid, rv);
        }
        if (wave_live) {
            const bool tail = (t + 1) * T16 > B;
#pragma unroll
            for (int ks = 0; ks < NT; ++ks) {
                const float4 ak = a_rows(Ks[buf], ks, l15, g), av = a_rows(Vs[buf], ks, l15, g);
                const float4 at = a_cols(Kt[buf], ks, l15, g);
                uint4 hc = make_uint4(0u, 0u, 0u, 0u);
                if (DROP) hc = *reinterpret_cast<const uint4*>(&Hc[buf][ks * 16 + 4 * g]);
                const uint32_t hcs[4] = {hc.x, hc.y, hc.z, hc.w};
#pragma unroll
                for (int n = 0; n < NS; ++n) {
                    // -lse and -delta of the lane's query are the accumulators' initial values: P = exp2(sc), dS = P * dp with no
                    // subtraction per element; a key beyond B starts at -inf and stays there (P = 0)
                    f32x4 s0 = f32x4{-lse2[n], -lse2[n], -lse2[n], -lse2[n]};
                    if (tail) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (t * T16 + ks * 16 + 4 * g + r >= B) s0[r] = -INFINITY;
                    }
                    const float d0 = DROP ? 0.f : -del[n];
                    PRIO_MFMA();
                    f32x4 sc = mfma16x4(ak, qb[n], s0);                              // S^T[key][q] - lse[q]
                    f32x4 dp = mfma16x4(av, dob[n], f32x4{d0, d0, d0, d0});          // dP^T[key][q] (- delta[q])
                    PRIO_VALU();
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = rlt_exp2(sc[r]);
                        if (DROP) dp[r] = p * ((rlt_keep_rc(hq[n], hcs[r], a.drop_thr) ? dp[r] * inv_keep : 0.f) - del[n]);
                        else dp[r] = p * dp[r];                                      // dS^T
                    }
                    PRIO_MFMA();
                    dq[n] = mfma16x4(at, dp, dq[n]);                                 // dQ^T[d][q] += K^T dS^T
                    PRIO_VALU();
                }
            }
        }
        if (t + 1 < nt) {
            stage_rows(Ks[buf ^ 1], tid, rk);
            stage_cols(Kt[buf ^ 1], tid, rk);
            stage_rows(Vs[buf ^ 1], tid, rv);
            if (DROP && tid < T16) Hc[buf ^ 1][tid] = rlt_col_hash(ps, (uint32_t)((t + 1) * T16 + tid));
        }
        __syncthreads();
    }
    if (!wave_live) return;
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        const int q = row0 + n * 16 + l15;
        if (q < B)
            *reinterpret_cast<float4*>(a.dqkv + ((size_t)s * B + q) * ld + h * 16 + 4 * g) =
                make_float4(dq[n][0] * a.scale, dq[n][1] * a.scale, dq[n][2] * a.scale, dq[n][3] * a.scale);
    }
}

// ------------------------------------------------------------------------------------------ dK, dV
template <bool DROP>
__global__ __launch_bounds__(256, RLT_A16_OCC) void attn16_bwd_dkv_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Qs[2][T16 * LDR];      // Q row-major (scores), transposed (dK)
    __shared__ __attribute__((aligned(16))) float Qt[2][16 * LDT];
    __shared__ __attribute__((aligned(16))) float Ds[2][T16 * LDR];      // dO row-major (dP), transposed (dV)
    __shared__ __attribute__((aligned(16))) float Dt[2][16 * LDT];
    __shared__ __attribute__((aligned(16))) float Ls[2][T16];            // -lse * log2e of the tile's queries (-inf beyond B)
    __shared__ __attribute__((aligned(16))) float Es[2][T16];            // -delta
    __shared__ __attribute__((aligned(16))) uint32_t Hr[2][T16];         // dropout: row (query) hashes of the tile
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int B = a.B, H = a.H, E = H * 16;
    const size_t ld = (size_t)3 * E;
    int pair, ktile;
    map_block(blockIdx.x, a.S * H, rlt_cdiv_dev(B, GROWS), pair, ktile);
    const int s = pair / H, h = pair % H;
    const float* base = a.qkv + (size_t)s * B * ld + h * 16;
    const float* dobase = a.dout + (size_t)s * B * E + h * 16;
    const float* lsebase = a.lse + ((size_t)s * H + h) * B;
    const float* delbase = a.delta + ((size_t)s * H + h) * B;
    const int row0 = ktile * GROWS + wv * WROWS;
    const bool wave_live = row0 < B;
    const uint32_t ps = pair_seed(a.seed, pair);

    float4 kb[NS], vb[NS];
    uint32_t hk[NS];
    f32x4 dk[NS], dv[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        const int key = row0 + n * 16 + l15, kc = min(key, B - 1);
        kb[n] = own_row(base + (size_t)kc * ld + E, g, a.scale * LOG2E);
        vb[n] = own_row(base + (size_t)kc * ld + 2 * E, g, 1.f);
        hk[n] = DROP ? rlt_col_hash(ps, (uint32_t)key) : 0u;
        dk[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float inv_keep = DROP ? 1.f / (1.f - a.drop_p) : 1.f;

    const int nt = rlt_cdiv_dev(B, T16);
    Stage rq, rd;
    float rl = 0.f, re = 0.f;
    auto load_tile = [&](int r0) {
        stage_load(base, ld, r0, B, tid, rq);
        stage_load(dobase, (size_t)E, r0, B, tid, rd);
        if (tid < T16) {
            const int qi = r0 + tid, qc = min(qi, B - 1);
            const float l = lsebase[qc], e = delbase[qc];
            rl = qi < B ? -l * LOG2E : -INFINITY;          // negated: initial values of the score / dP accumulators
            re = qi < B ? -e : 0.f;
        }
    };
    auto store_tile = [&](int b, int r0) {
        stage_rows(Qs[b], tid, rq);
        stage_cols(Qt[b], tid, rq);
        stage_rows(Ds[b], tid, rd);
        stage_cols(Dt[b], tid, rd);
        if (tid < T16) {
            Ls[b][tid] = rl; Es[b][tid] = re;
            if (DROP) Hr[b][tid] = rlt_row_hash(ps, (uint32_t)(r0 + tid));
        }
    };
    load_tile(0);
    store_tile(0, 0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tile((t + 1) * T16);
        if (wave_live) {
#pragma unroll
            for (int qs = 0; qs < NT; ++qs) {
                const float4 aq = a_rows(Qs[buf], qs, l15, g), ad = a_rows(Ds[buf], qs, l15, g);
                const float4 tq = a_cols(Qt[buf], qs, l15, g), td = a_cols(Dt[buf], qs, l15, g);
                const float4 l4 = *reinterpret_cast<const float4*>(&Ls[buf][qs * 16 + 4 * g]);
                const float4 e4 = *reinterpret_cast<const float4*>(&Es[buf][qs * 16 + 4 * g]);
                const float es[4] = {e4.x, e4.y, e4.z, e4.w};
                uint4 hr = make_uint4(0u, 0u, 0u, 0u);
                if (DROP) hr = *reinterpret_cast<const uint4*>(&Hr[buf][qs * 16 + 4 * g]);
                const uint32_t hrs[4] = {hr.x, hr.y, hr.z, hr.w};
#pragma unroll
                for (int n = 0; n < NS; ++n) {
                    // the accumulators start from -lse and -delta of their ROW (the tile's queries: the LDS tables hold the
                    // negated values, -inf for a query beyond B, whose weights are then exp2(-inf) = 0)
                    PRIO_MFMA();
                    f32x4 sc = mfma16x4(aq, kb[n], f32x4{l4.x, l4.y, l4.z, l4.w});                       // S[q][key] - lse[q]
                    f32x4 dp = mfma16x4(ad, vb[n], DROP ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{e4.x, e4.y, e4.z, e4.w});   // dP (- delta)
                    PRIO_VALU();
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = rlt_exp2(sc[r]);
                        if (DROP) {
                            const float m = rlt_keep_rc(hrs[r], hk[n], a.drop_thr) ? inv_keep : 0.f;
                            sc[r] = p * m;                                           // dropped P (feeds dV)
                            dp[r] = p * (dp[r] * m + es[r]);                         // dS
                        } else {
                            sc[r] = p;
                            dp[r] = p * dp[r];
                        }
                    }
                    PRIO_MFMA();
                    dv[n] = mfma16x4(td, sc, dv[n]);                                 // dV^T[d][key] += dO^T P
                    dk[n] = mfma16x4(tq, dp, dk[n]);                                 // dK^T[d][key] += Q^T dS
                    PRIO_VALU();
                }
            }
        }
        if (t + 1 < nt) store_tile(buf ^ 1, (t + 1) * T16);
        __syncthreads();
    }
    if (!wave_live) return;
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        const int key = row0 + n * 16 + l15;
        if (key < B) {
            float* drow = a.dqkv + ((size_t)s * B + key) * ld + h * 16 + 4 * g;
            *reinterpret_cast<float4*>(drow + E) =
                make_float4(dk[n][0] * a.scale, dk[n][1] * a.scale, dk[n][2] * a.scale, dk[n][3] * a.scale);
            *reinterpret_cast<float4*>(drow + 2 * E) = make_float4(dv[n][0], dv[n][1], dv[n][2], dv[n][3]);
        }
    }
}

template <bool DROP>
int launch16(int which, const AttnArgs& a, hipStream_t st) {
    const int grid = a.S * a.H * rlt_cdiv(a.B, GROWS);
    if (which == 0) hipLaunchKernelGGL((attn16_fwd_kernel<DROP>), dim3(grid), dim3(256), 0, st, a);
    else if (which == 1) hipLaunchKernelGGL((attn16_bwd_dkv_kernel<DROP>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((attn16_bwd_dq_kernel<DROP>), dim3(grid), dim3(256), 0, st, a);
    return RLT_LAUNCH_RESULT();
}

}  // namespace

// which = 0 forward, 1 dK/dV, 2 dQ (head dim 16, exact fp32)
int rlt_attn16_run(int which, const AttnArgs& a, hipStream_t st) {
    return a.drop_p > 0.f ? launch16<true>(which, a, st) : launch16<false>(which, a, st);
}

// List-axis multi-head self-attention (row M3 of SURVEY.md section 8a), forward and backward.
//
// In the reference every nn.TransformerEncoderLayer is batch_first=False but receives (B,S,E)
// (models/AttnCut.py:9,17-18), so at each position s and head h the B lists of the mini-batch
// attend to each other: a dense B x B score matrix per (s,h) - 1200 of them at AttnCut's
// S=300, H=4.  With position-major activations the B rows of one position are contiguous.
//
// Flash-style, fp32 throughout on the f32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32 products):
// the score matrix is never written to memory.  One wavefront owns 32 queries (or 32 keys in the
// dK/dV kernel); scores are produced TRANSPOSED (S^T = K Q^T) so that a query is a lane: the online
// softmax over keys is a per-lane loop over the 16 accumulator registers (its reference moves lazily:
// a cross-half shuffle and a rescale of O only when a score exceeds it by 2^8), and the probability
// registers are directly the B operand of the P.V product - no LDS round trip, no conversion.  K/V
// (or Q/dO) tiles of 64 rows are double-buffered in LDS with a register prefetch, shared by the 4
// wavefronts of a workgroup.  The fp32 MFMA occupies the vector ALU on gfx950 (tools/micro/
// mfma_valu_overlap.hip), so the kernels are written for the fewest vector instructions per score:
// backward accumulators seeded with -lse / -delta, -inf seeds instead of range tests, select-free
// staging of whole tiles.  Head dim 16 has its own file (attention16.hip: 16x16x4 MFMA).
//
// Backward recomputes the probabilities from the saved log-sum-exp (no B x B stash):
//   kernel dKV: workgroup = 128 keys, loops over queries, accumulates dK^T, dV^T in registers
//   kernel dQ : workgroup = 128 queries, loops over keys, accumulates dQ^T in registers
// so every gradient element is written exactly once: deterministic, no float atomics.
//
// FLOPs per (s,h): forward 4*B^2*HD; backward 14*B^2*HD (7 products; 5 would need atomics).
#include "attention_common.h"
#include <stdlib.h>
#include <string.h>

namespace {

// stage a [KT][HD] tile (row stride ld floats in global, HD+4 in LDS) through registers
template <int HD>
struct TileRegs { float4 v[(KT * HD / 4 + 255) / 256]; };

template <int HD>
__device__ __forceinline__ void tile_load(const float* __restrict__ base, size_t ld, int row0, int nrows, int tid,
                                          TileRegs<HD>& r) {
    constexpr int PER_ROW = HD / 4, N4 = KT * PER_ROW, NI = (N4 + 255) / 256;
    static_assert(N4 % 256 == 0, "tile must be a whole number of 256-thread passes");
    if (row0 + KT <= nrows) {            // whole tile in range (uniform): plain loads, no per-element select - the vector
                                         // ALU time of those selects is not hidden behind the fp32 MFMAs on gfx950
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = tid + 256 * i;
            r.v[i] = *reinterpret_cast<const float4*>(base + (size_t)(row0 + idx / PER_ROW) * ld + 4 * (idx % PER_ROW));
        }
        return;
    }
    // last tile: clamp the row into the matrix, load unconditionally, zero out-of-range rows with a select (a guarded
    // load makes hipcc branch around it and drain vmcnt, serialising the burst)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = tid + 256 * i;
        const int row = row0 + idx / PER_ROW, c4 = idx % PER_ROW;
        const bool ok = row < nrows;
        const float4 t = *reinterpret_cast<const float4*>(base + (size_t)min(row, nrows - 1) * ld + 4 * c4);
        r.v[i] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    }
}
template <int HD>
__device__ __forceinline__ void tile_store(float* __restrict__ lds, int tid, const TileRegs<HD>& r) {
    constexpr int PER_ROW = HD / 4, N4 = KT * PER_ROW, NI = (N4 + 255) / 256, LD = HD + 4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx / PER_ROW, c4 = idx % PER_ROW;
        *reinterpret_cast<float4*>(lds + row * LD + 4 * c4) = r.v[i];
    }
}

// this lane's half of a row: regs[ks] = row[hh*HD/2 + ks] * mul, ks < HD/2
template <int HD>
__device__ __forceinline__ void row_half_load(const float* __restrict__ rowp, int hh, float mul, float (&regs)[HD / 2]) {
#pragma unroll
    for (int j = 0; j < HD / 8; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(rowp + hh * (HD / 2) + 4 * j);
        regs[4 * j + 0] = v.x * mul; regs[4 * j + 1] = v.y * mul;
        regs[4 * j + 2] = v.z * mul; regs[4 * j + 3] = v.w * mul;
    }
}

// acc += Tile[rows sub*32 + (lane&31)][all d] (A operand, from LDS) x regs (B operand):
// result D[row = tile row][col = lane&31]
template <int HD>
__device__ __forceinline__ f32x16 mma_tile_rows(const float* __restrict__ tile, int sub, int l31, int hh,
                                                const float (&regs)[HD / 2], f32x16 acc) {
    constexpr int LD = HD + 4;
    const float* rp = tile + (sub * 32 + l31) * LD + hh * (HD / 2);
    // the A operands of group j+1 are read from LDS BEFORE the four MFMAs of group j: hipcc otherwise issues each read right
    // in front of its MFMAs and waits out the LDS latency once per group (ISA: `D W M4 D W M4 ...`; round 3: dK+dV 82.9 ->
    // 79.4 ms, 256 registers + scratch -> 240.  Chaining the first group of the NEXT product into the current one as well:
    // no further gain, not kept)
    float4 a = *reinterpret_cast<const float4*>(rp);
#pragma unroll
    for (int j = 0; j < HD / 8; ++j) {
        float4 an = a;
        if (j + 1 < HD / 8) an = *reinterpret_cast<const float4*>(rp + 4 * (j + 1));
        __builtin_amdgcn_sched_barrier(0);
        acc = mfma32(a.x, regs[4 * j + 0], acc);
        acc = mfma32(a.y, regs[4 * j + 1], acc);
        acc = mfma32(a.z, regs[4 * j + 2], acc);
        acc = mfma32(a.w, regs[4 * j + 3], acc);
        a = an;
    }
    return acc;
}
// acc[dt] += sum over the 32 rows of sub-tile `sub`:  Tile[row][dt*32 + i]^T-ish product
//   D[row = d (dt*32 + i)][col = lane&31] += sum_r Tile[sub*32 + acc_row(r,hh)][dt*32 + (lane&31)] * w[r]
// i.e. A operand = Tile^T read column-wise (ds_read_b32, consecutive lanes -> consecutive d),
//      B operand = the accumulator registers w of a previous product (rows of that product = k).
template <int HD>
__device__ __forceinline__ void mma_tile_cols(const float* __restrict__ tile, int sub, int l31, int hh,
                                              const f32x16& w, f32x16 (&acc)[(HD + 31) / 32]) {
    constexpr int LD = HD + 4, DT = (HD + 31) / 32;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int d = dt * 32 + l31;
        const bool ok = d < HD;
        const float* cp = tile + (sub * 32 + 4 * hh) * LD + (ok ? d : 0);
        // operands of the next four MFMAs read ahead of the current four (see mma_tile_rows)
        float a[4], an[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = cp[(i + 8 * 0) * LD];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) an[i] = cp[(i + 8 * (g + 1)) * LD];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[dt] = mfma32(ok ? a[i] : 0.f, w[4 * g + i], acc[dt]);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = an[i];
        }
    }
}

// ------------------------------------------------------------------------------------------ forward
// SB ("single buffer", head dim 64): ONE LDS copy of the K / V tile (35 KB per workgroup) and a 168-register cap, so that THREE
// workgroups = three wavefronts per SIMD share a CU instead of two (the double-buffered form needs 70 KB); the price is a
// second barrier per tile, which two other workgroups on the SIMD cover.
template <int HD, bool DROP, bool SB>
__global__ __launch_bounds__(256, SB ? 3 : 1) void attn_fwd_kernel(AttnArgs a) {
    constexpr int LD = HD + 4, DT = (HD + 31) / 32, NB = SB ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                       // [NB][KT*LD]
    float* Vs = smem + NB * KT * LD;        // [NB][KT*LD]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int B = a.B, H = a.H, E = H * HD;
    const size_t ld = (size_t)3 * E;
    const int ntile = rlt_cdiv_dev(B, QT);
    int pair, qt;
    map_block(blockIdx.x, a.S * H, ntile, pair, qt);
    const int s = pair / H, h = pair % H;
    const float* base = a.qkv + (size_t)s * B * ld + h * HD;     // row b of this position: base + b*ld
    const int q = qt * QT + wv * 32 + l31;
    const bool wave_live = qt * QT + wv * 32 < B;
    const int qc = min(q, B - 1);

    float qreg[HD / 2];
    row_half_load<HD>(base + (size_t)qc * ld, hh, a.scale * LOG2E, qreg);

    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    TileRegs<HD> rk, rv;
    const int nt = rlt_cdiv_dev(B, KT);
    tile_load<HD>(base + E, ld, 0, B, tid, rk);
    tile_load<HD>(base + 2 * E, ld, 0, B, tid, rv);
    tile_store<HD>(Ks, tid, rk);
    tile_store<HD>(Vs, tid, rv);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = SB ? 0 : (t & 1);
        if (!SB && t + 1 < nt) {
            tile_load<HD>(base + E, ld, (t + 1) * KT, B, tid, rk);
            tile_load<HD>(base + 2 * E, ld, (t + 1) * KT, B, tid, rv);
        }
        if (wave_live) {
            const float* kt_ = Ks + buf * KT * LD;
            const float* vt_ = Vs + buf * KT * LD;
            f32x16 sc[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[sub][r] = 0.f;
                sc[sub] = mma_tile_rows<HD>(kt_, sub, l31, hh, qreg, sc[sub]);     // S^T[key][q], log2 domain
            }
            if ((t + 1) * KT > B) {               // last tile only: keys beyond B
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * KT + sub * 32 + acc_row(r, hh) >= B) sc[sub][r] = -INFINITY;
            }
            // lazy rescaling: the reference m_run of the weights exp2(sc - m_run) moves only when a score of this tile exceeds
            // it by more than 8 (O / l does not depend on the reference, and weights up to 2^8 are comfortable in fp32), so the
            // common tile has no cross-lane step, no rescale of O and l.  On gfx950 the fp32 MFMA and the vector ALU do not
            // overlap (tools/micro/mfma_valu_overlap.hip: their times add): every vector instruction removed is time won.
            float tmax = -INFINITY;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sc[sub][r]);
            if (__any(t == 0 || tmax > m_run + 8.f)) {          // wave-uniform; the first tile sets the reference
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = rlt_exp2(m_run - m_new);
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
            }
            float psum = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = rlt_exp2(sc[sub][r] - m_run);
                    sc[sub][r] = p;
                    psum += p;
                }
            l_run += psum;
            if (DROP) {                    // dropout acts on the normalised probabilities: the normaliser keeps all keys
                // (compiled out at p = 0: hipcc if-converts a run-time test and executes the hash regardless)
                const uint32_t ps = pair_seed(a.seed, pair);
                const uint32_t hq = rlt_row_hash(ps, (uint32_t)q);
                const float inv_keep = 1.f / (1.f - a.drop_p);
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = t * KT + sub * 32 + acc_row(r, hh);
                        sc[sub][r] = rlt_keep_rc(hq, rlt_col_hash(ps, (uint32_t)key), a.drop_thr) ? sc[sub][r] * inv_keep : 0.f;
                    }
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) mma_tile_cols<HD>(vt_, sub, l31, hh, sc[sub], oacc);   // O^T[d][q]
        }
        if (SB) {                                // loads after the tile body (no staging registers live across it), then swap
            if (t + 1 < nt) {
                tile_load<HD>(base + E, ld, (t + 1) * KT, B, tid, rk);
                tile_load<HD>(base + 2 * E, ld, (t + 1) * KT, B, tid, rv);
            }
            __syncthreads();
        }
        if (t + 1 < nt) {
            tile_store<HD>(Ks + (SB ? 0 : (buf ^ 1)) * KT * LD, tid, rk);
            tile_store<HD>(Vs + (SB ? 0 : (buf ^ 1)) * KT * LD, tid, rv);
        }
        __syncthreads();
    }
    if (!wave_live) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (q < B) {
        store_acc_T<HD>(a.o + ((size_t)s * B + q) * E + h * HD, hh, oacc, 1.f / l_tot);
        if (hh == 0) a.lse_o[((size_t)s * H + h) * B + q] = (m_run + log2f(l_tot)) * LN2;
    }
}

// ------------------------------------------------------------------------------------------ delta
// delta[s][h][b] = sum_d dO[t][h*HD+d] * O[t][h*HD+d]
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ dout,
                                                         int S, int B, int H, int HD, float* __restrict__ delta) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t T = (size_t)S * B;
    const int E = H * HD;
    for (size_t t = (size_t)blockIdx.x * 4 + wv; t < T; t += (size_t)gridDim.x * 4) {
        const int s = (int)(t / B), b = (int)(t % B);
        for (int h = 0; h < H; ++h) {
            float v = 0.f;
            for (int d = lane; d < HD; d += 64) v += o[t * E + h * HD + d] * dout[t * E + h * HD + d];
            v = wave_sum(v);
            if (lane == 0) delta[((size_t)s * H + h) * B + b] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------ dK, dV
template <int HD, int OCC, bool DROP>
__global__ __launch_bounds__(256, OCC) void attn_bwd_dkv_kernel(AttnArgs a) {
    constexpr int LD = HD + 4, DT = (HD + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;                         // [2][KT*LD]
    float* Ds = smem + 2 * KT * LD;           // [2][KT*LD]   dO
    float* Ls = smem + 4 * KT * LD;           // [2][KT]      -lse * log2e (-inf beyond B)
    float* Es = Ls + 2 * KT;                  // [2][KT]      -delta
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int B = a.B, H = a.H, E = H * HD;
    const size_t ld = (size_t)3 * E;
    const int ntile = rlt_cdiv_dev(B, QT);
    int pair, ktile;
    map_block(blockIdx.x, a.S * H, ntile, pair, ktile);
    const int s = pair / H, h = pair % H;
    const float* base = a.qkv + (size_t)s * B * ld + h * HD;
    const float* dobase = a.dout + (size_t)s * B * E + h * HD;
    const float* lsebase = a.lse + ((size_t)s * H + h) * B;
    const float* delbase = a.delta + ((size_t)s * H + h) * B;
    const int key = ktile * QT + wv * 32 + l31;
    const bool wave_live = ktile * QT + wv * 32 < B;
    const int kc = min(key, B - 1);

    float kreg[HD / 2], vreg[HD / 2];
    row_half_load<HD>(base + (size_t)kc * ld + E, hh, a.scale * LOG2E, kreg);
    row_half_load<HD>(base + (size_t)kc * ld + 2 * E, hh, 1.f, vreg);

    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }

    TileRegs<HD> rq, rd;
    float rl = 0.f, re = 0.f;
    const int nt = rlt_cdiv_dev(B, KT);
    auto load_small = [&](int row0) {
        if (tid < KT) {
            const int qi = row0 + tid, qc = min(qi, B - 1);
            const float l = lsebase[qc], e = delbase[qc];
            rl = qi < B ? -l * LOG2E : -INFINITY;       // negated: the initial values of the score / dP accumulators
            re = qi < B ? -e : 0.f;
        }
    };
    tile_load<HD>(base, ld, 0, B, tid, rq);
    tile_load<HD>(dobase, (size_t)E, 0, B, tid, rd);
    load_small(0);
    tile_store<HD>(Qs, tid, rq);
    tile_store<HD>(Ds, tid, rd);
    if (tid < KT) { Ls[tid] = rl; Es[tid] = re; }
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) {
            tile_load<HD>(base, ld, (t + 1) * KT, B, tid, rq);
            tile_load<HD>(dobase, (size_t)E, (t + 1) * KT, B, tid, rd);
            load_small((t + 1) * KT);
        }
        if (wave_live) {
            const float* qt_ = Qs + buf * KT * LD;
            const float* dt_ = Ds + buf * KT * LD;
            const float* lt_ = Ls + buf * KT;
            const float* et_ = Es + buf * KT;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                // the accumulators start from -lse and -delta of their ROW (the LDS tables hold the negated values, -inf for a
                // query beyond B, whose weights are then exp2(-inf) = 0): P = exp2(sc), dS = P * dp - no subtraction and no
                // range test per element.  On gfx950 the fp32 MFMA and the vector ALU do not overlap
                // (tools/micro/mfma_valu_overlap.hip: their times ADD), so each vector instruction removed is time won.
                f32x16 sc, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = sub * 32 + acc_row(r, hh);
                    sc[r] = lt_[ql];
                    dp[r] = DROP ? 0.f : et_[ql];
                }
                sc = mma_tile_rows<HD>(qt_, sub, l31, hh, kreg, sc);    // S[q][key] - lse[q] (log2 domain)
                dp = mma_tile_rows<HD>(dt_, sub, l31, hh, vreg, dp);    // dP[q][key] (- delta[q])
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = rlt_exp2(sc[r]);
                    if (DROP) {
                        const int ql = sub * 32 + acc_row(r, hh);
                        const bool keep = rlt_keep(pair_seed(a.seed, pair), (uint32_t)(t * KT + ql), (uint32_t)key, a.drop_thr);
                        const float m = keep ? 1.f / (1.f - a.drop_p) : 0.f;
                        sc[r] = p * m;                                       // dropped P (feeds dV)
                        dp[r] = p * (dp[r] * m + et_[ql]);                   // dS
                    } else {
                        sc[r] = p;
                        dp[r] = p * dp[r];
                    }
                }
                mma_tile_cols<HD>(dt_, sub, l31, hh, sc, dv);            // dV^T[d][key] += dO^T P
                mma_tile_cols<HD>(qt_, sub, l31, hh, dp, dk);            // dK^T[d][key] += Q^T dS
            }
        }
        if (t + 1 < nt) {
            tile_store<HD>(Qs + (buf ^ 1) * KT * LD, tid, rq);
            tile_store<HD>(Ds + (buf ^ 1) * KT * LD, tid, rd);
            if (tid < KT) { Ls[(buf ^ 1) * KT + tid] = rl; Es[(buf ^ 1) * KT + tid] = re; }
        }
        __syncthreads();
    }
    if (!wave_live || key >= B) return;
    float* drow = a.dqkv + ((size_t)s * B + key) * ld + h * HD;
    store_acc_T<HD>(drow + E, hh, dk, a.scale);
    store_acc_T<HD>(drow + 2 * E, hh, dv, 1.f);
}

// ------------------------------------------------------------------------------------------ dQ
template <int HD, bool DROP, bool SB>
__global__ __launch_bounds__(256, SB ? 3 : (HD <= 64 ? 2 : 1)) void attn_bwd_dq_kernel(AttnArgs a) {
    constexpr int LD = HD + 4, DT = (HD + 31) / 32, NB = SB ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + NB * KT * LD;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int B = a.B, H = a.H, E = H * HD;
    const size_t ld = (size_t)3 * E;
    const int ntile = rlt_cdiv_dev(B, QT);
    int pair, qt;
    map_block(blockIdx.x, a.S * H, ntile, pair, qt);
    const int s = pair / H, h = pair % H;
    const float* base = a.qkv + (size_t)s * B * ld + h * HD;
    const int q = qt * QT + wv * 32 + l31;
    const bool wave_live = qt * QT + wv * 32 < B;
    const int qc = min(q, B - 1);

    float qreg[HD / 2], doreg[HD / 2];
    row_half_load<HD>(base + (size_t)qc * ld, hh, a.scale * LOG2E, qreg);
    row_half_load<HD>(a.dout + ((size_t)s * B + qc) * E + h * HD, hh, 1.f, doreg);
    const float lse2 = a.lse[((size_t)s * H + h) * B + qc] * LOG2E;
    const float del = a.delta[((size_t)s * H + h) * B + qc];

    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

    TileRegs<HD> rk, rv;
    const int nt = rlt_cdiv_dev(B, KT);
    tile_load<HD>(base + E, ld, 0, B, tid, rk);
    tile_load<HD>(base + 2 * E, ld, 0, B, tid, rv);
    tile_store<HD>(Ks, tid, rk);
    tile_store<HD>(Vs, tid, rv);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = SB ? 0 : (t & 1);
        if (!SB && t + 1 < nt) {
            tile_load<HD>(base + E, ld, (t + 1) * KT, B, tid, rk);
            tile_load<HD>(base + 2 * E, ld, (t + 1) * KT, B, tid, rv);
        }
        if (wave_live) {
            const float* kt_ = Ks + buf * KT * LD;
            const float* vt_ = Vs + buf * KT * LD;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                // -lse and -delta of the lane's query are the accumulators' initial values (see the dK/dV kernel); a key beyond
                // B starts at -inf (P = 0)
                f32x16 sc, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sc[r] = -lse2; dp[r] = DROP ? 0.f : -del; }
                if ((t + 1) * KT > B) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * KT + sub * 32 + acc_row(r, hh) >= B) sc[r] = -INFINITY;
                }
                sc = mma_tile_rows<HD>(kt_, sub, l31, hh, qreg, sc);     // S^T[key][q] - lse[q]
                dp = mma_tile_rows<HD>(vt_, sub, l31, hh, doreg, dp);    // dP^T[key][q] (- delta[q])
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = rlt_exp2(sc[r]);
                    if (DROP) {
                        const int kidx = t * KT + sub * 32 + acc_row(r, hh);
                        const float dpr = rlt_keep(pair_seed(a.seed, pair), (uint32_t)q, (uint32_t)kidx, a.drop_thr) ? dp[r] / (1.f - a.drop_p) : 0.f;
                        dp[r] = p * (dpr - del);
                    } else {
                        dp[r] = p * dp[r];                                // dS^T
                    }
                }
                mma_tile_cols<HD>(kt_, sub, l31, hh, dp, dq);             // dQ^T[d][q] += K^T dS^T
            }
        }
        if (SB) {
            if (t + 1 < nt) {
                tile_load<HD>(base + E, ld, (t + 1) * KT, B, tid, rk);
                tile_load<HD>(base + 2 * E, ld, (t + 1) * KT, B, tid, rv);
            }
            __syncthreads();
        }
        if (t + 1 < nt) {
            tile_store<HD>(Ks + (SB ? 0 : (buf ^ 1)) * KT * LD, tid, rk);
            tile_store<HD>(Vs + (SB ? 0 : (buf ^ 1)) * KT * LD, tid, rv);
        }
        __syncthreads();
    }
    if (!wave_live || q >= B) return;
    store_acc_T<HD>(a.dqkv + ((size_t)s * B + q) * ld + h * HD, hh, dq, a.scale);
}

template <int HD> size_t fwd_smem(bool sb = false) { return (size_t)(sb ? 2 : 4) * KT * (HD + 4) * sizeof(float); }
// head dim 64 forward / dQ: the single-buffer, three-workgroups-per-CU form (RLT_ATTN_SB=0: the double-buffered one)
static bool attn_sb() {
    static const bool v = [] { const char* e = getenv("RLT_ATTN_SB"); return !e || atoi(e) != 0; }();
    return v;
}
template <int HD> size_t dkv_smem() { return (size_t)(4 * KT * (HD + 4) + 4 * KT) * sizeof(float); }

template <int HD, bool DROP>
int launch_fwd_t(const AttnArgs& a, hipStream_t st) {
    const int grid = a.S * a.H * rlt_cdiv(a.B, QT);
    if (HD == 64 && attn_sb()) {
        int rc = rlt_allow_lds(attn_fwd_kernel<HD, DROP, true>, fwd_smem<HD>(true));
        if (rc) return rc;
        hipLaunchKernelGGL((attn_fwd_kernel<HD, DROP, true>), dim3(grid), dim3(256), fwd_smem<HD>(true), st, a);
        return RLT_LAUNCH_RESULT();
    }
    int rc = rlt_allow_lds(attn_fwd_kernel<HD, DROP, false>, fwd_smem<HD>());
    if (rc) return rc;
    hipLaunchKernelGGL((attn_fwd_kernel<HD, DROP, false>), dim3(grid), dim3(256), fwd_smem<HD>(), st, a);
    return RLT_LAUNCH_RESULT();
}
template <int HD, bool DROP>
int launch_dkv_t(const AttnArgs& a, hipStream_t st) {
    const int grid = a.S * a.H * rlt_cdiv(a.B, QT);
    static const int occ = [] { const char* e = getenv("RLT_DKV_OCC"); return (e && atoi(e) == 1) ? 1 : 2; }();
    if (occ == 2 && HD <= 64) {          // head dim 128 needs the 512-register form
        int rc = rlt_allow_lds(attn_bwd_dkv_kernel<HD, 2, DROP>, dkv_smem<HD>());
        if (rc) return rc;
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, 2, DROP>), dim3(grid), dim3(256), dkv_smem<HD>(), st, a);
    } else {
        int rc = rlt_allow_lds(attn_bwd_dkv_kernel<HD, 1, DROP>, dkv_smem<HD>());
        if (rc) return rc;
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, 1, DROP>), dim3(grid), dim3(256), dkv_smem<HD>(), st, a);
    }
    return RLT_LAUNCH_RESULT();
}
template <int HD, bool DROP>
int launch_dq_t(const AttnArgs& a, hipStream_t st) {
    const int grid = a.S * a.H * rlt_cdiv(a.B, QT);
    if (HD == 64 && attn_sb() && getenv("RLT_ATTN_SB_DQ")) {       // (at the 168-register cap the dQ kernel spills 80 registers: off)
        int rc = rlt_allow_lds(attn_bwd_dq_kernel<HD, DROP, true>, fwd_smem<HD>(true));
        if (rc) return rc;
        hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, DROP, true>), dim3(grid), dim3(256), fwd_smem<HD>(true), st, a);
        return RLT_LAUNCH_RESULT();
    }
    int rc = rlt_allow_lds(attn_bwd_dq_kernel<HD, DROP, false>, fwd_smem<HD>());
    if (rc) return rc;
    hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, DROP, false>), dim3(grid), dim3(256), fwd_smem<HD>(), st, a);
    return RLT_LAUNCH_RESULT();
}
// dropout is a template parameter: hipcc if-converts a run-time `drop_p > 0` test and executes the hash regardless
template <int HD> int launch_fwd(const AttnArgs& a, hipStream_t st) {
    return a.drop_p > 0.f ? launch_fwd_t<HD, true>(a, st) : launch_fwd_t<HD, false>(a, st);
}
template <int HD> int launch_dkv(const AttnArgs& a, hipStream_t st) {
    return a.drop_p > 0.f ? launch_dkv_t<HD, true>(a, st) : launch_dkv_t<HD, false>(a, st);
}
template <int HD> int launch_dq(const AttnArgs& a, hipStream_t st) {
    return a.drop_p > 0.f ? launch_dq_t<HD, true>(a, st) : launch_dq_t<HD, false>(a, st);
}

AttnArgs bwd_args(const float* qkv, const float* dout, const float* lse, const float* delta,
                  int S, int B, int H, int HD, float drop_p, uint32_t seed, float* dqkv) {
    AttnArgs a{};
    a.qkv = qkv; a.dout = dout; a.lse = lse; a.delta = delta; a.dqkv = dqkv;
    a.S = S; a.B = B; a.H = H;
    a.scale = 1.0f / sqrtf((float)HD);
    a.drop_p = drop_p; a.drop_thr = rlt_drop_threshold(drop_p); a.seed = seed;
    return a;
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint32_t seed, size_t rows, int cols, float p, float* out) {
    const uint32_t thr = rlt_drop_threshold(p);
    const size_t n = rows * cols;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = rlt_keep(seed, (uint32_t)(i / cols), (uint32_t)(i % cols), thr) ? 1.f / (1.f - p) : 0.f;
}
__global__ __launch_bounds__(256) void attn_dropout_mask_kernel(uint32_t seed, int pair0, int npair, int B, float p, float* out) {
    const uint32_t thr = rlt_drop_threshold(p);
    const size_t bb = (size_t)B * B, n = (size_t)npair * bb;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int pair = pair0 + (int)(i / bb);
        const size_t rem = i % bb;
        out[i] = rlt_keep(pair_seed(seed, pair), (uint32_t)(rem / B), (uint32_t)(rem % B), thr) ? 1.f / (1.f - p) : 0.f;
    }
}

}  // namespace

// RLT_ATTN_MODE=fp32|bf16x3 forces the attention family into one mode whatever the call's precision (A/B runs): -1 = not set
static int attn_forced() {
    static const int forced = [] {
        const char* e = getenv("RLT_ATTN_MODE");
        if (!e) return -1;
        return (!strcmp(e, "bf16x3") || !strcmp(e, "1")) ? 1 : 0;
    }();
    return forced;
}
// 0 = exact fp32 MFMA kernels (parity mode) or, in bf16x6 mode, the six-product kernels (attn6_use), 1 = split-bf16 kernels of
// attention3.hip
static int attn_mode_any() {
    const int forced = attn_forced();
    return forced >= 0 ? forced : (rlt_precision() == RLT_PRECISION_BF16X3 ? 1 : 0);
}
// head dims 16 / 32 / 64 have split-bf16 kernels; 128 (PLECut: d_model 256, 2 heads, models/PLECut.py:56) runs on the
// exact-fp32 kernels in either mode (a correct, unhurried instantiation: it is not on a benchmarked configuration)
static int attn_mode(int HD) { return HD <= 64 ? attn_mode_any() : 0; }
static bool hd_ok(int HD) { return HD == 16 || HD == 32 || HD == 64 || HD == 128; }
// bf16x6 mode: the six-product kernels of attention6.hip where they exist (head dims 16 / 32 / 64); RLT_ATTN6=0 - or a forced
// RLT_ATTN_MODE - keeps them out (A/B runs)
static bool attn6_use(int HD, float drop_p) {
    static const bool on = [] { const char* e = getenv("RLT_ATTN6"); return !e || atoi(e) != 0; }();
    (void)drop_p;
    return on && attn_forced() < 0 && rlt_precision() == RLT_PRECISION_BF16X6 && HD <= 64;
}
// RLT_ATTN6_IMG=1: ... staged from pre-split tile images (a prepare pass per call, LDS-DMA in the kernels) instead of every
// workgroup splitting its tiles itself.  Off by default: measured at 4096 x 60 positions it takes the 176-352 split instructions per
// tile out of the kernels (dQ 7.16 -> 6.82 ms, dK+dV 10.63 -> 10.50, ping-pong forward 9,900 -> 9,140 cycles per tile) and gives the
// same time back in the two prepare passes (0.45 + 0.22 ms) - the kernels wait at their two barriers per tile for the staging
// LATENCY, not for its instructions (profiles/r03_notes.md).  It is the staging a double-buffered form of dQ / dK+dV needs.
static bool attn6_img(int HD) {
    static const bool on = [] { const char* e = getenv("RLT_ATTN6_IMG"); return e && atoi(e) != 0; }();
    return on && attn6_use(HD, 0.f);
}

// head dim 16 in the exact-fp32 mode: the 16x16x4-MFMA kernels of attention16.hip (RLT_ATTN16=0: the 32x32x2 kernels of this
// file, whose d-indexed products are half padding at 16)
static bool attn16_use() {
    static const bool on = [] { const char* e = getenv("RLT_ATTN16"); return !e || atoi(e) != 0; }();
    return on;
}

extern "C" {

int rlt_dropout_mask(uint32_t seed, size_t rows, int cols, float p, float* out, void* stream) {
    RLT_CHECK_ARG(out && rows > 0 && cols > 0 && p >= 0.f && p < 1.f);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(1024), dim3(256), 0, rlt_stream(stream), seed, rows, cols, p, out);
    return RLT_LAUNCH_RESULT();
}

int rlt_attention_dropout_mask(uint32_t seed, int S, int B, int H, float p, float* out, void* stream) {
    RLT_CHECK_ARG(out && S > 0 && B > 0 && H > 0 && p >= 0.f && p < 1.f);
    hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3(1024), dim3(256), 0, rlt_stream(stream), seed, 0, S * H, B, p, out);
    return RLT_LAUNCH_RESULT();
}

int rlt_attention_dropout_mask_range(uint32_t seed, int pair0, int npair, int B, float p, float* out, void* stream) {
    RLT_CHECK_ARG(out && pair0 >= 0 && npair > 0 && B > 0 && p >= 0.f && p < 1.f);
    hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3(1024), dim3(256), 0, rlt_stream(stream), seed, pair0, npair, B, p, out);
    return RLT_LAUNCH_RESULT();
}

// ---- workspace layout of the backward pass: [ delta (S,H,B) floats, padded to 1 KiB | dO tile records (bf16x3) ]
static size_t delta_bytes(int S, int B, int H) { return ((size_t)S * H * B * sizeof(float) + 1023) / 1024 * 1024; }
// bf16x6 at head dim 16, 512 lists and more: the pipelined backward kernels of attention6n.hip stage pre-split tile images of
// Q, K, V, dO and the rows' seeds from the workspace (behind delta): dO + seeds are written by _bwd_prepare, Q by _bwd_dkv,
// K / V by _bwd_dq - each part prepares what it reads, so the three entry points stay callable on their own.  RLT_A6N_IMG=0:
// the two-wavefront kernels (A/B runs).  Train-mode launches (drop_p > 0) do not use the images.
static bool a6n_images(int HD, int B) {
    static const bool on = [] {
        for (const char* v : {"RLT_A6N", "RLT_A6N_1", "RLT_A6N_IMG"}) { const char* e = getenv(v); if (e && atoi(e) == 0) return false; }
        return true;
    }();
    return on && HD == 16 && B >= 512 && attn6_use(HD, 0.f) && !attn6_img(HD);
}

// bf16x6 at head dim 16, 512 lists and more in whole 128-row tiles, no dropout: the pipelined forward kernel of attention6n.hip stages
// pre-split K / V tile images (written by two prepare passes of the call) and leaves a flag word per workgroup for the fix-up
// launch - both in the forward's `images` buffer (K images | V images | flags).
// RLT_A6N_F1=0: the two-wavefront kernel (A/B runs)
static bool a6n_fwd_images(int HD, int B) {
    static const bool on = [] { const char* e = getenv("RLT_A6N_F1"); return !e || atoi(e) != 0; }();
    return on && a6n_images(HD, B) && B % 128 == 0;
}
static size_t a6n_flags_bytes(int S, int B, int H) { return ((size_t)S * H * rlt_cdiv(B, 256) * sizeof(uint32_t) + 255) / 256 * 256; }

// bf16x6 at head dim 64, 512 lists and more in whole 64-row tiles, with or without dropout: the pipelined forward kernel of attention6h.hip, same
// scheme - K / V tile images of the call + a flag word per 256-query workgroup in the forward's `images` buffer, fix-up launch of
// attention6.hip's ping-pong kernel for the flagged workgroups.  RLT_A6H=0: attention6.hip's kernel alone (A/B runs)
static bool a6h_fwd_images(int HD, int B) {
    static const bool on = [] { const char* e = getenv("RLT_A6H"); return !e || atoi(e) != 0; }();
    return on && HD == 64 && B >= 512 && B % 64 == 0 && attn6_use(HD, 0.f) && !attn6_img(HD);
}

size_t rlt_list_attention_fwd_workspace(int S, int B, int H, int HD, float drop_p, int precision) {
    RLT_PREC_SCOPE_SZ(precision);
    if (S <= 0 || B <= 0 || H <= 0) return 0;
    if (!hd_ok(HD)) return 0;
    if (attn_mode(HD) == 1) return rlt_attn3_images_bytes(S, B, H, HD, 3);
    // (the pipelined forward kernels have no train-mode form: a call with dropout does not use - and need not be given - their images)
    if (!(drop_p > 0.f) && a6n_fwd_images(HD, B)) return rlt_attn6n_fwd_images_bytes(S, B, H) + a6n_flags_bytes(S, B, H);
    if (a6h_fwd_images(HD, B)) return rlt_attn6h_fwd_images_bytes(S, B, H) + a6n_flags_bytes(S, B, H);       // (train mode included)
    return attn6_img(HD) ? rlt_attn6_images_bytes(S, B, H, HD, 3) : 0;
}

// 1: the backward entry points read the forward's `images` (split-bf16 tile records; the RLT_ATTN6_IMG staging) - the caller keeps
// the buffer until the backward pass; 0: `images` is scratch of the forward call (the pipelined bf16x6 forward kernels) or empty
int rlt_list_attention_images_retained(int S, int B, int H, int HD, int precision) {
    RLT_PREC_SCOPE_SZ(precision);
    if (S <= 0 || B <= 0 || H <= 0 || !hd_ok(HD)) return 0;
    return attn_mode(HD) == 1 || attn6_img(HD) ? 1 : 0;
}

int rlt_list_attention_fwd(const float* qkv, int S, int B, int H, int HD, float drop_p, uint32_t seed,
                           float* out, float* lse, void* images, size_t images_bytes, int precision, void* stream) {
    RLT_PREC_SCOPE(precision);
    RLT_CHECK_ARG(qkv && out && lse && S > 0 && B > 0 && H > 0 && drop_p >= 0.f && drop_p < 1.f);
    RLT_CHECK_SHAPE(hd_ok(HD));
    if (!(rlt_aligned16(qkv) && rlt_aligned16(out))) return RLT_E_ALIGN;
    AttnArgs a{};
    a.qkv = qkv; a.o = out; a.lse_o = lse; a.S = S; a.B = B; a.H = H;
    a.scale = 1.0f / sqrtf((float)HD);
    a.drop_p = drop_p; a.drop_thr = rlt_drop_threshold(drop_p); a.seed = seed;
    hipStream_t st = rlt_stream(stream);
    if (attn_mode(HD) == 1 && images) {      // split-bf16: needs room for the Q/K/V tile records
        if (images_bytes < rlt_attn3_images_bytes(S, B, H, HD, 3)) return RLT_E_WORKSPACE;
        if (!rlt_aligned16(images)) return RLT_E_ALIGN;
        return rlt_attn3_run(0, a, HD, images, nullptr, st);
    }
    if (attn6_use(HD, drop_p)) {
        if (drop_p <= 0.f && a6n_fwd_images(HD, B) && images && rlt_aligned16(images) &&
            images_bytes >= rlt_attn6n_fwd_images_bytes(S, B, H) + a6n_flags_bytes(S, B, H)) {
            a.img = images;
            a.redo = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(images) + rlt_attn6n_fwd_images_bytes(S, B, H));
            int rc = rlt_attn6n_prepare_at(1, 0, a, st);          // K, V tile images of this call
            if (!rc) rc = rlt_attn6n_prepare_at(2, 1, a, st);
            if (rc) return rc;
            return rlt_attn6n_run(0, a, st);
        }
        if (a6h_fwd_images(HD, B) && images && rlt_aligned16(images) &&
            images_bytes >= rlt_attn6h_fwd_images_bytes(S, B, H) + a6n_flags_bytes(S, B, H)) {
            AttnArgs b = a;
            b.img = images;
            b.redo = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(images) + rlt_attn6h_fwd_images_bytes(S, B, H));
            int rc = rlt_attn6h_prepare2(1, 0, 2, 1, b, st);        // K, V tile images of this call
            if (!rc) rc = rlt_attn6h_run(0, b, st);
            if (rc) return rc;
            b.img = nullptr;                                        // the fix-up launch stages its tiles itself
            return rlt_attn6_run(0, b, HD, st);
        }
        if (attn6_img(HD) && images) {       // the backward kernels will stage from these images: they must all be written
            if (images_bytes < rlt_attn6_images_bytes(S, B, H, HD, 3)) return RLT_E_WORKSPACE;
            if (!rlt_aligned16(images)) return RLT_E_ALIGN;
            a.img = images;
            const int rc = rlt_attn6_run(3, a, HD, st);          // Q / K / V tile images (Q for the backward pass)
            if (rc) return rc;
        }
        return rlt_attn6_run(0, a, HD, st);
    }
    if (HD == 128) return launch_fwd<128>(a, st);
    if (HD == 64) return launch_fwd<64>(a, st);
    if (HD == 32) return launch_fwd<32>(a, st);
    if (attn16_use()) return rlt_attn16_run(0, a, st);
    return launch_fwd<16>(a, st);
}

// bytes behind delta: the dO tile records (split-bf16), the dO images (RLT_ATTN6_IMG) or the four image blocks + seeds of the pipelined
// head-dim-16 kernels (no dropout only: the train-mode kernels stage their tiles themselves)
static size_t bwd_ws_extra(int S, int B, int H, int HD, float drop_p) {
    return attn_mode(HD) == 1 ? rlt_attn3_images_bytes(S, B, H, HD, 1)
         : attn6_img(HD) ? rlt_attn6_images_bytes(S, B, H, HD, 1)
         : a6n_images(HD, B) && !(drop_p > 0.f) ? rlt_attn6n_images_bytes(S, B, H) : 0;
}
size_t rlt_list_attention_bwd_workspace(int S, int B, int H, int HD, float drop_p, int precision) {
    RLT_PREC_SCOPE_SZ(precision);
    if (S <= 0 || B <= 0 || H <= 0) return 0;
    if (!hd_ok(HD)) return 0;
    return delta_bytes(S, B, H) + bwd_ws_extra(S, B, H, HD, drop_p);
}

static int bwd_prepare(const float* out, const float* dout, const float* lse, int S, int B, int H, int HD,
                       const void* images, void* ws, size_t ws_bytes, float drop_p, void* stream) {
    RLT_CHECK_ARG(out && dout && lse && ws && S > 0 && B > 0 && H > 0 && drop_p >= 0.f && drop_p < 1.f);
    RLT_CHECK_SHAPE(hd_ok(HD));
    const bool split = attn_mode(HD) == 1 && images;
    const bool img6 = !split && attn6_img(HD) && images;      // bwd_part hands ws + delta to the kernels as the dO images
    const bool img6n = !split && a6n_images(HD, B) && drop_p <= 0.f;
    if (ws_bytes < delta_bytes(S, B, H) + (split ? rlt_attn3_images_bytes(S, B, H, HD, 1)
                                                 : img6 ? rlt_attn6_images_bytes(S, B, H, HD, 1)
                                                 : img6n ? rlt_attn6n_images_bytes(S, B, H) : 0)) return RLT_E_WORKSPACE;
    if (!rlt_aligned16(ws)) return RLT_E_ALIGN;
    hipStream_t st = rlt_stream(stream);
    const size_t T = (size_t)S * B;
    const int dgrid = (int)((T + 3) / 4 > 4096 ? 4096 : (T + 3) / 4);
    if (!split) {
        hipLaunchKernelGGL(attn_delta_kernel, dim3(dgrid), dim3(256), 0, st, out, dout, S, B, H, HD, (float*)ws);
        int rc = RLT_LAUNCH_RESULT();
        if (!rc && img6) {
            AttnArgs a{};                                         // bf16x6 mode: the dO tile images behind delta
            a.dout = dout; a.S = S; a.B = B; a.H = H;
            a.dimg = (uint8_t*)ws + delta_bytes(S, B, H);
            rc = rlt_attn6_run(4, a, HD, st);
        }
        if (!rc && img6n) {                                       // head dim 16: the dO tile images and the rows' seeds (-lse, -delta)
            AttnArgs a{};
            a.dout = dout; a.lse = lse; a.delta = (const float*)ws; a.S = S; a.B = B; a.H = H;
            a.img = (uint8_t*)ws + delta_bytes(S, B, H);
            rc = rlt_attn6n_prepare(3, a, st);
            if (!rc) rc = rlt_attn6n_prepare(4, a, st);
        }
        return rc;
    }
    AttnArgs a{};       // split-bf16 mode: the pass that writes the dO records computes delta from the tiles it has in registers
    a.dout = dout; a.lse = lse; a.delta = (const float*)ws; a.S = S; a.B = B; a.H = H;
    a.o = const_cast<float*>(out);
    a.drop_p = drop_p;
    return rlt_attn3_run(3, a, HD, nullptr, (uint8_t*)ws + delta_bytes(S, B, H), st);
}

int rlt_list_attention_bwd_prepare(const float* out, const float* dout, const float* lse, int S, int B, int H, int HD, float drop_p,
                                   const void* images, void* ws, size_t ws_bytes, int precision, void* stream) {
    RLT_PREC_SCOPE(precision);
    return bwd_prepare(out, dout, lse, S, B, H, HD, images, ws, ws_bytes, drop_p, stream);
}

static int bwd_part(int which, const float* qkv, const float* dout, const float* lse, const void* images, void* ws, size_t ws_bytes,
                    int S, int B, int H, int HD, float drop_p, uint32_t seed, float* dqkv, void* stream) {
    RLT_CHECK_ARG(qkv && dout && lse && ws && dqkv && S > 0 && B > 0 && H > 0 && drop_p >= 0.f && drop_p < 1.f);
    RLT_CHECK_SHAPE(hd_ok(HD));
    if (!(rlt_aligned16(qkv) && rlt_aligned16(dout) && rlt_aligned16(dqkv) && rlt_aligned16(ws))) return RLT_E_ALIGN;
    // the part reads delta - and, by mode, reads or WRITES tile images behind it: the same size rule as _bwd_prepare, evaluated in THIS
    // call's precision scope (a workspace sized under another mode, or by an older delta-only rule, is refused instead of overrun)
    if (ws_bytes < delta_bytes(S, B, H) + (images || attn_mode(HD) != 1 ? bwd_ws_extra(S, B, H, HD, drop_p) : 0)) return RLT_E_WORKSPACE;
    const AttnArgs a = bwd_args(qkv, dout, lse, (const float*)ws, S, B, H, HD, drop_p, seed, dqkv);
    hipStream_t st = rlt_stream(stream);
    if (attn_mode(HD) == 1 && images)
        return rlt_attn3_run(which, a, HD, const_cast<void*>(images), (uint8_t*)ws + delta_bytes(S, B, H), st);
    if (attn6_use(HD, drop_p)) {
        AttnArgs b = a;
        if (attn6_img(HD) && images) {                            // (forward wrote the Q / K / V images, bwd_prepare the dO images)
            b.img = images;
            b.dimg = (const uint8_t*)ws + delta_bytes(S, B, H);
        }
        if (a6n_images(HD, B) && drop_p == 0.f) {                 // the pipelined head-dim-16 kernels: this part's own images first
            b.img = (const uint8_t*)ws + delta_bytes(S, B, H);    // (the workspace is scratch: dO images + seeds are there already)
            int rc = 0;
            if (which == 1) rc = rlt_attn6n_prepare(0, b, st);                       // dK+dV stages Q (and dO)
            else { rc = rlt_attn6n_prepare(1, b, st); if (!rc) rc = rlt_attn6n_prepare(2, b, st); }     // dQ stages K and V
            return rc ? rc : rlt_attn6n_run(which, b, st);
        }
        return rlt_attn6_run(which, b, HD, st);
    }
    if (HD == 16 && attn16_use()) return rlt_attn16_run(which, a, st);
    if (which == 1) {
        if (HD == 128) return launch_dkv<128>(a, st);
        if (HD == 64) return launch_dkv<64>(a, st);
        if (HD == 32) return launch_dkv<32>(a, st);
        return launch_dkv<16>(a, st);
    }
    if (HD == 128) return launch_dq<128>(a, st);
    if (HD == 64) return launch_dq<64>(a, st);
    if (HD == 32) return launch_dq<32>(a, st);
    return launch_dq<16>(a, st);
}

int rlt_list_attention_bwd_dkv(const float* qkv, const float* dout, const float* lse, const void* images, void* ws, size_t ws_bytes,
                               int S, int B, int H, int HD, float drop_p, uint32_t seed, float* dqkv, int precision, void* stream) {
    RLT_PREC_SCOPE(precision);
    return bwd_part(1, qkv, dout, lse, images, ws, ws_bytes, S, B, H, HD, drop_p, seed, dqkv, stream);
}

int rlt_list_attention_bwd_dq(const float* qkv, const float* dout, const float* lse, const void* images, void* ws, size_t ws_bytes,
                              int S, int B, int H, int HD, float drop_p, uint32_t seed, float* dqkv, int precision, void* stream) {
    RLT_PREC_SCOPE(precision);
    return bwd_part(2, qkv, dout, lse, images, ws, ws_bytes, S, B, H, HD, drop_p, seed, dqkv, stream);
}

int rlt_list_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse,
                           int S, int B, int H, int HD, float drop_p, uint32_t seed, const void* images, float* dqkv,
                           void* ws, size_t ws_bytes, int precision, void* stream) {
    RLT_PREC_SCOPE(precision);
    RLT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
    int rc = bwd_prepare(out, dout, lse, S, B, H, HD, images, ws, ws_bytes, drop_p, stream);     // knows whether dO^T is needed
    if (!rc) rc = rlt_list_attention_bwd_dkv(qkv, dout, lse, images, ws, ws_bytes, S, B, H, HD, drop_p, seed, dqkv, RLT_PRECISION_DEFAULT, stream);
    if (!rc) rc = rlt_list_attention_bwd_dq(qkv, dout, lse, images, ws, ws_bytes, S, B, H, HD, drop_p, seed, dqkv, RLT_PRECISION_DEFAULT, stream);
    return rc;
}

}  // extern "C"

// List-axis attention at head dim 16 (Choopy / MtChoopy: d_model 128, 8 heads - reference models/Choopy.py:7,11-12,19-21;
// BASELINE configs[2]: 8192 lists x 300), fp32-FAITHFUL six-product arithmetic ("bf16x6", see attention6.hip) on the NARROW bf16
// MFMA, v_mfma_f32_16x16x32_bf16.  Same interface, same algorithm (flash-style, scores produced with the wavefront's own rows on
// the lanes, deterministic, no atomics) and the same fp32 softmax arithmetic as attention6.hip; what changes is how the products
// map onto the matrix pipe, because at head dim 16 the 32x32x16 kernels pay for padding and for the vector ALU:
//
//  1. No padded axis.  A d-indexed OUTPUT (P.V, dV, dK, dQ) on the 32-row MFMA is half padding at d = 16; the 16-row MFMA has
//     none (6 instead of 12 MFMA-equivalents per 32 x 32 block and product).
//  2. Plane PAIRS in the contraction.  A d-CONTRACTED product (S = Q K^T, dP = dO V^T) has K = 16, the 16x16x32 MFMA K = 32: two
//     of the six plane products share one MFMA - A = [a_m | a_l], B = [b_m ; b_h] gives a_m b_m + a_l b_h, [a_h | a_m] x [b_l ; b_h]
//     gives a_h b_l + a_m b_h, [a_h | a_h] x [b_m ; b_h] gives a_h b_m + a_h b_h: three full MFMAs per 16 x 16 tile, smallest terms
//     first as before.  The pairing costs nothing at staging time: lane group g = lane >> 4 simply reads plane (g < 2 ? X : Y).
//  3. The three-way split of the FRESH operand (P, dS: one per score) on the matrix pipe.  x = h + m + l needs, per value, a
//     conversion, an unpack and a subtraction per level in vector code (5.5 instructions: what bounds the 32x32x16 kernels at
//     this head dim).  But the packed h of two 16 x 16 score tiles IS already the B operand of the next product, so the
//     residual comes from an MFMA: R1 = X - SEL h, with SEL a constant selection matrix of -1.0 entries (C = X, A = SEL, B = h).
//     fp32 accumulation of one exact bf16 product into X is exact (the difference is representable), so h, m, l are BIT-IDENTICAL
//     to the vector split (tools/micro/mfma_resid.hip: 0 mismatches over every exponent, exp2 weights, denormals, worst-split).
//     1.5 vector instructions per value (three v_cvt_pk_bf16_f32 per pair) + four 16-cycle MFMAs per 512 values.
//
// Tile images: [plane h | m | l][64 rows][16 d] bf16, 32-byte rows, unpadded (row reads as ds_read_b128 and transposed reads as
// ds_read_b64_tr_b16 are both conflict-free on it), split once per workgroup at staging time, double-buffered (one barrier per
// tile).  A wavefront owns NB blocks of 16 of its own rows (queries in forward / dQ, keys in dK+dV): fragments of a tile are read
// once per wavefront and used NB times.  Two 16 x 16 score tiles (keys 16 kb + 4 g + r of lane group g, register r) form one B
// operand of the next product: k slot (g, j) <-> tile row 16 (j >> 2) + 4 g + (j & 3), and the transposed reads deliver the A
// operand in exactly that order (block rows 4 g .. 4 g + 3 and + 16).
#include "attention_common.h"
#include "split6.h"
#include <stdlib.h>

#ifndef RLT_A6N_NB
#define RLT_A6N_NB 4        // 16-row blocks owned by a wavefront (forward / dQ)
#endif
#ifndef RLT_A6N_NBK
#define RLT_A6N_NBK 2       // ... in dK+dV (twice the stationary state per block)
#endif
#ifndef RLT_A6N_OCC
#define RLT_A6N_OCC 2       // wavefronts per SIMD the register budget is declared for
#endif

// SEED (template parameter of the two-wavefront kernels; the pipelined kernels always): the row / lane constants (-m_run, -lse,
// -delta) as the INITIAL accumulators of the score / dP products (as attention16.hip) - no subtraction per score.  It rounds every
// partial sum at the magnitude of the constant: invisible next to the accumulation error of thousands of lists, 2-3x the error
// of the unseeded form on a few dozen lists (profiles/r05_notes.md) - so launches of fewer than 512 lists run unseeded.  The same
// launches also keep ONE accumulator per output (RLT_A6N_2ACC below): with one or two key blocks the second accumulator only adds
// a rounding at full magnitude (whole-model gradient error 3.5x at 16 lists), what it removes needs thousands of lists to build up.
#ifndef RLT_A6N_2ACC
#define RLT_A6N_2ACC 1      // 1: the five small plane products of the list-contracted outputs (O, dQ, dK, dV) accumulate in their OWN accumulator, added to the h h' accumulator once at the end
#endif
#ifndef RLT_A6N_ABL
#define RLT_A6N_ABL 0       // timing-only ablations of the dQ kernel (wrong results): 1 no element-wise work, 2 no split MFMAs, 4 no score / dP products, 8 no dQ product
#endif
#ifndef RLT_A6N_SGB
#define RLT_A6N_SGB 0       // 1: sched_group_barrier pattern (1 MFMA, 2 vector) over the block body of dQ / dK+dV
#endif
// interleave hint for the scheduling region that ends here: NM x (one MFMA, NV vector instructions)
template <int NM, int NV>
__device__ __forceinline__ void interleave_hint() {
#if RLT_A6N_SGB
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
    }
#endif
}

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));

constexpr int PLN = KT * 16;          // bf16 elements per plane of a tile image
constexpr int IMGN = 3 * PLN;         // ... per image (h | m | l): 6 KiB

__device__ __forceinline__ uint32_t pk2n(float a, float b) {
    // the cast form: hipcc emits v_cvt_pk_bf16_f32 and inserts the wait states an MFMA needs behind a vector write of its operand
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    const v2 t = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ float lo16(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi16(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
#ifndef RLT_A6N_PAD
#define RLT_A6N_PAD -1      // >= 0: `s_nop PAD` behind every MFMA (the issuing wavefront steps back from the SIMD's issue port: attention6.hip, RLT_A6_PP_PAD)
#endif
__device__ __forceinline__ f32x4 mm(bf16x8 a, bf16x8 b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    if (RLT_A6N_PAD >= 0) asm volatile("s_nop %1" : "+v"(c) : "n"(RLT_A6N_PAD >= 0 ? RLT_A6N_PAD : 0));
    return c;
}
__device__ __forceinline__ bf16x8 frag4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return __builtin_bit_cast(bf16x8, make_uint4(a, b, c, d)); }

// exact three-way split in vector code (staging and the stationary fragments only): four values -> three packed pairs
__device__ __forceinline__ void split4v(float a, float b, float c, float d, uint2& h, uint2& m, uint2& l) {
    h.x = pk2n(a, b); h.y = pk2n(c, d);
    const float ra = a - lo16(h.x), rb = b - hi16(h.x), rc = c - lo16(h.y), rd = d - hi16(h.y);
    m.x = pk2n(ra, rb); m.y = pk2n(rc, rd);
    l.x = pk2n(ra - lo16(m.x), rb - hi16(m.x)); l.y = pk2n(rc - lo16(m.y), rd - hi16(m.y));
}

struct Planes { bf16x8 h, m, l; };
// ... of two 16 x 16 accumulator tiles on the matrix pipe (see the header): sel0 / sel1 select the tile
__device__ __forceinline__ bf16x8 pack8(const f32x4& t0, const f32x4& t1) {
    return frag4(pk2n(t0[0], t0[1]), pk2n(t0[2], t0[3]), pk2n(t1[0], t1[1]), pk2n(t1[2], t1[3]));
}
__device__ __forceinline__ Planes split_mx(f32x4 t0, f32x4 t1, bf16x8 sel0, bf16x8 sel1) {
    Planes p;
    p.h = pack8(t0, t1);
    if (RLT_A6N_ABL & 2) { p.m = pack8(t1, t0); p.l = pack8(t0, t0); return p; }
    t0 = mm(sel0, p.h, t0); t1 = mm(sel1, p.h, t1);
    p.m = pack8(t0, t1);
    t0 = mm(sel0, p.m, t0); t1 = mm(sel1, p.m, t1);
    p.l = pack8(t0, t1);
    return p;
}
// acc += A^T-image planes x fresh planes: the six products, smallest first
// ... with the small products in their own accumulator (RLT_A6N_2ACC)
__device__ __forceinline__ void mm6_2(const Planes& a, const Planes& b, f32x4& big, f32x4& small) {
    small = mm(a.m, b.m, small);
    small = mm(a.l, b.h, small);
    small = mm(a.h, b.l, small);
    small = mm(a.m, b.h, small);
    small = mm(a.h, b.m, small);
    big = mm(a.h, b.h, big);
}
__device__ __forceinline__ f32x4 mm6(const Planes& a, const Planes& b, f32x4 c) {
    c = mm(a.m, b.m, c);
    c = mm(a.l, b.h, c);
    c = mm(a.h, b.l, c);
    c = mm(a.m, b.h, c);
    c = mm(a.h, b.m, c);
    return mm(a.h, b.h, c);
}

// per-lane constants
struct LaneN {
    int l15, g;
    int offA, offB, offC;     // element offsets of the three row-fragment reads of a 16-row block: [m|l], [h|m], [h|h]
    int offT;                 // ... of the transposed reads of a 32-row block (plane 0, first half)
    bf16x8 sel0, sel1;        // selection operands of the matrix-pipe split
};
__device__ __forceinline__ LaneN lane_consts(int lane) {
    LaneN c;
    c.l15 = lane & 15; c.g = lane >> 4;
    const int ro = c.l15 * 16 + 8 * (c.g & 1);
    c.offA = (c.g < 2 ? 1 : 2) * PLN + ro;
    c.offB = (c.g < 2 ? 0 : 1) * PLN + ro;
    c.offC = ro;
    c.offT = (4 * c.g + (c.l15 >> 2)) * 16 + 4 * (c.l15 & 3);
    uint32_t s0[4] = {0u, 0u, 0u, 0u};
    if ((c.l15 >> 2) == c.g) s0[(c.l15 & 3) >> 1] = (c.l15 & 1) ? 0xBF800000u : 0x0000BF80u;     // -1.0 at element l15 & 3
    c.sel0 = frag4(s0[0], s0[1], 0u, 0u);
    c.sel1 = frag4(0u, 0u, s0[0], s0[1]);
    return c;
}

// the wavefront's own row as the stationary B fragments of the row products: bmh = [x_m ; x_h], blh = [x_l ; x_h]
__device__ __forceinline__ void own_frags(const float* __restrict__ rowp, const LaneN& c, float mul, bf16x8& bmh, bf16x8& blh) {
    const float4 v0 = *reinterpret_cast<const float4*>(rowp + 8 * (c.g & 1));
    const float4 v1 = *reinterpret_cast<const float4*>(rowp + 8 * (c.g & 1) + 4);
    uint2 h0, m0, l0, h1, m1, l1;
    split4v(v0.x * mul, v0.y * mul, v0.z * mul, v0.w * mul, h0, m0, l0);
    split4v(v1.x * mul, v1.y * mul, v1.z * mul, v1.w * mul, h1, m1, l1);
    const bool lo = c.g < 2;
    bmh = frag4(lo ? m0.x : h0.x, lo ? m0.y : h0.y, lo ? m1.x : h1.x, lo ? m1.y : h1.y);
    blh = frag4(lo ? l0.x : h0.x, lo ? l0.y : h0.y, lo ? l1.x : h1.x, lo ? l1.y : h1.y);
}

// row fragments of the 16-row block `blk` of an image
struct RowFr { bf16x8 a, b, c; };
__device__ __forceinline__ RowFr row_fr(const uint16_t* __restrict__ img, int blk, const LaneN& c) {
    RowFr f;
    f.a = *reinterpret_cast<const bf16x8*>(img + c.offA + blk * 256);
    f.b = *reinterpret_cast<const bf16x8*>(img + c.offB + blk * 256);
    f.c = *reinterpret_cast<const bf16x8*>(img + c.offC + blk * 256);
    return f;
}
// tile[row = block row][col = lane] = sum_d image[row][d] * own[col][d]
__device__ __forceinline__ f32x4 row_prod(const RowFr& f, bf16x8 bmh, bf16x8 blh, f32x4 c) {
    c = mm(f.a, bmh, c);
    c = mm(f.b, blh, c);
    return mm(f.c, bmh, c);
}
// transposed fragments of the 32-row block `b32` of an image (A[d][k slot])
__device__ __forceinline__ bf16x8 tr_pair(const uint16_t* p) {
    typedef short v8s __attribute__((ext_vector_type(8)));
    const v4s x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p));
    const v4s y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p + 256));
    const v8s v = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ Planes tr_fr(const uint16_t* __restrict__ img, int b32, const LaneN& c) {
    Planes f;
    f.h = tr_pair(img + c.offT + b32 * 512);
    f.m = tr_pair(img + PLN + c.offT + b32 * 512);
    f.l = tr_pair(img + 2 * PLN + c.offT + b32 * 512);
    return f;
}

// staging: one float4 of a [64][16] fp32 tile per thread (row tid >> 2, d 4 (tid & 3)); rows beyond nrows read as zero
__device__ __forceinline__ float4 stage_ld(const float* __restrict__ base, size_t ld, int row0, int nrows, int tid) {
    const int row = row0 + (tid >> 2);
    const float4 t = *reinterpret_cast<const float4*>(base + (size_t)min(row, nrows - 1) * ld + 4 * (tid & 3));
    const bool ok = row < nrows;
    return make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
}
__device__ __forceinline__ void stage_st(uint16_t* __restrict__ img, int tid, const float4& v) {
    uint2 h, m, l;
    split4v(v.x, v.y, v.z, v.w, h, m, l);
    const int off = (tid >> 2) * 16 + 4 * (tid & 3);
    *reinterpret_cast<uint2*>(img + off) = h;
    *reinterpret_cast<uint2*>(img + PLN + off) = m;
    *reinterpret_cast<uint2*>(img + 2 * PLN + off) = l;
}
// max / sum over the four lanes (l & 15) + 16 {0, 1, 2, 3} that share a column
__device__ __forceinline__ float col_max4(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float col_sum4(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ------------------------------------------------------------------------------------------ forward
template <int NB, bool DROP, bool SEED>
__global__ __launch_bounds__(256, RLT_A6N_OCC) void attn6n_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* img0 = reinterpret_cast<uint16_t*>(smem);           // [2 buffers][K image | V image]
    uint32_t* htab = reinterpret_cast<uint32_t*>(img0 + 4 * IMGN);   // [2][KT] column hashes of the tile's keys (DROP)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const LaneN c = lane_consts(lane);
    const int B = a.B, H = a.H, E = H * 16;
    const size_t ld = (size_t)3 * E;
    constexpr int WROWS = 16 * NB, GROWS = 4 * WROWS;
    int pair, qt;
    map_block(blockIdx.x, a.S * H, rlt_cdiv_dev(B, GROWS), pair, qt);
    const int s = pair / H, h = pair % H;
    const float* base = a.qkv + (size_t)s * B * ld + h * 16;
    const int row0 = qt * GROWS + wv * WROWS;
    const bool wave_live = row0 < B;
    const uint32_t ps = DROP ? pair_seed(a.seed, pair) : 0u;
    // second launch behind the pipelined forward kernel: only the workgroups it flagged (same grid, same block -> rows map)
    if (a.redo && a.redo[blockIdx.x] == 0u) return;

    bf16x8 qmh[NB], qlh[NB];
    uint32_t hq[NB];
    f32x4 o[NB], o2[NB], seed[NB];                      // seed: -m_run in all four registers, the initial value of the score accumulators
    float m_run[NB], l_run[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int q = row0 + 16 * n + c.l15;
        own_frags(base + (size_t)min(q, B - 1) * ld, c, a.scale * LOG2E, qmh[n], qlh[n]);
        hq[n] = DROP ? rlt_row_hash(ps, (uint32_t)q) : 0u;
        o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        o2[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        seed[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        m_run[n] = 0.f; l_run[n] = 0.f;          // m_run: the reference of the weights, set by the first block
    }
    const float inv_keep = DROP ? 1.f / (1.f - a.drop_p) : 1.f;

    const int nt = rlt_cdiv_dev(B, KT);
    float4 rk = stage_ld(base + E, ld, 0, B, tid), rv = stage_ld(base + 2 * E, ld, 0, B, tid);
    stage_st(img0, tid, rk);
    stage_st(img0 + IMGN, tid, rv);
    if (DROP && tid < KT) htab[tid] = rlt_col_hash(ps, (uint32_t)tid);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const uint16_t* Ki = img0 + buf * 2 * IMGN;
        const uint16_t* Vi = Ki + IMGN;
        if (t + 1 < nt) {
            rk = stage_ld(base + E, ld, (t + 1) * KT, B, tid);
            rv = stage_ld(base + 2 * E, ld, (t + 1) * KT, B, tid);
        }
        if (wave_live) {
            const bool tail = (t + 1) * KT > B;
#pragma unroll
            for (int b32 = 0; b32 < KT / 32; ++b32) {
                const RowFr k0 = row_fr(Ki, 2 * b32, c), k1 = row_fr(Ki, 2 * b32 + 1, c);
                const Planes vt = tr_fr(Vi, b32, c);
                f32x4 sc[NB][2];                                   // S^T[key][q], log2 domain
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    sc[n][0] = row_prod(k0, qmh[n], qlh[n], seed[n]);          // (seeded: scores relative to the reference)
                    sc[n][1] = row_prod(k1, qmh[n], qlh[n], seed[n]);
                }
                if (tail) {                                        // last tile only: keys beyond B
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (t * KT + b32 * 32 + kb * 16 + 4 * c.g + r >= B) {
#pragma unroll
                                for (int n = 0; n < NB; ++n) sc[n][kb][r] = -INFINITY;
                            }
                }
                // lazy rescaling as in attention16.hip: the weights are exp2(score - m_run) with the reference m_run moving only
                // when a weight would leave the comfortable fp32 range - the common block has no max search and no cross-lane step
                f32x4 pe[NB][2];
                float psum[NB];
                const bool first = t == 0 && b32 == 0;
                bool redo = first;
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    psum[n] = 0.f;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            pe[n][kb][r] = rlt_exp2(SEED ? sc[n][kb][r] : sc[n][kb][r] - m_run[n]);
                            psum[n] += pe[n][kb][r];
                        }
                    redo |= !(psum[n] <= 4096.f);
                }
                if (__any(redo)) {                                 // wave-uniform; rare after the first block
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        float tmax = -INFINITY;
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, sc[n][kb][r]);
                        tmax = col_max4(tmax);
                        // (seeded: scores are relative to the old reference, d = the move of the reference)
                        const float mo = SEED ? 0.f : m_run[n];
                        const float m_new = first ? tmax : fmaxf(tmax, mo);
                        const float alpha = first ? 0.f : rlt_exp2(mo - m_new);
                        psum[n] = 0.f;
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                pe[n][kb][r] = rlt_exp2(sc[n][kb][r] - m_new);
                                psum[n] += pe[n][kb][r];
                            }
                        l_run[n] *= alpha;
                        m_run[n] = SEED ? m_run[n] + m_new : m_new;
                        if (SEED) seed[n] = f32x4{-m_run[n], -m_run[n], -m_run[n], -m_run[n]};
                        o[n] *= alpha;
                        if (RLT_A6N_2ACC && SEED) o2[n] *= alpha;
                    }
                }
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    l_run[n] += psum[n];
                    if (DROP) {                                    // on the normalised probabilities: the normaliser keeps every key
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb) {
                            const uint4 hc = *reinterpret_cast<const uint4*>(htab + buf * KT + b32 * 32 + kb * 16 + 4 * c.g);
                            pe[n][kb][0] = rlt_keep_rc(hq[n], hc.x, a.drop_thr) ? pe[n][kb][0] * inv_keep : 0.f;
                            pe[n][kb][1] = rlt_keep_rc(hq[n], hc.y, a.drop_thr) ? pe[n][kb][1] * inv_keep : 0.f;
                            pe[n][kb][2] = rlt_keep_rc(hq[n], hc.z, a.drop_thr) ? pe[n][kb][2] * inv_keep : 0.f;
                            pe[n][kb][3] = rlt_keep_rc(hq[n], hc.w, a.drop_thr) ? pe[n][kb][3] * inv_keep : 0.f;
                        }
                    }
                    const Planes pp = split_mx(pe[n][0], pe[n][1], c.sel0, c.sel1);
                    if (RLT_A6N_2ACC && SEED) mm6_2(vt, pp, o[n], o2[n]);
                    else o[n] = mm6(vt, pp, o[n]);                  // O^T[d][q] += V^T P^T
                }
            }
        }
        if (t + 1 < nt) {
            stage_st(img0 + (buf ^ 1) * 2 * IMGN, tid, rk);
            stage_st(img0 + (buf ^ 1) * 2 * IMGN + IMGN, tid, rv);
            if (DROP && tid < KT) htab[(buf ^ 1) * KT + tid] = rlt_col_hash(ps, (uint32_t)((t + 1) * KT + tid));
        }
        __syncthreads();
    }
    if (!wave_live) return;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const float l_tot = col_sum4(l_run[n]);
        const int q = row0 + 16 * n + c.l15;
        if (RLT_A6N_2ACC && SEED) o[n] += o2[n];
        if (q < B) {
            const float inv = 1.f / l_tot;
            *reinterpret_cast<float4*>(a.o + ((size_t)s * B + q) * E + h * 16 + 4 * c.g) =
                make_float4(o[n][0] * inv, o[n][1] * inv, o[n][2] * inv, o[n][3] * inv);
            if (c.g == 0) a.lse_o[((size_t)s * H + h) * B + q] = (m_run[n] + log2f(l_tot)) * LN2;
        }
    }
}

// ------------------------------------------------------------------------------------------ dQ
// Keys beyond B: their K and V rows are staged as zeros, so whatever dS they get multiplies a zero column of K^T.
template <int NB, bool DROP, bool SEED>
__global__ __launch_bounds__(256, RLT_A6N_OCC) void attn6n_bwd_dq_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* img0 = reinterpret_cast<uint16_t*>(smem);           // [2 buffers][K image | V image]
    uint32_t* htab = reinterpret_cast<uint32_t*>(img0 + 4 * IMGN);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const LaneN c = lane_consts(lane);
    const int B = a.B, H = a.H, E = H * 16;
    const size_t ld = (size_t)3 * E;
    constexpr int WROWS = 16 * NB, GROWS = 4 * WROWS;
    int pair, qt;
    map_block(blockIdx.x, a.S * H, rlt_cdiv_dev(B, GROWS), pair, qt);
    const int s = pair / H, h = pair % H;
    const float* base = a.qkv + (size_t)s * B * ld + h * 16;
    const int row0 = qt * GROWS + wv * WROWS;
    const bool wave_live = row0 < B;
    const uint32_t ps = DROP ? pair_seed(a.seed, pair) : 0u;

    bf16x8 qmh[NB], qlh[NB], dmh[NB], dlh[NB];
    float lse2[NB], del[NB];
    uint32_t hq[NB];
    f32x4 dq[NB], dq2[NB], seed_s[NB], seed_d[NB];       // seeds: -lse / -delta of the lane's query in all four registers
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int q = row0 + 16 * n + c.l15, qc = min(q, B - 1);
        own_frags(base + (size_t)qc * ld, c, a.scale * LOG2E, qmh[n], qlh[n]);
        own_frags(a.dout + ((size_t)s * B + qc) * E + h * 16, c, 1.f, dmh[n], dlh[n]);
        lse2[n] = a.lse[((size_t)s * H + h) * B + qc] * LOG2E;
        del[n] = a.delta[((size_t)s * H + h) * B + qc];
        const float s0 = SEED ? -lse2[n] : 0.f, d0 = SEED && !DROP ? -del[n] : 0.f;
        seed_s[n] = f32x4{s0, s0, s0, s0};
        seed_d[n] = f32x4{d0, d0, d0, d0};
        hq[n] = DROP ? rlt_row_hash(ps, (uint32_t)q) : 0u;
        dq[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        dq2[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float inv_keep = DROP ? 1.f / (1.f - a.drop_p) : 1.f;

    const int nt = rlt_cdiv_dev(B, KT);
    float4 rk = stage_ld(base + E, ld, 0, B, tid), rv = stage_ld(base + 2 * E, ld, 0, B, tid);
    stage_st(img0, tid, rk);
    stage_st(img0 + IMGN, tid, rv);
    if (DROP && tid < KT) htab[tid] = rlt_col_hash(ps, (uint32_t)tid);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const uint16_t* Ki = img0 + buf * 2 * IMGN;
        const uint16_t* Vi = Ki + IMGN;
        if (t + 1 < nt) {
            rk = stage_ld(base + E, ld, (t + 1) * KT, B, tid);
            rv = stage_ld(base + 2 * E, ld, (t + 1) * KT, B, tid);
        }
        if (wave_live) {
#pragma unroll
            for (int b32 = 0; b32 < KT / 32; ++b32) {
                const RowFr k0 = row_fr(Ki, 2 * b32, c), k1 = row_fr(Ki, 2 * b32 + 1, c);
                const RowFr v0 = row_fr(Vi, 2 * b32, c), v1 = row_fr(Vi, 2 * b32 + 1, c);
                const Planes kt = tr_fr(Ki, b32, c);
                uint4 hc[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
                if (DROP) {
                    hc[0] = *reinterpret_cast<const uint4*>(htab + buf * KT + b32 * 32 + 4 * c.g);
                    hc[1] = *reinterpret_cast<const uint4*>(htab + buf * KT + b32 * 32 + 16 + 4 * c.g);
                }
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    f32x4 sc[2], dp[2];
                    if (RLT_A6N_ABL & 4) {
                        sc[0] = f32x4{lse2[n], del[n], lse2[n], del[n]}; sc[1] = sc[0] * 0.5f; dp[0] = sc[0] + 1.f; dp[1] = sc[1] - 1.f;
                        asm volatile("" : "+v"(sc[0]), "+v"(sc[1]), "+v"(dp[0]), "+v"(dp[1]));
                    } else {
                    sc[0] = row_prod(k0, qmh[n], qlh[n], seed_s[n]);                     // S^T[key][q] (- lse[q])
                    sc[1] = row_prod(k1, qmh[n], qlh[n], seed_s[n]);
                    dp[0] = row_prod(v0, dmh[n], dlh[n], seed_d[n]);                     // dP^T[key][q] (- delta[q])
                    dp[1] = row_prod(v1, dmh[n], dlh[n], seed_d[n]);
                    }
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        const uint32_t hcs[4] = {hc[kb].x, hc[kb].y, hc[kb].z, hc[kb].w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (RLT_A6N_ABL & 1) continue;
                            const float p = rlt_exp2(SEED ? sc[kb][r] : sc[kb][r] - lse2[n]);
                            float dpr = dp[kb][r];
                            if (DROP) dpr = rlt_keep_rc(hq[n], hcs[r], a.drop_thr) ? dpr * inv_keep : 0.f;
                            dp[kb][r] = SEED && !DROP ? p * dpr : p * (dpr - del[n]);     // dS^T
                        }
                    }
                    const Planes ds = split_mx(dp[0], dp[1], c.sel0, c.sel1);
                    if (RLT_A6N_ABL & 8) { asm volatile("" :: "v"(ds.h), "v"(ds.m), "v"(ds.l)); continue; }
                    if (RLT_A6N_2ACC && SEED) mm6_2(kt, ds, dq[n], dq2[n]);
                    else dq[n] = mm6(kt, ds, dq[n]);                                 // dQ^T[d][q] += K^T dS^T
                }
                interleave_hint<22 * NB, 2>();
            }
        }
        if (t + 1 < nt) {
            stage_st(img0 + (buf ^ 1) * 2 * IMGN, tid, rk);
            stage_st(img0 + (buf ^ 1) * 2 * IMGN + IMGN, tid, rv);
            if (DROP && tid < KT) htab[(buf ^ 1) * KT + tid] = rlt_col_hash(ps, (uint32_t)((t + 1) * KT + tid));
        }
        __syncthreads();
    }
    if (!wave_live) return;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int q = row0 + 16 * n + c.l15;
        if (RLT_A6N_2ACC && SEED) dq[n] += dq2[n];
        if (q < B)
            *reinterpret_cast<float4*>(a.dqkv + ((size_t)s * B + q) * ld + h * 16 + 4 * c.g) =
                make_float4(dq[n][0] * a.scale, dq[n][1] * a.scale, dq[n][2] * a.scale, dq[n][3] * a.scale);
    }
}

// ------------------------------------------------------------------------------------------ dK, dV
// Queries beyond B: their Q / dO rows are staged as zeros and their lse entry is +inf, so P = dS = 0.
template <int NB, bool DROP, bool SEED>
__global__ __launch_bounds__(256, RLT_A6N_OCC) void attn6n_bwd_dkv_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* img0 = reinterpret_cast<uint16_t*>(smem);           // [2 buffers][Q image | dO image]
    float* tab0 = reinterpret_cast<float*>(img0 + 4 * IMGN);      // [2 buffers][lse * log2e | delta | row hashes][KT]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const LaneN c = lane_consts(lane);
    const int B = a.B, H = a.H, E = H * 16;
    const size_t ld = (size_t)3 * E;
    constexpr int WROWS = 16 * NB, GROWS = 4 * WROWS;
    int pair, ktile;
    map_block(blockIdx.x, a.S * H, rlt_cdiv_dev(B, GROWS), pair, ktile);
    const int s = pair / H, h = pair % H;
    const float* base = a.qkv + (size_t)s * B * ld + h * 16;
    const float* dobase = a.dout + (size_t)s * B * E + h * 16;
    const float* lsebase = a.lse + ((size_t)s * H + h) * B;
    const float* delbase = a.delta + ((size_t)s * H + h) * B;
    const int row0 = ktile * GROWS + wv * WROWS;
    const bool wave_live = row0 < B;
    const uint32_t ps = DROP ? pair_seed(a.seed, pair) : 0u;

    bf16x8 kmh[NB], klh[NB], vmh[NB], vlh[NB];
    uint32_t hk[NB];
    f32x4 dk[NB], dv[NB], dk2[NB], dv2[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        dk2[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv2[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int key = row0 + 16 * n + c.l15, kc = min(key, B - 1);
        own_frags(base + (size_t)kc * ld + E, c, a.scale * LOG2E, kmh[n], klh[n]);
        own_frags(base + (size_t)kc * ld + 2 * E, c, 1.f, vmh[n], vlh[n]);
        hk[n] = DROP ? rlt_col_hash(ps, (uint32_t)key) : 0u;
        dk[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float inv_keep = DROP ? 1.f / (1.f - a.drop_p) : 1.f;

    const int nt = rlt_cdiv_dev(B, KT);
    float4 rq, rd;
    float rl = 0.f, re = 0.f;
    auto load_tile = [&](int r0) {
        rq = stage_ld(base, ld, r0, B, tid);
        rd = stage_ld(dobase, (size_t)E, r0, B, tid);
        if (tid < KT) {
            const int qi = r0 + tid, qc = min(qi, B - 1);
            const float l = lsebase[qc], e = delbase[qc];
            rl = qi < B ? l * LOG2E : INFINITY;
            re = qi < B ? e : 0.f;
            if (SEED) { rl = -rl; re = -re; }       // negated: initial values of the score / dP accumulators
        }
    };
    auto store_tile = [&](int b, int r0) {
        stage_st(img0 + b * 2 * IMGN, tid, rq);
        stage_st(img0 + b * 2 * IMGN + IMGN, tid, rd);
        if (tid < KT) {
            float* tb = tab0 + b * 3 * KT;
            tb[tid] = rl; tb[KT + tid] = re;
            if (DROP) reinterpret_cast<uint32_t*>(tb)[2 * KT + tid] = rlt_row_hash(ps, (uint32_t)(r0 + tid));
        }
    };
    load_tile(0);
    store_tile(0, 0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const uint16_t* Qi = img0 + buf * 2 * IMGN;
        const uint16_t* Di = Qi + IMGN;
        const float* tb = tab0 + buf * 3 * KT;
        if (t + 1 < nt) load_tile((t + 1) * KT);
        if (wave_live) {
#pragma unroll
            for (int b32 = 0; b32 < KT / 32; ++b32) {
                const RowFr q0 = row_fr(Qi, 2 * b32, c), q1 = row_fr(Qi, 2 * b32 + 1, c);
                const RowFr d0 = row_fr(Di, 2 * b32, c), d1 = row_fr(Di, 2 * b32 + 1, c);
                const Planes qt_ = tr_fr(Qi, b32, c), dt_ = tr_fr(Di, b32, c);
                float4 l4[2], e4[2];
                uint4 hr[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    l4[kb] = *reinterpret_cast<const float4*>(tb + b32 * 32 + kb * 16 + 4 * c.g);
                    e4[kb] = *reinterpret_cast<const float4*>(tb + KT + b32 * 32 + kb * 16 + 4 * c.g);
                    if (DROP) hr[kb] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint32_t*>(tb) + 2 * KT + b32 * 32 + kb * 16 + 4 * c.g);
                }
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    f32x4 sc[2], dp[2];
                    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
                    sc[0] = row_prod(q0, kmh[n], klh[n], SEED ? f32x4{l4[0].x, l4[0].y, l4[0].z, l4[0].w} : z4);     // S[q][key] (- lse[q])
                    sc[1] = row_prod(q1, kmh[n], klh[n], SEED ? f32x4{l4[1].x, l4[1].y, l4[1].z, l4[1].w} : z4);
                    const f32x4 e0 = SEED && !DROP ? f32x4{e4[0].x, e4[0].y, e4[0].z, e4[0].w} : z4;
                    const f32x4 e1 = SEED && !DROP ? f32x4{e4[1].x, e4[1].y, e4[1].z, e4[1].w} : z4;
                    dp[0] = row_prod(d0, vmh[n], vlh[n], e0);                             // dP[q][key] (- delta[q])
                    dp[1] = row_prod(d1, vmh[n], vlh[n], e1);
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        const float ls[4] = {l4[kb].x, l4[kb].y, l4[kb].z, l4[kb].w};
                        const float es[4] = {e4[kb].x, e4[kb].y, e4[kb].z, e4[kb].w};
                        const uint32_t hrs[4] = {hr[kb].x, hr[kb].y, hr[kb].z, hr[kb].w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = rlt_exp2(SEED ? sc[kb][r] : sc[kb][r] - ls[r]);
                            if (DROP) {
                                const float m = rlt_keep_rc(hrs[r], hk[n], a.drop_thr) ? inv_keep : 0.f;
                                sc[kb][r] = p * m;                                   // dropped P (feeds dV)
                                dp[kb][r] = SEED ? p * (dp[kb][r] * m + es[r]) : p * (dp[kb][r] * m - es[r]);   // dS
                            } else {
                                sc[kb][r] = p;
                                dp[kb][r] = SEED ? p * dp[kb][r] : p * (dp[kb][r] - es[r]);
                            }
                        }
                    }
                    const Planes pp = split_mx(sc[0], sc[1], c.sel0, c.sel1);
                    if (RLT_A6N_2ACC && SEED) mm6_2(dt_, pp, dv[n], dv2[n]);
                    else dv[n] = mm6(dt_, pp, dv[n]);                                // dV^T[d][key] += dO^T P
                    const Planes ds = split_mx(dp[0], dp[1], c.sel0, c.sel1);
                    if (RLT_A6N_2ACC && SEED) mm6_2(qt_, ds, dk[n], dk2[n]);
                    else dk[n] = mm6(qt_, ds, dk[n]);                                // dK^T[d][key] += Q^T dS
                }
                interleave_hint<32 * NB, 2>();
            }
        }
        if (t + 1 < nt) store_tile(buf ^ 1, (t + 1) * KT);
        __syncthreads();
    }
    if (!wave_live) return;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int key = row0 + 16 * n + c.l15;
        if (RLT_A6N_2ACC && SEED) { dk[n] += dk2[n]; dv[n] += dv2[n]; }
        if (key < B) {
            float* drow = a.dqkv + ((size_t)s * B + key) * ld + h * 16 + 4 * c.g;
            *reinterpret_cast<float4*>(drow + E) =
                make_float4(dk[n][0] * a.scale, dk[n][1] * a.scale, dk[n][2] * a.scale, dk[n][3] * a.scale);
            *reinterpret_cast<float4*>(drow + 2 * E) = make_float4(dv[n][0], dv[n][1], dv[n][2], dv[n][3]);
        }
    }
}


// ------------------------------------------------------------------------------------------ dQ / dK+dV: one wavefront per SIMD
// The two-wavefront kernels above leave the matrix pipe half idle: a wavefront's vector work (exp2, P dP, the conversions) and
// its MFMAs depend on each other step by step, and a SIMD's second wavefront does not fill the holes - its own stream has the same
// shape, and the times ADD (ablations in profiles/r05_notes.md: element-wise work 40 % of the dQ kernel's time, every MFMA group
// its full matrix time on top).  Here ONE 256-thread workgroup per CU owns 256 of its own rows (a wavefront 64 = four blocks of
// 16), and the tile body is a SOFTWARE PIPELINE over items (a 32-row block of the tile x one own block = 512 scores): in slot s the
// wavefront issues the row products of item s, the matrix-pipe residuals of item s - 2 and the output products of item s - 3,
// every MFMA followed by a fenced gap that holds <= ~8 cycles of the vector work of items s - 1 .. s - 3 (what a 16x16x32 MFMA
// hides: tools/micro/mfma16_gap.hip), the LDS fragment reads of the next 32-row block and the staging of the next tile - placed
// by tools/gen_attn6n_body.py (dependences, register-ring hazards and operand margins checked at generation time).  The pipeline
// runs ACROSS tile boundaries (tiles of 128 rows, double-buffered images, one barrier per tile; every LDS read of a tile is
// issued before the barrier that ends it) and drains on one extra, empty tile (rows beyond B are staged as zeros - in dK+dV with
// lse = +inf -, so they add nothing: no masks, no special last tile).  No dropout here (train-mode launches take the kernels above).
#ifndef RLT_A6N_OCC1
#define RLT_A6N_OCC1 1       // one wavefront per SIMD, 512 registers (compiled with -mllvm -amdgpu-mfma-vgpr-form: rlt_hip/build.py)
#endif
#ifndef RLT_A6N_DQ1_BODY          // (timing experiments compile other generated bodies: tools/gen_attn6n_body.py with GEN_OMIT)
#define RLT_A6N_DQ1_BODY "attention6n_dq1_body.inc"
#define RLT_A6N_DKV1_BODY "attention6n_dkv1_body.inc"
#endif
#ifdef RLT_A6N_STAMPS
// diagnostic build only: s_memtime at every slot of tiles 8..11 of one workgroup (tools/bench_kernels.py a6n_stamps); entries 16 / 17:
// before / behind the barrier
__device__ unsigned long long a6n_stamps[4 * 4 * 18];
#define A6N_STAMP(k) do { if (blockIdx.x == 64 && lane == 0 && t >= 8 && t < 12) \
    a6n_stamps[(wv * 4 + (t - 8)) * 18 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define A6N_STAMP(k) do { } while (0)
#endif
constexpr int KTN1 = 128;                   // rows per tile
constexpr int PLT = KTN1 * 16;              // bf16 elements per plane
constexpr int IMGT = 3 * PLT;               // ... per image (12 KiB)

// ---- pre-split tile images for the pipelined kernels ---------------------------------------------------------------------------
// Without them every workgroup of a (position, head) pair re-splits the same 128-row tiles (32 workgroups per pair at 8192
// lists), and the split + LDS stores + loads are a fifth of the dK+dV kernel's time (profiles/r05_notes.md).  A prepare pass
// writes each tile of Q, K, V and dO ONCE as the three-plane image in exactly the LDS layout (12 KiB = twelve 1-KiB LDS-DMA
// pieces), and the seeds of a tile's rows (-lse log2e | -delta: 1 KiB) next to them; the kernels stage a tile with
// `global_load_lds_dwordx4` - no registers, no vector instructions.  ntile + 1 records per pair: the last one lies wholly beyond
// B (zero rows, lse = +inf) - the tile the pipeline drains on.
//   images: record (matrix m in {Q, K, V, dO}, pair, tile) at (((m * npair + pair) * (ntile + 1)) + tile) * 12288 bytes
//   seeds:  record (pair, tile) at (pair * (ntile + 1) + tile) * 1024 bytes, behind the four image blocks
constexpr int RECB = IMGT * 2;                 // bytes per image record
__host__ __device__ inline size_t a6n_img_block(int npair, int ntile) { return (size_t)npair * (ntile + 1) * RECB; }
__global__ __launch_bounds__(256) void attn6n_prepare_kernel(const float* __restrict__ src, size_t ld, int cols_per_pos, int S, int B, int H,
                                                             uint8_t* __restrict__ img) {
    const int tid = threadIdx.x, ntile = rlt_cdiv_dev(B, KTN1);
    const int pair = blockIdx.x / (ntile + 1), tile = blockIdx.x % (ntile + 1);
    const int s_ = pair / H, h = pair % H;
    (void)cols_per_pos;
    const float* base = src + (size_t)s_ * B * ld + h * 16;
    uint16_t* rec = reinterpret_cast<uint16_t*>(img + ((size_t)pair * (ntile + 1) + tile) * RECB);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = 64 * u + (tid >> 2), row = tile * KTN1 + r, d0 = 4 * (tid & 3);
        const float4 t = *reinterpret_cast<const float4*>(base + (size_t)min(row, B - 1) * ld + d0);
        const bool ok = row < B;
        uint2 hh_, mm_, ll_;
        split4v(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f, hh_, mm_, ll_);
        uint16_t* im = rec + r * 16 + d0;
        *reinterpret_cast<uint2*>(im) = hh_;
        *reinterpret_cast<uint2*>(im + PLT) = mm_;
        *reinterpret_cast<uint2*>(im + 2 * PLT) = ll_;
    }
}
__global__ __launch_bounds__(256) void attn6n_seed_kernel(const float* __restrict__ lse, const float* __restrict__ delta, int S, int B, int H,
                                                          float* __restrict__ tab) {
    const int tid = threadIdx.x, ntile = rlt_cdiv_dev(B, KTN1);
    const int pair = blockIdx.x / (ntile + 1), tile = blockIdx.x % (ntile + 1);
    const int qi = tile * KTN1 + (tid & (KTN1 - 1)), qc = min(qi, B - 1);
    const float v = tid < KTN1 ? lse[(size_t)pair * B + qc] * LOG2E : delta[(size_t)pair * B + qc];
    tab[((size_t)pair * (ntile + 1) + tile) * 256 + tid] = qi < B ? -v : (tid < KTN1 ? -INFINITY : 0.f);
}

template <bool DKV>
__global__ __launch_bounds__(256, RLT_A6N_OCC1) void attn6n_bwd1_kernel(AttnArgs a) {
    constexpr int NB = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* img0 = reinterpret_cast<uint16_t*>(smem);           // [2 buffers][matrix 0 | matrix 1]: dQ: K, V; dK+dV: Q, dO
    float* tab0 = reinterpret_cast<float*>(img0 + 4 * IMGT);      // dK+dV: [2 buffers][-lse * log2e | -delta][KTN1]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int B = a.B, H = a.H, E = H * 16;
    const size_t ld = (size_t)3 * E;
    int pair, rt;
    map_block(blockIdx.x, a.S * H, rlt_cdiv_dev(B, 256), pair, rt);
    const int s_ = pair / H, h = pair % H;
    const float* base = a.qkv + (size_t)s_ * B * ld + h * 16;
    const float* dobase = a.dout + (size_t)s_ * B * E + h * 16;
    const float* lsebase = a.lse + ((size_t)s_ * H + h) * B;
    const float* delbase = a.delta + ((size_t)s_ * H + h) * B;
    const int row0 = rt * 256 + wv * 64;

    // per-lane constants
    const int ro = l15 * 16 + 8 * (g & 1);
    const int offW[3] = {(g < 2 ? 1 : 2) * PLT + ro, (g < 2 ? 0 : 1) * PLT + ro, ro};      // row fragments [m|l], [h|m], [h|h]
    const int offT = (4 * g + (l15 >> 2)) * 16 + 4 * (l15 & 3);
    bf16x8 sel[2];
    {
        uint32_t s0[2] = {0u, 0u};
        if ((l15 >> 2) == g) s0[(l15 & 3) >> 1] = (l15 & 1) ? 0xBF800000u : 0x0000BF80u;
        sel[0] = frag4(s0[0], s0[1], 0u, 0u);
        sel[1] = frag4(0u, 0u, s0[0], s0[1]);
    }
    LaneN c;
    c.l15 = l15; c.g = g;

    // stationary fragments of the own rows: dQ: Q (scaled) and dO of the queries; dK+dV: K (scaled) and V of the keys
    bf16x8 amh[NB], alh[NB], bmh[NB], blh[NB];
    f32x4 seed_s[DKV ? 1 : NB], seed_d[DKV ? 1 : NB];           // dQ: -lse / -delta of the lane's query in all four registers
    f32x4 acc[DKV ? 2 : 1][NB], acc2[DKV ? 2 : 1][NB];          // dQ: [0] = dQ; dK+dV: [0] = dV, [1] = dK; acc2: the small plane products
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int r = row0 + 16 * n + l15, rc = min(r, B - 1);
        if (DKV) {
            own_frags(base + (size_t)rc * ld + E, c, a.scale * LOG2E, amh[n], alh[n]);
            own_frags(base + (size_t)rc * ld + 2 * E, c, 1.f, bmh[n], blh[n]);
        } else {
            own_frags(base + (size_t)rc * ld, c, a.scale * LOG2E, amh[n], alh[n]);
            own_frags(dobase + (size_t)rc * E, c, 1.f, bmh[n], blh[n]);
            const float s0 = -lsebase[rc] * LOG2E, d0 = -delbase[rc];
            seed_s[n] = f32x4{s0, s0, s0, s0};
            seed_d[n] = f32x4{d0, d0, d0, d0};
        }
#pragma unroll
        for (int w = 0; w < (DKV ? 2 : 1); ++w) {
            acc[w][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc2[w][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    const int nt = rlt_cdiv_dev(B, KTN1);
    // staging: LDS-DMA of the pre-split tile records (attn6n_prepare_kernel): 12 pieces of 1 KiB per image, the two images of a
    // tile adjacent in LDS; dK+dV: + the piece of the tile's seeds.
    const int npair = a.S * H;
    const uint8_t* rec0 = reinterpret_cast<const uint8_t*>(a.img) + ((size_t)(DKV ? 0 : 1) * npair + pair) * (size_t)(nt + 1) * RECB;
    const uint8_t* rec1 = reinterpret_cast<const uint8_t*>(a.img) + ((size_t)(DKV ? 3 : 2) * npair + pair) * (size_t)(nt + 1) * RECB;
    const uint8_t* recs = reinterpret_cast<const uint8_t*>(a.img) + 4 * a6n_img_block(npair, nt) + (size_t)pair * (nt + 1) * 1024;
    // (a wavefront copies pieces 3 wv .. 3 wv + 2 of each image: the record has the LDS layout, so the instruction's immediate offset
    // moves source and destination together and M0 - the LDS base - is written once per image, not per piece (attention6h.hip: 2 % of
    // the launch there).  Nothing else in this kernel uses M0; the statements neither save nor restore it.)
    auto dma = [&](int j, int tile, uint16_t* ibuf, float* tbuf) __attribute__((always_inline)) {
        if (j < 6) {                                            // pieces 3 wv + (j % 3) of matrix j / 3 (no branch: j is a constant)
            const uint8_t* rec = (j < 3 ? rec0 : rec1) + (size_t)tile * RECB + 3 * wv * 1024 + lane * 16;
            if (j % 3 == 0) {
                const uint32_t dst = __builtin_amdgcn_readfirstlane(
                    (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(reinterpret_cast<uint8_t*>(ibuf) + (j / 3) * (IMGT * 2) + 3 * wv * 1024));
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(dst) : "memory");
            }
            asm volatile("global_load_lds_dwordx4 %0, off offset:%1" :: "v"(rec), "n"((j % 3) * 1024) : "memory");
        } else if (DKV) {                                       // the seeds: every wavefront copies the same KiB (no branch on the wavefront)
            const uint8_t* rec = recs + (size_t)tile * 1024 + lane * 16;
            const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(reinterpret_cast<uint8_t*>(tbuf)));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(rec) : "memory");
        }
    };
    // prologue: tile 0 -> buffer 0
#pragma unroll
    for (int j = 0; j < 7; ++j) dma(j, 0, img0, tab0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // the pipeline's registers: RING item sets (scores / dP in fp32, the planes of P and dS) and the fragments of the current block
    f32x4 sc[4][2], dp[4][2];
    uint32_t pln[2][3][4][4];                                   // [P | dS][h, m, l][ring][dword]
    bf16x8 fr[2][2][2][3];                                      // row fragments [buffer = 32-row block & 1][matrix][16-row block][which]
    v4s trf[2][2][3][2];                                        // transposed fragments [buffer][matrix][plane][half]
    float4 tabv[2][2][2];                                       // dK+dV: [buffer][lse | delta][16-row block] seeds of the block's rows
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) { sc[i][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[i][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int i = 0; i < 2 * 3 * 4 * 4; ++i) (&pln[0][0][0][0])[i] = 0u;
#pragma unroll
    for (int i = 0; i < 24; ++i) (&fr[0][0][0][0])[i] = frag4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int i = 0; i < 24; ++i) (&trf[0][0][0][0])[i] = v4s{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) (&tabv[0][0][0])[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    for (int t = 0; t <= nt; ++t) {                             // nt + 1 bodies: the last one drains the pipeline on an empty tile
        const int cur = t & 1;
        const uint16_t* Ic = img0 + cur * 2 * IMGT;
        uint16_t* In = img0 + (cur ^ 1) * 2 * IMGT;
        const float* Tc = tab0 + cur * 2 * KTN1;
        float* Tn = tab0 + (cur ^ 1) * 2 * KTN1;
        const int t_next = min(t + 1, nt);           // (the body of the drain tile stages that tile once more: never read)
#pragma unroll
        for (int n = 0; n < NB; ++n) {              // (their only uses are "a" operands: keep them in AGPRs across the back edge)
            asm volatile("" : "+a"(amh[n]), "+a"(alh[n]), "+a"(bmh[n]), "+a"(blh[n]));
#pragma unroll
            for (int w = 0; w < (DKV ? 2 : 1); ++w) asm volatile("" : "+a"(acc[w][n]), "+a"(acc2[w][n]));
        }
        asm volatile("" : "+a"(sel[0]), "+a"(sel[1]));
#define GAP_END __builtin_amdgcn_sched_barrier(0)
        // The MFMAs of the body are asm statements with the accumulator tied in place ("+v"): hipcc's own choice puts the result of
        // an accumulate somewhere else and reuses the old registers at once - a write-after-read hazard it then pads with s_nop
        // (80 per tile body).  It sees no MFMA in an asm statement and pads nothing, so the schedule keeps the distances itself
        // (tools/gen_attn6n_body.py): LAG gaps from an MFMA to the first vector read of its result, MARGIN gaps from a vector /
        // LDS write of an operand to the MFMA that reads it, and no register of a fragment or tile is rewritten in the gap
        // behind the MFMA that reads it.  A dependent accumulate straight behind its producer needs no wait states.
        // Register classes: what the vector ALU touches (score / dP tiles, planes, seeds, staging) in VGPRs; what only MFMAs touch in
        // AGPRs - the stationary fragments and selection constants ("a" B / A operands), the LDS fragments (read straight into
        // AGPRs) and the output accumulators (C / D in AGPRs: "+a") - so that nothing is copied between the two halves of the file.
        auto mma_aa = [&](f32x4& d, bf16x8 av, bf16x8 bv) __attribute__((always_inline)) {      // tile += A(agpr) B(agpr)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(av), "a"(bv));
        };
        auto mma_aa_c = [&](f32x4& d, bf16x8 av, bf16x8 bv, const f32x4& cv) __attribute__((always_inline)) {      // tile = A B + seed
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(d) : "a"(av), "a"(bv), "v"(cv));
        };
        auto mma_av = [&](f32x4& d, bf16x8 av, bf16x8 bv) __attribute__((always_inline)) {      // tile += A(agpr) B(vgpr)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(av), "v"(bv));
        };
        // (the transposed fragments stay in VGPRs: ds_read_b64_tr_b16 pairs are joined into one operand, and as an "a" operand hipcc
        // would copy them over with v_accvgpr_write right in front of the MFMA - unpadded in front of an asm statement)
        auto mma_out = [&](f32x4& d, bf16x8 av, bf16x8 bv) __attribute__((always_inline)) {     // output accumulator (agpr) += A(vgpr) B(vgpr)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(av), "v"(bv));
        };
        auto m_s = [&](int it, int n, int k, int fb) __attribute__((always_inline)) {
            const int kb = k / 3, j = k % 3;
            if (j == 0) {
                if (DKV) mma_aa_c(sc[it][kb], fr[fb][0][kb][0], amh[n], f32x4{tabv[fb][0][kb].x, tabv[fb][0][kb].y, tabv[fb][0][kb].z, tabv[fb][0][kb].w});
                else mma_aa_c(sc[it][kb], fr[fb][0][kb][0], amh[n], seed_s[DKV ? 0 : n]);
            } else {
                mma_aa(sc[it][kb], fr[fb][0][kb][j], j == 1 ? alh[n] : amh[n]);
            }
        };
        auto m_d = [&](int it, int n, int k, int fb) __attribute__((always_inline)) {
            const int kb = k / 3, j = k % 3;
            if (j == 0) {
                if (DKV) mma_aa_c(dp[it][kb], fr[fb][1][kb][0], bmh[n], f32x4{tabv[fb][1][kb].x, tabv[fb][1][kb].y, tabv[fb][1][kb].z, tabv[fb][1][kb].w});
                else mma_aa_c(dp[it][kb], fr[fb][1][kb][0], bmh[n], seed_d[DKV ? 0 : n]);
            } else {
                mma_aa(dp[it][kb], fr[fb][1][kb][j], j == 1 ? blh[n] : bmh[n]);
            }
        };
        auto plane = [&](int which, int lvl, int it) __attribute__((always_inline)) {
            return frag4(pln[which][lvl][it][0], pln[which][lvl][it][1], pln[which][lvl][it][2], pln[which][lvl][it][3]);
        };
        auto m_r = [&](int it, int which, int level, int kb) __attribute__((always_inline)) {
            mma_av(which ? dp[it][kb] : sc[it][kb], sel[kb], plane(which, level - 1, it));
        };
        auto m_o = [&](int it, int n, int which, int k, int fb) __attribute__((always_inline)) {
            const int ap = k == 0 || k == 3 ? 1 : k == 1 ? 2 : 0, bp = k == 0 || k == 4 ? 1 : k == 2 ? 2 : 0;
            const int mat = DKV ? (which ? 0 : 1) : 0;          // dQ: K^T; dV: dO^T, dK: Q^T
            const int w = DKV ? which : 0;
            typedef short v8s __attribute__((ext_vector_type(8)));
            const v4s x = trf[fb][mat][ap][0], y = trf[fb][mat][ap][1];
            const v8s av = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
            mma_out(k < 5 ? acc2[w][n] : acc[w][n], __builtin_bit_cast(bf16x8, av), plane(which, bp, it));
        };
        auto e_exp = [&](int it, int kb, int r) __attribute__((always_inline)) { sc[it][kb][r] = rlt_exp2(sc[it][kb][r]); };
        auto e_mul = [&](int it, int kb, int r) __attribute__((always_inline)) { dp[it][kb][r] *= sc[it][kb][r]; };
        auto c_pk = [&](int it, int which, int lvl, int j) __attribute__((always_inline)) {
            const f32x4& tile = which ? dp[it][j >> 1] : sc[it][j >> 1];
            pln[which][lvl][it][j] = pk2n(tile[2 * (j & 1)], tile[2 * (j & 1) + 1]);
        };
        auto rd_row = [&](int fb, int mat, int kb, int w, int b32) __attribute__((always_inline)) {
            fr[fb][mat][kb][w] = *reinterpret_cast<const bf16x8*>(Ic + mat * IMGT + offW[w] + (2 * b32 + kb) * 256);
        };
        auto rd_tr = [&](int fb, int mat, int pl, int half, int b32) __attribute__((always_inline)) {
            trf[fb][mat][pl][half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (v4s __attribute__((address_space(3)))*)(Ic + mat * IMGT + pl * PLT + offT + b32 * 512 + half * 256));
        };
        auto rd_tab = [&](int fb, int which, int kb, int b32) __attribute__((always_inline)) {
            tabv[fb][which][kb] = *reinterpret_cast<const float4*>(Tc + which * KTN1 + b32 * 32 + kb * 16 + 4 * g);
        };
        auto st_dma = [&](int j) __attribute__((always_inline)) { dma(j, t_next, In, Tn); };
        if constexpr (DKV) {
#include RLT_A6N_DKV1_BODY
        } else {
#include RLT_A6N_DQ1_BODY
        }
#undef GAP_END
        A6N_STAMP(16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wavefront's LDS-DMA pieces of the next tile have landed
        __syncthreads();
        A6N_STAMP(17);
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int r = row0 + 16 * n + l15;
        if (r < B) {
            float* drow = a.dqkv + ((size_t)s_ * B + r) * ld + h * 16 + 4 * g;
            if (DKV) {
                const f32x4 dv = acc[0][n] + acc2[0][n], dk = acc[1][n] + acc2[1][n];
                *reinterpret_cast<float4*>(drow + E) = make_float4(dk[0] * a.scale, dk[1] * a.scale, dk[2] * a.scale, dk[3] * a.scale);
                *reinterpret_cast<float4*>(drow + 2 * E) = make_float4(dv[0], dv[1], dv[2], dv[3]);
            } else {
                const f32x4 dq = acc[0][n] + acc2[0][n];
                *reinterpret_cast<float4*>(drow) = make_float4(dq[0] * a.scale, dq[1] * a.scale, dq[2] * a.scale, dq[3] * a.scale);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ forward: one wavefront per SIMD
// The same item pipeline as the backward kernels above, for the FORWARD pass: S(s) x6 | R1(s-2) x2 | R2(s-2) x2 | O(s-3) x6 - 16 MFMAs per
// item of 512 scores - with exp2, the row sums, the plane conversions, the fragment reads and the LDS-DMA staging of the pre-split K / V tile
// images in the gaps (tools/gen_attn6n_body.py fwd).  A pipeline four items deep cannot move a running maximum, so the reference of a
// query's weights is FIXED before it starts: the maximum of the scores against the first 32 keys (what the two-wavefront kernel
// above uses too, until a block's weights leave the comfortable range).  fp32 and the exact split are scale-free, so the result is
// the same whatever the reference - unless a weight overflows: a workgroup whose normalisers end up non-finite, zero or above 2^100
// raises its flag in a.redo, and a second launch of the two-wavefront kernel (same grid, same block -> rows map) redoes exactly the
// flagged workgroups with the moving reference (it returns at once everywhere else).  Needs B % 128 == 0 (the drain tile multiplies
// its row sums by a zero flag; a partly filled tile would need a mask per key), no dropout.
__global__ __launch_bounds__(256, RLT_A6N_OCC1) void attn6n_fwd1_kernel(AttnArgs a) {
    constexpr int NB = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* img0 = reinterpret_cast<uint16_t*>(smem);           // [2 buffers][K image | V image]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int B = a.B, H = a.H, E = H * 16;
    const size_t ld = (size_t)3 * E;
    int pair, rt;
    map_block(blockIdx.x, a.S * H, rlt_cdiv_dev(B, 256), pair, rt);
    const int s_ = pair / H, h = pair % H;
    const float* base = a.qkv + (size_t)s_ * B * ld + h * 16;
    const int row0 = rt * 256 + wv * 64;

    const int ro = l15 * 16 + 8 * (g & 1);
    const int offW[3] = {(g < 2 ? 1 : 2) * PLT + ro, (g < 2 ? 0 : 1) * PLT + ro, ro};      // row fragments [m|l], [h|m], [h|h]
    const int offT = (4 * g + (l15 >> 2)) * 16 + 4 * (l15 & 3);
    bf16x8 sel[2];
    {
        uint32_t s0[2] = {0u, 0u};
        if ((l15 >> 2) == g) s0[(l15 & 3) >> 1] = (l15 & 1) ? 0xBF800000u : 0x0000BF80u;
        sel[0] = frag4(s0[0], s0[1], 0u, 0u);
        sel[1] = frag4(0u, 0u, s0[0], s0[1]);
    }
    LaneN c;
    c.l15 = l15; c.g = g;

    bf16x8 amh[NB], alh[NB];                                      // the lane's queries (scaled), stationary B operands of the score products
    f32x4 seed_s[NB], acc[NB], acc2[NB];
    float l_run[NB], m_ref[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int r = row0 + 16 * n + l15, rc = min(r, B - 1);
        own_frags(base + (size_t)rc * ld, c, a.scale * LOG2E, amh[n], alh[n]);
        acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc2[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        l_run[n] = 0.f;
    }
    const int nt = B / KTN1;
    const int npair = a.S * H;
    const uint8_t* rec0 = reinterpret_cast<const uint8_t*>(a.img) + ((size_t)0 * npair + pair) * (size_t)(nt + 1) * RECB;      // K images (block 0 of the forward's buffer)
    const uint8_t* rec1 = reinterpret_cast<const uint8_t*>(a.img) + ((size_t)1 * npair + pair) * (size_t)(nt + 1) * RECB;      // V images
    auto dma = [&](int j, int tile, uint16_t* ibuf) __attribute__((always_inline)) {       // (as in attn6n_bwd1_kernel: M0 once per image)
        const uint8_t* rec = (j < 3 ? rec0 : rec1) + (size_t)tile * RECB + 3 * wv * 1024 + lane * 16;
        if (j % 3 == 0) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(
                (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(reinterpret_cast<uint8_t*>(ibuf) + (j / 3) * (IMGT * 2) + 3 * wv * 1024));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(dst) : "memory");
        }
        asm volatile("global_load_lds_dwordx4 %0, off offset:%1" :: "v"(rec), "n"((j % 3) * 1024) : "memory");
    };
#pragma unroll
    for (int j = 0; j < 6; ++j) dma(j, 0, img0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // the reference of each query: its largest score against the first 32 keys (log2 domain)
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        float tmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            t = mm(*reinterpret_cast<const bf16x8*>(img0 + offW[0] + kb * 256), amh[n], t);
            t = mm(*reinterpret_cast<const bf16x8*>(img0 + offW[1] + kb * 256), alh[n], t);
            t = mm(*reinterpret_cast<const bf16x8*>(img0 + offW[2] + kb * 256), amh[n], t);
            tmax = fmaxf(fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3])), tmax);
        }
        m_ref[n] = col_max4(tmax);
        seed_s[n] = f32x4{-m_ref[n], -m_ref[n], -m_ref[n], -m_ref[n]};
    }

    f32x4 sc[4][2];
    uint32_t pln[1][3][4][4];                                   // [P][h, m, l][ring][dword]
    bf16x8 fr[2][1][2][3];                                      // K row fragments [buffer = 32-row block & 1][.][16-row block][which]
    v4s trf[2][2][3][2];                                        // V^T fragments [buffer][matrix (1 used)][plane][half]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) sc[i][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3 * 4 * 4; ++i) (&pln[0][0][0][0])[i] = 0u;
#pragma unroll
    for (int i = 0; i < 12; ++i) (&fr[0][0][0][0])[i] = frag4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int i = 0; i < 24; ++i) (&trf[0][0][0][0])[i] = v4s{0, 0, 0, 0};

    for (int t = 0; t <= nt; ++t) {                             // nt + 1 bodies: the last one drains the pipeline on the empty tile record
        const int cur = t & 1;
        const uint16_t* Ic = img0 + cur * 2 * IMGT;
        uint16_t* In = img0 + (cur ^ 1) * 2 * IMGT;
        const int t_next = min(t + 1, nt);
        const float livef = t < nt ? 1.f : 0.f;                  // the items that START in the drain body carry no keys
        const float prevf = t > 0 ? 1.f : 0.f;
#pragma unroll
        for (int n = 0; n < NB; ++n) asm volatile("" : "+a"(amh[n]), "+a"(alh[n]), "+a"(acc[n]), "+a"(acc2[n]));
        asm volatile("" : "+a"(sel[0]), "+a"(sel[1]));
#define GAP_END __builtin_amdgcn_sched_barrier(0)
        auto mma_aa = [&](f32x4& d, bf16x8 av, bf16x8 bv) __attribute__((always_inline)) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(av), "a"(bv));
        };
        auto mma_aa_c = [&](f32x4& d, bf16x8 av, bf16x8 bv, const f32x4& cv) __attribute__((always_inline)) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(d) : "a"(av), "a"(bv), "v"(cv));
        };
        auto mma_av = [&](f32x4& d, bf16x8 av, bf16x8 bv) __attribute__((always_inline)) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(av), "v"(bv));
        };
        auto mma_out = [&](f32x4& d, bf16x8 av, bf16x8 bv) __attribute__((always_inline)) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(av), "v"(bv));
        };
        auto m_s = [&](int it, int n, int k, int fb) __attribute__((always_inline)) {
            const int kb = k / 3, j = k % 3;
            if (j == 0) mma_aa_c(sc[it][kb], fr[fb][0][kb][0], amh[n], seed_s[n]);
            else mma_aa(sc[it][kb], fr[fb][0][kb][j], j == 1 ? alh[n] : amh[n]);
        };
        auto plane = [&](int which, int lvl, int it) __attribute__((always_inline)) {
            return frag4(pln[0][lvl][it][0], pln[0][lvl][it][1], pln[0][lvl][it][2], pln[0][lvl][it][3]);
        };
        auto m_r = [&](int it, int which, int level, int kb) __attribute__((always_inline)) {
            mma_av(sc[it][kb], sel[kb], plane(0, level - 1, it));
        };
        auto m_o = [&](int it, int n, int which, int k, int fb) __attribute__((always_inline)) {
            const int ap = k == 0 || k == 3 ? 1 : k == 1 ? 2 : 0, bp = k == 0 || k == 4 ? 1 : k == 2 ? 2 : 0;
            typedef short v8s __attribute__((ext_vector_type(8)));
            const v4s x = trf[fb][1][ap][0], y = trf[fb][1][ap][1];
            const v8s av = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
            mma_out(k < 5 ? acc2[n] : acc[n], __builtin_bit_cast(bf16x8, av), plane(0, bp, it));
        };
        auto e_exp = [&](int it, int kb, int r) __attribute__((always_inline)) { sc[it][kb][r] = rlt_exp2(sc[it][kb][r]); };
        // (chunks of the previous tile's last items: in the first body there is no previous tile - the ring holds zeros - and the weight must be 0, not exp2(0))
        auto e_exp_p = [&](int it, int kb, int r) __attribute__((always_inline)) { sc[it][kb][r] = rlt_exp2(sc[it][kb][r]) * prevf; };
        auto e_sum = [&](int it, int n, int kb, int r) __attribute__((always_inline)) { l_run[n] = __builtin_fmaf(sc[it][kb][r], livef, l_run[n]); };
        auto e_sum_p = [&](int it, int n, int kb, int r) __attribute__((always_inline)) { l_run[n] += sc[it][kb][r]; };
        auto c_pk = [&](int it, int which, int lvl, int j) __attribute__((always_inline)) {
            const f32x4& tile = sc[it][j >> 1];
            pln[0][lvl][it][j] = pk2n(tile[2 * (j & 1)], tile[2 * (j & 1) + 1]);
        };
        auto rd_row = [&](int fb, int mat, int kb, int w, int b32) __attribute__((always_inline)) {
            fr[fb][0][kb][w] = *reinterpret_cast<const bf16x8*>(Ic + offW[w] + (2 * b32 + kb) * 256);
        };
        auto rd_tr = [&](int fb, int mat, int pl, int half, int b32) __attribute__((always_inline)) {
            trf[fb][1][pl][half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (v4s __attribute__((address_space(3)))*)(Ic + IMGT + pl * PLT + offT + b32 * 512 + half * 256));
        };
        auto st_dma = [&](int j) __attribute__((always_inline)) { dma(j, t_next, In); };
#include "attention6n_fwd1_body.inc"
#undef GAP_END
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wavefront's LDS-DMA pieces of the next tile have landed
        __syncthreads();
    }
    bool bad = false;
    float inv[NB], lse[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const float l_tot = col_sum4(l_run[n]);
        bad |= !(l_tot > 0.f && l_tot <= 1.2676506e30f);         // 2^100: non-finite, vanished or far out of range -> redo with the moving reference
        inv[n] = 1.f / l_tot;
        lse[n] = (m_ref[n] + log2f(l_tot)) * LN2;
    }
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    if (tid == 0 && a.redo) a.redo[blockIdx.x] = any_bad ? 1u : 0u;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int q = row0 + 16 * n + l15;
        const f32x4 o = acc[n] + acc2[n];
        if (q < B) {
            *reinterpret_cast<float4*>(a.o + ((size_t)s_ * B + q) * E + h * 16 + 4 * g) =
                make_float4(o[0] * inv[n], o[1] * inv[n], o[2] * inv[n], o[3] * inv[n]);
            if (g == 0) a.lse_o[((size_t)s_ * H + h) * B + q] = lse[n];
        }
    }
}

template <bool DROP>
int launch6n(int which, const AttnArgs& a, hipStream_t st) {
    constexpr int NB = RLT_A6N_NB, NBK = RLT_A6N_NBK;
    const size_t shm = (size_t)4 * IMGN * sizeof(uint16_t) + (which == 1 ? 2 * 3 * KT * sizeof(float) : 2 * KT * sizeof(uint32_t));
    // backward without dropout at 512 lists and more: the one-wavefront pipelined kernels (RLT_A6N_1=0: the kernels above, A/B runs)
    static const bool one = [] { const char* e = getenv("RLT_A6N_1"); return !e || atoi(e) != 0; }();
    if (!DROP && one && which != 0 && a.B >= 512 && a.img) {
        const size_t shm1 = (size_t)4 * IMGT * sizeof(uint16_t) + (which == 1 ? 2 * 2 * KTN1 * sizeof(float) : 0);
        const dim3 grid1(a.S * a.H * rlt_cdiv(a.B, 256));
        int rc;
        if (which == 1) {
            if ((rc = rlt_allow_lds(attn6n_bwd1_kernel<true>, shm1))) return rc;
            hipLaunchKernelGGL((attn6n_bwd1_kernel<true>), grid1, dim3(256), shm1, st, a);
        } else {
            if ((rc = rlt_allow_lds(attn6n_bwd1_kernel<false>, shm1))) return rc;
            hipLaunchKernelGGL((attn6n_bwd1_kernel<false>), grid1, dim3(256), shm1, st, a);
        }
        return RLT_LAUNCH_RESULT();
    }
    const bool seed = a.B >= 512;
    const dim3 gq(a.S * a.H * rlt_cdiv(a.B, 64 * NB)), gk(a.S * a.H * rlt_cdiv(a.B, 64 * NBK));
    // forward without dropout, 512 lists and more in whole 128-row tiles, K / V images prepared (a.img) and a flag word per workgroup
    // (a.redo): the pipelined kernel, then the two-wavefront kernel for the workgroups it flagged (RLT_A6N_F1=0: the latter alone)
    static const bool fwd1 = [] { const char* e = getenv("RLT_A6N_F1"); return !e || atoi(e) != 0; }();
    if (!DROP && one && fwd1 && which == 0 && a.B >= 512 && a.B % KTN1 == 0 && a.img && a.redo && NB == 4) {
        const size_t shm1 = (size_t)4 * IMGT * sizeof(uint16_t);
        const int rc = rlt_allow_lds(attn6n_fwd1_kernel, shm1);
        if (rc) return rc;
        hipLaunchKernelGGL(attn6n_fwd1_kernel, gq, dim3(256), shm1, st, a);
        hipLaunchKernelGGL((attn6n_fwd_kernel<NB, DROP, true>), gq, dim3(256), shm, st, a);
        return RLT_LAUNCH_RESULT();
    }
    if (which == 0) {
        if (seed) hipLaunchKernelGGL((attn6n_fwd_kernel<NB, DROP, true>), gq, dim3(256), shm, st, a);
        else hipLaunchKernelGGL((attn6n_fwd_kernel<NB, DROP, false>), gq, dim3(256), shm, st, a);
    } else if (which == 1) {
        if (seed) hipLaunchKernelGGL((attn6n_bwd_dkv_kernel<NBK, DROP, true>), gk, dim3(256), shm, st, a);
        else hipLaunchKernelGGL((attn6n_bwd_dkv_kernel<NBK, DROP, false>), gk, dim3(256), shm, st, a);
    } else {
        if (seed) hipLaunchKernelGGL((attn6n_bwd_dq_kernel<NB, DROP, true>), gq, dim3(256), shm, st, a);
        else hipLaunchKernelGGL((attn6n_bwd_dq_kernel<NB, DROP, false>), gq, dim3(256), shm, st, a);
    }
    return RLT_LAUNCH_RESULT();
}

}  // namespace

#ifdef RLT_A6N_STAMPS
extern "C" int rlt_debug_a6n_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(a6n_stamps), sizeof(unsigned long long) * 4 * 4 * 18);
}
#endif

// images of the pipelined backward kernels (see attn6n_prepare_kernel): bytes, and the passes that write them - what = 0 Q, 1 K,
// 2 V (columns of a.qkv), 3 dO (a.dout), 4 the seeds (a.lse, a.delta) - into a.img
size_t rlt_attn6n_images_bytes(int S, int B, int H) {
    const int npair = S * H, ntile = rlt_cdiv(B, KTN1);
    return 4 * a6n_img_block(npair, ntile) + (size_t)npair * (ntile + 1) * 1024;
}
// the forward's buffer: K images | V images (two blocks)
size_t rlt_attn6n_fwd_images_bytes(int S, int B, int H) { return 2 * a6n_img_block(S * H, rlt_cdiv(B, KTN1)); }
int rlt_attn6n_prepare(int what, const AttnArgs& a, hipStream_t st) { return rlt_attn6n_prepare_at(what, what, a, st); }
// ... matrix `what` into image block `slot` of a.img
int rlt_attn6n_prepare_at(int what, int slot, const AttnArgs& a, hipStream_t st) {
    const int npair = a.S * a.H, ntile = rlt_cdiv(a.B, KTN1), E = a.H * 16;
    uint8_t* img = reinterpret_cast<uint8_t*>(const_cast<void*>(a.img));
    const dim3 grid(npair * (ntile + 1));
    if (what < 3)
        hipLaunchKernelGGL(attn6n_prepare_kernel, grid, dim3(256), 0, st, a.qkv + what * E, (size_t)3 * E, 0, a.S, a.B, a.H, img + slot * a6n_img_block(npair, ntile));
    else if (what == 3)
        hipLaunchKernelGGL(attn6n_prepare_kernel, grid, dim3(256), 0, st, a.dout, (size_t)E, 0, a.S, a.B, a.H, img + 3 * a6n_img_block(npair, ntile));
    else
        hipLaunchKernelGGL(attn6n_seed_kernel, grid, dim3(256), 0, st, a.lse, a.delta, a.S, a.B, a.H,
                           reinterpret_cast<float*>(img + 4 * a6n_img_block(npair, ntile)));
    return RLT_LAUNCH_RESULT();
}

// which = 0 forward, 1 dK/dV, 2 dQ (head dim 16, bf16x6 arithmetic)
int rlt_attn6n_run(int which, const AttnArgs& a, hipStream_t st) {
    return a.drop_p > 0.f ? launch6n<true>(which, a, st) : launch6n<false>(which, a, st);
}

// List-axis attention, split-bf16 ("bf16x3") variant of attention.hip: same algorithm, same interface,
// same fp32 inputs/outputs/softmax/accumulators, but every MFMA product a*b is evaluated as
// a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with hi = bf16(x), lo = bf16(x - hi) on v_mfma_f32_32x32x16_bf16
// (16x the rate of the f32 MFMA per product; ~2^-16 relative error per product).
//
// Layout notes (32x32x16 bf16 MFMA: A[row = l&31][k = 8h + j], B[k = 8h + j][col = l&31], h = l>>5, j < 8;
// D[row = (r&3) + 8(r>>2) + 4h][col = l&31]):
//  * operands whose contraction index is contiguous in memory (K, Q, dO rows over d) are staged in LDS as
//    [row][d] bf16 hi / lo images: a lane's fragment is ONE ds_read_b128;
//  * products that contract over the key / query index take their B operand straight from the accumulator
//    registers of the previous product (registers 8s..8s+7 -> k-step s; the k order inside a step is
//    key = 16s + 8(j>>2) + 4h + (j&3)), so the other operand is staged TRANSPOSED, [d][kpos], with the keys of
//    each 16-block stored in that permuted order: again one ds_read_b128 per fragment;
//  * fp32 -> (hi, lo) splitting happens once per workgroup per tile, at staging time (4x4 register transpose
//    for the transposed image), not per MFMA.
#include "attention_common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LDT3 = 72;                 // bf16 elements per row of a transposed [d][64 rows] image (144 B)

__device__ __forceinline__ uint32_t pk2(float a, float b) {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    v2 t = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ void split4(float a, float b, float c, float d, uint2& hi, uint2& lo) {
    hi.x = pk2(a, b);
    hi.y = pk2(c, d);
    lo.x = pk2(a - __builtin_bit_cast(float, hi.x << 16), b - __builtin_bit_cast(float, hi.x & 0xffff0000u));
    lo.y = pk2(c - __builtin_bit_cast(float, hi.y << 16), d - __builtin_bit_cast(float, hi.y & 0xffff0000u));
}
__device__ __forceinline__ bf16x8 as_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

// 8 fp32 -> hi / lo bf16x8 fragments
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
    uint2 h0, l0, h1, l1;
    split4(x[0], x[1], x[2], x[3], h0, l0);
    split4(x[4], x[5], x[6], x[7], h1, l1);
    hi = as_frag(make_uint4(h0.x, h0.y, h1.x, h1.y));
    lo = as_frag(make_uint4(l0.x, l0.y, l1.x, l1.y));
}

__device__ __forceinline__ f32x16 mfma3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
    return c;
}

// position of row `key` (0..63) inside a transposed image row: 16-blocks keep their place, inside a block
// bits 2 and 3 are swapped (so that accumulator registers 8s..8s+7 of lane-half h are 8 consecutive slots)
__device__ __forceinline__ constexpr int kpos(int key) { return (key & ~12) | ((key & 4) << 1) | ((key & 8) >> 1); }

// ---- staging: a [64 rows][HD] fp32 tile -> registers (each thread: 4 consecutive rows x 4 consecutive d) ----
template <int HD>
struct Stage { float4 v[4]; };

template <int HD>
__device__ __forceinline__ bool stage_active(int tid) { return tid < 16 * (HD / 4); }

template <int HD>
__device__ __forceinline__ void stage_load(const float* __restrict__ base, size_t ld, int row0, int nrows, int tid, Stage<HD>& st) {
    // thread -> (rb = 4-row block, dq = 4-column block); threads beyond the tile idle (HD < 64)
    const int rb = tid / (HD / 4), dq = tid % (HD / 4);
    if (!stage_active<HD>(tid)) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + 4 * rb + i;
        const float4 t = *reinterpret_cast<const float4*>(base + (size_t)min(row, nrows - 1) * ld + 4 * dq);
        const bool ok = row < nrows;
        st.v[i] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    }
}
// rows image: [row][HD + 8] hi / lo
template <int HD>
__device__ __forceinline__ void stage_store_rows(uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, int tid, const Stage<HD>& st) {
    constexpr int LD = HD + 8;
    const int rb = tid / (HD / 4), dq = tid % (HD / 4);
    if (!stage_active<HD>(tid)) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint2 h, l;
        split4(st.v[i].x, st.v[i].y, st.v[i].z, st.v[i].w, h, l);
        *reinterpret_cast<uint2*>(hi + (4 * rb + i) * LD + 4 * dq) = h;
        *reinterpret_cast<uint2*>(lo + (4 * rb + i) * LD + 4 * dq) = l;
    }
}
// transposed image: [d][LDT3] hi / lo, rows permuted by kpos
template <int HD>
__device__ __forceinline__ void stage_store_T(uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, int tid, const Stage<HD>& st) {
    const int rb = tid / (HD / 4), dq = tid % (HD / 4);
    if (!stage_active<HD>(tid)) return;
    const float* f0 = reinterpret_cast<const float*>(&st.v[0]);
    const float* f1 = reinterpret_cast<const float*>(&st.v[1]);
    const float* f2 = reinterpret_cast<const float*>(&st.v[2]);
    const float* f3 = reinterpret_cast<const float*>(&st.v[3]);
    const int kp = kpos(4 * rb);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint2 h, l;
        split4(f0[c], f1[c], f2[c], f3[c], h, l);
        *reinterpret_cast<uint2*>(hi + (4 * dq + c) * LDT3 + kp) = h;
        *reinterpret_cast<uint2*>(lo + (4 * dq + c) * LDT3 + kp) = l;
    }
}

// ---- this lane's half of a row as B-operand fragments over d: frag[ks] covers d = 16ks + 8h + j ----
template <int HD>
__device__ __forceinline__ void row_frags(const float* __restrict__ rowp, int hh, float mul, bf16x8 (&fh)[HD / 16], bf16x8 (&fl)[HD / 16]) {
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
        const float4 v0 = *reinterpret_cast<const float4*>(rowp + 16 * ks + 8 * hh);
        const float4 v1 = *reinterpret_cast<const float4*>(rowp + 16 * ks + 8 * hh + 4);
        const float x[8] = {v0.x * mul, v0.y * mul, v0.z * mul, v0.w * mul, v1.x * mul, v1.y * mul, v1.z * mul, v1.w * mul};
        split8(x, fh[ks], fl[ks]);
    }
}

// acc (D[row = tile row][col = lane]) += rows-image tile (A, rows sub*32 + l31) x register fragments (B)
template <int HD>
__device__ __forceinline__ f32x16 mma_rows(const uint16_t* __restrict__ hi, const uint16_t* __restrict__ lo, int sub, int l31, int hh,
                                           const bf16x8 (&bh)[HD / 16], const bf16x8 (&bl)[HD / 16], f32x16 acc) {
    constexpr int LD = HD + 8;
    const int off = (sub * 32 + l31) * LD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(hi + off + 16 * ks);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(lo + off + 16 * ks);
        acc = mfma3(ah, al, bh[ks], bl[ks], acc);
    }
    return acc;
}

// acc[dt] (D[row = d][col = lane]) += transposed-image tile (A, row d, 32 permuted rows of sub-tile `sub`)
//                                     x the accumulator registers w of a previous product (B, rows of w = k)
template <int HD>
__device__ __forceinline__ void mma_T(const uint16_t* __restrict__ thi, const uint16_t* __restrict__ tlo, int sub, int l31, int hh,
                                      const f32x16& w, f32x16 (&acc)[(HD + 31) / 32]) {
    constexpr int DT = (HD + 31) / 32;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float x[8] = {w[8 * s + 0], w[8 * s + 1], w[8 * s + 2], w[8 * s + 3], w[8 * s + 4], w[8 * s + 5], w[8 * s + 6], w[8 * s + 7]};
        bf16x8 wh, wl;
        split8(x, wh, wl);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = dt * 32 + l31;
            const bool ok = d < HD;
            const int off = min(d, HD - 1) * LDT3 + sub * 32 + 16 * s + 8 * hh;
            uint4 ah = *reinterpret_cast<const uint4*>(thi + off);
            uint4 al = *reinterpret_cast<const uint4*>(tlo + off);
            if (HD < 32 && !ok) { ah = make_uint4(0, 0, 0, 0); al = make_uint4(0, 0, 0, 0); }
            acc[dt] = mfma3(as_frag(ah), as_frag(al), wh, wl, acc[dt]);
        }
    }
}

template <int HD> constexpr int rows_elems() { return KT * (HD + 8); }     // one hi or lo rows image
template <int HD> constexpr int T_elems() { return HD * LDT3; }             // one hi or lo transposed image

// ------------------------------------------------------------------------------------------ forward
template <int HD>
__global__ __launch_bounds__(256, 2) void attn3_fwd_kernel(AttnArgs a) {
    constexpr int DT = (HD + 31) / 32;
    constexpr int STAGE = 2 * rows_elems<HD>() + 2 * T_elems<HD>();       // K hi,lo (rows) | V hi,lo (transposed)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* lds = reinterpret_cast<uint16_t*>(smem);
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

    bf16x8 qh[HD / 16], ql[HD / 16];
    row_frags<HD>(base + (size_t)qc * ld, hh, a.scale * LOG2E, qh, ql);

    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    auto k_hi = [&](int buf) { return lds + buf * STAGE; };
    auto k_lo = [&](int buf) { return lds + buf * STAGE + rows_elems<HD>(); };
    auto v_hi = [&](int buf) { return lds + buf * STAGE + 2 * rows_elems<HD>(); };
    auto v_lo = [&](int buf) { return lds + buf * STAGE + 2 * rows_elems<HD>() + T_elems<HD>(); };

    Stage<HD> rk, rv;
    const int nt = rlt_cdiv_dev(B, KT);
    stage_load<HD>(base + E, ld, 0, B, tid, rk);
    stage_load<HD>(base + 2 * E, ld, 0, B, tid, rv);
    stage_store_rows<HD>(k_hi(0), k_lo(0), tid, rk);
    stage_store_T<HD>(v_hi(0), v_lo(0), tid, rv);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) {
            stage_load<HD>(base + E, ld, (t + 1) * KT, B, tid, rk);
            stage_load<HD>(base + 2 * E, ld, (t + 1) * KT, B, tid, rv);
        }
        if (wave_live) {
            f32x16 sc[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[sub][r] = 0.f;
                sc[sub] = mma_rows<HD>(k_hi(buf), k_lo(buf), sub, l31, hh, qh, ql, sc[sub]);     // S^T[key][q], log2 domain
            }
            float tmax = -INFINITY;
            if (t == nt - 1) {            // only the last tile can hold keys beyond B
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * KT + sub * 32 + acc_row(r, hh) >= B) sc[sub][r] = -INFINITY;
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sc[sub][r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = exp2f(m_run - m_new);
            float psum = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = exp2f(sc[sub][r] - m_new);
                    sc[sub][r] = p;
                    psum += p;
                }
            l_run = l_run * alpha + psum;
            m_run = m_new;
            if (a.drop_p > 0.f) {
                const uint32_t ps = pair_seed(a.seed, pair);
                const float inv_keep = 1.f / (1.f - a.drop_p);
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = t * KT + sub * 32 + acc_row(r, hh);
                        sc[sub][r] = rlt_keep(ps, (uint32_t)q, (uint32_t)key, a.drop_thr) ? sc[sub][r] * inv_keep : 0.f;
                    }
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) mma_T<HD>(v_hi(buf), v_lo(buf), sub, l31, hh, sc[sub], oacc);   // O^T[d][q]
        }
        if (t + 1 < nt) {
            stage_store_rows<HD>(k_hi(buf ^ 1), k_lo(buf ^ 1), tid, rk);
            stage_store_T<HD>(v_hi(buf ^ 1), v_lo(buf ^ 1), tid, rv);
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

// ------------------------------------------------------------------------------------------ dQ
// workgroup = 128 queries; loops over 64-key tiles.  K is staged in both images (rows for S^T = K Q^T,
// transposed for dQ^T = K^T dS^T), V in the rows image (dP^T = V dO^T).
template <int HD>
__global__ __launch_bounds__(256, 2) void attn3_bwd_dq_kernel(AttnArgs a) {
    constexpr int DT = (HD + 31) / 32;
    constexpr int STAGE = 4 * rows_elems<HD>() + 2 * T_elems<HD>();       // K rows hi,lo | V rows hi,lo | K^T hi,lo
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* lds = reinterpret_cast<uint16_t*>(smem);
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

    bf16x8 qh[HD / 16], ql[HD / 16], doh[HD / 16], dol[HD / 16];
    row_frags<HD>(base + (size_t)qc * ld, hh, a.scale * LOG2E, qh, ql);
    row_frags<HD>(a.dout + ((size_t)s * B + qc) * E + h * HD, hh, 1.f, doh, dol);
    const float lse2 = a.lse[((size_t)s * H + h) * B + qc] * LOG2E;
    const float del = a.delta[((size_t)s * H + h) * B + qc];

    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

    uint16_t* kr_hi = lds;
    uint16_t* kr_lo = lds + rows_elems<HD>();
    uint16_t* vr_hi = lds + 2 * rows_elems<HD>();
    uint16_t* vr_lo = lds + 3 * rows_elems<HD>();
    uint16_t* kt_hi = lds + 4 * rows_elems<HD>();
    uint16_t* kt_lo = lds + 4 * rows_elems<HD>() + T_elems<HD>();
    (void)STAGE;

    Stage<HD> rk, rv;
    const int nt = rlt_cdiv_dev(B, KT);
    stage_load<HD>(base + E, ld, 0, B, tid, rk);
    stage_load<HD>(base + 2 * E, ld, 0, B, tid, rv);
    for (int t = 0; t < nt; ++t) {
        // single LDS stage: (stores for tile t) | barrier | (prefetch t+1 into registers, multiply tile t) | barrier
        stage_store_rows<HD>(kr_hi, kr_lo, tid, rk);
        stage_store_T<HD>(kt_hi, kt_lo, tid, rk);
        stage_store_rows<HD>(vr_hi, vr_lo, tid, rv);
        __syncthreads();
        if (t + 1 < nt) {
            stage_load<HD>(base + E, ld, (t + 1) * KT, B, tid, rk);
            stage_load<HD>(base + 2 * E, ld, (t + 1) * KT, B, tid, rv);
        }
        if (wave_live) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x16 sc, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
                sc = mma_rows<HD>(kr_hi, kr_lo, sub, l31, hh, qh, ql, sc);       // S^T[key][q]
                dp = mma_rows<HD>(vr_hi, vr_lo, sub, l31, hh, doh, dol, dp);     // dP^T[key][q]
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kidx = t * KT + sub * 32 + acc_row(r, hh);
                    const float p = kidx < B ? exp2f(sc[r] - lse2) : 0.f;
                    float dpr = dp[r];
                    if (a.drop_p > 0.f)
                        dpr = rlt_keep(pair_seed(a.seed, pair), (uint32_t)q, (uint32_t)kidx, a.drop_thr) ? dpr / (1.f - a.drop_p) : 0.f;
                    dp[r] = p * (dpr - del);                                       // dS^T
                }
                mma_T<HD>(kt_hi, kt_lo, sub, l31, hh, dp, dq);                     // dQ^T[d][q] += K^T dS^T
            }
        }
        __syncthreads();
    }
    if (!wave_live || q >= B) return;
    store_acc_T<HD>(a.dqkv + ((size_t)s * B + q) * ld + h * HD, hh, dq, a.scale);
}

// ------------------------------------------------------------------------------------------ dK, dV
// workgroup = 128 keys; loops over 64-query tiles.  Q and dO are staged in both images (rows for
// S = Q K^T and dP = dO V^T, transposed for dV^T = dO^T P and dK^T = Q^T dS).
template <int HD>
__global__ __launch_bounds__(256, (HD > 32 ? 1 : 2)) void attn3_bwd_dkv_kernel(AttnArgs a) {
    constexpr int DT = (HD + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* lds = reinterpret_cast<uint16_t*>(smem);
    uint16_t* qr_hi = lds;
    uint16_t* qr_lo = lds + rows_elems<HD>();
    uint16_t* dr_hi = lds + 2 * rows_elems<HD>();
    uint16_t* dr_lo = lds + 3 * rows_elems<HD>();
    uint16_t* qt_hi = lds + 4 * rows_elems<HD>();
    uint16_t* qt_lo = qt_hi + T_elems<HD>();
    uint16_t* dt_hi = qt_hi + 2 * T_elems<HD>();
    uint16_t* dt_lo = qt_hi + 3 * T_elems<HD>();
    float* Ls = reinterpret_cast<float*>(qt_hi + 4 * T_elems<HD>());       // [KT] lse * log2e
    float* Es = Ls + KT;                                                     // [KT] delta
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

    bf16x8 kh[HD / 16], kl[HD / 16], vh[HD / 16], vl[HD / 16];
    row_frags<HD>(base + (size_t)kc * ld + E, hh, a.scale * LOG2E, kh, kl);
    row_frags<HD>(base + (size_t)kc * ld + 2 * E, hh, 1.f, vh, vl);

    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }

    Stage<HD> rq, rd;
    float rl = 0.f, re = 0.f;
    const int nt = rlt_cdiv_dev(B, KT);
    auto load_small = [&](int row0) {
        if (tid < KT) {
            const int qi = row0 + tid, qcl = min(qi, B - 1);
            const float l = lsebase[qcl], e = delbase[qcl];
            rl = qi < B ? l * LOG2E : 0.f;
            re = qi < B ? e : 0.f;
        }
    };
    stage_load<HD>(base, ld, 0, B, tid, rq);
    stage_load<HD>(dobase, (size_t)E, 0, B, tid, rd);
    load_small(0);
    for (int t = 0; t < nt; ++t) {
        stage_store_rows<HD>(qr_hi, qr_lo, tid, rq);
        stage_store_T<HD>(qt_hi, qt_lo, tid, rq);
        stage_store_rows<HD>(dr_hi, dr_lo, tid, rd);
        stage_store_T<HD>(dt_hi, dt_lo, tid, rd);
        if (tid < KT) { Ls[tid] = rl; Es[tid] = re; }
        __syncthreads();
        if (t + 1 < nt) {
            stage_load<HD>(base, ld, (t + 1) * KT, B, tid, rq);
            stage_load<HD>(dobase, (size_t)E, (t + 1) * KT, B, tid, rd);
            load_small((t + 1) * KT);
        }
        if (wave_live) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x16 sc, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
                sc = mma_rows<HD>(qr_hi, qr_lo, sub, l31, hh, kh, kl, sc);       // S[q][key]
                dp = mma_rows<HD>(dr_hi, dr_lo, sub, l31, hh, vh, vl, dp);       // dP[q][key]
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = sub * 32 + acc_row(r, hh);
                    const bool ok = t * KT + ql < B;
                    const float p = ok ? exp2f(sc[r] - Ls[ql]) : 0.f;
                    float pd = p, dpr = dp[r];
                    if (a.drop_p > 0.f) {
                        const bool keep = rlt_keep(pair_seed(a.seed, pair), (uint32_t)(t * KT + ql), (uint32_t)key, a.drop_thr);
                        const float m = keep ? 1.f / (1.f - a.drop_p) : 0.f;
                        pd = p * m;
                        dpr *= m;
                    }
                    sc[r] = pd;                                                    // (dropped) P, feeds dV
                    dp[r] = p * (dpr - Es[ql]);                                    // dS
                }
                mma_T<HD>(dt_hi, dt_lo, sub, l31, hh, sc, dv);                     // dV^T[d][key] += dO^T P
                mma_T<HD>(qt_hi, qt_lo, sub, l31, hh, dp, dk);                     // dK^T[d][key] += Q^T dS
            }
        }
        __syncthreads();
    }
    if (!wave_live || key >= B) return;
    float* drow = a.dqkv + ((size_t)s * B + key) * ld + h * HD;
    store_acc_T<HD>(drow + E, hh, dk, a.scale);
    store_acc_T<HD>(drow + 2 * E, hh, dv, 1.f);
}

// ------------------------------------------------------------------------------------------ dV / dK (split)
// The fused dK/dV kernel above needs ~340 registers (spills at 2 wavefronts/SIMD).  Split form: two kernels
// of the dQ kernel's shape, each within 256 registers: dV = P^T dO (recomputes S), dK = dS^T Q (recomputes S, dP).
// 5 MFMA products instead of 4, but both run at occupancy 2 with no scratch traffic.
template <int HD>
__global__ __launch_bounds__(256, 2) void attn3_bwd_dv_kernel(AttnArgs a) {
    constexpr int DT = (HD + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* lds = reinterpret_cast<uint16_t*>(smem);
    uint16_t* qr_hi = lds;
    uint16_t* qr_lo = lds + rows_elems<HD>();
    uint16_t* dt_hi = lds + 2 * rows_elems<HD>();
    uint16_t* dt_lo = dt_hi + T_elems<HD>();
    float* Ls = reinterpret_cast<float*>(dt_hi + 2 * T_elems<HD>());       // [KT] lse * log2e
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
    const int key = ktile * QT + wv * 32 + l31;
    const bool wave_live = ktile * QT + wv * 32 < B;
    const int kc = min(key, B - 1);

    bf16x8 kh[HD / 16], kl[HD / 16];
    row_frags<HD>(base + (size_t)kc * ld + E, hh, a.scale * LOG2E, kh, kl);
    f32x16 dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[dt][r] = 0.f;

    Stage<HD> rq, rd;
    float rl = 0.f;
    const int nt = rlt_cdiv_dev(B, KT);
    auto load_small = [&](int row0) {
        if (tid < KT) {
            const int qi = row0 + tid;
            const float l = lsebase[min(qi, B - 1)];
            rl = qi < B ? l * LOG2E : 0.f;
        }
    };
    stage_load<HD>(base, ld, 0, B, tid, rq);
    stage_load<HD>(dobase, (size_t)E, 0, B, tid, rd);
    load_small(0);
    for (int t = 0; t < nt; ++t) {
        stage_store_rows<HD>(qr_hi, qr_lo, tid, rq);
        stage_store_T<HD>(dt_hi, dt_lo, tid, rd);
        if (tid < KT) Ls[tid] = rl;
        __syncthreads();
        if (t + 1 < nt) {
            stage_load<HD>(base, ld, (t + 1) * KT, B, tid, rq);
            stage_load<HD>(dobase, (size_t)E, (t + 1) * KT, B, tid, rd);
            load_small((t + 1) * KT);
        }
        if (wave_live) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x16 sc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = 0.f;
                sc = mma_rows<HD>(qr_hi, qr_lo, sub, l31, hh, kh, kl, sc);       // S[q][key]
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = sub * 32 + acc_row(r, hh);
                    float p = (t * KT + ql < B) ? exp2f(sc[r] - Ls[ql]) : 0.f;
                    if (a.drop_p > 0.f)
                        p = rlt_keep(pair_seed(a.seed, pair), (uint32_t)(t * KT + ql), (uint32_t)key, a.drop_thr) ? p / (1.f - a.drop_p) : 0.f;
                    sc[r] = p;                                                     // (dropped) P
                }
                mma_T<HD>(dt_hi, dt_lo, sub, l31, hh, sc, dv);                     // dV^T[d][key] += dO^T P
            }
        }
        __syncthreads();
    }
    if (!wave_live || key >= B) return;
    store_acc_T<HD>(a.dqkv + ((size_t)s * B + key) * ld + h * HD + 2 * E, hh, dv, 1.f);
}

template <int HD>
__global__ __launch_bounds__(256, 2) void attn3_bwd_dk_kernel(AttnArgs a) {
    constexpr int DT = (HD + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* lds = reinterpret_cast<uint16_t*>(smem);
    uint16_t* qr_hi = lds;
    uint16_t* qr_lo = lds + rows_elems<HD>();
    uint16_t* dr_hi = lds + 2 * rows_elems<HD>();
    uint16_t* dr_lo = lds + 3 * rows_elems<HD>();
    uint16_t* qt_hi = lds + 4 * rows_elems<HD>();
    uint16_t* qt_lo = qt_hi + T_elems<HD>();
    float* Ls = reinterpret_cast<float*>(qt_hi + 2 * T_elems<HD>());       // [KT] lse * log2e
    float* Es = Ls + KT;                                                     // [KT] delta
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

    bf16x8 kh[HD / 16], kl[HD / 16], vh[HD / 16], vl[HD / 16];
    row_frags<HD>(base + (size_t)kc * ld + E, hh, a.scale * LOG2E, kh, kl);
    row_frags<HD>(base + (size_t)kc * ld + 2 * E, hh, 1.f, vh, vl);
    f32x16 dk[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[dt][r] = 0.f;

    Stage<HD> rq, rd;
    float rl = 0.f, re = 0.f;
    const int nt = rlt_cdiv_dev(B, KT);
    auto load_small = [&](int row0) {
        if (tid < KT) {
            const int qi = row0 + tid, qcl = min(qi, B - 1);
            const float l = lsebase[qcl], e = delbase[qcl];
            rl = qi < B ? l * LOG2E : 0.f;
            re = qi < B ? e : 0.f;
        }
    };
    stage_load<HD>(base, ld, 0, B, tid, rq);
    stage_load<HD>(dobase, (size_t)E, 0, B, tid, rd);
    load_small(0);
    for (int t = 0; t < nt; ++t) {
        stage_store_rows<HD>(qr_hi, qr_lo, tid, rq);
        stage_store_T<HD>(qt_hi, qt_lo, tid, rq);
        stage_store_rows<HD>(dr_hi, dr_lo, tid, rd);
        if (tid < KT) { Ls[tid] = rl; Es[tid] = re; }
        __syncthreads();
        if (t + 1 < nt) {
            stage_load<HD>(base, ld, (t + 1) * KT, B, tid, rq);
            stage_load<HD>(dobase, (size_t)E, (t + 1) * KT, B, tid, rd);
            load_small((t + 1) * KT);
        }
        if (wave_live) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x16 sc, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
                sc = mma_rows<HD>(qr_hi, qr_lo, sub, l31, hh, kh, kl, sc);       // S[q][key]
                dp = mma_rows<HD>(dr_hi, dr_lo, sub, l31, hh, vh, vl, dp);       // dP[q][key]
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = sub * 32 + acc_row(r, hh);
                    const float p = (t * KT + ql < B) ? exp2f(sc[r] - Ls[ql]) : 0.f;
                    float dpr = dp[r];
                    if (a.drop_p > 0.f)
                        dpr = rlt_keep(pair_seed(a.seed, pair), (uint32_t)(t * KT + ql), (uint32_t)key, a.drop_thr) ? dpr / (1.f - a.drop_p) : 0.f;
                    dp[r] = p * (dpr - Es[ql]);                                    // dS
                }
                mma_T<HD>(qt_hi, qt_lo, sub, l31, hh, dp, dk);                     // dK^T[d][key] += Q^T dS
            }
        }
        __syncthreads();
    }
    if (!wave_live || key >= B) return;
    store_acc_T<HD>(a.dqkv + ((size_t)s * B + key) * ld + h * HD + E, hh, dk, a.scale);
}

template <int HD> size_t dv3_smem() { return (size_t)(2 * rows_elems<HD>() + 2 * T_elems<HD>()) * sizeof(uint16_t) + KT * sizeof(float); }
template <int HD> size_t dk3_smem() { return (size_t)(4 * rows_elems<HD>() + 2 * T_elems<HD>()) * sizeof(uint16_t) + 2 * KT * sizeof(float); }
template <int HD> size_t fwd3_smem() { return (size_t)2 * (2 * rows_elems<HD>() + 2 * T_elems<HD>()) * sizeof(uint16_t); }
template <int HD> size_t dq3_smem() { return (size_t)(4 * rows_elems<HD>() + 2 * T_elems<HD>()) * sizeof(uint16_t); }
template <int HD> size_t dkv3_smem() { return (size_t)(4 * rows_elems<HD>() + 4 * T_elems<HD>()) * sizeof(uint16_t) + 2 * KT * sizeof(float); }

template <int HD>
int launch3(int which, const AttnArgs& a, hipStream_t st) {
    const int grid = a.S * a.H * rlt_cdiv(a.B, QT);
    int rc;
    if (which == 0) {
        if ((rc = rlt_allow_lds(attn3_fwd_kernel<HD>, fwd3_smem<HD>()))) return rc;
        hipLaunchKernelGGL(attn3_fwd_kernel<HD>, dim3(grid), dim3(256), fwd3_smem<HD>(), st, a);
    } else if (which == 1) {
        static const int fused = [] { const char* e = getenv("RLT_ATTN_DKV_FUSED"); return e ? atoi(e) : 0; }();
        if (fused || HD <= 32) {       // small head dims fit the fused kernel at occupancy 2
            if ((rc = rlt_allow_lds(attn3_bwd_dkv_kernel<HD>, dkv3_smem<HD>()))) return rc;
            hipLaunchKernelGGL(attn3_bwd_dkv_kernel<HD>, dim3(grid), dim3(256), dkv3_smem<HD>(), st, a);
        } else {
            if ((rc = rlt_allow_lds(attn3_bwd_dv_kernel<HD>, dv3_smem<HD>()))) return rc;
            if ((rc = rlt_allow_lds(attn3_bwd_dk_kernel<HD>, dk3_smem<HD>()))) return rc;
            hipLaunchKernelGGL(attn3_bwd_dv_kernel<HD>, dim3(grid), dim3(256), dv3_smem<HD>(), st, a);
            hipLaunchKernelGGL(attn3_bwd_dk_kernel<HD>, dim3(grid), dim3(256), dk3_smem<HD>(), st, a);
        }
    } else {
        if ((rc = rlt_allow_lds(attn3_bwd_dq_kernel<HD>, dq3_smem<HD>()))) return rc;
        hipLaunchKernelGGL(attn3_bwd_dq_kernel<HD>, dim3(grid), dim3(256), dq3_smem<HD>(), st, a);
    }
    return RLT_LAUNCH_RESULT();
}

int dispatch3(int which, const AttnArgs& a, int HD, hipStream_t st) {
    if (HD == 64) return launch3<64>(which, a, st);
    if (HD == 32) return launch3<32>(which, a, st);
    if (HD == 16) return launch3<16>(which, a, st);
    return RLT_E_SHAPE;
}

}  // namespace

int rlt_attn3_fwd(const AttnArgs& a, int HD, hipStream_t st) { return dispatch3(0, a, HD, st); }
int rlt_attn3_bwd_dkv(const AttnArgs& a, int HD, hipStream_t st) { return dispatch3(1, a, HD, st); }
int rlt_attn3_bwd_dq(const AttnArgs& a, int HD, hipStream_t st) { return dispatch3(2, a, HD, st); }

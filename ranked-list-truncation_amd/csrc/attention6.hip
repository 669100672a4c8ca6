// List-axis attention, fp32-FAITHFUL variant on the bf16 matrix pipe ("bf16x6", the attention half of that mode; the GEMM
// half is gemm6_kernel in gemm.hip): same algorithm, same interface and the same fp32 softmax arithmetic as the exact-fp32
// kernels of attention.hip, but every MFMA product a*b is evaluated from an EXACT three-way bf16 split of both operands,
//   x = h + m + l  (h = bf16(x), m = bf16(x - h), l = x - h - m: 8 + 8 + 8 significand bits),
//   a*b ~ m*m' + l*h' + h*l' + m*h' + h*m' + h*h'   (six v_mfma_f32_32x32x16_bf16, fp32 accumulate, smallest first);
// what is dropped is < 2^-23 |a b| in the worst case, 2^-29 typically (tests/test_split6.py): under one fp32 ulp of the
// product.  (On operands that all share the worst-case low bits the ROUNDINGS of the small plane products into the large running
// sums add coherently in a one-accumulator kernel: what made dQ / dK at head dim 64 4-8x the f32 MFMA kernels' error through round 5
// was the coherent error of THIS file's forward in O, amplified through delta = rowsum(dO o O); the head-dim-16 kernels and the
// head-dim-64 forward at 512 lists and more (attention6n.hip, attention6h.hip) keep the small products in an accumulator of their
// own, and with them every head-dim-64 gradient is within 2x of the f32 kernels on every class - tools/gpu_probe.py x6_adversarial,
// profiles/r06_notes.md.)  Six bf16 products cost 6/16 of one f32 MFMA
// product (2500 / 6 = 417 TFLOP/s of fp32-level peak against 157.3).
//
// Head dims 64 (the AttnCut / MMOECut family), 32 and 16 (Choopy / MtChoopy), with and without train-mode dropout; head dim
// 128 (PLECut) stays on the exact-fp32 kernels in that mode.  Layout as in attention.hip: one wavefront owns 32 queries (32 keys in the dK/dV
// kernel), scores are produced TRANSPOSED so that the softmax is a per-lane loop over accumulator registers and the
// probabilities are directly the B operand of the next product (registers 8s..8s+7 = k-step s).  Tiles of 64 rows are
// split once per workgroup at staging time into [row][d] bf16 images (three planes, 144-byte rows); products that contract
// over the tile's ROW index read their A operand transposed from the same image with ds_read_b64_tr_b16, so no transposed
// image exists.  One LDS copy of a tile pair (55 KB): two workgroups per CU, their tile phases uncorrelated.
#include "attention_common.h"
#include "split6.h"
#include <stdlib.h>
#include <utility>

#ifndef RLT_A6_DKV_PREFETCH
#define RLT_A6_DKV_PREFETCH 0   // dK+dV kernel, 1: next tile's global loads issued BEFORE the tile body (32 staging registers live across it); measured 48.9 ms against 47.6 with the loads after the body (the partner workgroup covers their latency)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));

// per head dim: bf16 elements per image row (HD + 8: 16-byte aligned fragments, rows 16 bytes apart in the banks), per plane,
// per tile image [h | m | l]
template <int HD> constexpr int ldr6() { return HD + 8; }
template <int HD> constexpr int plane6() { return KT * (HD + 8); }
template <int HD> constexpr int img6() { return 3 * KT * (HD + 8); }

__device__ __forceinline__ uint32_t pk2_6(float a, float b) {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    v2 t = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, t);
}
// exact three-way split of four values (see gemm.hip split4x3)
__device__ __forceinline__ void split4x3_6(float a, float b, float c, float d, uint2& hi, uint2& mid, uint2& lo) {
    hi.x = pk2_6(a, b);
    hi.y = pk2_6(c, d);
    asm("" : "+v"(hi.x), "+v"(hi.y));
    const float ra = a - __builtin_bit_cast(float, hi.x << 16), rb = b - __builtin_bit_cast(float, hi.x & 0xffff0000u);
    const float rc = c - __builtin_bit_cast(float, hi.y << 16), rd = d - __builtin_bit_cast(float, hi.y & 0xffff0000u);
    mid.x = pk2_6(ra, rb);
    mid.y = pk2_6(rc, rd);
    asm("" : "+v"(mid.x), "+v"(mid.y));
    lo.x = pk2_6(ra - __builtin_bit_cast(float, mid.x << 16), rb - __builtin_bit_cast(float, mid.x & 0xffff0000u));
    lo.y = pk2_6(rc - __builtin_bit_cast(float, mid.y << 16), rd - __builtin_bit_cast(float, mid.y & 0xffff0000u));
}
struct Frag3 { bf16x8 h, m, l; };
__device__ __forceinline__ Frag3 split8x3(const float (&x)[8]) {
    uint2 h0, m0, l0, h1, m1, l1;
    split4x3_6(x[0], x[1], x[2], x[3], h0, m0, l0);
    split4x3_6(x[4], x[5], x[6], x[7], h1, m1, l1);
    Frag3 f;
    f.h = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
    f.m = __builtin_bit_cast(bf16x8, make_uint4(m0.x, m0.y, m1.x, m1.y));
    f.l = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
    return f;
}
__device__ __forceinline__ f32x16 mfma6(const Frag3& a, const Frag3& b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, c, 0, 0, 0);
    return c;
}

// ---- staging: a [64 rows][HD] fp32 tile -> registers (HD / 16 float4 per thread) -> three-plane image
template <int HD> struct Stage6 { float4 v[HD / 16]; };
template <int HD>
__device__ __forceinline__ void stage6_load(const float* __restrict__ base, size_t ld, int row0, int nrows, int tid, Stage6<HD>& st) {
#pragma unroll
    for (int i = 0; i < HD / 16; ++i) {
        const int idx = tid + 256 * i;
        const int row = row0 + idx / (HD / 4), dq = idx % (HD / 4);
        const float4 t = *reinterpret_cast<const float4*>(base + (size_t)min(row, nrows - 1) * ld + 4 * dq);
        const bool ok = row < nrows;
        st.v[i] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    }
}
template <int HD>
__device__ __forceinline__ void stage6_store(uint16_t* __restrict__ img, int tid, const Stage6<HD>& st, float mul) {
#pragma unroll
    for (int i = 0; i < HD / 16; ++i) {
        const int idx = tid + 256 * i;
        uint2 h, m, l;
        split4x3_6(st.v[i].x * mul, st.v[i].y * mul, st.v[i].z * mul, st.v[i].w * mul, h, m, l);
        const int off = (idx / (HD / 4)) * ldr6<HD>() + 4 * (idx % (HD / 4));
        *reinterpret_cast<uint2*>(img + off) = h;
        *reinterpret_cast<uint2*>(img + plane6<HD>() + off) = m;
        *reinterpret_cast<uint2*>(img + 2 * plane6<HD>() + off) = l;
    }
}

// ---- pre-split tile images in HBM (IMG kernels) ---------------------------------------------------------------------------
// Without them every workgroup re-splits the same K / V (Q / dO) tiles - 16 to 32 workgroups per (position, head) pair - and the
// split is vector work in kernels that are vector-ISSUE-bound (profiles/r03_notes.md).  A prepare pass writes each 64-row tile of
// Q, K, V (forward) and dO (backward) ONCE as the three-plane image in exactly the LDS layout (padding included; 27 / 15 / 9 KiB at
// head dim 64 / 32 / 16, whole 1 KiB LDS-DMA pieces), and the attention kernels stage a tile with `global_load_lds_dwordx4` - no
// registers, no vector instructions.  Record (matrix m, pair, tile) at ((m * npair + pair) * ntile + tile) * img6_bytes.
template <int HD> constexpr int img6_bytes() { return img6<HD>() * 2; }
static_assert(img6_bytes<64>() % 1024 == 0 && img6_bytes<32>() % 1024 == 0 && img6_bytes<16>() % 1024 == 0, "whole LDS-DMA pieces");
template <int HD>
__device__ __forceinline__ const uint8_t* rec6(const void* img, int mat, int npair, int ntile, int pair, int tile) {
    return reinterpret_cast<const uint8_t*>(img) + ((size_t)((size_t)mat * npair + pair) * ntile + tile) * img6_bytes<HD>();
}
// LDS-DMA copy of one image by NW wavefronts (inline assembly: invisible to hipcc's wait-count insertion - the caller waits with
// dma_fence6() before the barrier that publishes the image)
template <int HD, int NW>
__device__ __forceinline__ void dma_image6(uint16_t* lds_img, const uint8_t* __restrict__ rec, int wv, int lane) {
    constexpr int NP = img6_bytes<HD>() / 1024;
#pragma unroll
    for (int c = 0; c < (NP + NW - 1) / NW; ++c) {
        const int chunk = wv + NW * c;
        if (chunk < NP) {
            // (wave-uniform: the LDS destination travels in M0)
            const uint32_t dst = __builtin_amdgcn_readfirstlane(
                (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(reinterpret_cast<uint8_t*>(lds_img) + chunk * 1024));
            const uint8_t* src = rec + chunk * 1024 + lane * 16;
            RLT_DMA_ASM(dst, src);
        }
    }
}
__device__ __forceinline__ void dma_fence6() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// the prepare pass: one workgroup per (pair, tile), blockIdx.y = matrix
template <int HD>
__global__ __launch_bounds__(256) void attn6_prepare_kernel(const float* __restrict__ src, size_t ld, int col0, int cstep, int S, int B, int H,
                                                            uint8_t* __restrict__ img) {
    const int tid = threadIdx.x, ntile = rlt_cdiv_dev(B, KT), npair = S * H;
    const int pair = blockIdx.x / ntile, tile = blockIdx.x % ntile, mat = blockIdx.y;
    const int s = pair / H, h = pair % H;
    const float* base = src + (size_t)s * B * ld + col0 + mat * cstep + h * HD;
    Stage6<HD> st;
    stage6_load<HD>(base, ld, tile * KT, B, tid, st);
    uint16_t* rec = reinterpret_cast<uint16_t*>(img + ((size_t)((size_t)mat * npair + pair) * ntile + tile) * img6_bytes<HD>());
#pragma unroll
    for (int i = 0; i < HD / 16; ++i) {
        const int idx = tid + 256 * i;
        uint2 hh_, mm_, ll_;
        split4x3_6(st.v[i].x, st.v[i].y, st.v[i].z, st.v[i].w, hh_, mm_, ll_);
        const int off = (idx / (HD / 4)) * ldr6<HD>() + 4 * (idx % (HD / 4));
        *reinterpret_cast<uint2*>(rec + off) = hh_;
        *reinterpret_cast<uint2*>(rec + plane6<HD>() + off) = mm_;
        *reinterpret_cast<uint2*>(rec + 2 * plane6<HD>() + off) = ll_;
    }
}

// this lane's half of a global row as B-operand fragments over d: frag[ks] covers d = 16 ks + 8 hh + j
template <int HD>
__device__ __forceinline__ void row_frags6(const float* __restrict__ rowp, int hh, float mul, Frag3 (&f)[HD / 16]) {
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
        const float4 v0 = *reinterpret_cast<const float4*>(rowp + 16 * ks + 8 * hh);
        const float4 v1 = *reinterpret_cast<const float4*>(rowp + 16 * ks + 8 * hh + 4);
        const float x[8] = {v0.x * mul, v0.y * mul, v0.z * mul, v0.w * mul, v1.x * mul, v1.y * mul, v1.z * mul, v1.w * mul};
        f[ks] = split8x3(x);
    }
}

// acc (D[row = tile row][col = lane]) += image tile (A, rows sub*32 + l31, contraction over d) x register fragments (B)
template <int HD>
__device__ __forceinline__ f32x16 mma_rows6(const uint16_t* __restrict__ img, int sub, int l31, int hh,
                                            const Frag3 (&b)[HD / 16], f32x16 acc) {
    const int off = (sub * 32 + l31) * ldr6<HD>() + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
        Frag3 a;
        a.h = *reinterpret_cast<const bf16x8*>(img + off + 16 * ks);
        a.m = *reinterpret_cast<const bf16x8*>(img + plane6<HD>() + off + 16 * ks);
        a.l = *reinterpret_cast<const bf16x8*>(img + 2 * plane6<HD>() + off + 16 * ks);
        acc = mfma6(a, b[ks], acc);
    }
    return acc;
}

__device__ __forceinline__ v4s tr_read6(const uint16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p));
}
__device__ __forceinline__ bf16x8 cat_frag6(v4s a, v4s b) {
    typedef short v8s __attribute__((ext_vector_type(8)));
    const v8s v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// acc[dt] (D[row = d][col = lane]) += sum over the 32 rows of sub-tile `sub` of  image[row][d] * w[row][lane]:
// A operand = the image read TRANSPOSED (ds_read_b64_tr_b16: the 16-lane group (lane >> 4) covers d = 32 dt + 16 (group & 1) +
// (lane & 15); the k slots of lane half hh are rows 16 s + 4 hh + {0..3} and + 8 - the rows registers 8s..8s+7 of w hold),
// B operand = the accumulator registers w of a previous product, split three ways here
// Head dim 16: the 32-row MFMA output has 16 rows to spare; the lanes that would supply d = 16..31 read d - 16 again (a
// valid address) and their output rows are never stored.
template <int HD>
__device__ __forceinline__ void mma_cols6(const uint16_t* __restrict__ img, int sub, int lane, const f32x16& w,
                                          f32x16 (&acc)[(HD + 31) / 32]) {
    constexpr int DT = (HD + 31) / 32, LDR = ldr6<HD>(), PL = plane6<HD>();
    const int hh = lane >> 5;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float x[8] = {w[8 * s + 0], w[8 * s + 1], w[8 * s + 2], w[8 * s + 3], w[8 * s + 4], w[8 * s + 5], w[8 * s + 6], w[8 * s + 7]};
        const Frag3 b = split8x3(x);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int row = sub * 32 + 16 * s + 4 * hh + ((lane & 15) >> 2);
            const int off = row * LDR + 32 * dt + (HD >= 32 ? 16 * ((lane >> 4) & 1) : 0) + 4 * (lane & 3);
            Frag3 a;
            a.h = cat_frag6(tr_read6(img + off), tr_read6(img + off + 8 * LDR));
            a.m = cat_frag6(tr_read6(img + PL + off), tr_read6(img + PL + off + 8 * LDR));
            a.l = cat_frag6(tr_read6(img + 2 * PL + off), tr_read6(img + 2 * PL + off + 8 * LDR));
            acc[dt] = mfma6(a, b, acc[dt]);
        }
    }
}

// ------------------------------------------------------------------------------------------ forward
// DROP: train-mode dropout of the attention probabilities (same counter-based masks as the other two kernel families:
// keep(pair seed, query, key) = row_hash(query) * col_hash(key) >= threshold; the hashes of a tile's rows / keys sit in a
// 64-entry LDS table written at staging time, the lane's own hash in a register)
template <int HD, bool DROP, bool IMG>
__global__ __launch_bounds__(256, 2) void attn6_fwd_kernel(AttnArgs a) {
    constexpr int DT = (HD + 31) / 32, IMG6 = img6<HD>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* Ki = reinterpret_cast<uint16_t*>(smem);          // K tile image
    uint16_t* Vi = Ki + IMG6;                                   // V tile image
    uint32_t* htab = reinterpret_cast<uint32_t*>(Vi + IMG6);    // [KT] column hashes of the tile's keys (DROP)
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

    Frag3 qf[HD / 16];
    row_frags6<HD>(base + (size_t)qc * ld, hh, a.scale * LOG2E, qf);

    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const uint32_t ps = DROP ? pair_seed(a.seed, pair) : 0u;
    const uint32_t hq = DROP ? rlt_row_hash(ps, (uint32_t)q) : 0u;
    const float inv_keep = DROP ? 1.f / (1.f - a.drop_p) : 1.f;
    Stage6<HD> rk, rv;
    const int nt = rlt_cdiv_dev(B, KT);
    const uint8_t* reck = IMG ? rec6<HD>(a.img, 1, a.S * H, nt, pair, 0) : nullptr;      // K / V tile images of this pair
    const uint8_t* recv = IMG ? rec6<HD>(a.img, 2, a.S * H, nt, pair, 0) : nullptr;
    if (IMG) {
        dma_image6<HD, 4>(Ki, reck, wv, lane);
        dma_image6<HD, 4>(Vi, recv, wv, lane);
        dma_fence6();
    } else {
        stage6_load<HD>(base + E, ld, 0, B, tid, rk);
        stage6_load<HD>(base + 2 * E, ld, 0, B, tid, rv);
        stage6_store<HD>(Ki, tid, rk, 1.f);
        stage6_store<HD>(Vi, tid, rv, 1.f);
    }
    if (DROP && tid < KT) htab[tid] = rlt_col_hash(ps, (uint32_t)tid);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        if (!IMG && t + 1 < nt) {
            stage6_load<HD>(base + E, ld, (t + 1) * KT, B, tid, rk);
            stage6_load<HD>(base + 2 * E, ld, (t + 1) * KT, B, tid, rv);
        }
        if (wave_live) {
            f32x16 sc[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[sub][r] = 0.f;
                sc[sub] = mma_rows6<HD>(Ki, sub, l31, hh, qf, sc[sub]);            // S^T[key][q], log2 domain
            }
            if ((t + 1) * KT > B) {               // keys beyond B exist in the last tile only (their rows are zero-filled)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * KT + sub * 32 + acc_row(r, hh) >= B) sc[sub][r] = -INFINITY;
            }
            // lazy rescaling (see attention.hip): the reference m_run moves only when a score exceeds it by more than 8, so the
            // common tile has no cross-lane step and no rescale of O and l
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
            if (DROP) {                          // dropout acts on the normalised probabilities: the normaliser keeps all keys
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        sc[sub][r] = rlt_keep_rc(hq, htab[sub * 32 + acc_row(r, hh)], a.drop_thr) ? sc[sub][r] * inv_keep : 0.f;
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) mma_cols6<HD>(Vi, sub, lane, sc[sub], oacc);   // O^T[d][q] += V^T P^T
        }
        __syncthreads();                         // every wavefront is done with the tile
        if (t + 1 < nt) {
            if (IMG) {
                dma_image6<HD, 4>(Ki, reck + (size_t)(t + 1) * img6_bytes<HD>(), wv, lane);
                dma_image6<HD, 4>(Vi, recv + (size_t)(t + 1) * img6_bytes<HD>(), wv, lane);
                dma_fence6();
            } else {
                stage6_store<HD>(Ki, tid, rk, 1.f);
                stage6_store<HD>(Vi, tid, rv, 1.f);
            }
            if (DROP && tid < KT) htab[tid] = rlt_col_hash(ps, (uint32_t)((t + 1) * KT + tid));
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

// ------------------------------------------------------------------------------------------ forward, ping-pong form
// A SIMD's vector issue port blocks at its head (tools/micro/mfma_valu_overlap.hip): the MFMA bursts of one wavefront and the
// vector work of its SIMD partner ADD unless (a) the partner really is in a vector phase while this one bursts and (b) the
// bursting wave steps back from the port for a few cycles after each MFMA.  Two independent 256-thread workgroups per CU give
// neither.  This form runs ONE 512-thread workgroup per CU (256 queries) as two groups of four wavefronts that share the K / V
// tile images and execute the same four segments per tile, group B one segment behind group A, a barrier after every segment:
//     X: the 48 MFMAs of S^T = K Q^T            (matrix)        Y: softmax + dropout + the 3-way split of P   (vector)
//     Z: the 48 MFMAs of O^T += V^T P^T         (matrix)        W: split + LDS store of the group's half of tile t+1 (vector)
//   slot:   4t     4t+1   4t+2   4t+3   4t+4
//   A:      X(t)   Y(t)   Z(t)   W(t)   X(t+1)       (A stages K, into the other image buffer)
//   B:      W(t-1) X(t)   Y(t)   Z(t)   W(t)         (B stages V)
// so every slot pairs a matrix segment with a vector segment on each SIMD; `s_nop` behind every MFMA of X and Z (RLT_A6_PP_PAD).
// Measured (4096 x 60 positions, s_memtime stamps of a -DRLT_PP_STAMPS build, tools/pp_stamps.py): the four slots of a tile take
// ~1830 / 2110 / 1950-2200 / 2100-2500 cycles + 200-600 at each barrier, 9,700 per tile against 2 x 5,550 for two tiles of the
// two-workgroup form: 5.55 -> 5.25 ms.  The slots are now as long as their VECTOR segments: the ~610 vector instructions per
// wavefront and tile (softmax 130, split of P 176, exponentials 32, staging split + addresses 150, ...) do not fit the ~5 issue
// slots that each of the partner's 96 MFMAs leaves - the kernel is vector-issue-bound, which is what is left to shorten.
#ifndef RLT_A6_PP_PAD
#define RLT_A6_PP_PAD 1
#endif
__device__ __forceinline__ void pp_pad(f32x16& c) {
    if (RLT_A6_PP_PAD >= 0) asm volatile("s_nop %1" : "+v"(c) : "n"(RLT_A6_PP_PAD >= 0 ? RLT_A6_PP_PAD : 0));
}
__device__ __forceinline__ f32x16 mfma6_pp(const Frag3& a, const Frag3& b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, c, 0, 0, 0); pp_pad(c);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, c, 0, 0, 0); pp_pad(c);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, c, 0, 0, 0); pp_pad(c);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, c, 0, 0, 0); pp_pad(c);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, c, 0, 0, 0); pp_pad(c);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, c, 0, 0, 0); pp_pad(c);
    return c;
}
constexpr int QT_PP = 256;     // queries per ping-pong workgroup
#ifdef RLT_PP_STAMPS
// diagnostic build only: s_memtime at every segment boundary of workgroup 0, wavefronts 0 (group A) and 4 (group B), tiles 8..15
__device__ unsigned long long pp_stamps[2 * 8 * 8];
#define PP_STAMP(k) do { if (blockIdx.x == 0 && (wv == 0 || wv == 4) && lane == 0 && t >= 8 && t < 16) \
    pp_stamps[((wv >> 2) * 8 + (t - 8)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PP_STAMP(k) do { } while (0)
#endif

template <int HD, bool DROP, bool IMG>
__global__ __launch_bounds__(512, 1) void attn6_fwd_pp_kernel(AttnArgs a) {
    constexpr int DT = (HD + 31) / 32, IMG6 = img6<HD>(), LDR = ldr6<HD>(), PL = plane6<HD>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* Ki = reinterpret_cast<uint16_t*>(smem);          // [2] K tile images
    uint16_t* Vi = Ki + 2 * IMG6;                               // [2] V tile images
    uint32_t* htab = reinterpret_cast<uint32_t*>(Vi + 2 * IMG6);   // [2][KT] column hashes of the tile's keys (DROP)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int grp = __builtin_amdgcn_readfirstlane(wv >> 2), tid2 = tid & 255;
    const int B = a.B, H = a.H, E = H * HD;
    const size_t ld = (size_t)3 * E;
    const int ntile = rlt_cdiv_dev(B, QT_PP);
    int pair, qt;
    map_block(blockIdx.x, a.S * H, ntile, pair, qt);
    const int s = pair / H, h = pair % H;
    const float* base = a.qkv + (size_t)s * B * ld + h * HD;
    const int q = qt * QT_PP + wv * 32 + l31;
    const bool wave_live = qt * QT_PP + wv * 32 < B;
    const int qc = min(q, B - 1);
    // second launch behind the pipelined forward kernel of attention6h.hip: only the workgroups it flagged (same grid, same block -> rows map)
    if (a.redo && a.redo[blockIdx.x] == 0u) return;

    Frag3 qf[HD / 16];
    row_frags6<HD>(base + (size_t)qc * ld, hh, a.scale * LOG2E, qf);
    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const uint32_t ps = DROP ? pair_seed(a.seed, pair) : 0u;
    const uint32_t hq = DROP ? rlt_row_hash(ps, (uint32_t)q) : 0u;
    const float inv_keep = DROP ? 1.f / (1.f - a.drop_p) : 1.f;

    const int nt = rlt_cdiv_dev(B, KT);
    const float* src = base + (grp == 0 ? E : 2 * E);          // group A stages K, group B stages V
    uint16_t* mine = grp == 0 ? Ki : Vi;
    // IMG: the group's tile images go global -> LDS by LDS-DMA, two tiles ahead (image t + 2 is requested in W(t), when its buffer -
    // that of tile t - has been read by both groups, and awaited in W(t + 1)): the W segment has no vector work left
    const uint8_t* rec = IMG ? rec6<HD>(a.img, grp == 0 ? 1 : 2, a.S * H, nt, pair, 0) : nullptr;
    const int wv4 = wv & 3;
    Stage6<HD> st;
    if (IMG) {
        dma_image6<HD, 4>(mine, rec, wv4, lane);
        if (nt > 1) dma_image6<HD, 4>(mine + IMG6, rec + img6_bytes<HD>(), wv4, lane);
        dma_fence6();
    } else {
        stage6_load<HD>(src, ld, 0, B, tid2, st);
        stage6_store<HD>(mine, tid2, st, 1.f);
        if (nt > 1) stage6_load<HD>(src, ld, KT, B, tid2, st);
    }
    if (DROP && tid < KT) htab[tid] = rlt_col_hash(ps, (uint32_t)tid);
    __syncthreads();

    f32x16 sc[2];
    Frag3 pf[2][2];                                              // P of the tile, split: [sub][s] (rows 16 s .. 16 s + 15 of the sub-tile)
    // the matrix segments read the fragments of MFMA group g + 1 BEFORE the six MFMAs of group g (a scheduling fence keeps hipcc
    // from sinking the reads to their use): with the vector work in the partner's shadow, what is left of a slot is the matrix
    // segment itself, and an exposed LDS latency per group made it 2000-2100 cycles for 1536 of MFMAs (s_memtime stamps)
    auto seg_x = [&](int t) {
        const uint16_t* img = Ki + (t & 1) * IMG6;
        auto frag = [&](int g) {
            constexpr int KS_ = HD / 16;                                                    // g = sub * KS_ + ks
            const int off = ((g / KS_) * 32 + l31) * LDR + 8 * hh + 16 * (g % KS_);
            Frag3 f;
            f.h = *reinterpret_cast<const bf16x8*>(img + off);
            f.m = *reinterpret_cast<const bf16x8*>(img + PL + off);
            f.l = *reinterpret_cast<const bf16x8*>(img + 2 * PL + off);
            return f;
        };
        constexpr int NG = 2 * (HD / 16);
        Frag3 fr[2];
        fr[0] = frag(0);
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            constexpr int KS = HD / 16;
            if (g + 1 < NG) fr[(g + 1) & 1] = frag(g + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (g % KS == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            }
            acc = mfma6_pp(fr[g & 1], qf[g % KS], acc);
            if (g % KS == KS - 1) sc[g / KS] = acc;
        }
    };
    auto seg_y = [&](int t) {
        if ((t + 1) * KT > B) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * KT + sub * 32 + acc_row(r, hh) >= B) sc[sub][r] = -INFINITY;
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sc[sub][r]);
        if (__any(t == 0 || tmax > m_run + 8.f)) {              // lazy rescaling (attention.hip)
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
        // transcendental (and packed-fp32, and integer-multiply) instructions do NOT run beside a partner's MFMAs - one of them
        // in eight serialises the whole stream (tools/micro/mfma_valu_overlap.hip) - while plain fp32 / integer / convert
        // instructions do: the 32 exponentials are issued as one burst between scheduling fences, the subtractions before and the
        // sum and the three-way split after them stay overlappable (this file is compiled without SLP packing)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[sub][r] -= m_run;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[sub][r] = rlt_exp2(sc[sub][r]);
        __builtin_amdgcn_sched_barrier(0);
        float psum = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) psum += sc[sub][r];
        l_run += psum;
        if (DROP) {
            const uint32_t* ht = htab + (t & 1) * KT;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sc[sub][r] = rlt_keep_rc(hq, ht[sub * 32 + acc_row(r, hh)], a.drop_thr) ? sc[sub][r] * inv_keep : 0.f;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const float x[8] = {sc[sub][8 * s2 + 0], sc[sub][8 * s2 + 1], sc[sub][8 * s2 + 2], sc[sub][8 * s2 + 3],
                                    sc[sub][8 * s2 + 4], sc[sub][8 * s2 + 5], sc[sub][8 * s2 + 6], sc[sub][8 * s2 + 7]};
                pf[sub][s2] = split8x3(x);
            }
    };
    auto seg_z = [&](int t) {
        const uint16_t* img = Vi + (t & 1) * IMG6;
        auto frag = [&](int g) {                                 // g = (sub * 2 + s2) * DT + dt
            const int dt = g % DT, s2 = (g / DT) & 1, sub = g / (2 * DT);
            const int row = sub * 32 + 16 * s2 + 4 * hh + ((lane & 15) >> 2);
            const int off = row * LDR + 32 * dt + (HD >= 32 ? 16 * ((lane >> 4) & 1) : 0) + 4 * (lane & 3);
            Frag3 f;
            f.h = cat_frag6(tr_read6(img + off), tr_read6(img + off + 8 * LDR));
            f.m = cat_frag6(tr_read6(img + PL + off), tr_read6(img + PL + off + 8 * LDR));
            f.l = cat_frag6(tr_read6(img + 2 * PL + off), tr_read6(img + 2 * PL + off + 8 * LDR));
            return f;
        };
        constexpr int NG = 4 * DT;
        Frag3 fr[2];
        fr[0] = frag(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) fr[(g + 1) & 1] = frag(g + 1);
            __builtin_amdgcn_sched_barrier(0);
            oacc[g % DT] = mfma6_pp(fr[g & 1], pf[g / (2 * DT)][(g / DT) & 1], oacc[g % DT]);
        }
    };
    auto seg_w = [&](int t) {
        if (t + 1 < nt) {
            if (IMG) {
                dma_fence6();                                    // image t + 1 (requested in W(t - 1) or the prologue) has landed
                if (t + 2 < nt) dma_image6<HD, 4>(mine + (t & 1) * IMG6, rec + (size_t)(t + 2) * img6_bytes<HD>(), wv4, lane);
            } else {
                stage6_store<HD>(mine + ((t + 1) & 1) * IMG6, tid2, st, 1.f);
                if (t + 2 < nt) stage6_load<HD>(src, ld, (t + 2) * KT, B, tid2, st);
            }
            if (DROP && tid < KT) htab[((t + 1) & 1) * KT + tid] = rlt_col_hash(ps, (uint32_t)((t + 1) * KT + tid));
        }
    };
    // every wavefront executes 4 nt + 1 barriers
    if (grp == 1) __syncthreads();
    for (int t = 0; t < nt; ++t) {
        PP_STAMP(0);
        if (wave_live) seg_x(t);
        PP_STAMP(1);
        __syncthreads();
        PP_STAMP(2);
        if (wave_live) seg_y(t);
        PP_STAMP(3);
        __syncthreads();
        PP_STAMP(4);
        if (wave_live) seg_z(t);
        PP_STAMP(5);
        __syncthreads();
        PP_STAMP(6);
        seg_w(t);
        PP_STAMP(7);
        __syncthreads();
    }
    if (grp == 0) __syncthreads();
    if (!wave_live) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (q < B) {
        store_acc_T<HD>(a.o + ((size_t)s * B + q) * E + h * HD, hh, oacc, 1.f / l_tot);
        if (hh == 0) a.lse_o[((size_t)s * H + h) * B + q] = (m_run + log2f(l_tot)) * LN2;
    }
}

// ------------------------------------------------------------------------------------------ dK, dV
template <int HD, bool DROP, bool IMG>
__global__ __launch_bounds__(256, 2) void attn6_bwd_dkv_kernel(AttnArgs a) {
    constexpr int DT = (HD + 31) / 32, IMG6 = img6<HD>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* Qi = reinterpret_cast<uint16_t*>(smem);          // Q tile image (unscaled: the scale sits in the K fragments)
    uint16_t* Di = Qi + IMG6;                                   // dO tile image
    float* Ls = reinterpret_cast<float*>(Di + IMG6);            // [KT] lse * log2e
    float* Es = Ls + KT;                                        // [KT] delta
    uint32_t* htab = reinterpret_cast<uint32_t*>(Es + KT);      // [KT] row hashes of the tile's queries (DROP)
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

    Frag3 kf[HD / 16], vf[HD / 16];
    row_frags6<HD>(base + (size_t)kc * ld + E, hh, a.scale * LOG2E, kf);
    row_frags6<HD>(base + (size_t)kc * ld + 2 * E, hh, 1.f, vf);

    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }

    const uint32_t ps = DROP ? pair_seed(a.seed, pair) : 0u;
    const uint32_t hk = DROP ? rlt_col_hash(ps, (uint32_t)key) : 0u;
    const float inv_keep = DROP ? 1.f / (1.f - a.drop_p) : 1.f;
    Stage6<HD> rq, rd;
    float rl = 0.f, re = 0.f;
    const int nt = rlt_cdiv_dev(B, KT);
    auto load_small = [&](int row0) {
        if (tid < KT) {
            const int qi = row0 + tid, qc = min(qi, B - 1);
            const float l = lsebase[qc], e = delbase[qc];
            rl = qi < B ? l * LOG2E : INFINITY;
            re = qi < B ? e : 0.f;
        }
    };
    const uint8_t* recq = IMG ? rec6<HD>(a.img, 0, a.S * H, nt, pair, 0) : nullptr;       // Q / dO tile images of this pair
    const uint8_t* recd = IMG ? rec6<HD>(a.dimg, 0, a.S * H, nt, pair, 0) : nullptr;
    if (IMG) {
        dma_image6<HD, 4>(Qi, recq, wv, lane);
        dma_image6<HD, 4>(Di, recd, wv, lane);
    } else {
        stage6_load<HD>(base, ld, 0, B, tid, rq);
        stage6_load<HD>(dobase, (size_t)E, 0, B, tid, rd);
    }
    load_small(0);
    if (IMG) dma_fence6();
    else {
        stage6_store<HD>(Qi, tid, rq, 1.f);
        stage6_store<HD>(Di, tid, rd, 1.f);
    }
    if (tid < KT) { Ls[tid] = rl; Es[tid] = re; }
    if (DROP && tid < KT) htab[tid] = rlt_row_hash(ps, (uint32_t)tid);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        if (!IMG && RLT_A6_DKV_PREFETCH && t + 1 < nt) {
            stage6_load<HD>(base, ld, (t + 1) * KT, B, tid, rq);
            stage6_load<HD>(dobase, (size_t)E, (t + 1) * KT, B, tid, rd);
            load_small((t + 1) * KT);
        }
        if (wave_live) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x16 sc, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
                sc = mma_rows6<HD>(Qi, sub, l31, hh, kf, sc);               // S[q][key] (log2 domain)
                dp = mma_rows6<HD>(Di, sub, l31, hh, vf, dp);               // dP[q][key]
                // (seeding the accumulators with -lse / -delta, as attention3.hip does, saves two subtractions per score but
                // rounds every partial sum at the magnitude of lse: measured 10-25 % more error against fp64 - not here; a
                // last-tile-only branch for the row mask made hipcc duplicate the tile body and spill 86 registers; the mask is
                // in the lse table instead)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = sub * 32 + acc_row(r, hh);
                    // Ls = +inf for a query beyond B, so its weight is 0 without a range test per element; at head dim 64 the
                    // test stays: without it hipcc's schedule of this body spills 200 instead of 70 bytes and runs 2 % slower
                    const bool ok = HD < 64 || t * KT + ql < B;
                    const float p = ok ? rlt_exp2(sc[r] - Ls[ql]) : 0.f;
                    float pd = p, dpr = dp[r];
                    if (DROP) {
                        const float m = rlt_keep_rc(htab[ql], hk, a.drop_thr) ? inv_keep : 0.f;
                        pd = p * m;
                        dpr *= m;
                    }
                    sc[r] = pd;                                          // (dropped) P (feeds dV)
                    dp[r] = p * (dpr - Es[ql]);                          // dS
                }
                mma_cols6<HD>(Di, sub, lane, sc, dv);                        // dV^T[d][key] += dO^T P
                mma_cols6<HD>(Qi, sub, lane, dp, dk);                        // dK^T[d][key] += Q^T dS
            }
        }
        if (!IMG && !RLT_A6_DKV_PREFETCH && t + 1 < nt) {       // no registers held across the tile body; the partner workgroup covers the latency
            stage6_load<HD>(base, ld, (t + 1) * KT, B, tid, rq);
            stage6_load<HD>(dobase, (size_t)E, (t + 1) * KT, B, tid, rd);
            load_small((t + 1) * KT);
        }
        if (IMG && t + 1 < nt) load_small((t + 1) * KT);
        __syncthreads();
        if (t + 1 < nt) {
            if (IMG) {
                dma_image6<HD, 4>(Qi, recq + (size_t)(t + 1) * img6_bytes<HD>(), wv, lane);
                dma_image6<HD, 4>(Di, recd + (size_t)(t + 1) * img6_bytes<HD>(), wv, lane);
                dma_fence6();
            } else {
                stage6_store<HD>(Qi, tid, rq, 1.f);
                stage6_store<HD>(Di, tid, rd, 1.f);
            }
            if (tid < KT) { Ls[tid] = rl; Es[tid] = re; }
            if (DROP && tid < KT) htab[tid] = rlt_row_hash(ps, (uint32_t)((t + 1) * KT + tid));
        }
        __syncthreads();
    }
    if (!wave_live || key >= B) return;
    float* drow = a.dqkv + ((size_t)s * B + key) * ld + h * HD;
    store_acc_T<HD>(drow + E, hh, dk, a.scale);
    store_acc_T<HD>(drow + 2 * E, hh, dv, 1.f);
}

// ------------------------------------------------------------------------------------------ dK, dV: one wavefront per SIMD
// Head dim 64.  The two-workgroup kernel above holds 160 stationary registers (K, V fragments, dK, dV) of the 256 a wavefront
// has at two per SIMD; hipcc spills K fragments into the S chain (scratch reloads and `s_waitcnt vmcnt(0)` between the MFMAs) and
// its tile body is blocks of MFMAs followed by blocks of vector work that the SIMD partner does not absorb.  A bf16 MFMA hides four
// plain vector instructions of the SAME wavefront when they sit right behind it (tools/micro/mfma_split.hip).  So here ONE
// 256-thread workgroup per CU owns 256 keys, a wavefront 64 (two halves of 32), with 512 registers: K (h, m) / V fragments and the
// 8 accumulator blocks stay in registers (the l plane of the K fragments in a wavefront-private LDS block), the Q / dO tile images
// are double-buffered (one barrier per tile), and the tile body is ONE basic block of 64 steps of six MFMAs, each MFMA followed by
// a gap with at most one chunk (~4 vector instructions) of the element-wise work and one LDS read of the next step.  The chunks -
// P = exp2(S - lse) and dS = P (dP - delta) per register, the three-way splits of P and dS in six parts, split + LDS store of the
// next tile (Q, then dO through the same 16 staging registers) - are placed by tools/gen_attn6_body.py (earliest deadline first,
// dependences checked there); `GAP_END` fences keep hipcc from reordering them:
//   phases, block b = (query sub-tile b >> 1, key half b & 1):  X0 | X1 | Y0 | X2 | Y1 | X3 | Y2 | Y3
//   X(b): S (4 steps) and dP (4 steps) of block b into ONE score / dP accumulator pair;  Y(b): dV, dK of block b (8 steps)
constexpr int QT1 = 256;
struct Frag2 { bf16x8 h, m; };
// LDS tile image of the two one-wavefront kernels below (they stage and split their tiles themselves, so the layout is theirs alone):
// [plane][64 rows][64 d] bf16 with 128-byte rows, the 16-byte unit c of row r at c ^ swz1(r).  The padded 144-byte rows of the other kernels
// of this file cost these two 21-27 % of their LDS cycles in bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r06_notes.md:
// the transposed reads of four rows x 64 bytes wrap onto each other at a 36-bank stride); this one is conflict-free for the row fragments
// (`ds_read_b128`: 16-lane groups of rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} at one unit), the transposed fragments
// (`ds_read_b64_tr_b16`: 4 rows x 4 units per 32 lanes) and the staging stores (a row per 16 lanes) - tests/test_a6h_layout.py enumerates it.
// RLT_A6_SWZ=0 (build switch): the padded layout (A/B runs).
#ifndef RLT_A6_SWZ
#define RLT_A6_SWZ 1
#endif
constexpr int LDR1 = RLT_A6_SWZ ? 64 : ldr6<64>(), PL1 = KT * LDR1, IMG1 = 3 * PL1;
__host__ __device__ constexpr int swz1(int row) { return (((row >> 1) & 1) << 2) | ((row >> 3) & 3); }
// element offset of bf16 element `col` (a multiple of 4) of row `row` in a plane
__device__ __forceinline__ int img1_off(int row, int col) {
#if RLT_A6_SWZ
    return row * 64 + ((((col >> 3) ^ swz1(row)) << 3) | (col & 7));
#else
    return row * LDR1 + col;
#endif
}
// prologue staging of a whole tile into that layout (the tile body stages through its generated chunks)
__device__ __forceinline__ void stage1_store(uint16_t* __restrict__ img, int tid, const Stage6<64>& st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        uint2 h, m, l;
        split4x3_6(st.v[i].x, st.v[i].y, st.v[i].z, st.v[i].w, h, m, l);
        const int off = img1_off(idx / 16, 4 * (idx % 16));
        *reinterpret_cast<uint2*>(img + off) = h;
        *reinterpret_cast<uint2*>(img + PL1 + off) = m;
        *reinterpret_cast<uint2*>(img + 2 * PL1 + off) = l;
    }
}
#ifndef RLT_A6_MLAST
#define RLT_A6_MLAST 0     // 1: the plane of the first product is read LAST so that one wait covers a step - measured 1 % slower
#endif
#ifndef RLT_A6_ACCV
#define RLT_A6_ACCV 1      // S / dP products of the one-wavefront kernels as asm MFMAs with VGPR accumulators (0: builtins, A/B switch)
#endif
#ifndef RLT_DKV1_PIN
#define RLT_DKV1_PIN 1
#endif
__device__ __forceinline__ bf16x8 cat2_6(uint2 a, uint2 b) { return __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y)); }
#ifdef RLT_DKV1_STAMPS
// diagnostic build only: s_memtime at every step of tiles 8..11 of one workgroup (tools/bench_kernels.py dkv1_stamps); entries 64 /
// 65: before / behind the barrier
__device__ unsigned long long dkv1_stamps[4 * 4 * 66];
#define DKV1_STAMP(k) do { if (blockIdx.x == 64 && lane == 0 && t >= 8 && t < 12) \
    dkv1_stamps[(wv * 4 + (t - 8)) * 66 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DKV1_STAMP(k) do { } while (0)
#endif
template <bool DROP>
__global__ __launch_bounds__(256, 1) void attn6_bwd_dkv1_kernel(AttnArgs a) {
    constexpr int HD = 64, IMG6 = IMG1, PL = PL1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* img0 = reinterpret_cast<uint16_t*>(smem);              // [2 buffers][Q image | dO image]
    float* tab0 = reinterpret_cast<float*>(img0 + 4 * IMG6);         // [2 buffers][lse * log2e | delta | row hashes][KT]
    uint4* klp = reinterpret_cast<uint4*>(tab0 + 2 * 3 * KT);        // [4 wavefronts][2 key halves][4 k-steps][64 lanes]: K fragments, l plane
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int B = a.B, H = a.H, E = H * HD;
    const size_t ld = (size_t)3 * E;
    const int ntile = rlt_cdiv_dev(B, QT1);
    int pair, ktile;
    map_block(blockIdx.x, a.S * H, ntile, pair, ktile);
    const int s_ = pair / H, h = pair % H;
    const float* base = a.qkv + (size_t)s_ * B * ld + h * HD;
    const float* dobase = a.dout + (size_t)s_ * B * E + h * HD;
    const float* lsebase = a.lse + ((size_t)s_ * H + h) * B;
    const float* delbase = a.delta + ((size_t)s_ * H + h) * B;
    const int key0 = ktile * QT1 + wv * 64 + l31;                    // key of half 0; half 1: + 32

    Frag2 kf[2][4];
    Frag3 vf[2][4];
    uint4* klw = klp + (size_t)wv * 2 * 4 * 64 + lane;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        const int kc = min(key0 + 32 * kh, B - 1);
        Frag3 t3[4];
        row_frags6<HD>(base + (size_t)kc * ld + E, hh, a.scale * LOG2E, t3);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[kh][ks].h = t3[ks].h;
            kf[kh][ks].m = t3[ks].m;
            klw[(kh * 4 + ks) * 64] = __builtin_bit_cast(uint4, t3[ks].l);
        }
        row_frags6<HD>(base + (size_t)kc * ld + 2 * E, hh, 1.f, vf[kh]);
    }
    f32x16 dk[2][2], dv[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[i >> 1][i & 1][r] = 0.f; dv[i >> 1][i & 1][r] = 0.f; }

    const uint32_t ps = DROP ? pair_seed(a.seed, pair) : 0u;
    const uint32_t hk0 = DROP ? rlt_col_hash(ps, (uint32_t)key0) : 0u, hk1 = DROP ? rlt_col_hash(ps, (uint32_t)(key0 + 32)) : 0u;
    const float inv_keep = DROP ? 1.f / (1.f - a.drop_p) : 1.f;
    const int nt = rlt_cdiv_dev(B, KT);
    Stage6<HD> rs;                                                  // ONE staging set: Q of the next tile, then dO of the next tile, ...
    float rl, re;
    const int trow = tid & (KT - 1);
    auto load_small = [&](int row0) {
        const int qi = row0 + trow, qc = min(qi, B - 1);
        const float l = lsebase[qc], e = delbase[qc];
        rl = qi < B ? l * LOG2E : INFINITY;
        re = qi < B ? e : 0.f;
    };
    auto store_small = [&](float* tb, int row0) {               // (the four wavefronts write the same 64 values)
        tb[trow] = rl;
        tb[KT + trow] = re;
        if (DROP) reinterpret_cast<uint32_t*>(tb)[2 * KT + trow] = rlt_row_hash(ps, (uint32_t)(row0 + trow));
    };
    auto load_unit = [&](const float* src, size_t lds_, int row0, int i) {
        const int idx = tid + 256 * i;
        const int row = row0 + idx / (HD / 4), dq_ = idx % (HD / 4);
        // rows beyond B repeat the last row (finite values): their lse entry is +inf, so P = dS = 0 and nothing reaches dK / dV;
        // no select here - it would wait for the load where it is issued.  32-bit element offset from the (uniform) matrix
        // base: B * ld < 2^24 (host-checked for this kernel), one v_mad_u32_u24 instead of a 64-bit multiply per load
        const uint32_t off = __umul24((uint32_t)min(row, B - 1), (uint32_t)lds_) + 4u * (uint32_t)dq_;
        rs.v[i] = *reinterpret_cast<const float4*>(src + off);
    };
    // prologue: tile 0 -> buffer 0; Q of tile 1 -> staging registers, lse / delta of tile 1 -> rl / re
    {
        Stage6<HD> r0;
        stage6_load<HD>(base, ld, 0, B, tid, rs);
        stage6_load<HD>(dobase, (size_t)E, 0, B, tid, r0);
        load_small(0);
        stage1_store(img0, tid, rs);
        stage1_store(img0 + IMG6, tid, r0);
        store_small(tab0, 0);
        const int r1 = min(1, nt - 1) * KT;
        stage6_load<HD>(base, ld, r1, B, tid, rs);
        load_small(r1);
    }
    __syncthreads();

    const uint4* klr_base = klp + (size_t)wv * 2 * 4 * 64 + lane;
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const uint16_t* Qc = img0 + cur * 2 * IMG6;
        const uint16_t* Dc = Qc + IMG6;
        uint16_t* Qn = img0 + (cur ^ 1) * 2 * IMG6;
        uint16_t* Dn = Qn + IMG6;
        const float* Tc = tab0 + cur * 3 * KT;
        float* Tn = tab0 + (cur ^ 1) * 3 * KT;
        const int row_n1 = min(t + 1, nt - 1) * KT, row_n2 = min(t + 2, nt - 1) * KT;

#if RLT_DKV1_PIN
        // (keeps the eight accumulator blocks in AGPRs across the back edge: without it hipcc shares their registers with the
        // S / dP accumulators and moves 64 of them out and back every tile)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("" : "+a"(dk[i >> 1][i & 1]));
            asm volatile("" : "+a"(dv[i >> 1][i & 1]));
        }
#if RLT_A6_ACCV
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("" : "+a"(kf[i >> 2][i & 3].h));
            asm volatile("" : "+a"(kf[i >> 2][i & 3].m));
            asm volatile("" : "+a"(vf[i >> 2][i & 3].h));
        }
#endif
#endif
        f32x16 sc, dp;                               // S / dP accumulators of the block in its X phase
        float pv[16], gv[16];                        // P (dropped) and dS of the block whose splits are under way
        Split6 st[2][2][2];                          // [P | dS][k-step][half]: the split units (state and result)
        Split6 sg[4];                                // staging units
        Frag3 afr[2];                                // A fragments of an X step, read one step ahead
        v4s trv[2][3][2];                            // A fragments of a Y step: [buffer][plane h, m, l][half]
        uint4 klr[2];
        float4 lvr[2], evr[2];
        uint4 hvr[2];
#define GAP_END __builtin_amdgcn_sched_barrier(0)
#define ATTN6_STAMP DKV1_STAMP
// (asking for the finished S / dP accumulators in VGPRs here - asm("" : "+v"(sc)) - makes hipcc copy all 16 registers in one
// burst behind the last MFMA; left to itself it reads each register where its chunk needs it, one v_accvgpr_read per element)
#define ACC_DONE(w) do { } while (0)
        // LDS read k of step j of X(b) / Y(b) into buffer u (issued one step ahead)
        auto rx = [&](int u, int k, int b, int j) __attribute__((always_inline)) {
            const int sub = b >> 1, kh = b & 1;
            const uint16_t* img = j < 4 ? Qc : Dc;
            const int off = img1_off(sub * 32 + l31, 8 * hh + 16 * (j & 3));
            // issue order: the plane the FIRST product takes (m) LAST - LDS reads complete in order, so the one wait in front of
            // that product covers the whole step (issued m, l, h: a wait before each of the first three products)
            const int kk = RLT_A6_MLAST ? (j < 4 ? k : k + 1) : (k < 3 ? 3 - k : 0);
            if (kk == 0) klr[u] = klr_base[(kh * 4 + j) * 64];
            else if (kk == 1) afr[u].h = *reinterpret_cast<const bf16x8*>(img + off);
            else if (kk == 2) afr[u].l = *reinterpret_cast<const bf16x8*>(img + 2 * PL + off);
            else afr[u].m = *reinterpret_cast<const bf16x8*>(img + PL + off);
        };
        auto ry = [&](int u, int k, int b, int j) __attribute__((always_inline)) {
            const int sub = b >> 1, s = j >> 2, which = (j >> 1) & 1, dt = j & 1;
            const uint16_t* img = which ? Qc : Dc;
            const int row = sub * 32 + 16 * s + 4 * hh + ((lane & 15) >> 2) + 8 * (k & 1);
            const int off = img1_off(row, 32 * dt + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
            const int pl = RLT_A6_MLAST ? (k < 2 ? 0 : k < 4 ? 2 : 1) : (k < 2 ? 1 : k < 4 ? 2 : 0);     // h, l, m: the plane of the first product last
            trv[u][pl][k & 1] = tr_read6(img + pl * PL + off);
        };
        auto tl = [&](int b, int c) __attribute__((always_inline)) {
            lvr[c & 1] = *reinterpret_cast<const float4*>(Tc + (b >> 1) * 32 + 8 * c + 4 * hh);
        };
        auto te = [&](int b, int c) __attribute__((always_inline)) {
            evr[c & 1] = *reinterpret_cast<const float4*>(Tc + KT + (b >> 1) * 32 + 8 * c + 4 * hh);
            if (DROP) hvr[c & 1] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint32_t*>(Tc) + 2 * KT + (b >> 1) * 32 + 8 * c + 4 * hh);
        };
        // product i of a step: (a.m, b.m), (a.l, b.h), (a.h, b.l), (a.m, b.h), (a.h, b.m), (a.h, b.h); planes 0 h, 1 m, 2 l
        // The S / dP products as asm statements: accumulator in VGPRs (the element-wise chunks read it there - as a builtin hipcc
        // puts it into AGPRs and every element costs a v_accvgpr_read), the K fragments and the h plane of V in AGPRs ("a": their
        // only uses, pinned at the loop top; with dK / dV that is 224 of the 256).  hipcc does not see an MFMA in the asm and
        // inserts no wait states for it: the accumulator is first read ACC_LAG = 3 gaps (3 MFMAs + their chunks: > 11 wait
        // states) behind the last product (tools/gen_attn6_body.py); a product never follows a vector write of its operands
        // (fragments come from LDS / the setup).
        auto mfma_v = [&](f32x16& c, bf16x8 av, bf16x8 bv, bool zero, bool b_agpr) __attribute__((always_inline)) {
#if !RLT_A6_ACCV
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, zero ? z : c, 0, 0, 0);
            return;
#endif
            if (zero) {
                if (b_agpr) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(av), "a"(bv));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(av), "v"(bv));
            } else {
                if (b_agpr) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "a"(bv));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
            }
        };
        auto mx = [&](int u, int i, int b, int j) __attribute__((always_inline)) {
            const int kh = b & 1;
            const int ap = i == 0 || i == 3 ? 1 : i == 1 ? 2 : 0, bp = i == 0 || i == 4 ? 1 : i == 2 ? 2 : 0;
            const bf16x8 av = ap == 0 ? afr[u].h : ap == 1 ? afr[u].m : afr[u].l;
            if (j < 4) {
                if (bp == 2) mfma_v(sc, av, __builtin_bit_cast(bf16x8, klr[u]), false, false);
                else mfma_v(sc, av, bp == 0 ? kf[kh][j].h : kf[kh][j].m, j == 0 && i == 0, true);
            } else {
                const Frag3& vv = vf[kh][j - 4];
                if (bp == 0) mfma_v(dp, av, vv.h, false, true);
                else mfma_v(dp, av, bp == 1 ? vv.m : vv.l, j == 4 && i == 0, false);
            }
        };
        auto my = [&](int u, int i, int b, int j) __attribute__((always_inline)) {
            const int kh = b & 1, s = j >> 2, which = (j >> 1) & 1, dt = j & 1;
            const int ap = i == 0 || i == 3 ? 1 : i == 1 ? 2 : 0, bp = i == 0 || i == 4 ? 1 : i == 2 ? 2 : 0;
            const bf16x8 av = cat_frag6(trv[u][ap][0], trv[u][ap][1]);
            const Split6& u0 = st[which][s][0];
            const Split6& u1 = st[which][s][1];
            const bf16x8 bv = bp == 0 ? cat2_6(u0.hi, u1.hi) : bp == 1 ? cat2_6(u0.mid, u1.mid) : cat2_6(u0.lo, u1.lo);
            if (which == 0) dv[kh][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, dv[kh][dt], 0, 0, 0);
            else dk[kh][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, dk[kh][dt], 0, 0, 0);
        };
        // register r of block b: P = exp2(S - lse) / dS = P (dP - delta) and the dropout of P
        auto ea = [&](int b, int r) __attribute__((always_inline)) {
            const float4& l4 = lvr[(r >> 2) & 1];
            const float lv = (r & 3) == 0 ? l4.x : (r & 3) == 1 ? l4.y : (r & 3) == 2 ? l4.z : l4.w;
            pv[r] = rlt_exp2(sc[r] - lv);
        };
        auto eb = [&](int b, int r) __attribute__((always_inline)) {
            const float4& e4 = evr[(r >> 2) & 1];
            const float ev = (r & 3) == 0 ? e4.x : (r & 3) == 1 ? e4.y : (r & 3) == 2 ? e4.z : e4.w;
            float dpr = dp[r];
            if (DROP) {
                const uint4& h4 = hvr[(r >> 2) & 1];
                const uint32_t hv = (r & 3) == 0 ? h4.x : (r & 3) == 1 ? h4.y : (r & 3) == 2 ? h4.z : h4.w;
                const float m = rlt_keep_rc(hv, (b & 1) ? hk1 : hk0, a.drop_thr) ? inv_keep : 0.f;
                dpr *= m;
                gv[r] = pv[r] * (dpr - ev);
                pv[r] *= m;
            } else {
                gv[r] = pv[r] * (dpr - ev);
            }
        };
        auto sp = [&](int m, int s, int half, int part) __attribute__((always_inline)) {
            const float* w = m ? gv : pv;
            const int r0 = 8 * s + 4 * half;
            split6_part(st[m][s][half], w[r0], w[r0 + 1], w[r0 + 2], w[r0 + 3], part);
        };
        // staging unit i (rows idx / 16 of the tile): which = 0: Q of tile t + 1 -> image, then load dO of tile t + 1 into the same
        // registers; which = 1: dO of tile t + 1 -> image, then load Q of tile t + 2
        auto stg = [&](int which, int i, int part) __attribute__((always_inline)) {
            split6_part(sg[i], rs.v[i].x, rs.v[i].y, rs.v[i].z, rs.v[i].w, part);
            if (part == 5) {
                uint16_t* img = which ? Dn : Qn;
                const int idx = tid + 256 * i;
                const int off = img1_off(idx / (HD / 4), 4 * (idx % (HD / 4)));
                *reinterpret_cast<uint2*>(img + off) = sg[i].hi;
                *reinterpret_cast<uint2*>(img + PL + off) = sg[i].mid;
                *reinterpret_cast<uint2*>(img + 2 * PL + off) = sg[i].lo;
                if (which == 0) load_unit(dobase, (size_t)E, row_n1, i);
                else load_unit(base, ld, row_n2, i);
            }
        };
        store_small(Tn, row_n1);                     // lse / delta of tile t + 1 (loaded during tile t - 1)
        load_small(row_n2);
        rx(0, 0, 0, 0); rx(0, 1, 0, 0); rx(0, 2, 0, 0); rx(0, 3, 0, 0);
        GAP_END;
        if constexpr (DROP) {
#include "attention6_dkv1_body_drop.inc"
        } else {
#include "attention6_dkv1_body.inc"
        }
#undef GAP_END
#undef ATTN6_STAMP
#undef ACC_DONE
        DKV1_STAMP(64);
        __syncthreads();
        DKV1_STAMP(65);
    }
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        const int key = key0 + 32 * kh;
        if (key < B) {
            float* drow = a.dqkv + ((size_t)s_ * B + key) * ld + h * HD;
            store_acc_T<HD>(drow + E, hh, dk[kh], a.scale);
            store_acc_T<HD>(drow + 2 * E, hh, dv[kh], 1.f);
        }
    }
}

// ------------------------------------------------------------------------------------------ dQ: one wavefront per SIMD
// The same construction for dQ (head dim 64): a workgroup owns 256 queries, a wavefront 64 (two halves of 32) with the Q (h, m) /
// dO fragments, lse, delta and the four dQ accumulator blocks in registers (l plane of the Q fragments in LDS); K / V tile images
// double-buffered; block b = (key sub-tile b >> 1, query half b & 1): X(b) = S^T and dP^T (8 steps), Y(b) = dQ^T += K^T dS^T
// (4 steps: k-step s x d tile); tile body generated by tools/gen_attn6_body.py dq.  Keys beyond B: their K rows are staged as
// zeros, so whatever dS they get multiplies a zero column of K^T - no mask on the scores.
template <bool DROP>
__global__ __launch_bounds__(256, 1) void attn6_bwd_dq1_kernel(AttnArgs a) {
    constexpr int HD = 64, IMG6 = IMG1, PL = PL1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* img0 = reinterpret_cast<uint16_t*>(smem);              // [2 buffers][K image | V image]
    uint32_t* tab0 = reinterpret_cast<uint32_t*>(img0 + 4 * IMG6);   // [2 buffers][KT] column hashes of the tile's keys (DROP)
    uint4* qlp = reinterpret_cast<uint4*>(tab0 + 2 * KT);            // [4 wavefronts][2 query halves][4 k-steps][64 lanes]: Q fragments, l plane
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int B = a.B, H = a.H, E = H * HD;
    const size_t ld = (size_t)3 * E;
    const int ntile = rlt_cdiv_dev(B, QT1);
    int pair, qt;
    map_block(blockIdx.x, a.S * H, ntile, pair, qt);
    const int s_ = pair / H, h = pair % H;
    const float* base = a.qkv + (size_t)s_ * B * ld + h * HD;
    const int q0 = qt * QT1 + wv * 64 + l31;                        // query of half 0; half 1: + 32

    Frag2 qf[2][4];
    Frag3 dof[2][4];
    float lse2[2], del[2];
    uint4* qlw = qlp + (size_t)wv * 2 * 4 * 64 + lane;
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
        const int qc = min(q0 + 32 * qh, B - 1);
        Frag3 t3[4];
        row_frags6<HD>(base + (size_t)qc * ld, hh, a.scale * LOG2E, t3);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[qh][ks].h = t3[ks].h;
            qf[qh][ks].m = t3[ks].m;
            qlw[(qh * 4 + ks) * 64] = __builtin_bit_cast(uint4, t3[ks].l);
        }
        row_frags6<HD>(a.dout + ((size_t)s_ * B + qc) * E + h * HD, hh, 1.f, dof[qh]);
        lse2[qh] = a.lse[((size_t)s_ * H + h) * B + qc] * LOG2E;
        del[qh] = a.delta[((size_t)s_ * H + h) * B + qc];
    }
    f32x16 dq[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i >> 1][i & 1][r] = 0.f;

    const uint32_t ps = DROP ? pair_seed(a.seed, pair) : 0u;
    const uint32_t hq0 = DROP ? rlt_row_hash(ps, (uint32_t)q0) : 0u, hq1 = DROP ? rlt_row_hash(ps, (uint32_t)(q0 + 32)) : 0u;
    const float inv_keep = DROP ? 1.f / (1.f - a.drop_p) : 1.f;
    const int nt = rlt_cdiv_dev(B, KT);
    Stage6<HD> rs;                                                  // ONE staging set: K of the next tile, then V of the next tile, ...
    const int trow = tid & (KT - 1);
    auto load_unit = [&](const float* src, int row0, int i) {
        const int idx = tid + 256 * i;
        const int row = row0 + idx / (HD / 4), dq_ = idx % (HD / 4);
        const uint32_t off = __umul24((uint32_t)min(row, B - 1), (uint32_t)ld) + 4u * (uint32_t)dq_;     // (see attn6_bwd_dkv1_kernel)
        rs.v[i] = *reinterpret_cast<const float4*>(src + off);
    };
    {   // prologue: tile 0 -> buffer 0; K of tile 1 -> staging registers
        Stage6<HD> r0;
        stage6_load<HD>(base + E, ld, 0, B, tid, rs);               // (zero rows beyond B)
        stage6_load<HD>(base + 2 * E, ld, 0, B, tid, r0);
        stage1_store(img0, tid, rs);
        stage1_store(img0 + IMG6, tid, r0);
        if (DROP) tab0[trow] = rlt_col_hash(ps, (uint32_t)trow);
        const int r1 = min(1, nt - 1) * KT;
#pragma unroll
        for (int i = 0; i < 4; ++i) load_unit(base + E, r1, i);
    }
    __syncthreads();

    const uint4* qlr_base = qlp + (size_t)wv * 2 * 4 * 64 + lane;
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const uint16_t* Kc = img0 + cur * 2 * IMG6;
        const uint16_t* Vc = Kc + IMG6;
        uint16_t* Kn = img0 + (cur ^ 1) * 2 * IMG6;
        uint16_t* Vn = Kn + IMG6;
        const uint32_t* Tc = tab0 + cur * KT;
        uint32_t* Tn = tab0 + (cur ^ 1) * KT;
        const int row_n1 = min(t + 1, nt - 1) * KT, row_n2 = min(t + 2, nt - 1) * KT;
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(dq[i >> 1][i & 1]));     // (see attn6_bwd_dkv1_kernel)
#if RLT_A6_ACCV
#pragma unroll
        for (int i = 0; i < 8; ++i) {               // the stationary fragments live in AGPRs (their only uses are "a" operands)
            asm volatile("" : "+a"(qf[i >> 2][i & 3].h));
            asm volatile("" : "+a"(qf[i >> 2][i & 3].m));
            asm volatile("" : "+a"(dof[i >> 2][i & 3].h));
            asm volatile("" : "+a"(dof[i >> 2][i & 3].m));
            asm volatile("" : "+a"(dof[i >> 2][i & 3].l));
        }
#endif
        f32x16 sc, dp;                               // S^T / dP^T accumulators of the block in its X phase
        float pv[16], gv[16];                        // P and dS^T of the block whose element-wise work / splits are under way
        Split6 st[2][2][2];                          // [1][k-step][half]: the split units of dS^T (state and result)
        Split6 sg[4];                                // staging units
        Frag3 afr[2];
        v4s trv[2][3][2];
        uint4 qlr[2];
        uint4 hvr[2];
#define GAP_END __builtin_amdgcn_sched_barrier(0)
#define ATTN6_STAMP DKV1_STAMP
// (asking for the finished S / dP accumulators in VGPRs here - asm("" : "+v"(sc)) - makes hipcc copy all 16 registers in one
// burst behind the last MFMA; left to itself it reads each register where its chunk needs it, one v_accvgpr_read per element)
#define ACC_DONE(w) do { } while (0)
        auto rx = [&](int u, int k, int b, int j) __attribute__((always_inline)) {
            const int sub = b >> 1, qh = b & 1;
            const uint16_t* img = j < 4 ? Kc : Vc;
            const int off = img1_off(sub * 32 + l31, 8 * hh + 16 * (j & 3));
            const int kk = RLT_A6_MLAST ? (j < 4 ? k : k + 1) : (k < 3 ? 3 - k : 0);     // (m last: see attn6_bwd_dkv1_kernel)
            if (kk == 0) qlr[u] = qlr_base[(qh * 4 + j) * 64];
            else if (kk == 1) afr[u].h = *reinterpret_cast<const bf16x8*>(img + off);
            else if (kk == 2) afr[u].l = *reinterpret_cast<const bf16x8*>(img + 2 * PL + off);
            else afr[u].m = *reinterpret_cast<const bf16x8*>(img + PL + off);
        };
        auto ry = [&](int u, int k, int b, int j) __attribute__((always_inline)) {
            const int sub = b >> 1, s = j >> 1, dt = j & 1;
            const int row = sub * 32 + 16 * s + 4 * hh + ((lane & 15) >> 2) + 8 * (k & 1);
            const int off = img1_off(row, 32 * dt + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
            const int pl = RLT_A6_MLAST ? (k < 2 ? 0 : k < 4 ? 2 : 1) : (k < 2 ? 1 : k < 4 ? 2 : 0);
            trv[u][pl][k & 1] = tr_read6(Kc + pl * PL + off);
        };
        auto te = [&](int b, int c) __attribute__((always_inline)) {
            if (DROP) hvr[c & 1] = *reinterpret_cast<const uint4*>(Tc + (b >> 1) * 32 + 8 * c + 4 * hh);
        };
        // The S^T / dP^T products as asm statements: accumulator in VGPRs (the element-wise chunks read it there - as a builtin
        // hipcc puts it into AGPRs and every element costs a v_accvgpr_read), stationary fragments in AGPRs ("a": the only uses of
        // qf / dof, so they live there).  hipcc does not see an MFMA in the asm and inserts no wait states for it: the
        // accumulator is first read RLT_A6_ACC_LAG gaps (>= 3 MFMAs + their chunks: > 11 wait states) behind the last product
        // (tools/gen_attn6_body.py), a product never follows a vector write of its operands (fragments come from LDS / setup).
        auto mfma_v = [&](f32x16& c, bf16x8 av, bf16x8 bv, bool zero, bool b_agpr) __attribute__((always_inline)) {
#if !RLT_A6_ACCV
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, zero ? z : c, 0, 0, 0);
            return;
#endif
            if (zero) {
                if (b_agpr) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(av), "a"(bv));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(av), "v"(bv));
            } else {
                if (b_agpr) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "a"(bv));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
            }
        };
        auto mx = [&](int u, int i, int b, int j) __attribute__((always_inline)) {
            const int qh = b & 1;
            const int ap = i == 0 || i == 3 ? 1 : i == 1 ? 2 : 0, bp = i == 0 || i == 4 ? 1 : i == 2 ? 2 : 0;
            const bf16x8 av = ap == 0 ? afr[u].h : ap == 1 ? afr[u].m : afr[u].l;
            if (j < 4) {
                if (bp == 2) mfma_v(sc, av, __builtin_bit_cast(bf16x8, qlr[u]), false, false);
                else mfma_v(sc, av, bp == 0 ? qf[qh][j].h : qf[qh][j].m, j == 0 && i == 0, true);
            } else {
                const Frag3& vv = dof[qh][j - 4];
                mfma_v(dp, av, bp == 0 ? vv.h : bp == 1 ? vv.m : vv.l, j == 4 && i == 0, true);
            }
        };
        auto my = [&](int u, int i, int b, int j) __attribute__((always_inline)) {
            const int qh = b & 1, s = j >> 1, dt = j & 1;
            const int ap = i == 0 || i == 3 ? 1 : i == 1 ? 2 : 0, bp = i == 0 || i == 4 ? 1 : i == 2 ? 2 : 0;
            const bf16x8 av = cat_frag6(trv[u][ap][0], trv[u][ap][1]);
            const Split6& u0 = st[1][s][0];
            const Split6& u1 = st[1][s][1];
            const bf16x8 bv = bp == 0 ? cat2_6(u0.hi, u1.hi) : bp == 1 ? cat2_6(u0.mid, u1.mid) : cat2_6(u0.lo, u1.lo);
            dq[qh][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, dq[qh][dt], 0, 0, 0);
        };
        auto ea = [&](int b, int r) __attribute__((always_inline)) { pv[r] = rlt_exp2(sc[r] - lse2[b & 1]); };
        auto eb = [&](int b, int r) __attribute__((always_inline)) {
            float dpr = dp[r];
            if (DROP) {
                const uint4& h4 = hvr[(r >> 2) & 1];
                const uint32_t hv = (r & 3) == 0 ? h4.x : (r & 3) == 1 ? h4.y : (r & 3) == 2 ? h4.z : h4.w;
                dpr = rlt_keep_rc((b & 1) ? hq1 : hq0, hv, a.drop_thr) ? dpr * inv_keep : 0.f;
            }
            gv[r] = pv[r] * (dpr - del[b & 1]);
        };
        auto sp = [&](int m, int s, int half, int part) __attribute__((always_inline)) {
            const int r0 = 8 * s + 4 * half;
            split6_part(st[1][s][half], gv[r0], gv[r0 + 1], gv[r0 + 2], gv[r0 + 3], part);
        };
        // staging unit i: which = 0: K of tile t + 1 (rows beyond B as zeros) -> image, then load V of tile t + 1 into the same
        // registers; which = 1: V of tile t + 1 -> image, then load K of tile t + 2
        auto stg = [&](int which, int i, int part) __attribute__((always_inline)) {
            const int idx = tid + 256 * i;
            if (which == 0 && part == 0) {
                const bool ok = row_n1 + idx / (HD / 4) < B;
                rs.v[i] = make_float4(ok ? rs.v[i].x : 0.f, ok ? rs.v[i].y : 0.f, ok ? rs.v[i].z : 0.f, ok ? rs.v[i].w : 0.f);
            }
            split6_part(sg[i], rs.v[i].x, rs.v[i].y, rs.v[i].z, rs.v[i].w, part);
            if (part == 5) {
                uint16_t* img = which ? Vn : Kn;
                const int off = img1_off(idx / (HD / 4), 4 * (idx % (HD / 4)));
                *reinterpret_cast<uint2*>(img + off) = sg[i].hi;
                *reinterpret_cast<uint2*>(img + PL + off) = sg[i].mid;
                *reinterpret_cast<uint2*>(img + 2 * PL + off) = sg[i].lo;
                if (which == 0) load_unit(base + 2 * E, row_n1, i);
                else load_unit(base + E, row_n2, i);
            }
        };
        if (DROP) Tn[trow] = rlt_col_hash(ps, (uint32_t)(row_n1 + trow));
        rx(0, 0, 0, 0); rx(0, 1, 0, 0); rx(0, 2, 0, 0); rx(0, 3, 0, 0);
        GAP_END;
        if constexpr (DROP) {
#include "attention6_dq1_body_drop.inc"
        } else {
#include "attention6_dq1_body.inc"
        }
#undef GAP_END
#undef ATTN6_STAMP
#undef ACC_DONE
        DKV1_STAMP(64);
        __syncthreads();
        DKV1_STAMP(65);
    }
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
        const int q = q0 + 32 * qh;
        if (q < B) store_acc_T<HD>(a.dqkv + ((size_t)s_ * B + q) * ld + h * HD, hh, dq[qh], a.scale);
    }
}

// ------------------------------------------------------------------------------------------ dQ
template <int HD, bool DROP, bool IMG>
__global__ __launch_bounds__(256, 2) void attn6_bwd_dq_kernel(AttnArgs a) {
    constexpr int DT = (HD + 31) / 32, IMG6 = img6<HD>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* Ki = reinterpret_cast<uint16_t*>(smem);
    uint16_t* Vi = Ki + IMG6;
    uint32_t* htab = reinterpret_cast<uint32_t*>(Vi + IMG6);    // [KT] column hashes of the tile's keys (DROP)
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

    Frag3 qf[HD / 16], dof[HD / 16];
    row_frags6<HD>(base + (size_t)qc * ld, hh, a.scale * LOG2E, qf);
    row_frags6<HD>(a.dout + ((size_t)s * B + qc) * E + h * HD, hh, 1.f, dof);
    const float lse2 = a.lse[((size_t)s * H + h) * B + qc] * LOG2E;
    const float del = a.delta[((size_t)s * H + h) * B + qc];

    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

    const uint32_t ps = DROP ? pair_seed(a.seed, pair) : 0u;
    const uint32_t hq = DROP ? rlt_row_hash(ps, (uint32_t)q) : 0u;
    const float inv_keep = DROP ? 1.f / (1.f - a.drop_p) : 1.f;
    Stage6<HD> rk, rv;
    const int nt = rlt_cdiv_dev(B, KT);
    const uint8_t* reck = IMG ? rec6<HD>(a.img, 1, a.S * H, nt, pair, 0) : nullptr;
    const uint8_t* recv = IMG ? rec6<HD>(a.img, 2, a.S * H, nt, pair, 0) : nullptr;
    if (IMG) {
        dma_image6<HD, 4>(Ki, reck, wv, lane);
        dma_image6<HD, 4>(Vi, recv, wv, lane);
        dma_fence6();
    } else {
        stage6_load<HD>(base + E, ld, 0, B, tid, rk);
        stage6_load<HD>(base + 2 * E, ld, 0, B, tid, rv);
        stage6_store<HD>(Ki, tid, rk, 1.f);
        stage6_store<HD>(Vi, tid, rv, 1.f);
    }
    if (DROP && tid < KT) htab[tid] = rlt_col_hash(ps, (uint32_t)tid);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        if (!IMG && t + 1 < nt) {
            stage6_load<HD>(base + E, ld, (t + 1) * KT, B, tid, rk);
            stage6_load<HD>(base + 2 * E, ld, (t + 1) * KT, B, tid, rv);
        }
        if (wave_live) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x16 sc, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
                sc = mma_rows6<HD>(Ki, sub, l31, hh, qf, sc);                // S^T[key][q]
                dp = mma_rows6<HD>(Vi, sub, l31, hh, dof, dp);               // dP^T[key][q]
                if ((t + 1) * KT > B) {                                   // keys beyond B exist in the last tile only
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * KT + sub * 32 + acc_row(r, hh) >= B) sc[r] = -INFINITY;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = rlt_exp2(sc[r] - lse2);
                    float dpr = dp[r];
                    if (DROP) dpr = rlt_keep_rc(hq, htab[sub * 32 + acc_row(r, hh)], a.drop_thr) ? dpr * inv_keep : 0.f;
                    dp[r] = p * (dpr - del);                              // dS^T
                }
                mma_cols6<HD>(Ki, sub, lane, dp, dq);                         // dQ^T[d][q] += K^T dS^T
            }
        }
        __syncthreads();
        if (t + 1 < nt) {
            if (IMG) {
                dma_image6<HD, 4>(Ki, reck + (size_t)(t + 1) * img6_bytes<HD>(), wv, lane);
                dma_image6<HD, 4>(Vi, recv + (size_t)(t + 1) * img6_bytes<HD>(), wv, lane);
                dma_fence6();
            } else {
                stage6_store<HD>(Ki, tid, rk, 1.f);
                stage6_store<HD>(Vi, tid, rv, 1.f);
            }
            if (DROP && tid < KT) htab[tid] = rlt_col_hash(ps, (uint32_t)((t + 1) * KT + tid));
        }
        __syncthreads();
    }
    if (!wave_live || q >= B) return;
    store_acc_T<HD>(a.dqkv + ((size_t)s * B + q) * ld + h * HD, hh, dq, a.scale);
}

}  // namespace

template <int HD, bool DROP, bool IMG>
static int attn6_launch(int which, const AttnArgs& a, hipStream_t st) {
    const int grid = a.S * a.H * rlt_cdiv(a.B, QT);
    const size_t shm = (size_t)2 * img6<HD>() * sizeof(uint16_t) + (which == 1 ? 2 * KT * sizeof(float) : 0) + KT * sizeof(uint32_t);
    int rc;
    static const bool pp = [] { const char* e = getenv("RLT_A6_PP"); return !e || atoi(e) != 0; }();    // RLT_A6_PP=0: the two-workgroup form
    if constexpr (HD == 64) {                 // ping-pong form (instantiated for head dim 64 only): one 512-thread workgroup per CU, 256 queries
        if (which == 0 && (pp || a.redo)) {          // (a.redo: the fix-up launch behind attention6h.hip's forward is this kernel's grid)
            const size_t shm_pp = (size_t)4 * img6<HD>() * sizeof(uint16_t) + 2 * KT * sizeof(uint32_t);
            if ((rc = rlt_allow_lds(attn6_fwd_pp_kernel<HD, DROP, IMG>, shm_pp))) return rc;
            hipLaunchKernelGGL((attn6_fwd_pp_kernel<HD, DROP, IMG>), dim3(a.S * a.H * rlt_cdiv(a.B, QT_PP)), dim3(512), shm_pp, st, a);
            return RLT_LAUNCH_RESULT();
        }
    }
    if constexpr (HD == 64) {                 // dK+dV with one wavefront per SIMD: 256 keys per workgroup, double-buffered tiles
        static const bool dkv1 = [] { const char* e = getenv("RLT_A6_DKV1"); return !e || atoi(e) != 0; }();
        static const bool dq1 = [] { const char* e = getenv("RLT_A6_DQ1"); return !e || atoi(e) != 0; }();
        const bool small24 = (long long)a.B * 3 * a.H * HD < (1ll << 24);     // their loaders form B * ld in 24-bit multiplies
        if (which == 2 && dq1 && small24) {
            const size_t shm1 = (size_t)4 * IMG1 * sizeof(uint16_t) + 2 * KT * sizeof(uint32_t) + (size_t)4 * 2 * 4 * 64 * sizeof(uint4);
            if ((rc = rlt_allow_lds(attn6_bwd_dq1_kernel<DROP>, shm1))) return rc;
            hipLaunchKernelGGL((attn6_bwd_dq1_kernel<DROP>), dim3(a.S * a.H * rlt_cdiv(a.B, QT1)), dim3(256), shm1, st, a);
            return RLT_LAUNCH_RESULT();
        }
        if (which == 1 && dkv1 && small24) {
            const size_t shm1 = (size_t)4 * IMG1 * sizeof(uint16_t) + 2 * 3 * KT * sizeof(float) + (size_t)4 * 2 * 4 * 64 * sizeof(uint4);
            if ((rc = rlt_allow_lds(attn6_bwd_dkv1_kernel<DROP>, shm1))) return rc;
            hipLaunchKernelGGL((attn6_bwd_dkv1_kernel<DROP>), dim3(a.S * a.H * rlt_cdiv(a.B, QT1)), dim3(256), shm1, st, a);
            return RLT_LAUNCH_RESULT();
        }
    }
    if (which == 0) {
        if ((rc = rlt_allow_lds(attn6_fwd_kernel<HD, DROP, IMG>, shm))) return rc;
        hipLaunchKernelGGL((attn6_fwd_kernel<HD, DROP, IMG>), dim3(grid), dim3(256), shm, st, a);
    } else if (which == 1) {
        if ((rc = rlt_allow_lds(attn6_bwd_dkv_kernel<HD, DROP, IMG>, shm))) return rc;
        hipLaunchKernelGGL((attn6_bwd_dkv_kernel<HD, DROP, IMG>), dim3(grid), dim3(256), shm, st, a);
    } else {
        if ((rc = rlt_allow_lds(attn6_bwd_dq_kernel<HD, DROP, IMG>, shm))) return rc;
        hipLaunchKernelGGL((attn6_bwd_dq_kernel<HD, DROP, IMG>), dim3(grid), dim3(256), shm, st, a);
    }
    return RLT_LAUNCH_RESULT();
}
template <int HD>
static int attn6_prepare(int which, const AttnArgs& a, hipStream_t st) {
    const int npair = a.S * a.H, ntile = rlt_cdiv(a.B, KT), E = a.H * HD;
    if (which == 3)        // Q, K, V images from the packed qkv rows (column blocks 0, E, 2E)
        hipLaunchKernelGGL((attn6_prepare_kernel<HD>), dim3(npair * ntile, 3), dim3(256), 0, st, a.qkv, (size_t)3 * E, 0, E, a.S, a.B, a.H,
                           (uint8_t*)const_cast<void*>(a.img));
    else                   // dO images
        hipLaunchKernelGGL((attn6_prepare_kernel<HD>), dim3(npair * ntile, 1), dim3(256), 0, st, a.dout, (size_t)E, 0, 0, a.S, a.B, a.H,
                           (uint8_t*)const_cast<void*>(a.dimg));
    return RLT_LAUNCH_RESULT();
}
template <int HD, bool DROP>
static int attn6_dispatch(int which, const AttnArgs& a, hipStream_t st) {
    if (which >= 3) return attn6_prepare<HD>(which, a, st);
    const bool img = a.img != nullptr && (which != 1 || a.dimg != nullptr);
    return img ? attn6_launch<HD, DROP, true>(which, a, st) : attn6_launch<HD, DROP, false>(which, a, st);
}

size_t rlt_attn6_images_bytes(int S, int B, int H, int HD, int nmat) {
    const size_t per = HD == 64 ? img6_bytes<64>() : HD == 32 ? img6_bytes<32>() : img6_bytes<16>();
    return (size_t)nmat * S * H * rlt_cdiv(B, KT) * per;
}

#ifdef RLT_DKV1_STAMPS
extern "C" int rlt_debug_dkv1_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dkv1_stamps), sizeof(unsigned long long) * 4 * 4 * 66);
}
#endif
#ifdef RLT_PP_STAMPS
extern "C" int rlt_debug_pp_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pp_stamps), sizeof(unsigned long long) * 128);
}
#endif

// which: 0 forward, 1 dK/dV, 2 dQ, 3 / 4 the prepare passes; head dim 16, 32 or 64 (the caller checks).  Dropout is a template
// parameter: hipcc if-converts a run-time `drop_p > 0` test and executes the hashes regardless.
int rlt_attn6_run(int which, const AttnArgs& a, int HD, hipStream_t st) {
    const bool drop = a.drop_p > 0.f;
    if (HD == 64) return drop ? attn6_dispatch<64, true>(which, a, st) : attn6_dispatch<64, false>(which, a, st);
    if (HD == 32) return drop ? attn6_dispatch<32, true>(which, a, st) : attn6_dispatch<32, false>(which, a, st);
    // head dim 16: the 16x16x32 kernels of attention6n.hip (RLT_A6N=0: the 32x32x16 kernels of this file, A/B runs; the
    // pre-split-image staging exists in this file only)
    static const bool a6n = [] { const char* e = getenv("RLT_A6N"); return !e || atoi(e) != 0; }();
    if (a6n && which < 3 && !(a.img != nullptr && (which != 1 || a.dimg != nullptr))) return rlt_attn6n_run(which, a, st);
    return drop ? attn6_dispatch<16, true>(which, a, st) : attn6_dispatch<16, false>(which, a, st);
}

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
#ifndef RLT_ASM_DMA
#define RLT_ASM_DMA 1         // head dim 64: LDS-DMA issued as inline assembly (see dma_copy)
#endif
#ifndef RLT_ASM_DMA_HD16
#define RLT_ASM_DMA_HD16 0    // also at head dims 16 / 32
#endif
#ifndef RLT_SPREAD_HD16
#define RLT_SPREAD_HD16 0     // head dims 16 / 32 (needs RLT_ASM_DMA_HD16): dQ and dK+dV kernels spread the pieces over the tile body
#endif
#ifndef RLT_FWD_TRV
#define RLT_FWD_TRV 1         // forward kernel, head dim 64: V^T by transposed reads from the V rows image (no transposed image of V)
#endif
#ifndef RLT_DQ_STEPPED
#define RLT_DQ_STEPPED 1      // dQ kernel, head dim 64, no dropout: stepped tile body
#endif
#ifndef RLT_DQ_TRREAD
#define RLT_DQ_TRREAD 1       // ... with K^T read transposed from the K rows image (K^T image not copied)
#endif
#ifndef RLT_DQ_SPREAD_EVERY
#define RLT_DQ_SPREAD_EVERY 4 // ... one LDS-DMA piece every so many of its 24 matrix steps (5 pieces per wavefront)
#endif
#ifndef RLT_DKV_TRREAD
#define RLT_DKV_TRREAD 1      // stepped dK+dV body: transposed operands by ds_read_b64_tr_b16 from the rows images (no transposed images copied)
#endif
#ifndef RLT_STEPPED_SPREAD
#define RLT_STEPPED_SPREAD 1  // stepped tile body: LDS-DMA pieces of the next tile spread over the matrix steps
#endif
#ifndef RLT_FWD_SPREAD
#define RLT_FWD_SPREAD 0      // forward kernel, head dim 64: LDS-DMA pieces spread over the tile body (measured +2 %: off)
#endif
#ifndef RLT_DQ_SPREAD
#define RLT_DQ_SPREAD 1       // dQ kernel likewise (measured -2 %)
#endif
#ifndef RLT_SPREAD_EVERY
#define RLT_SPREAD_EVERY 6    // one piece every so many matrix steps (5 pieces per wavefront and tile with RLT_DKV_TRREAD, 32 steps)
#endif
#ifndef RLT_STEPPED
#define RLT_STEPPED 1         // dK+dV, head dim 64: stepped tile body with this fragment prefetch distance (matrix steps; 1 measured best: 4.63 vs 4.69 ms at 2); 0 = compiler-scheduled body
#endif
#ifndef RLT_HD16_SMALL_MFMA
#define RLT_HD16_SMALL_MFMA 1        // head dim 16: dV / dK products on v_mfma_f32_16x16x32_bf16 (0: the padded 32x32x16 form)
#endif
constexpr float LAZY_TH = 8.f;           // forward: the running-maximum reference moves when a tile exceeds it by > 2^8
constexpr int LDT3 = 72;                 // bf16 elements per row of a transposed [d][64 rows] image (144 B)
// rows of a transposed image: the HD rows of the tile, plus - where the 32-row MFMA operand has rows to spare (head
// dim 16) - one row of ones: the product that forms O^T = V^T P^T then also delivers sum_k P[k][q] in its row HD, i.e.
// the softmax normaliser, and the forward kernel keeps no running sum of its own (32 adds per 64 keys less)
template <int HD> constexpr int T_rows() { return HD < 32 ? HD + 1 : HD; }

__device__ __forceinline__ uint32_t pk2(float a, float b) {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    v2 t = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ void split4(float a, float b, float c, float d, uint2& hi, uint2& lo) {
    hi.x = pk2(a, b);
    hi.y = pk2(c, d);
    // opaque to the optimiser: otherwise hipcc re-derives the low element's bf16 with a second v_cvt_pk (x, 0) instead
    // of shifting the packed pair (one extra VALU instruction per two elements)
    asm("" : "+v"(hi.x), "+v"(hi.y));
    lo.x = pk2(a - __builtin_bit_cast(float, hi.x << 16), b - __builtin_bit_cast(float, hi.x & 0xffff0000u));
    lo.y = pk2(c - __builtin_bit_cast(float, hi.y << 16), d - __builtin_bit_cast(float, hi.y & 0xffff0000u));
}
__device__ __forceinline__ bf16x8 as_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of 16-bit elements, delivered column-major (lane i of the
// group gets column i of the 4 rows); every lane's address 8-byte aligned, EXEC all ones (cdna_hip_programming.md T10)
typedef short v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4s tr_read(const uint16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p));
}
__device__ __forceinline__ bf16x8 cat_frag(v4s a, v4s b) {
    typedef short v8s __attribute__((ext_vector_type(8)));
    const v8s v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

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
// the same for images consumed by v_mfma_f32_16x16x32_bf16 (head dim 16, dK+dV kernel): after the row exchange of mma_T16
// k-group g = lane >> 4 of the B operand holds rows {0,16,4,20}[g] + (j&3) + 8(j>>2) of a 32-row block in element j, so
// row q = b4 b3 b2 b1 b0 sits at position 16 b2 + 8 b4 + 4 b3 + b1b0 of the block
__device__ __forceinline__ constexpr int kpos16(int key) {
    return (key & ~28) | ((key & 4) << 2) | ((key & 16) >> 1) | ((key & 8) >> 1);
}

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
__device__ __forceinline__ void stage_store_T(uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, int tid, const Stage<HD>& st,
                                              bool perm16 = false) {
    const int rb = tid / (HD / 4), dq = tid % (HD / 4);
    if (!stage_active<HD>(tid)) return;
    const float* f0 = reinterpret_cast<const float*>(&st.v[0]);
    const float* f1 = reinterpret_cast<const float*>(&st.v[1]);
    const float* f2 = reinterpret_cast<const float*>(&st.v[2]);
    const float* f3 = reinterpret_cast<const float*>(&st.v[3]);
    const int kp = perm16 ? kpos16(4 * rb) : kpos(4 * rb);
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
            // head dim 16: lanes 16..31 supply A rows d >= HD.  They read the row of ones (clamped address): output
            // row d depends on A row d only, and store_acc_T never writes rows >= HD, so no zeroing (a divergent branch
            // and 8 moves per 8 scores when it was there)
            const int d = dt * 32 + l31;
            const int off = min(d, T_rows<HD>() - 1) * LDT3 + sub * 32 + 16 * s + 8 * hh;
            const uint4 ah = *reinterpret_cast<const uint4*>(thi + off);
            const uint4 al = *reinterpret_cast<const uint4*>(tlo + off);
            acc[dt] = mfma3(as_frag(ah), as_frag(al), wh, wl, acc[dt]);
        }
    }
}

// The same product with the A operand read TRANSPOSED from a rows image [row][HD + 8] (ds_read_b64_tr_b16): the 16-lane
// group (lane >> 4) covers d = 32 dt + 16 (group & 1) + (lane & 15); the k slots of lane half hh are rows 16 s + 4 hh + {0..3}
// and + 8 of the 32-row sub-tile - the rows registers 8s..8s+7 of w hold.  No transposed image of the matrix is needed.
template <int HD>
__device__ __forceinline__ void mma_Ttr(const uint16_t* __restrict__ rhi, const uint16_t* __restrict__ rlo, int sub, int lane,
                                        const f32x16& w, f32x16 (&acc)[(HD + 31) / 32]) {
    constexpr int DT = (HD + 31) / 32;
    const int hh = lane >> 5;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float x[8] = {w[8 * s + 0], w[8 * s + 1], w[8 * s + 2], w[8 * s + 3], w[8 * s + 4], w[8 * s + 5], w[8 * s + 6], w[8 * s + 7]};
        bf16x8 wh, wl;
        split8(x, wh, wl);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int row = sub * 32 + 16 * s + 4 * hh + ((lane & 15) >> 2);
            const int off = row * (HD + 8) + 32 * dt + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
            const bf16x8 ah = cat_frag(tr_read(rhi + off), tr_read(rhi + off + 8 * (HD + 8)));
            const bf16x8 al = cat_frag(tr_read(rlo + off), tr_read(rlo + off + 8 * (HD + 8)));
            acc[dt] = mfma3(ah, al, wh, wl, acc[dt]);
        }
    }
}

// Head dim 16: acc[half] (D[row = d][col = key of the 16-key half]) += transposed-image tile (A, 16 rows d, the 32 rows of
// sub-tile `sub` in kpos16 order) x the accumulator registers w of a 32x32x16 product (B, rows of w = k) on
// v_mfma_f32_16x16x32_bf16: M = 16 is the head dimension exactly, where the 32x32x16 form multiplies 16 rows of padding.
// w has its column (key) on the lane and rows 8s..8s+7 (+4 for the upper lane half) in registers; the 16x16x32 B operand
// wants, per 16-key half, all 32 rows spread over the four 16-lane groups: one v_permlane16_swap per packed register
// pair moves registers 8..15 of lanes 0-15 / 32-47 to lanes 16-31 / 48-63 and registers 0..7 the other way.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mma_T16(const uint16_t* __restrict__ thi, const uint16_t* __restrict__ tlo, int sub, int lane,
                                        const f32x16& w, f32x4v (&acc)[2], f32x4v* ksum = nullptr) {
    const float x[8] = {w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]};
    const float y[8] = {w[8], w[9], w[10], w[11], w[12], w[13], w[14], w[15]};
    bf16x8 xh, xl, yh, yl;
    split8(x, xh, xl);
    split8(y, yh, yl);
    uint4 XH = __builtin_bit_cast(uint4, xh), XL = __builtin_bit_cast(uint4, xl);
    uint4 YH = __builtin_bit_cast(uint4, yh), YL = __builtin_bit_cast(uint4, yl);
    uint32_t* xhp = reinterpret_cast<uint32_t*>(&XH); uint32_t* xlp = reinterpret_cast<uint32_t*>(&XL);
    uint32_t* yhp = reinterpret_cast<uint32_t*>(&YH); uint32_t* ylp = reinterpret_cast<uint32_t*>(&YL);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        auto rh = __builtin_amdgcn_permlane16_swap(xhp[i], yhp[i], false, false);
        auto rl = __builtin_amdgcn_permlane16_swap(xlp[i], ylp[i], false, false);
        xhp[i] = rh[0]; yhp[i] = rh[1];
        xlp[i] = rl[0]; ylp[i] = rl[1];
    }
    const int off = (lane & 15) * LDT3 + sub * 32 + 8 * (lane >> 4);
    const bf16x8 ah = as_frag(*reinterpret_cast<const uint4*>(thi + off));
    const bf16x8 al = as_frag(*reinterpret_cast<const uint4*>(tlo + off));
    const bf16x8 b0h = as_frag(XH), b0l = as_frag(XL), b1h = as_frag(YH), b1l = as_frag(YL);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, b0h, acc[0], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b0l, acc[0], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b0h, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, b1h, acc[1], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b1l, acc[1], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b1h, acc[1], 0, 0, 0);
    if (ksum) {          // every row of ksum[half] += sum over the 32 k entries of w (hi + lo), per column: the softmax normaliser
        const bf16x8 ones = as_frag(make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u));
        ksum[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, b0l, ksum[0], 0, 0, 0);
        ksum[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, b0h, ksum[0], 0, 0, 0);
        ksum[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, b1l, ksum[1], 0, 0, 0);
        ksum[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, b1h, ksum[1], 0, 0, 0);
    }
}

template <int HD> constexpr int rows_elems() { return KT * (HD + 8); }     // one hi or lo rows image
template <int HD> constexpr int T_elems() { return T_rows<HD>() * LDT3; }   // one hi or lo transposed image

// ---- pre-split tile images in HBM -----------------------------------------------------------------------
// A one-time "prepare" pass splits each 64-row tile of Q, K, V (and dO) into bf16 hi/lo ONCE and writes it in
// exactly the LDS image layout; the attention kernels then stage tiles with LDS-DMA (global_load_lds, no
// registers, no VALU).  Without it every workgroup re-splits the same K/V tiles (32x redundant at B = 4096).
// Record of one (matrix, position*head pair, 64-row tile):
//   [ rows image hi | rows image lo | pad ] RP bytes   [ transposed image hi | lo | pad ] TP bytes   [ aux 1 KiB ]
// aux (dO records only): lse*log2e [64 floats], delta [64 floats].
constexpr int rup1k(int b) { return (b + 1023) / 1024 * 1024; }
template <int HD> struct Rec {
    static constexpr int RP = rup1k(2 * rows_elems<HD>() * 2);
    static constexpr int TP = rup1k(2 * T_elems<HD>() * 2);
    static constexpr int AUX = 1024;
    static constexpr int BYTES = RP + TP + AUX;
};
constexpr int QT3 = 256;                  // rows owned by a workgroup of the split-bf16 kernels (8 wavefronts x 32)

// LDS-DMA copy of NBYTES (multiple of 1 KiB) global -> LDS by the 8 wavefronts of the workgroup.
// ASM: issued as inline assembly.  For a global_load_lds it can see, hipcc's wait-count insertion treats every later
// ds_read of the workgroup's LDS object as possibly aliasing the copy in flight: it waits with lgkmcnt(0) instead of a
// counted lgkmcnt(N) after EVERY group of fragment reads, and it may put an s_waitcnt vmcnt(0) in the middle of the tile
// body, i.e. wait there for the NEXT tile's copy.  The copy of tile t+1 goes to the other stage, so the only ordering
// needed is dma_wait_barrier<true>() before that stage is read.  Measured at head dim 64 (r02_notes.md): dK+dV kernel -2 %,
// forward kernel -1.5 %, dQ kernel -1 % (with frags_ready() below; without it the forward kernel got 5 % slower).
template <int HD> constexpr bool asm_dma() { return RLT_ASM_DMA != 0 && (HD == 64 || RLT_ASM_DMA_HD16 != 0); }
template <int HD> constexpr bool spread_dma() { return asm_dma<HD>() && (HD == 64 || RLT_SPREAD_HD16 != 0); }

template <int NBYTES, bool ASM = false>
__device__ __forceinline__ void dma_copy(uint8_t* lds_dst, const uint8_t* __restrict__ gsrc, int wv, int lane) {
    static_assert(NBYTES % 1024 == 0, "LDS-DMA pieces are 1 KiB per wavefront instruction");
#pragma unroll
    for (int c = 0; c < (NBYTES / 1024 + 7) / 8; ++c) {
        const int chunk = wv + 8 * c;
        if (chunk < NBYTES / 1024) {
            if constexpr (ASM) {
                const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(lds_dst + chunk * 1024);
                const uint8_t* src = gsrc + chunk * 1024 + lane * 16;
                RLT_DMA_ASM(dst, src);
            } else {
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gsrc + chunk * 1024 + lane * 16),
                                                 (void __attribute__((address_space(3)))*)(lds_dst + chunk * 1024), 16, 0, 0);
            }
        }
    }
}
// one piece (`c`-th 1 KiB chunk of this wavefront) of dma_copy, inline-assembly form: for tile bodies that spread the copy
// of the next tile over their matrix steps
template <int NBYTES>
__device__ __forceinline__ void dma_piece(uint8_t* lds_dst, const uint8_t* __restrict__ gsrc, int wv, int lane, int c) {
    const int chunk = wv + 8 * c;
    if (chunk < NBYTES / 1024) {
        const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(lds_dst + chunk * 1024);
        const uint8_t* src = gsrc + chunk * 1024 + lane * 16;
        RLT_DMA_ASM(dst, src);
    }
}

// branch-free form: the chunk index is clamped (a wavefront beyond the end re-copies the last chunk: the same bytes to the
// same place), so that a tile body that issues pieces stays ONE basic block (sched_group_barrier patterns need that)
template <int NBYTES>
__device__ __forceinline__ void dma_piece_clamped(uint8_t* lds_dst, const uint8_t* __restrict__ gsrc, int wv, int lane, int c) {
    const int chunk = min(wv + 8 * c, NBYTES / 1024 - 1);
    const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(lds_dst + chunk * 1024);
    const uint8_t* src = gsrc + chunk * 1024 + lane * 16;
    RLT_DMA_ASM(dst, src);
}

// Register fragments fetched from HBM before the tile loop must have ARRIVED before the loop: an empty asm statement
// that consumes them makes hipcc wait here.  Otherwise it places the s_waitcnt vmcnt(N) at their first use inside the
// loop, with N counting only the loads it knows of - the inline-assembly LDS-DMA pieces are not among them, so from
// the second tile on that wait stalls the tile body until the NEXT tile's copy has landed.
template <int N>
__device__ __forceinline__ void frags_ready(const bf16x8 (&f)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" :: "v"(f[i]));
}
// every wavefront waits for its own pieces, then the workgroup barrier makes all pieces visible to all wavefronts
template <bool ASM = false>
__device__ __forceinline__ void dma_wait_barrier() {
    if constexpr (ASM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// fragments of one row of a rows image straight from HBM (no VALU): frag[ks] covers d = 16ks + 8h + j
template <int HD>
__device__ __forceinline__ void image_row_frags(const uint8_t* __restrict__ rec, int row, int hh, bf16x8 (&fh)[HD / 16], bf16x8 (&fl)[HD / 16]) {
    const uint16_t* hi = reinterpret_cast<const uint16_t*>(rec);
    const uint16_t* lo = hi + rows_elems<HD>();
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
        fh[ks] = as_frag(*reinterpret_cast<const uint4*>(hi + row * (HD + 8) + 16 * ks + 8 * hh));
        fl[ks] = as_frag(*reinterpret_cast<const uint4*>(lo + row * (HD + 8) + 16 * ks + 8 * hh));
    }
}

struct PrepArgs {
    const float* src[3]; size_t ld; float mul[3];      // up to 3 matrices (row-major tiles of HD columns), scale factors
    const float* lse; const float* delta;              // optional aux (dO records)
    const float* o; float* delta_out;                  // dO records: delta = rowsum(dO * O) is computed here (and written out)
    uint8_t* out;                                       // records: [matrix][pair][tile]
    int S, B, H, nmat;
    int perm16[3];                                      // transposed image of matrix m in kpos16 order (head dim 16: Q, dO)
    int skip_t[3];                                      // matrix m: no transposed image (its bytes of the record stay unwritten)
};

template <int HD>
__global__ __launch_bounds__(256) void attn3_prepare_kernel(PrepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint8_t* rec = reinterpret_cast<uint8_t*>(smem);
    uint16_t* r_hi = reinterpret_cast<uint16_t*>(rec);
    uint16_t* r_lo = r_hi + rows_elems<HD>();
    uint16_t* t_hi = reinterpret_cast<uint16_t*>(rec + Rec<HD>::RP);
    uint16_t* t_lo = t_hi + T_elems<HD>();
    float* aux = reinterpret_cast<float*>(rec + Rec<HD>::RP + Rec<HD>::TP);
    const int tid = threadIdx.x;
    const int nt = rlt_cdiv_dev(a.B, KT);
    const int pair = blockIdx.x / nt, tile = blockIdx.x % nt, m = blockIdx.y;
    const int s = pair / a.H, h = pair % a.H;
    // zero the pads so that the records are deterministic
    for (int i = tid; i < Rec<HD>::BYTES / 16; i += 256) reinterpret_cast<uint4*>(rec)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const float* base = a.src[m] + (size_t)s * a.B * a.ld + h * HD;
    Stage<HD> st;
    stage_load<HD>(base, a.ld, tile * KT, a.B, tid, st);
    if (a.o) {
        // delta[q] = sum_d dO[q][d] * O[q][d] of this head, from the tile already in registers (a separate pass over dO and O
        // did this before): the HD/4 threads of a 4-row block reduce their partial dot products by shuffles
        Stage<HD> so;
        stage_load<HD>(a.o + (size_t)s * a.B * a.ld + h * HD, a.ld, tile * KT, a.B, tid, so);
        float part[4] = {0.f, 0.f, 0.f, 0.f};
        if (stage_active<HD>(tid)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                part[i] = st.v[i].x * so.v[i].x + st.v[i].y * so.v[i].y + st.v[i].z * so.v[i].z + st.v[i].w * so.v[i].w;
        }
#pragma unroll
        for (int off = 1; off < HD / 4; off <<= 1)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[i] += __shfl_xor(part[i], off, 64);
        if (stage_active<HD>(tid) && tid % (HD / 4) == 0) {
            const int rb = tid / (HD / 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = tile * KT + 4 * rb + i;
                aux[KT + 4 * rb + i] = q < a.B ? -part[i] : 0.f;
                if (q < a.B) a.delta_out[((size_t)s * a.H + h) * a.B + q] = part[i];
            }
        }
    }
    if (stage_active<HD>(tid)) {
        const float mul = a.mul[m];
#pragma unroll
        for (int i = 0; i < 4; ++i) { st.v[i].x *= mul; st.v[i].y *= mul; st.v[i].z *= mul; st.v[i].w *= mul; }
    }
    stage_store_rows<HD>(r_hi, r_lo, tid, st);
    const bool skip_t = a.skip_t[m] != 0;
    if (!skip_t) stage_store_T<HD>(t_hi, t_lo, tid, st, a.perm16[m] != 0);
    if (T_rows<HD>() > HD && tid < KT) t_hi[HD * LDT3 + tid] = 0x3F80;      // bf16 1.0 (lo plane stays 0)
    if (a.lse && tid < KT) {
        const int q = tile * KT + tid, qc = min(q, a.B - 1);
        const float l = a.lse[((size_t)s * a.H + h) * a.B + qc];
        aux[tid] = q < a.B ? -l * LOG2E : 0.f;          // negated: both seed MFMA accumulators (S - lse, dP - delta)
        if (!a.o) aux[KT + tid] = q < a.B ? -a.delta[((size_t)s * a.H + h) * a.B + qc] : 0.f;
    }
    __syncthreads();
    uint4* dst = reinterpret_cast<uint4*>(a.out + (((size_t)m * a.S * a.H + pair) * nt + tile) * Rec<HD>::BYTES);
    for (int i = tid; i < Rec<HD>::BYTES / 16; i += 256)
        if (!skip_t || i < Rec<HD>::RP / 16 || i >= (Rec<HD>::RP + Rec<HD>::TP) / 16) dst[i] = reinterpret_cast<const uint4*>(rec)[i];
}

#if defined(RLT_STAMPS)
// timeline instrumentation (variant builds only): s_memtime at tile start / before the tile barrier, per wavefront, for
// the first 16 tiles of one workgroup; read back with rlt_debug_stamps()
__device__ unsigned long long rlt_stamp_buf[8 * 16 * 2];
#define RLT_STAMP(slot) do { if (blockIdx.x == gridDim.x / 2 && t < 16 && lane == 0) \
    rlt_stamp_buf[(wv * 16 + t) * 2 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define RLT_STAMP(slot) do {} while (0)
#endif

struct Attn3Args {
    AttnArgs a;
    const uint8_t* img;       // Q | K | V records
    const uint8_t* dimg;      // dO records (backward)
};
template <int HD>
__device__ __forceinline__ const uint8_t* record(const uint8_t* img, int m, int npair, int nt, int pair, int tile) {
    return img + (((size_t)m * npair + pair) * nt + tile) * Rec<HD>::BYTES;
}

// ------------------------------------------------------------------------------------------ forward
template <int HD, bool DROP>
__global__ __launch_bounds__(512, 2) void attn3_fwd_kernel(Attn3Args g) {
    const AttnArgs& a = g.a;
    constexpr int DT = (HD + 31) / 32;
    constexpr int STAGE = Rec<HD>::RP + Rec<HD>::TP;                        // K rows pair | V transposed pair
    // TRV (head dim 64): P.V takes V^T by transposed reads from the V ROWS image, which is copied in place of the transposed
    // image (same size at head dim 64); the prepare pass then writes no transposed image of V either
    constexpr bool TRV = RLT_FWD_TRV != 0 && HD == 64;
    static_assert(!TRV || Rec<HD>::RP <= Rec<HD>::TP, "the V rows image must fit the stage slot of the transposed image");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint8_t* lds = reinterpret_cast<uint8_t*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hh = lane >> 5;
    const int B = a.B, H = a.H, E = H * HD, npair = a.S * H;
    const int nt = rlt_cdiv_dev(B, KT);
    const int ntile = rlt_cdiv_dev(B, QT3);
    int pair, qt;
    map_block(blockIdx.x, npair, ntile, pair, qt);
    const int s = pair / H, h = pair % H;
    const int q = qt * QT3 + wv * 32 + l31;
    const bool wave_live = qt * QT3 + wv * 32 < B;

    bf16x8 qh[HD / 16], ql[HD / 16];
    {
        const int qtile = min(q >> 6, nt - 1);
        image_row_frags<HD>(record<HD>(g.img, 0, npair, nt, pair, qtile), q & 63, hh, qh, ql);   // Q pre-scaled by scale*log2e
        frags_ready(qh); frags_ready(ql);
    }
    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = 0.f, l_run = 0.f;          // m_run: the reference the scores are taken relative to (set by the first tile)
    f32x16 c_m;
#pragma unroll
    for (int r = 0; r < 16; ++r) c_m[r] = 0.f;
    // head dim 16: P.V on v_mfma_f32_16x16x32_bf16 (mma_T16), accumulators and normaliser per 16-query half
    constexpr bool S16 = HD == 16 && RLT_HD16_SMALL_MFMA;
    // padded 32x32x16 form at head dim 16 without dropout: the normaliser is row HD of O^T (T_rows above; with dropout
    // P.V runs on the dropped P); 16x16x32 form without dropout: a ones-operand product sums P (LMFMA)
    constexpr bool LROW = T_rows<HD>() > HD && !DROP && !S16;
    constexpr bool LMFMA = S16 && !DROP;
    f32x4v o16[2], l16[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r) { o16[hf][r] = 0.f; l16[hf][r] = 0.f; }

    // dropout: hash of this lane's query once; hashes of a tile's keys in LDS, written while the tile is in flight
    uint32_t* htab = reinterpret_cast<uint32_t*>(lds + 2 * STAGE);
    const uint32_t ps = DROP ? pair_seed(a.seed, pair) : 0u;
    const uint32_t hq = DROP ? rlt_row_hash(ps, (uint32_t)q) : 0u;
    const float inv_keep = DROP ? 1.f / (1.f - a.drop_p) : 1.f;
    auto issue = [&](int t, int buf) {
        dma_copy<Rec<HD>::RP, asm_dma<HD>()>(lds + buf * STAGE, record<HD>(g.img, 1, npair, nt, pair, t), wv, lane);
        if (TRV) dma_copy<Rec<HD>::RP, asm_dma<HD>()>(lds + buf * STAGE + Rec<HD>::RP, record<HD>(g.img, 2, npair, nt, pair, t), wv, lane);
        else dma_copy<Rec<HD>::TP, asm_dma<HD>()>(lds + buf * STAGE + Rec<HD>::RP, record<HD>(g.img, 2, npair, nt, pair, t) + Rec<HD>::RP, wv, lane);
        if (DROP && tid < KT) htab[buf * KT + tid] = rlt_col_hash(ps, (uint32_t)(t * KT + tid));
    };
    issue(0, 0);
    dma_wait_barrier<asm_dma<HD>()>();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        // head dim 64: the 4-5 LDS-DMA pieces of a wavefront are spread over the tile body instead of all 8 wavefronts
        // issuing everything before their first MFMA (see the dK+dV kernel); a wavefront without queries issues at the top
        constexpr bool SPREAD = RLT_FWD_SPREAD != 0 && HD == 64 && asm_dma<HD>() && !TRV;
        const bool more = t + 1 < nt;
        if (!SPREAD || !wave_live) { if (more) issue(t + 1, buf ^ 1); }
        else if (DROP && more && tid < KT) htab[(buf ^ 1) * KT + tid] = rlt_col_hash(ps, (uint32_t)((t + 1) * KT + tid));
        const uint8_t* nk = record<HD>(g.img, 1, npair, nt, pair, min(t + 1, nt - 1));
        const uint8_t* nv = record<HD>(g.img, 2, npair, nt, pair, min(t + 1, nt - 1)) + Rec<HD>::RP;
        uint8_t* nl = lds + (buf ^ 1) * STAGE;
        auto piece = [&](int pc) {                    // 0..2: K rows image, 3..5: V transposed image
            if (!SPREAD || !more) return;
            if (pc < 3) dma_piece<Rec<HD>::RP>(nl, nk, wv, lane, pc);
            else dma_piece<Rec<HD>::TP>(nl + Rec<HD>::RP, nv, wv, lane, pc - 3);
        };
        if (wave_live) {
            piece(0);
            const uint16_t* k_hi = reinterpret_cast<const uint16_t*>(lds + buf * STAGE);
            const uint16_t* k_lo = k_hi + rows_elems<HD>();
            const uint16_t* v_hi = reinterpret_cast<const uint16_t*>(lds + buf * STAGE + Rec<HD>::RP);
            const uint16_t* v_lo = v_hi + (TRV ? rows_elems<HD>() : T_elems<HD>());
            f32x16 sc[2];
            // Lazily rescaled running maximum: the score products start from accumulators holding -m_ref (a 16-register
            // block, the MFMA's C operand), so they come out as s - m_ref with no subtraction per score, and m_ref moves -
            // with the rescale of O and of the normaliser - only when a tile's maximum exceeds it by more than 2^LAZY_TH
            // (first tile: always).  exp2(s - m_ref) <= 2^LAZY_TH then; O, the normaliser and the LSE carry the same
            // reference, so the result is the same softmax.  Per tile this saves 32 subtractions and 16*DT multiplies.
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) sc[sub] = mma_rows<HD>(k_hi, k_lo, sub, l31, hh, qh, ql, c_m);   // S^T[key][q] - m_ref
            piece(1);
            if (t == nt - 1) {            // only the last tile can hold keys beyond B
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
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const bool move = t == 0 || tmax > LAZY_TH;
            if (__any(move)) {            // wave-uniform branch; lanes that do not move shift by 0
                const float shift = move ? tmax : 0.f;
                const float alpha = rlt_exp2(-shift);
                if (S16) {        // the 16x16 accumulators of a lane belong to queries (lane & 15) and 16 + (lane & 15)
                    const float a0 = __shfl(alpha, lane & 15, 64), a1 = __shfl(alpha, 16 + (lane & 15), 64);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { o16[0][r] *= a0; o16[1][r] *= a1; l16[0][r] *= a0; l16[1][r] *= a1; }
                } else {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
                }
                l_run *= alpha;
                m_run += shift;
#pragma unroll
                for (int r = 0; r < 16; ++r) c_m[r] = -m_run;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[sub][r] -= shift;
            }
            piece(2);
            // per 32-key sub-tile: exponentiate, (drop), feed P.V - the second sub-tile's VALU work is issued while
            // the first sub-tile's MFMAs execute
            float psum = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = rlt_exp2(sc[sub][r]);
                    sc[sub][r] = p;
                    if (!LROW && !LMFMA) psum += p;
                }
                if (DROP) {
                    const uint4* hk4 = reinterpret_cast<const uint4*>(htab + buf * KT + sub * 32 + 4 * hh);
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {                       // registers 4g..4g+3 = keys 8g+4hh+{0..3}
                        const uint4 hk = hk4[2 * gq];
                        sc[sub][4 * gq + 0] = rlt_keep_rc(hq, hk.x, a.drop_thr) ? sc[sub][4 * gq + 0] * inv_keep : 0.f;
                        sc[sub][4 * gq + 1] = rlt_keep_rc(hq, hk.y, a.drop_thr) ? sc[sub][4 * gq + 1] * inv_keep : 0.f;
                        sc[sub][4 * gq + 2] = rlt_keep_rc(hq, hk.z, a.drop_thr) ? sc[sub][4 * gq + 2] * inv_keep : 0.f;
                        sc[sub][4 * gq + 3] = rlt_keep_rc(hq, hk.w, a.drop_thr) ? sc[sub][4 * gq + 3] * inv_keep : 0.f;
                    }
                }
                piece(3 + 2 * sub);
                if (S16) mma_T16(v_hi, v_lo, sub, lane, sc[sub], o16, LMFMA ? l16 : nullptr);   // O^T[d][q] (+ sum of P)
                else if (TRV) mma_Ttr<HD>(v_hi, v_lo, sub, lane, sc[sub], oacc);                // O^T[d][q], V^T read transposed
                else mma_T<HD>(v_hi, v_lo, sub, l31, hh, sc[sub], oacc);                        // O^T[d][q]
                if (sub == 0) piece(4);
            }
            if (!LROW && !LMFMA) l_run += psum;
        }
        dma_wait_barrier<asm_dma<HD>()>();
    }
    if (!wave_live) return;
    if (S16) {
        // normaliser of this lane's own query (l31) and of the two queries its 16x16 accumulators belong to
        const float l_own = LMFMA ? (l31 < 16 ? l16[0][0] : l16[1][0]) : l_run + __shfl_xor(l_run, 32, 64);
        const float lq[2] = {LMFMA ? l16[0][0] : __shfl(l_own, lane & 15, 64), LMFMA ? l16[1][0] : __shfl(l_own, 16 + (lane & 15), 64)};
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int qq = qt * QT3 + wv * 32 + 16 * hf + (lane & 15);
            if (qq < B) {
                const float inv = 1.f / lq[hf];
                *reinterpret_cast<float4*>(a.o + ((size_t)s * B + qq) * E + h * HD + 4 * (lane >> 4)) =
                    make_float4(o16[hf][0] * inv, o16[hf][1] * inv, o16[hf][2] * inv, o16[hf][3] * inv);
            }
        }
        if (q < B && hh == 0) a.lse_o[((size_t)s * H + h) * B + q] = (m_run + log2f(l_own)) * LN2;
        return;
    }
    const float l_tot = LROW ? oacc[0][8] : l_run + __shfl_xor(l_run, 32, 64);   // register 8 = row 16 (20 for the upper lane half)
    if (q < B) {
        store_acc_T<HD>(a.o + ((size_t)s * B + q) * E + h * HD, hh, oacc, 1.f / l_tot);
        if (hh == 0) a.lse_o[((size_t)s * H + h) * B + q] = (m_run + log2f(l_tot)) * LN2;
    }
}

// ------------------------------------------------------------------------------------------ dQ
template <int HD, bool DROP>
__global__ __launch_bounds__(512, 2) void attn3_bwd_dq_kernel(Attn3Args g) {
    const AttnArgs& a = g.a;
    constexpr int DT = (HD + 31) / 32;
    constexpr int KREC = Rec<HD>::RP + Rec<HD>::TP;                         // K rows pair | K transposed pair
    constexpr int STAGE = KREC + Rec<HD>::RP;                               // ... | V rows pair
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint8_t* lds = reinterpret_cast<uint8_t*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hh = lane >> 5;
    const int B = a.B, H = a.H, E = H * HD, npair = a.S * H;
    const size_t ld = (size_t)3 * E;
    const int nt = rlt_cdiv_dev(B, KT);
    const int ntile = rlt_cdiv_dev(B, QT3);
    int pair, qt;
    map_block(blockIdx.x, npair, ntile, pair, qt);
    const int s = pair / H, h = pair % H;
    const int q = qt * QT3 + wv * 32 + l31;
    const bool wave_live = qt * QT3 + wv * 32 < B;
    const int qc = min(q, B - 1);
    const uint32_t ps = pair_seed(a.seed, pair);
    const float inv_keep = DROP ? 1.f / (1.f - a.drop_p) : 1.f;

    bf16x8 qh[HD / 16], ql[HD / 16], doh[HD / 16], dol[HD / 16];
    const int qtile = min(q >> 6, nt - 1);
    image_row_frags<HD>(record<HD>(g.img, 0, npair, nt, pair, qtile), q & 63, hh, qh, ql);
    image_row_frags<HD>(record<HD>(g.dimg, 0, npair, nt, pair, qtile), q & 63, hh, doh, dol);
    const float lse2 = a.lse[((size_t)s * H + h) * B + qc] * LOG2E;
    const float del = a.delta[((size_t)s * H + h) * B + qc];
    frags_ready(qh); frags_ready(ql); frags_ready(doh); frags_ready(dol);
    asm volatile("" :: "v"(lse2), "v"(del));

    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    constexpr bool S16 = HD == 16 && RLT_HD16_SMALL_MFMA;         // dQ^T on v_mfma_f32_16x16x32_bf16 (mma_T16)
    f32x4v dq16[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r) dq16[hf][r] = 0.f;
    // the S and dP products start from accumulators holding -lse and -delta of this lane's query (two register
    // blocks kept for the whole kernel: the MFMA reads them as its C operand), so they come out as S - lse and
    // dP - delta without a subtraction per element.  With dropout delta is subtracted after the mask.  Where the 16
    // extra registers would push the kernel past the 128 that let two workgroups share a CU (head dim 32, and head dim
    // 16 with dropout: 116 -> 132 registers, 73 -> 92 ms per Choopy launch) the plain form stays
    constexpr bool SEED = HD > 32 || (HD == 16 && !DROP);
    f32x16 c_lse, c_del;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c_lse[r] = SEED ? -lse2 : 0.f; c_del[r] = DROP ? 0.f : -del; }
    uint32_t* htab = reinterpret_cast<uint32_t*>(lds + 2 * STAGE);          // dropout: per-key hashes of the tile
    const uint32_t hq = DROP ? rlt_row_hash(ps, (uint32_t)q) : 0u;
    // head dim 64: the 7-8 LDS-DMA pieces of a wavefront spread over the tile body (see the dK+dV kernel)
    constexpr bool SPREAD = RLT_DQ_SPREAD != 0 && spread_dma<HD>();
    auto issue = [&](int t, int buf) {
        dma_copy<KREC, asm_dma<HD>()>(lds + buf * STAGE, record<HD>(g.img, 1, npair, nt, pair, t), wv, lane);
        dma_copy<Rec<HD>::RP, asm_dma<HD>()>(lds + buf * STAGE + KREC, record<HD>(g.img, 2, npair, nt, pair, t), wv, lane);
        if (DROP && tid < KT) htab[buf * KT + tid] = rlt_col_hash(ps, (uint32_t)(t * KT + tid));
    };
    // stepped body with transposed reads (head dim 64, no dropout): only the K and V ROWS images are copied - 36 chunks dealt
    // as one list to the 8 wavefronts, 5 slots each (4 re-copy the last chunk), branch-free (see the dK+dV kernel)
    constexpr bool DEAL = RLT_DQ_STEPPED != 0 && RLT_DQ_TRREAD != 0 && HD == 64 && !DROP && SPREAD;
    auto deal_piece = [&](const uint8_t* nk, const uint8_t* nv, uint8_t* nl, int pc) {
        constexpr int NR = Rec<HD>::RP / 1024;
        const int id = min(wv + 8 * pc, 2 * NR - 1);
        const bool isk = id < NR;
        const int ch = isk ? id : id - NR;
        const uint8_t* src = (isk ? nk : nv) + ch * 1024 + lane * 16;
        const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)nl + (isk ? 0 : KREC) + ch * 1024;
        RLT_DMA_ASM(dst, src);
    };
    issue(0, 0);
    dma_wait_barrier<asm_dma<HD>()>();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (DEAL && !wave_live) {                   // a wavefront without queries: its 5 slots of the dealt list, all at once
            if (t + 1 < nt) {
#pragma unroll
                for (int pc = 0; pc < 5; ++pc)
                    deal_piece(record<HD>(g.img, 1, npair, nt, pair, t + 1), record<HD>(g.img, 2, npair, nt, pair, t + 1), lds + (buf ^ 1) * STAGE, pc);
            }
        } else if (!SPREAD || !wave_live) { if (t + 1 < nt) issue(t + 1, buf ^ 1); }
        else if (DROP && t + 1 < nt && tid < KT) htab[(buf ^ 1) * KT + tid] = rlt_col_hash(ps, (uint32_t)((t + 1) * KT + tid));
        if (wave_live) {
            const uint16_t* kr_hi = reinterpret_cast<const uint16_t*>(lds + buf * STAGE);
            const uint16_t* kr_lo = kr_hi + rows_elems<HD>();
            const uint16_t* kt_hi = reinterpret_cast<const uint16_t*>(lds + buf * STAGE + Rec<HD>::RP);
            const uint16_t* kt_lo = kt_hi + T_elems<HD>();
            const uint16_t* vr_hi = reinterpret_cast<const uint16_t*>(lds + buf * STAGE + KREC);
            const uint16_t* vr_lo = vr_hi + rows_elems<HD>();
            const bool more = t + 1 < nt;
            const uint8_t* nk = record<HD>(g.img, 1, npair, nt, pair, min(t + 1, nt - 1));
            const uint8_t* nv = record<HD>(g.img, 2, npair, nt, pair, min(t + 1, nt - 1));
            uint8_t* nl = lds + ((t & 1) ^ 1) * STAGE;
            // stepped body (head dim 64, no dropout; see the dK+dV kernel): K^T by ds_read_b64_tr_b16 from the K rows image
            constexpr bool STEPPED = RLT_DQ_STEPPED != 0 && HD == 64 && !DROP && SPREAD;
            constexpr bool TRREAD = STEPPED && RLT_DQ_TRREAD != 0;
            auto piece = [&](int pc) {                // 0..4: K record (rows + transposed), 5..7: V rows image
                if (!SPREAD || (!more && !TRREAD)) return;      // (with TRREAD the last tile re-copies itself into the idle stage)
                if (TRREAD) {
                    if (pc < 5) deal_piece(nk, nv, nl, pc);
                } else if (pc < 5) dma_piece<KREC>(nl, nk, wv, lane, pc);
                else dma_piece<Rec<HD>::RP>(nl + KREC, nv, wv, lane, pc - 5);
            };
            if constexpr (STEPPED) {
                // 24 matrix steps per tile over the two 32-key sub-tiles a, b:  [S_a dP_a] [S_b dP_b | E_a] [dQ_a | E_b] [dQ_b]
                bf16x8 fh[24], fl[24];
                f32x16 sc2[2], dp2[2];
                bf16x8 gh[2][2], gl[2][2];            // split dS^T: [sub-tile][rows 8s..8s+7 of the block]
                auto frag = [&](int st) {
                    const int k = st & 3;
                    if (st < 16) {
                        const int sub = st >> 3;
                        const uint16_t* hi = (st & 4) ? vr_hi : kr_hi;
                        const uint16_t* lo = (st & 4) ? vr_lo : kr_lo;
                        const int off = (sub * 32 + l31) * (HD + 8) + 8 * hh + 16 * k;
                        fh[st] = *reinterpret_cast<const bf16x8*>(hi + off);
                        fl[st] = *reinterpret_cast<const bf16x8*>(lo + off);
                    } else if (TRREAD) {              // A[d][k slot] = K[key][d], k = 2 s + dt (mapping as in the dK+dV kernel)
                        const int sub = (st - 16) >> 2;
                        const int krow = sub * 32 + 16 * (k >> 1) + 4 * hh + ((lane & 15) >> 2);
                        const int off = krow * (HD + 8) + 32 * (k & 1) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
                        const v4s h0 = tr_read(kr_hi + off), h1 = tr_read(kr_hi + off + 8 * (HD + 8));
                        const v4s l0 = tr_read(kr_lo + off), l1 = tr_read(kr_lo + off + 8 * (HD + 8));
                        fh[st] = cat_frag(h0, h1);
                        fl[st] = cat_frag(l0, l1);
                    } else {
                        const int sub = (st - 16) >> 2;
                        const int off = ((k & 1) * 32 + l31) * LDT3 + sub * 32 + 16 * (k >> 1) + 8 * hh;
                        fh[st] = as_frag(*reinterpret_cast<const uint4*>(kt_hi + off));
                        fl[st] = as_frag(*reinterpret_cast<const uint4*>(kt_lo + off));
                    }
                };
                auto mm = [&](int st) {
                    const int k = st & 3;
                    if (st % RLT_DQ_SPREAD_EVERY == 0) piece(st / RLT_DQ_SPREAD_EVERY);
                    if (st + 2 < 24) frag(st + 2);
                    if (st < 16) {
                        const int sub = st >> 3;
                        if (st & 4) dp2[sub] = mfma3(fh[st], fl[st], doh[k], dol[k], k == 0 ? c_del : dp2[sub]);    // dP^T[key][q] - delta
                        else sc2[sub] = mfma3(fh[st], fl[st], qh[k], ql[k], k == 0 ? c_lse : sc2[sub]);             // S^T[key][q] - lse
                    } else {
                        const int sub = (st - 16) >> 2;
                        dq[k & 1] = mfma3(fh[st], fl[st], gh[sub][k >> 1], gl[sub][k >> 1], dq[k & 1]);             // dQ^T[d][q] += K^T dS^T
                    }
                };
                // element-wise work as fixed per-step units behind the MFMAs (see the dK+dV kernel):
                //   4..7 exp_a | 8..11 mul_a | 12..15 split dS_a + exp_b | 16 mul_b | 17..19 mul_b + split dS_b | 20 split dS_b
                // No mask for keys beyond B in the last tile: their K rows are zero (prepare pass), so whatever finite dS^T they
                // get is multiplied by zero in K^T dS^T.
                uint2 qh_[2][2][2], ql_[2][2][2];     // [sub-tile][half][quarter] -> hi / lo of 4 values of dS^T
                auto unit_exp = [&](int sub, int c) {
#pragma unroll
                    for (int r = 4 * c; r < 4 * c + 4; ++r) sc2[sub][r] = rlt_exp2(sc2[sub][r]);
                };
                auto unit_mul = [&](int sub, int c) {
#pragma unroll
                    for (int r = 4 * c; r < 4 * c + 4; ++r) dp2[sub][r] = sc2[sub][r] * dp2[sub][r];
                };
                auto unit_split = [&](int sub, int qd) {
                    const f32x16& w = dp2[sub];
                    split4(w[4 * qd], w[4 * qd + 1], w[4 * qd + 2], w[4 * qd + 3], qh_[sub][qd >> 1][qd & 1], ql_[sub][qd >> 1][qd & 1]);
                };
                auto operand = [&](int sub, int half) {
                    gh[sub][half] = as_frag(make_uint4(qh_[sub][half][0].x, qh_[sub][half][0].y, qh_[sub][half][1].x, qh_[sub][half][1].y));
                    gl[sub][half] = as_frag(make_uint4(ql_[sub][half][0].x, ql_[sub][half][0].y, ql_[sub][half][1].x, ql_[sub][half][1].y));
                };
                auto pattern = [] {
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                    __builtin_amdgcn_sched_barrier(0);
                };
                frag(0); frag(1);
#pragma unroll
                for (int st = 0; st < 4; ++st) { mm(st); pattern(); }                                   // S_a
#pragma unroll
                for (int st = 4; st < 8; ++st) { mm(st); unit_exp(0, st - 4); pattern(); }              // dP_a
#pragma unroll
                for (int st = 8; st < 12; ++st) { mm(st); unit_mul(0, st - 8); pattern(); }             // S_b
#pragma unroll
                for (int st = 12; st < 16; ++st) { mm(st); unit_split(0, st - 12); unit_exp(1, st - 12); pattern(); }   // dP_b
                operand(0, 0); operand(0, 1);
#pragma unroll
                for (int st = 16; st < 20; ++st) {                                                      // dQ_a
                    mm(st); unit_mul(1, st - 16);
                    if (st > 16) unit_split(1, st - 17);
                    pattern();
                }
#pragma unroll
                for (int st = 20; st < 24; ++st) {                                                      // dQ_b
                    if (st == 20) operand(1, 0);
                    if (st == 22) operand(1, 1);
                    mm(st);
                    if (st == 20) unit_split(1, 3);
                    pattern();
                }
            } else {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                piece(4 * sub);
                f32x16 sc = mma_rows<HD>(kr_hi, kr_lo, sub, l31, hh, qh, ql, c_lse);      // S^T[key][q] - lse
                piece(4 * sub + 1);
                f32x16 dp = mma_rows<HD>(vr_hi, vr_lo, sub, l31, hh, doh, dol, c_del);    // dP^T[key][q] (- delta)
                piece(4 * sub + 2);
                if (t == nt - 1) {            // only the last tile can hold keys beyond B
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * KT + sub * 32 + acc_row(r, hh) >= B) sc[r] = -INFINITY;
                }
                const uint4* hk4 = reinterpret_cast<const uint4*>(htab + buf * KT + sub * 32 + 4 * hh);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {                           // registers 4g..4g+3 = keys 8g+4hh+{0..3}
                    uint32_t hk[4] = {1u, 1u, 1u, 1u};
                    if (DROP) { const uint4 v = hk4[2 * gq]; hk[0] = v.x; hk[1] = v.y; hk[2] = v.z; hk[3] = v.w; }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * gq + i;
                        const float p = rlt_exp2(SEED ? sc[r] : sc[r] - lse2);
                        float dpr = dp[r];
                        if (DROP) dpr = (rlt_keep_rc(hq, hk[i], a.drop_thr) ? dpr * inv_keep : 0.f) - del;
                        dp[r] = p * dpr;                                           // dS^T
                    }
                }
                piece(4 * sub + 3);
                if (S16) mma_T16(kt_hi, kt_lo, sub, lane, dp, dq16);               // dQ^T[d][q] += K^T dS^T
                else mma_T<HD>(kt_hi, kt_lo, sub, l31, hh, dp, dq);
            }
            }
        }
        dma_wait_barrier<asm_dma<HD>()>();
    }
    if (!wave_live) return;
    if (S16) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int qq = qt * QT3 + wv * 32 + 16 * hf + (lane & 15);
            if (qq < B)
                *reinterpret_cast<float4*>(a.dqkv + ((size_t)s * B + qq) * ld + h * HD + 4 * (lane >> 4)) =
                    make_float4(dq16[hf][0] * a.scale, dq16[hf][1] * a.scale, dq16[hf][2] * a.scale, dq16[hf][3] * a.scale);
        }
        return;
    }
    if (q >= B) return;
    store_acc_T<HD>(a.dqkv + ((size_t)s * B + q) * ld + h * HD, hh, dq, a.scale);
}

// ------------------------------------------------------------------------------------------ dK + dV fused
// workgroup = 256 keys, loops over 64-query tiles.  One pass computes S and P (recomputed from the LSE) once for both
// gradients: dV^T[d][key] += dO^T P and dK^T[d][key] += (cQ)^T dS - 4 products per tile (separate dV and dK kernels
// need 5); ~210 unified VGPRs, one 512-thread workgroup per CU.
template <int HD, bool DROP>
__global__ __launch_bounds__(512) void attn3_bwd_dkv_kernel(Attn3Args g) {
    const AttnArgs& a = g.a;
    constexpr int DT = (HD + 31) / 32;
    constexpr int QREC = Rec<HD>::RP + Rec<HD>::TP;                         // Q rows pair | Q transposed pair
    constexpr int STAGE = 2 * QREC + Rec<HD>::AUX;                          // ... | dO rows pair | dO transposed pair | aux
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint8_t* lds = reinterpret_cast<uint8_t*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hh = lane >> 5;
    const int B = a.B, H = a.H, E = H * HD, npair = a.S * H;
    const size_t ld = (size_t)3 * E;
    const int nt = rlt_cdiv_dev(B, KT);
    const int ntile = rlt_cdiv_dev(B, QT3);
    int pair, ktile;
    map_block(blockIdx.x, npair, ntile, pair, ktile);
    const int s = pair / H, h = pair % H;
    const int key = ktile * QT3 + wv * 32 + l31;
    const bool wave_live = ktile * QT3 + wv * 32 < B;
    const uint32_t ps = pair_seed(a.seed, pair);
    const float inv_keep = DROP ? 1.f / (1.f - a.drop_p) : 1.f;

    bf16x8 kh[HD / 16], kl[HD / 16], vh[HD / 16], vl[HD / 16];
    const int ktl = min(key >> 6, nt - 1);
    image_row_frags<HD>(record<HD>(g.img, 1, npair, nt, pair, ktl), key & 63, hh, kh, kl);
    image_row_frags<HD>(record<HD>(g.img, 2, npair, nt, pair, ktl), key & 63, hh, vh, vl);
    frags_ready(kh); frags_ready(kl); frags_ready(vh); frags_ready(vl);
    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    // head dim 16: the dV / dK products run on v_mfma_f32_16x16x32_bf16 (mma_T16), accumulators per 16-key half
    f32x4v dk16[2], dv16[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r) { dk16[hf][r] = 0.f; dv16[hf][r] = 0.f; }

    uint32_t* htab = reinterpret_cast<uint32_t*>(lds + 2 * STAGE);          // dropout: per-query hashes of the tile
    const uint32_t hk = DROP ? rlt_col_hash(ps, (uint32_t)key) : 1u;
    auto issue = [&](int t, int buf) {
        dma_copy<QREC, asm_dma<HD>()>(lds + buf * STAGE, record<HD>(g.img, 0, npair, nt, pair, t), wv, lane);
        dma_copy<STAGE - QREC, asm_dma<HD>()>(lds + buf * STAGE + QREC, record<HD>(g.dimg, 0, npair, nt, pair, t), wv, lane);   // whole dO record
        if (DROP && tid < KT) htab[buf * KT + tid] = rlt_row_hash(ps, (uint32_t)(t * KT + tid));
    };
    // TRREAD (stepped head-dim-64 body): the dV / dK products take their A operands from the ROWS images with
    // ds_read_b64_tr_b16, so the transposed images are not copied at all.  The 37 chunks of a tile - Q rows image 18, dO rows
    // image 18, aux block 1 - are ONE list dealt to the 8 wavefronts, 5 slots each (3 of the 40 re-copy the aux chunk);
    // source and destination follow from the chunk id by selects: branch-free, so that the tile body stays one basic block.
    constexpr bool TRREAD = RLT_DKV_TRREAD != 0 && RLT_STEPPED != 0 && HD == 64 && !DROP;
    auto deal_piece = [&](const uint8_t* nq, const uint8_t* nd, uint8_t* nl, int pc) {
        constexpr int NR = Rec<HD>::RP / 1024;
        const int id = min(wv + 8 * pc, 2 * NR);
        const bool isq = id < NR, isaux = id == 2 * NR;
        const int ch = isq ? id : id - NR;
        const uint8_t* src = (isq ? nq : nd) + (isaux ? QREC : ch * 1024) + lane * 16;
        const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)nl + (isq ? 0 : QREC) + (isaux ? QREC : ch * 1024);
        RLT_DMA_ASM(dst, src);
    };
    issue(0, 0);
    dma_wait_barrier<asm_dma<HD>()>();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        RLT_STAMP(0);
        // The copy of tile t+1 is spread over the tile body (below): all 8 wavefronts issuing their ~10 pieces at the top of the
        // tile cost every one of them ~1,500 cycles before the first MFMA (timeline stamps).  A wavefront without keys runs no
        // matrix steps and issues its pieces here.
        constexpr bool SPREAD = RLT_STEPPED_SPREAD != 0 && spread_dma<HD>() && (HD != 64 || (RLT_STEPPED != 0 && !DROP));
        if (SPREAD && TRREAD && !wave_live) {        // its 5 slots of the dealt list, all at once
            if (t + 1 < nt) {
                const uint8_t* nq1 = record<HD>(g.img, 0, npair, nt, pair, t + 1);
                const uint8_t* nd1 = record<HD>(g.dimg, 0, npair, nt, pair, t + 1);
#pragma unroll
                for (int pc = 0; pc < 5; ++pc) deal_piece(nq1, nd1, lds + (buf ^ 1) * STAGE, pc);
            }
        } else if (!SPREAD || !wave_live) { if (t + 1 < nt) issue(t + 1, buf ^ 1); }
        else if (DROP && t + 1 < nt && tid < KT) htab[(buf ^ 1) * KT + tid] = rlt_row_hash(ps, (uint32_t)((t + 1) * KT + tid));
        if (wave_live) {
            const uint16_t* qr_hi = reinterpret_cast<const uint16_t*>(lds + buf * STAGE);
            const uint16_t* qr_lo = qr_hi + rows_elems<HD>();
            const uint16_t* qt_hi = reinterpret_cast<const uint16_t*>(lds + buf * STAGE + Rec<HD>::RP);
            const uint16_t* qt_lo = qt_hi + T_elems<HD>();
            const uint16_t* dr_hi = reinterpret_cast<const uint16_t*>(lds + buf * STAGE + QREC);
            const uint16_t* dr_lo = dr_hi + rows_elems<HD>();
            const uint16_t* dt_hi = reinterpret_cast<const uint16_t*>(lds + buf * STAGE + QREC + Rec<HD>::RP);
            const uint16_t* dt_lo = dt_hi + T_elems<HD>();
            const float* Ls = reinterpret_cast<const float*>(lds + buf * STAGE + 2 * QREC);
            const float* Es = Ls + KT;
            const bool more = SPREAD && t + 1 < nt;
            // piece pc (0..9) of this wavefront's share of the copy of tile t+1: 5 pieces of the Q record, 5 of the dO
            // record; the record addresses are computed once per tile (address arithmetic inside the bursts cost 15 %)
            const uint8_t* nq = record<HD>(g.img, 0, npair, nt, pair, min(t + 1, nt - 1));
            const uint8_t* nd = record<HD>(g.dimg, 0, npair, nt, pair, min(t + 1, nt - 1));
            uint8_t* nl = lds + (buf ^ 1) * STAGE;
            auto next_piece = [&](int pc) {
                if (!more && !TRREAD) return;
                if (TRREAD) {                     // (the last tile re-copies itself into the idle stage)
                    if (pc < 5) deal_piece(nq, nd, nl, pc);
                } else if (pc < 5) dma_piece<QREC>(nl, nq, wv, lane, pc);
                else dma_piece<STAGE - QREC>(nl + QREC, nd, wv, lane, pc - 5);
            };
#if RLT_STEPPED
            if constexpr (HD == 64 && !DROP) {     // (with dropout the stepped body is 18 % slower than the compiler-scheduled one: 6.55 vs 5.57 ms)
                // The 32 matrix steps of a tile (3 MFMAs each) as an explicit software pipeline over the two 32-query
                // sub-tiles a, b:   [S_a dP_a] [S_b dP_b | E_a] [dV_a dK_a | E_b] [dV_b dK_b]
                // E_x = exp, dS and the bf16 splits of sub-tile x, cut into per-step chunks of a few vector instructions that
                // sit in the MFMA gaps of the OTHER sub-tile's products (left to itself hipcc keeps program order: the
                // element-wise block runs between the products it separates and the matrix pipe idles meanwhile - measured
                // tile time = MFMA cycles + vector issue cycles).  The LDS fragments of step i + RLT_STEPPED are read while
                // step i multiplies; a full scheduling fence after every step pins all of it.
                bf16x8 fh[32], fl[32];
                f32x16 sc2[2], dp2[2];
                bf16x8 ph[2][2], pl[2][2], gh[2][2], gl[2][2];    // split P and dS: [sub-tile][rows 8s..8s+7 of the block]
                auto frag = [&](int st) {
                    const int sub = (st >> 3) & 1, prod = (st >> 4) * 2 + ((st >> 2) & 1), k = st & 3;
                    if (prod < 2) {
                        const uint16_t* hi = prod == 0 ? qr_hi : dr_hi;
                        const uint16_t* lo = prod == 0 ? qr_lo : dr_lo;
                        const int off = (sub * 32 + l31) * (HD + 8) + 8 * hh + 16 * k;
                        fh[st] = *reinterpret_cast<const bf16x8*>(hi + off);
                        fl[st] = *reinterpret_cast<const bf16x8*>(lo + off);
                    } else if (TRREAD) {
                        // A[d][k-slot j] = X[query][d] read transposed from the rows image: k = 2 s + dt; the 16-lane group
                        // (lane >> 4) covers d = 32 dt + 16 (group & 1) + (lane & 15); its lane 4q + p supplies the address of
                        // row q, columns 4p..4p+3 of the 4 x 16 block and receives column (lane & 15).  The k slots of lane half
                        // hh are queries 16 s + 4 hh + {0..3} (first read) and + 8 (second read): the rows the B operand
                        // (registers 8s..8s+7 of the score block) holds.
                        const uint16_t* hi = prod == 2 ? dr_hi : qr_hi;
                        const uint16_t* lo = prod == 2 ? dr_lo : qr_lo;
                        const int qrow = sub * 32 + 16 * (k >> 1) + 4 * hh + ((lane & 15) >> 2);
                        const int off = qrow * (HD + 8) + 32 * (k & 1) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
                        const v4s h0 = tr_read(hi + off), h1 = tr_read(hi + off + 8 * (HD + 8));
                        const v4s l0 = tr_read(lo + off), l1 = tr_read(lo + off + 8 * (HD + 8));
                        fh[st] = cat_frag(h0, h1);
                        fl[st] = cat_frag(l0, l1);
                    } else {
                        const uint16_t* hi = prod == 2 ? dt_hi : qt_hi;
                        const uint16_t* lo = prod == 2 ? dt_lo : qt_lo;
                        const int off = ((k & 1) * 32 + l31) * LDT3 + sub * 32 + 16 * (k >> 1) + 8 * hh;   // k = 2 s + dt
                        fh[st] = as_frag(*reinterpret_cast<const uint4*>(hi + off));
                        fl[st] = as_frag(*reinterpret_cast<const uint4*>(lo + off));
                    }
                };
                auto seed = [&](int sub) {                   // accumulators start at -lse[q] and -delta[q]
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ql = sub * 32 + acc_row(r, hh);
                        sc2[sub][r] = Ls[ql];
                        dp2[sub][r] = DROP ? 0.f : Es[ql];
                    }
                };
#pragma unroll
                for (int i = 0; i < RLT_STEPPED; ++i) frag(i);
                seed(0);
                auto mm = [&](int st) {                      // the fragment reads of step st + RLT_STEPPED and the 3 MFMAs of step st
                    const int sub = (st >> 3) & 1, prod = (st >> 4) * 2 + ((st >> 2) & 1), k = st & 3;
                    if (RLT_SPREAD_EVERY > 0 && st % (RLT_SPREAD_EVERY > 0 ? RLT_SPREAD_EVERY : 1) == 0 && st / (RLT_SPREAD_EVERY > 0 ? RLT_SPREAD_EVERY : 1) < 10)
                        next_piece(st / (RLT_SPREAD_EVERY > 0 ? RLT_SPREAD_EVERY : 1));
                    if (st + RLT_STEPPED < 32) frag(st + RLT_STEPPED);
                    if (prod == 0) sc2[sub] = mfma3(fh[st], fl[st], kh[k], kl[k], sc2[sub]);           // S[q][key] - lse
                    else if (prod == 1) dp2[sub] = mfma3(fh[st], fl[st], vh[k], vl[k], dp2[sub]);      // dP[q][key] (- delta)
                    else if (prod == 2) dv[k & 1] = mfma3(fh[st], fl[st], ph[sub][k >> 1], pl[sub][k >> 1], dv[k & 1]);   // dV^T += dO^T P
                    else dk[k & 1] = mfma3(fh[st], fl[st], gh[sub][k >> 1], gl[sub][k >> 1], dk[k & 1]);                 // dK^T += (c Q)^T dS
                };
                // Measured with one working wavefront per SIMD (r02_notes.md): the 262 vector instructions of a tile cost
                // their full ~1,050 cycles on top of the 3,072 MFMA cycles when they sit in blocks between the products,
                // ~500 in this form.  Every step carries a fixed unit of the element-wise work whose inputs are ready -
                // 4 exp, or 4 multiplies, plus half of a split8 (12 instructions) - and a sched_group_barrier pattern puts
                // a third of it behind each of the step's three MFMAs (~5 vector instructions per MFMA gap).
                //   exp(S_a) under dP_a (steps 4..7); dS_a = P dP under S_b; exp(S_b) under dP_b; dS_b under dV_a;
                //   the splits of P_a, dS_a, P_b, dS_b follow their inputs by two steps; all done by step 21.
                uint2 qh_[2][2][2][2], ql_[2][2][2][2];      // [matrix P/dS][sub-tile][half][quarter] -> hi / lo of 4 values
                auto unit_exp = [&](int sub, int c) {        // registers 4c..4c+3: P = exp2(S - lse)
#pragma unroll
                    for (int r = 4 * c; r < 4 * c + 4; ++r) sc2[sub][r] = rlt_exp2(sc2[sub][r]);
                };
                auto unit_mul = [&](int sub, int c) {        // dS = P (dP - delta)
#pragma unroll
                    for (int r = 4 * c; r < 4 * c + 4; ++r) dp2[sub][r] = sc2[sub][r] * dp2[sub][r];
                };
                auto unit_split = [&](int m, int sub, int qd) {   // quarter qd (registers 4qd..4qd+3) of P (m = 0) or dS (m = 1)
                    const f32x16& w = m ? dp2[sub] : sc2[sub];
                    split4(w[4 * qd], w[4 * qd + 1], w[4 * qd + 2], w[4 * qd + 3], qh_[m][sub][qd >> 1][qd & 1], ql_[m][sub][qd >> 1][qd & 1]);
                };
                auto operand = [&](int m, int sub, int half, bf16x8& hi, bf16x8& lo) {
                    hi = as_frag(make_uint4(qh_[m][sub][half][0].x, qh_[m][sub][half][0].y, qh_[m][sub][half][1].x, qh_[m][sub][half][1].y));
                    lo = as_frag(make_uint4(ql_[m][sub][half][0].x, ql_[m][sub][half][0].y, ql_[m][sub][half][1].x, ql_[m][sub][half][1].y));
                };
                auto pattern = [] {                          // one step: (MFMA, 5 VALU) x 3, the LDS reads first
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                    __builtin_amdgcn_sched_barrier(0);
                };
                // unit per step (issue cycles: exp unit 32, multiply unit 17, split quarter 52; budget 3 x 24 per step):
                //   4..7 exp_a | 8..11 mul_a + split P_a | 12..15 split dS_a | 16 exp_b | 17..19 exp_b + mul_b |
                //   20 mul_b + split P_b | 21..23 split P_b | 24..27 split dS_b
#pragma unroll
                for (int st = 0; st < 4; ++st) { mm(st); if (st == 3) seed(1); pattern(); }                           // S_a
#pragma unroll
                for (int st = 4; st < 8; ++st) { mm(st); unit_exp(0, st - 4); pattern(); }                            // dP_a
#pragma unroll
                for (int st = 8; st < 12; ++st) { mm(st); unit_mul(0, st - 8); unit_split(0, 0, st - 8); pattern(); } // S_b
#pragma unroll
                for (int st = 12; st < 16; ++st) { mm(st); unit_split(1, 0, st - 12); pattern(); }                    // dP_b
                operand(0, 0, 0, ph[0][0], pl[0][0]); operand(0, 0, 1, ph[0][1], pl[0][1]);
#pragma unroll
                for (int st = 16; st < 20; ++st) {                                                                    // dV_a
                    mm(st); unit_exp(1, st - 16);
                    if (st > 16) unit_mul(1, st - 17);
                    pattern();
                }
                operand(1, 0, 0, gh[0][0], gl[0][0]); operand(1, 0, 1, gh[0][1], gl[0][1]);
#pragma unroll
                for (int st = 20; st < 24; ++st) {                                                                    // dK_a
                    mm(st);
                    if (st == 20) unit_mul(1, 3);
                    unit_split(0, 1, st - 20);
                    pattern();
                }
                operand(0, 1, 0, ph[1][0], pl[1][0]); operand(0, 1, 1, ph[1][1], pl[1][1]);
#pragma unroll
                for (int st = 24; st < 28; ++st) { mm(st); unit_split(1, 1, st - 24); pattern(); }                    // dV_b
                operand(1, 1, 0, gh[1][0], gl[1][0]); operand(1, 1, 1, gh[1][1], gl[1][1]);
#pragma unroll
                for (int st = 28; st < 32; ++st) { mm(st); pattern(); }                                               // dK_b
            } else {
#endif
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                // the accumulators start at -lse[q] and -delta[q] (the aux block of the dO record holds them negated),
                // so the products come out as S - lse and dP - delta without a subtraction per element (with dropout
                // delta is subtracted after the mask)
                f32x16 sc, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = sub * 32 + acc_row(r, hh);
                    sc[r] = Ls[ql];
                    dp[r] = DROP ? 0.f : Es[ql];
                }
                next_piece(5 * sub);
                sc = mma_rows<HD>(qr_hi, qr_lo, sub, l31, hh, kh, kl, sc);       // S[q][key] - lse (Q carries scale*log2e)
                next_piece(5 * sub + 1);
                dp = mma_rows<HD>(dr_hi, dr_lo, sub, l31, hh, vh, vl, dp);       // dP[q][key] (- delta)
                next_piece(5 * sub + 2);
                // queries beyond B need no mask: their columns of the transposed Q / dO images are zero, p is finite
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = sub * 32 + acc_row(r, hh);
                    float p = rlt_exp2(sc[r]);
                    float dpr = dp[r];
                    if (DROP) {
                        const bool keep = rlt_keep_rc(htab[buf * KT + ql], hk, a.drop_thr);
                        dpr = keep ? dpr * inv_keep : 0.f;
                        dp[r] = p * (dpr + Es[ql]);                                // dS uses the undropped p
                        p = keep ? p * inv_keep : 0.f;
                    } else {
                        dp[r] = p * dpr;
                    }
                    sc[r] = p;
                }
                next_piece(5 * sub + 3);
                if (HD == 16 && RLT_HD16_SMALL_MFMA) {
                    mma_T16(dt_hi, dt_lo, sub, lane, sc, dv16);                    // dV^T[d][key] += dO^T P
                    next_piece(5 * sub + 4);
                    mma_T16(qt_hi, qt_lo, sub, lane, dp, dk16);                    // dK^T[d][key] += (c Q)^T dS
                } else {
                    mma_T<HD>(dt_hi, dt_lo, sub, l31, hh, sc, dv);                 // dV^T[d][key] += dO^T P
                    next_piece(5 * sub + 4);
                    mma_T<HD>(qt_hi, qt_lo, sub, l31, hh, dp, dk);                 // dK^T[d][key] += (c Q)^T dS
                }
            }
#if RLT_STEPPED
            }
#endif
        }
        RLT_STAMP(1);
        dma_wait_barrier<asm_dma<HD>()>();
    }
    if (!wave_live) return;
    if (HD == 16 && RLT_HD16_SMALL_MFMA) {          // D[row = d = 4 (lane >> 4) + r][col = lane & 15]: one float4 per lane and 16-key half
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int kk = ktile * QT3 + wv * 32 + 16 * hf + (lane & 15);
            if (kk < B) {
                float* row16 = a.dqkv + ((size_t)s * B + kk) * ld + h * HD + 4 * (lane >> 4);
                *reinterpret_cast<float4*>(row16 + E) = make_float4(dk16[hf][0] * LN2, dk16[hf][1] * LN2, dk16[hf][2] * LN2, dk16[hf][3] * LN2);
                *reinterpret_cast<float4*>(row16 + 2 * E) = make_float4(dv16[hf][0], dv16[hf][1], dv16[hf][2], dv16[hf][3]);
            }
        }
        return;
    }
    if (key >= B) return;
    float* row = a.dqkv + ((size_t)s * B + key) * ld + h * HD;
    store_acc_T<HD>(row + E, hh, dk, LN2);          // the Q image carries c = scale*log2e: dK = ln2 * dS^T (cQ)
    store_acc_T<HD>(row + 2 * E, hh, dv, 1.f);
}

// two tile stages + the [2][KT] dropout hash table
constexpr size_t HTAB = 2 * KT * sizeof(uint32_t);
template <int HD> size_t fwd3_smem() { return (size_t)2 * (Rec<HD>::RP + Rec<HD>::TP) + HTAB; }
template <int HD> size_t dq3_smem() { return (size_t)2 * (2 * Rec<HD>::RP + Rec<HD>::TP) + HTAB; }
template <int HD> size_t dkv3_smem() { return (size_t)2 * (2 * (Rec<HD>::RP + Rec<HD>::TP) + Rec<HD>::AUX) + HTAB; }

template <int HD>
int prepare3(const PrepArgs& p, hipStream_t st) {
    const int nt = rlt_cdiv(p.B, KT);
    int rc = rlt_allow_lds(attn3_prepare_kernel<HD>, Rec<HD>::BYTES);
    if (rc) return rc;
    hipLaunchKernelGGL(attn3_prepare_kernel<HD>, dim3(p.S * p.H * nt, p.nmat), dim3(256), Rec<HD>::BYTES, st, p);
    return RLT_LAUNCH_RESULT();
}

template <int HD, bool DROP>
int launch3(int which, const Attn3Args& g, hipStream_t st) {
    const AttnArgs& a = g.a;
    const int grid = a.S * a.H * rlt_cdiv(a.B, QT3);
    int rc;
    if (which == 0) {
        if ((rc = rlt_allow_lds(attn3_fwd_kernel<HD, DROP>, fwd3_smem<HD>()))) return rc;
        hipLaunchKernelGGL((attn3_fwd_kernel<HD, DROP>), dim3(grid), dim3(512), fwd3_smem<HD>(), st, g);
    } else if (which == 1) {
        if ((rc = rlt_allow_lds(attn3_bwd_dkv_kernel<HD, DROP>, dkv3_smem<HD>()))) return rc;
        hipLaunchKernelGGL((attn3_bwd_dkv_kernel<HD, DROP>), dim3(grid), dim3(512), dkv3_smem<HD>(), st, g);
    } else {
        if ((rc = rlt_allow_lds(attn3_bwd_dq_kernel<HD, DROP>, dq3_smem<HD>()))) return rc;
        hipLaunchKernelGGL((attn3_bwd_dq_kernel<HD, DROP>), dim3(grid), dim3(512), dq3_smem<HD>(), st, g);
    }
    return RLT_LAUNCH_RESULT();
}

template <int HD>
int run3(int which, const AttnArgs& a, void* images, void* dimages, hipStream_t st) {
    Attn3Args g;
    g.a = a; g.img = (const uint8_t*)images; g.dimg = (const uint8_t*)dimages;
    if (which == 0) {            // forward: split Q (scaled), K, V once, then attend
        PrepArgs p{};
        const int E = a.H * HD;
        p.src[0] = a.qkv; p.src[1] = a.qkv + E; p.src[2] = a.qkv + 2 * E;
        p.mul[0] = a.scale * LOG2E; p.mul[1] = 1.f; p.mul[2] = 1.f;
        p.ld = (size_t)3 * E; p.out = (uint8_t*)images; p.S = a.S; p.B = a.B; p.H = a.H; p.nmat = 3;
        p.perm16[0] = p.perm16[1] = p.perm16[2] = HD == 16 && RLT_HD16_SMALL_MFMA;   // head dim 16: every transposed image feeds mma_T16
        // head dim 64 without dropout: the stepped dK+dV and dQ bodies read Q^T and K^T transposed from the rows images
        // (ds_read_b64_tr_b16), so those two images are not written (V^T is: the forward kernel's P.V operand)
        constexpr bool TR_ALL = HD == 64 && RLT_STEPPED != 0 && RLT_DKV_TRREAD != 0 && RLT_STEPPED_SPREAD != 0 &&
                                RLT_DQ_STEPPED != 0 && RLT_DQ_TRREAD != 0 && RLT_DQ_SPREAD != 0 && spread_dma<HD>();
        p.skip_t[0] = p.skip_t[1] = TR_ALL && a.drop_p == 0.f;
        p.skip_t[2] = HD == 64 && RLT_FWD_TRV != 0;          // V^T: only the forward kernel's P.V used it
        int rc = prepare3<HD>(p, st);
        if (rc) return rc;
    } else if (which == 3) {     // backward prepare: split dO (+ lse, delta)
        PrepArgs p{};
        p.src[0] = a.dout; p.mul[0] = 1.f; p.ld = (size_t)a.H * HD;
        p.lse = a.lse; p.delta = a.delta;
        p.o = a.o; p.delta_out = const_cast<float*>(a.delta);     // with a.o set: delta is computed by the prepare pass
        p.out = (uint8_t*)dimages; p.S = a.S; p.B = a.B; p.H = a.H; p.nmat = 1;
        p.perm16[0] = HD == 16 && RLT_HD16_SMALL_MFMA;       // dO^T likewise
        // dO^T is read transposed from the rows image by the stepped dK+dV body; the caller says so with drop_p == 0
        // (rlt_list_attention_bwd and, since ABI 5, the stand-alone prepare entry point: both are given the forward call's drop_p)
        constexpr bool TR_DKV = HD == 64 && RLT_STEPPED != 0 && RLT_DKV_TRREAD != 0 && RLT_STEPPED_SPREAD != 0 && spread_dma<HD>();
        p.skip_t[0] = TR_DKV && a.drop_p == 0.f;
        return prepare3<HD>(p, st);
    }
    return a.drop_p > 0.f ? launch3<HD, true>(which, g, st) : launch3<HD, false>(which, g, st);
}

}  // namespace

#if defined(RLT_STAMPS)
extern "C" int rlt_debug_stamps(unsigned long long* host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rlt_stamp_buf), n * sizeof(unsigned long long));
}
#endif

size_t rlt_attn3_images_bytes(int S, int B, int H, int HD, int nmat) {
    const size_t nt = (size_t)rlt_cdiv(B, KT);
    const size_t rec = HD == 64 ? Rec<64>::BYTES : (HD == 32 ? Rec<32>::BYTES : Rec<16>::BYTES);
    return (size_t)nmat * S * H * nt * rec;
}

// which: 0 forward (writes `images`), 1 dK+dV, 2 dQ, 3 backward prepare (writes `dimages` from dout, lse, delta)
int rlt_attn3_run(int which, const AttnArgs& a, int HD, void* images, void* dimages, hipStream_t st) {
    if (HD == 64) return run3<64>(which, a, images, dimages, st);
    if (HD == 32) return run3<32>(which, a, images, dimages, st);
    if (HD == 16) return run3<16>(which, a, images, dimages, st);
    return RLT_E_SHAPE;
}

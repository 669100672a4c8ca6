// List-axis attention at head dim 64 (AttnCut / MtAttnCut / MMOECut: d_model 256, 4 heads - reference models/AttnCut.py:9-10,17-18;
// BASELINE configs[1]: 4096 lists x 300), fp32-FAITHFUL six-product arithmetic ("bf16x6", see attention6.hip) in the form of the
// head-dim-16 kernels of attention6n.hip: v_mfma_f32_16x16x32_bf16, the three-way split of the fresh operand (P, dS) on the MATRIX
// pipe (R1 = X - SEL h: bit-identical to the vector split, tools/micro/mfma_resid.hip), ONE wavefront per SIMD running a software
// pipeline over 512-score items across tile boundaries, tile images pre-split once per call and staged by LDS-DMA, and the five small
// plane products of every list-contracted output in an accumulator of their OWN (what keeps the `worst-split` class at the f32
// kernels' error: profiles/r05_notes.md).  Same interface, same algorithm (flash-style, deterministic, no atomics), same fp32
// softmax arithmetic as attention6.hip, which stays the path for ragged batches, fewer than 512 lists and the backward pass.
//
// What head dim 64 changes against attention6n.hip: a d-contracted product (S = Q K^T, dP = dO V^T) is two k-steps of each of the
// six plane products (12 MFMAs per 16 x 16 tile - no plane pairing), a d-indexed output (P V, dQ, dK, dV) four 16-row blocks; a slot
// of the pipeline is 52 (forward) MFMAs long, 48 of them the products themselves, and the vector work of an item (8 exp2, 8 row-sum
// adds, 12 conversions) fills a third of what the gaps hide.  Tiles are 64 rows (two images of 24 KiB, double-buffered: 96 KiB).
//
// Tile image: [plane h | m | l][64 rows][64 d] bf16, 128-byte rows, the 16-byte unit c of row r at unit c ^ (2 ((r >> 1) & 3)):
// conflict-free for both access patterns - the row fragments (ds_read_b128: lane (l15, g) reads unit 4 ks + g of row l15; the
// hardware's 16-lane groups mix rows {0-3, 12-15} at g with rows {4-11} at g ^ 1) and the transposed fragments
// (ds_read_b64_tr_b16: 32 lanes read 8 consecutive rows x 2 adjacent units).
#include "attention_common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));

constexpr int KTH = 64;                      // rows per tile
constexpr int PLH = KTH * 64;                // bf16 elements per plane of a tile image (8 KiB)
constexpr int IMGH = 3 * PLH;                // ... per image (h | m | l): 24 KiB
constexpr int RECH = IMGH * 2;               // bytes per image record: 24 LDS-DMA pieces of 1 KiB

__device__ __forceinline__ uint32_t pk2h(float a, float b) {
    // the cast form: hipcc emits v_cvt_pk_bf16_f32 and inserts the wait states an MFMA needs behind a vector write of its operand
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    const v2 t = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ float lo16h(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi16h(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
__device__ __forceinline__ bf16x8 frag4h(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return __builtin_bit_cast(bf16x8, make_uint4(a, b, c, d)); }
__device__ __forceinline__ f32x4 mmh(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// exact three-way split in vector code (prepare pass and the stationary fragments only): eight values -> three fragments
__device__ __forceinline__ void split8h(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    uint32_t hh[4], mm[4], ll[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        hh[i] = pk2h(a, b);
        const float ra = a - lo16h(hh[i]), rb = b - hi16h(hh[i]);
        mm[i] = pk2h(ra, rb);
        ll[i] = pk2h(ra - lo16h(mm[i]), rb - hi16h(mm[i]));
    }
    h = frag4h(hh[0], hh[1], hh[2], hh[3]);
    m = frag4h(mm[0], mm[1], mm[2], mm[3]);
    l = frag4h(ll[0], ll[1], ll[2], ll[3]);
}

// swizzle of the 16-byte unit index by the row
__host__ __device__ inline int swz(int row) { return 2 * ((row >> 1) & 3); }

// max / sum over the four lanes (l & 15) + 16 {0, 1, 2, 3} that share a column
__device__ __forceinline__ float col_max4h(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float col_sum4h(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ---- pre-split tile images -----------------------------------------------------------------------------------------------------
// One workgroup per (matrix, pair, tile): 64 rows x 64 d of fp32 -> the three-plane image in exactly the LDS layout.  ntile + 1
// records per pair: the last one lies wholly beyond B (zero rows) - the tile the pipeline drains on.
//   record (block m, pair, tile) at (((m * npair + pair) * (ntile + 1)) + tile) * 24576 bytes
__host__ __device__ inline size_t a6h_img_block(int npair, int ntile) { return (size_t)npair * (ntile + 1) * RECH; }
__global__ __launch_bounds__(256) void attn6h_prepare_kernel(const float* __restrict__ src0, const float* __restrict__ src1, size_t ld,
                                                             int S, int B, int H, uint8_t* __restrict__ img0, uint8_t* __restrict__ img1) {
    const int tid = threadIdx.x, ntile = rlt_cdiv_dev(B, KTH);
    const int pair = blockIdx.x / (ntile + 1), tile = blockIdx.x % (ntile + 1);
    const int s_ = pair / H, h = pair % H;
    const float* base = (blockIdx.y ? src1 : src0) + (size_t)s_ * B * ld + h * 64;
    uint16_t* rec = reinterpret_cast<uint16_t*>((blockIdx.y ? img1 : img0) + ((size_t)pair * (ntile + 1) + tile) * RECH);
    const int r = tid >> 2, row = tile * KTH + r, c0 = 2 * (tid & 3);
    const bool ok = row < B;
    const float* p = base + (size_t)min(row, B - 1) * ld + 8 * c0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float4 v0 = *reinterpret_cast<const float4*>(p + 8 * u), v1 = *reinterpret_cast<const float4*>(p + 8 * u + 4);
        const float x[8] = {ok ? v0.x : 0.f, ok ? v0.y : 0.f, ok ? v0.z : 0.f, ok ? v0.w : 0.f,
                            ok ? v1.x : 0.f, ok ? v1.y : 0.f, ok ? v1.z : 0.f, ok ? v1.w : 0.f};
        bf16x8 hh_, mm_, ll_;
        split8h(x, hh_, mm_, ll_);
        uint16_t* im = rec + r * 64 + ((c0 + u) ^ swz(r)) * 8;
        *reinterpret_cast<bf16x8*>(im) = hh_;
        *reinterpret_cast<bf16x8*>(im + PLH) = mm_;
        *reinterpret_cast<bf16x8*>(im + 2 * PLH) = ll_;
    }
}

#ifndef RLT_A6H_FWD1_BODY          // (timing experiments compile other generated bodies: tools/gen_attn6h_body.py with GEN_OMIT / GEN_DMA_STEP)
#define RLT_A6H_FWD1_BODY "attention6h_fwd1_body.inc"
#define RLT_A6H_FWD1_BODY_DROP "attention6h_fwd1_body_drop.inc"
#endif
#ifdef RLT_A6H_STAMPS
// diagnostic build only: s_memtime at every slot of tiles 8..11 of one workgroup; entries 16 / 17: before / behind the barrier
__device__ unsigned long long a6h_stamps[4 * 4 * 18];
#define A6H_STAMP(k) do { if (blockIdx.x == 64 && lane == 0 && t >= 8 && t < 12) \
    a6h_stamps[(wv * 4 + (t - 8)) * 18 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define A6H_STAMP(k) do { } while (0)
#endif

// the six plane products, smallest first: plane (0 = h, 1 = m, 2 = l) of the A and of the B operand
__device__ __forceinline__ constexpr int prod_a(int p) { return p == 0 ? 1 : p == 1 ? 2 : p == 2 ? 0 : p == 3 ? 1 : 0; }
__device__ __forceinline__ constexpr int prod_b(int p) { return p == 0 ? 1 : p == 1 ? 0 : p == 2 ? 2 : p == 3 ? 0 : p == 4 ? 1 : 0; }

// ------------------------------------------------------------------------------------------ forward: one wavefront per SIMD
// One 256-thread workgroup per CU owns 256 queries (a wavefront 64 = four blocks of 16).  Slot s of the tile body issues
// S(s) x24 (the item's scores against the wavefront's stationary, scaled Q fragments, seeded with minus the query's reference),
// R1(s-2) x2, R2(s-2) x2 (the residuals of P's split) and O(s-3) x24 (O^T += V^T P^T); exp2, the row sums, the conversions, the
// fragment reads and the LDS-DMA staging sit in the gaps (tools/gen_attn6h_body.py fwd).  A pipeline four items deep cannot move a
// running maximum, so the reference of a query's weights is FIXED before it starts: the maximum of its scores against the first 32
// keys.  fp32 and the exact split are scale-free, so the result is the same whatever the reference - unless a weight leaves the fp32
// range: a workgroup whose normalisers end up non-finite, zero or above 2^100 raises its flag in a.redo, and a second launch (the
// ping-pong kernel of attention6.hip, same grid, same block -> rows map) redoes exactly the flagged workgroups with a moving
// reference.  Needs B % 64 == 0 (a partly filled tile would need a mask per key).
// DROP: train-mode dropout of the attention probabilities, the same counter-based masks as every other kernel family (keep(pair seed,
// query, key) = row_hash(query) * col_hash(key) >= threshold): the lane's four query hashes in registers, the 64 column hashes of
// the NEXT tile computed once per body into a double-buffered LDS table, a block's eight per lane read with its fragments; the mask
// is applied to the weight after it went into the normaliser (e_drop: one integer multiply, a compare, a multiply, a select).
template <bool DROP>
__global__ __launch_bounds__(256, 1) void attn6h_fwd1_kernel(AttnArgs a) {
    constexpr int NB = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* img0 = reinterpret_cast<uint16_t*>(smem);           // [2 buffers][K image | V image]
    uint32_t* htab = reinterpret_cast<uint32_t*>(img0 + 4 * IMGH);   // DROP: [2 buffers][64] column hashes of the tile's keys
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int B = a.B, H = a.H, E = H * 64;
    const size_t ld = (size_t)3 * E;
    int pair, rt;
    map_block(blockIdx.x, a.S * H, rlt_cdiv_dev(B, 256), pair, rt);
    const int s_ = pair / H, h = pair % H;
    const float* base = a.qkv + (size_t)s_ * B * ld + h * 64;
    const int row0 = rt * 256 + wv * 64;

    // per-lane element offsets into an image: row fragments (k-step 0 / 1) and transposed fragments (d block 0 .. 3)
    int offR[2], offT[4];
    {
        const int swr = swz(l15);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) offR[ks] = l15 * 64 + (((4 * ks + g) ^ swr) * 8);
        const int rowt = 4 * g + (l15 >> 2), swt = swz(rowt);
#pragma unroll
        for (int db = 0; db < 4; ++db) offT[db] = rowt * 64 + (((2 * db) ^ swt) + ((l15 & 3) >> 1)) * 8 + 4 * (l15 & 1);
    }
    bf16x8 sel[2];
    {
        uint32_t s0[2] = {0u, 0u};
        if ((l15 >> 2) == g) s0[(l15 & 3) >> 1] = (l15 & 1) ? 0xBF800000u : 0x0000BF80u;       // -1.0 at element l15 & 3
        sel[0] = frag4h(s0[0], s0[1], 0u, 0u);
        sel[1] = frag4h(0u, 0u, s0[0], s0[1]);
    }

    // the lane's queries (scaled, log2 domain), stationary B operands of the score products: [own block][plane][k-step]
    bf16x8 qf[NB][3][2];
    f32x4 seed_s[NB], acc[NB][4], acc2[NB][4];
    float l_run[NB], m_ref[NB];
    uint32_t hq[NB];
    const uint32_t ps = DROP ? pair_seed(a.seed, pair) : 0u;
    const float inv_keep = DROP ? 1.f / (1.f - a.drop_p) : 1.f;
    const float qmul = a.scale * LOG2E;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int r = row0 + 16 * n + l15, rc = min(r, B - 1);
        hq[n] = DROP ? rlt_row_hash(ps, (uint32_t)r) : 0u;
        const float* qp = base + (size_t)rc * ld;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float4 v0 = *reinterpret_cast<const float4*>(qp + 32 * ks + 8 * g), v1 = *reinterpret_cast<const float4*>(qp + 32 * ks + 8 * g + 4);
            const float x[8] = {v0.x * qmul, v0.y * qmul, v0.z * qmul, v0.w * qmul, v1.x * qmul, v1.y * qmul, v1.z * qmul, v1.w * qmul};
            split8h(x, qf[n][0][ks], qf[n][1][ks], qf[n][2][ks]);
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) { acc[n][db] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[n][db] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        l_run[n] = 0.f;
    }
    const int nt = B / KTH;
    const int npair = a.S * H;
    const uint8_t* rec0 = reinterpret_cast<const uint8_t*>(a.img) + ((size_t)0 * npair + pair) * (size_t)(nt + 1) * RECH;      // K images
    const uint8_t* rec1 = reinterpret_cast<const uint8_t*>(a.img) + ((size_t)1 * npair + pair) * (size_t)(nt + 1) * RECH;      // V images
    // staging: a wavefront copies pieces 6 wv .. 6 wv + 5 of each image (24 pieces of 1 KiB).  The record has the LDS layout, so the
    // instruction's immediate offset moves the source and the destination together: M0 (the LDS base) is written once per group of
    // up to four pieces (j % 6 == 0 and == 4), not per piece.  Nothing else in this kernel uses M0 (no movrel, no other LDS-DMA
    // builtin); the statements neither save nor restore it.
    auto dma = [&](int j, int tile, uint16_t* ibuf) __attribute__((always_inline)) {
        const int q = j % 6, grp = q < 4 ? 0 : 4;
        const uint8_t* rec = (j < 6 ? rec0 : rec1) + (size_t)tile * RECH + (6 * wv + grp) * 1024 + lane * 16;
        if (q == grp) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(
                (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(reinterpret_cast<uint8_t*>(ibuf) + (j / 6) * RECH + (6 * wv + grp) * 1024));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(dst) : "memory");
        }
        asm volatile("global_load_lds_dwordx4 %0, off offset:%1" :: "v"(rec), "n"((q - grp) * 1024) : "memory");
    };
#pragma unroll
    for (int j = 0; j < 12; ++j) dma(j, 0, img0);
    if (DROP) htab[lane] = rlt_col_hash(ps, (uint32_t)lane);      // tile 0 (every wavefront writes the same 64 words)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // the reference of each query: its largest score against the first 32 keys (log2 domain)
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        float tmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    t = mmh(*reinterpret_cast<const bf16x8*>(img0 + prod_a(p) * PLH + 16 * kb * 64 + offR[ks]), qf[n][prod_b(p)][ks], t);
            tmax = fmaxf(fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3])), tmax);
        }
        m_ref[n] = col_max4h(tmax);
        seed_s[n] = f32x4{-m_ref[n], -m_ref[n], -m_ref[n], -m_ref[n]};
    }

    // the pipeline's registers: RING item sets (scores in fp32, the planes of P) and the single-buffered fragments of the current block
    f32x4 sc[4][2];
    uint32_t pln[3][4][4];                                      // [h, m, l][ring][dword]
    bf16x8 kf[2][3][2];                                         // K row fragments [16-row block][plane][k-step]
    v4s vt[3][4][2];                                            // V^T fragments [plane][d block][half]
    uint4 hcv[2][2];                                            // DROP: column hashes of the block's keys [block parity][16-row block]
    hcv[0][0] = hcv[0][1] = hcv[1][0] = hcv[1][1] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) sc[i][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3 * 4 * 4; ++i) (&pln[0][0][0])[i] = 0u;
#pragma unroll
    for (int i = 0; i < 12; ++i) (&kf[0][0][0])[i] = frag4h(0u, 0u, 0u, 0u);
#pragma unroll
    for (int i = 0; i < 24; ++i) (&vt[0][0][0])[i] = v4s{0, 0, 0, 0};

    for (int t = 0; t <= nt; ++t) {                             // nt + 1 bodies: the last one drains the pipeline on the empty tile record
        const int cur = t & 1;
        const uint16_t* Ic = img0 + cur * 2 * IMGH;
        uint16_t* In = img0 + (cur ^ 1) * 2 * IMGH;
        const int t_next = min(t + 1, nt);
        const float livef = t < nt ? 1.f : 0.f;                  // the items that START in the drain body carry no keys
        const float prevf = t > 0 ? 1.f : 0.f;
#pragma unroll
        for (int n = 0; n < NB; ++n) {              // (their only uses are MFMA operands: keep them in AGPRs across the back edge)
#pragma unroll
            for (int i = 0; i < 6; ++i) asm volatile("" : "+a"((&qf[n][0][0])[i]));
#pragma unroll
            for (int db = 0; db < 4; ++db) asm volatile("" : "+a"(acc[n][db]), "+a"(acc2[n][db]));
        }
        asm volatile("" : "+a"(sel[0]), "+a"(sel[1]));
#define GAP_END __builtin_amdgcn_sched_barrier(0)
        // MFMAs as asm statements with the accumulator tied in place; the schedule keeps the distances hipcc cannot see
        // (tools/gen_attn6h_body.py: LAG, MARGIN, WAR).  Register classes: what the vector ALU or LDS reads touch in VGPRs, what only
        // MFMAs touch (stationary fragments, selection constants, output accumulators) in AGPRs.
        auto mma_va = [&](f32x4& d, bf16x8 av, bf16x8 bv) __attribute__((always_inline)) {      // tile += A(vgpr) B(agpr)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(av), "a"(bv));
        };
        auto mma_va_c = [&](f32x4& d, bf16x8 av, bf16x8 bv, const f32x4& cv) __attribute__((always_inline)) {      // tile = A B + seed
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(av), "a"(bv), "v"(cv));
        };
        auto mma_av = [&](f32x4& d, bf16x8 av, bf16x8 bv) __attribute__((always_inline)) {      // tile += A(agpr) B(vgpr)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(av), "v"(bv));
        };
        auto mma_out = [&](f32x4& d, bf16x8 av, bf16x8 bv) __attribute__((always_inline)) {     // output accumulator (agpr) += A(vgpr) B(vgpr)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(av), "v"(bv));
        };
        auto m_s = [&](int it, int n, int k) __attribute__((always_inline)) {
            const int kb = k / 12, p = (k % 12) / 2, ks = k % 2;
            if (k % 12 == 0) mma_va_c(sc[it][kb], kf[kb][prod_a(p)][ks], qf[n][prod_b(p)][ks], seed_s[n]);
            else mma_va(sc[it][kb], kf[kb][prod_a(p)][ks], qf[n][prod_b(p)][ks]);
        };
        auto plane = [&](int lvl, int it) __attribute__((always_inline)) {
            return frag4h(pln[lvl][it][0], pln[lvl][it][1], pln[lvl][it][2], pln[lvl][it][3]);
        };
        auto m_r = [&](int it, int which, int level, int kb) __attribute__((always_inline)) {
            mma_av(sc[it][kb], sel[kb], plane(level - 1, it));
        };
        auto m_o = [&](int it, int n, int which, int k) __attribute__((always_inline)) {
            const int p = k >> 2, db = k & 3;
            const v4s x = vt[prod_a(p)][db][0], y = vt[prod_a(p)][db][1];
            const v8s av = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
            mma_out(p < 5 ? acc2[n][db] : acc[n][db], __builtin_bit_cast(bf16x8, av), plane(prod_b(p), it));
        };
        auto e_exp = [&](int it, int kb, int r) __attribute__((always_inline)) { sc[it][kb][r] = rlt_exp2(sc[it][kb][r]); };
        // (chunks of the previous tile's last items: in the first body there is no previous tile - the ring holds zeros - and the weight must be 0, not exp2(0))
        auto e_exp_p = [&](int it, int kb, int r) __attribute__((always_inline)) { sc[it][kb][r] = rlt_exp2(sc[it][kb][r]) * prevf; };
        auto e_sum = [&](int it, int n, int kb, int r) __attribute__((always_inline)) { l_run[n] = __builtin_fmaf(sc[it][kb][r], livef, l_run[n]); };
        auto e_sum_p = [&](int it, int n, int kb, int r) __attribute__((always_inline)) { l_run[n] += sc[it][kb][r]; };
        auto c_pk = [&](int it, int which, int lvl, int j) __attribute__((always_inline)) {
            const f32x4& tile = sc[it][j >> 1];
            pln[lvl][it][j] = pk2h(tile[2 * (j & 1)], tile[2 * (j & 1) + 1]);
        };
        auto rd_row = [&](int mat, int kb, int pl, int ks, int b32) __attribute__((always_inline)) {
            kf[kb][pl][ks] = *reinterpret_cast<const bf16x8*>(Ic + pl * PLH + (32 * b32 + 16 * kb) * 64 + offR[ks]);
        };
        auto rd_tr = [&](int mat, int pl, int db, int half, int b32) __attribute__((always_inline)) {
            vt[pl][db][half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (v4s __attribute__((address_space(3)))*)(Ic + IMGH + pl * PLH + (32 * b32 + 16 * half) * 64 + offT[db]));
        };
        auto st_dma = [&](int j) __attribute__((always_inline)) { dma(j, t_next, In); };
        const uint32_t* Hc = htab + cur * KTH;
        auto rd_hc = [&](int fb, int kb, int b32) __attribute__((always_inline)) {
            hcv[fb][kb] = *reinterpret_cast<const uint4*>(Hc + 32 * b32 + 16 * kb + 4 * g);
        };
        auto st_hcol = [&]() __attribute__((always_inline)) {
            htab[(cur ^ 1) * KTH + lane] = rlt_col_hash(ps, (uint32_t)(t_next * KTH + lane));
        };
        auto e_drop = [&](int it, int n, int kb, int r, int fb) __attribute__((always_inline)) {
            const uint4& h4 = hcv[fb][kb];
            const uint32_t hc = r == 0 ? h4.x : r == 1 ? h4.y : r == 2 ? h4.z : h4.w;
            sc[it][kb][r] = rlt_keep_rc(hq[n], hc, a.drop_thr) ? sc[it][kb][r] * inv_keep : 0.f;
        };
        if constexpr (DROP) {
#include RLT_A6H_FWD1_BODY_DROP
        } else {
#include RLT_A6H_FWD1_BODY
        }
#undef GAP_END
        A6H_STAMP(16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wavefront's LDS-DMA pieces of the next tile have landed
        __syncthreads();
        A6H_STAMP(17);
    }
    bool bad = false;
    float inv[NB], lse[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const float l_tot = col_sum4h(l_run[n]);
        bad |= !(l_tot > 0.f && l_tot <= 1.2676506e30f);         // 2^100: non-finite, vanished or far out of range -> redo with the moving reference
        inv[n] = 1.f / l_tot;
        lse[n] = (m_ref[n] + log2f(l_tot)) * LN2;
    }
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    if (tid == 0 && a.redo) a.redo[blockIdx.x] = any_bad ? 1u : 0u;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int q = row0 + 16 * n + l15;
        if (q < B) {
            float* orow = a.o + ((size_t)s_ * B + q) * E + h * 64 + 4 * g;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const f32x4 o = acc[n][db] + acc2[n][db];
                *reinterpret_cast<float4*>(orow + 16 * db) = make_float4(o[0] * inv[n], o[1] * inv[n], o[2] * inv[n], o[3] * inv[n]);
            }
            if (g == 0) a.lse_o[((size_t)s_ * H + h) * B + q] = lse[n];
        }
    }
}

}  // namespace

#ifdef RLT_A6H_STAMPS
extern "C" int rlt_debug_a6h_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(a6h_stamps), sizeof(unsigned long long) * 4 * 4 * 18);
}
#endif

// the forward's image buffer: K images | V images (two blocks of npair x (ntile + 1) records)
size_t rlt_attn6h_fwd_images_bytes(int S, int B, int H) { return 2 * a6h_img_block(S * H, rlt_cdiv(B, KTH)); }

// column blocks `what0`, `what1` of a.qkv (0 Q, 1 K, 2 V; 3: a.dout) into image blocks slot0, slot1 of a.img - one launch
int rlt_attn6h_prepare2(int what0, int slot0, int what1, int slot1, const AttnArgs& a, hipStream_t st) {
    const int npair = a.S * a.H, ntile = rlt_cdiv(a.B, KTH), E = a.H * 64;
    uint8_t* img = reinterpret_cast<uint8_t*>(const_cast<void*>(a.img));
    const size_t blk = a6h_img_block(npair, ntile);
    RLT_CHECK_ARG((what0 == 3) == (what1 == 3));               // one row stride per launch
    const float* s0 = what0 == 3 ? a.dout : a.qkv + what0 * E;
    const float* s1 = what1 == 3 ? a.dout : a.qkv + what1 * E;
    hipLaunchKernelGGL(attn6h_prepare_kernel, dim3(npair * (ntile + 1), what1 == what0 && slot1 == slot0 ? 1 : 2), dim3(256), 0, st,
                       s0, s1, what0 == 3 ? (size_t)E : (size_t)3 * E, a.S, a.B, a.H, img + slot0 * blk, img + slot1 * blk);
    return RLT_LAUNCH_RESULT();
}

// which = 0: the pipelined forward (K / V images in blocks 0 / 1 of a.img, one flag word per workgroup in a.redo); the caller
// follows it with the fix-up launch of attention6.hip's kernel
int rlt_attn6h_run(int which, const AttnArgs& a, hipStream_t st) {
    RLT_CHECK_ARG(which == 0 && a.img && a.redo && a.B % KTH == 0);
    const size_t shm = (size_t)4 * IMGH * sizeof(uint16_t) + 2 * KTH * sizeof(uint32_t);
    const dim3 grid(a.S * a.H * rlt_cdiv(a.B, 256));
    int rc;
    if (a.drop_p > 0.f) {            // (a template parameter: hipcc if-converts a run-time test and executes the hashes regardless)
        if ((rc = rlt_allow_lds(attn6h_fwd1_kernel<true>, shm))) return rc;
        hipLaunchKernelGGL(attn6h_fwd1_kernel<true>, grid, dim3(256), shm, st, a);
    } else {
        if ((rc = rlt_allow_lds(attn6h_fwd1_kernel<false>, shm))) return rc;
        hipLaunchKernelGGL(attn6h_fwd1_kernel<false>, grid, dim3(256), shm, st, a);
    }
    return RLT_LAUNCH_RESULT();
}

// Weights-stationary products of the fp32-faithful mode ("bf16x6") for the K = 256 Linear layers of the encoder and the LSTM input
// projection (row M3 of SURVEY.md section 8a: in_proj / out_proj / linear1 of nn.TransformerEncoderLayer, models/AttnCut.py:9-10; VERDICT
// r04 item 3): C[M x N] = A[M x 256] op(B) + bias with M = the 1.2 M (position, list) rows.
//
// Rounds 2-4 ran these as 256 x 256 output tiles with a 16-step K loop: a third of a tile's time is prologue, split of the
// stationary operand (again for every tile) and epilogue, and the matrix pipe stays 46-57 % busy.  Here NOTHING is tiled over K: a
// workgroup of four wavefronts (one per SIMD, 512 registers each) keeps a 256-column panel of the weight - all of K - resident, split
// once: every wavefront 64 columns = 4 blocks of 16 x 8 k-steps of fragments, h and m planes in its 256 AGPRs, l plane half in
// VGPRs (k-steps 0-3) and half in LDS (64 KB) - and streams 32-row blocks of A through it: A operand of v_mfma_f32_16x16x32_bf16 =
// weight columns (row l&15 = column of C, k = 32 ks + 8 (l>>4) + j), B operand = the block's rows from LDS (three planes, 16-byte
// chunks XOR-swizzled with the row: LDS is full, 160 KB), C: lane holds 4 consecutive columns of one row - a 16-byte store.
// A block is 384 MFMAs per wavefront, half-major (rows 0-15, then 16-31); the gaps behind them carry the split of the NEXT block
// into the other LDS buffer, the loads of the block after that, and the stores of the half that just finished (tools/gen_gemm6s_body.py).
// One barrier per block.  Row bounds are buffer bounds (rows past M are never loaded or stored).
#include "common.h"
#include "gemm6s.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// KS = K / 32 k-steps: 8 (K = 256: AttnCut's d_model) or 4 (K = 128: Choopy's; the whole l plane of the panel fits the VGPRs, no LDS part)
template <int KS> constexpr int gs_rowb() { return 64 * KS; }                  // bytes per row of a plane: K bf16
template <int KS> constexpr int gs_plane() { return 32 * gs_rowb<KS>(); }      // one plane of a 32-row block
template <int KS> constexpr int gs_buf() { return 3 * gs_plane<KS>(); }
template <int KS> constexpr int gs_wl() { return KS > 4 ? 4 * 4 * (KS - 4) * 64 * 16 : 0; }      // l plane, k-steps 4..: [wavefront][column block][k-step - 4][lane] x 16 B
template <int KS> constexpr size_t gs_lds() { return (size_t)2 * gs_buf<KS>() + gs_wl<KS>(); }
static_assert(gs_lds<8>() == 160 * 1024, "the K = 256 streaming product fills LDS exactly");

__device__ __forceinline__ uint32_t pk2s(float a, float b) {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    v2 t = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ float bfs_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bfs_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
__device__ __forceinline__ void split4s(float a, float b, float c, float d, uint2& hi, uint2& mid, uint2& lo) {
    hi.x = pk2s(a, b);
    hi.y = pk2s(c, d);
    asm("" : "+v"(hi.x), "+v"(hi.y));
    const float ra = a - bfs_lo(hi.x), rb = b - bfs_hi(hi.x), rc = c - bfs_lo(hi.y), rd = d - bfs_hi(hi.y);
    mid.x = pk2s(ra, rb);
    mid.y = pk2s(rc, rd);
    asm("" : "+v"(mid.x), "+v"(mid.y));
    lo.x = pk2s(ra - bfs_lo(mid.x), rb - bfs_hi(mid.x));
    lo.y = pk2s(rc - bfs_lo(mid.y), rd - bfs_hi(mid.y));
}
__device__ __forceinline__ bf16x8 frag8s(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return __builtin_bit_cast(bf16x8, make_uint4(a, b, c, d));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_s(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// TB: B stored [N][K] (y = x W^T, W row-major) or [K][N].  EPI: 0 plain, 1 C = max(C, 0), 2 the same + the 1-bit mask (C > 0) written,
// 3 C = mask bit ? C * mask_scale : 0.  A mask word holds the 32 rows of a block for one column: a half block writes its 16 bits as a
// short (ballot over the 16 lanes that hold a column's rows), reads its bit out of the word every lane of the column loads.
// NCB: 16-column blocks per wavefront - 4 (a 256-column panel per workgroup) or 2 (128 columns: N = 128, 384 at K = 128; plain epilogues)
template <int KS, int NCB, bool TB, int EPI>
__global__ __launch_bounds__(256, 1) void gemm6s_kernel(Gemm6sArgs g) {
    static_assert(NCB == 4 || (NCB == 2 && KS == 4 && EPI < 2), "128-column panels: K = 128, plain epilogues");
    constexpr int PW = 64 * NCB, WC = 16 * NCB;                    // columns per workgroup / per wavefront
    constexpr int GS_PLANE = gs_plane<KS>(), GS_BUF = gs_buf<KS>(), ROWB = gs_rowb<KS>(), CPR = 4 * KS, RPP = 256 / CPR, NI = 32 / RPP;
    extern __shared__ __attribute__((aligned(16))) uint8_t sm6s[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 15, q = lane >> 4;
    const int npanel = g.N / PW;
    // (integer division runs on the vector ALU: without the readfirstlane its results - and every buffer resource formed from them - sit
    //  in VGPRs, and hipcc wraps each buffer instruction in a waterfall loop)
    const int panel = __builtin_amdgcn_readfirstlane(blockIdx.x % npanel), stream = __builtin_amdgcn_readfirstlane(blockIdx.x / npanel),
              nstream = __builtin_amdgcn_readfirstlane(gridDim.x / npanel);
    const int nblk = (g.M + 31) / 32;
    uint4* wl_s = reinterpret_cast<uint4*>(sm6s + 2 * GS_BUF);
    (void)wl_s;

    // ---- the stationary panel: column 256 panel + 64 w + 16 cb + n, k = 32 ks + 8 q + j ----
    bf16x8 wh[NCB][KS], wm[NCB][KS], wlv[NCB][4];
    f32x4 bv[NCB];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {             // (second pass: the l fragments kept in VGPRs, when the others are in their AGPRs)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int col = PW * panel + WC * w + 16 * cb + n;
            const float* wp = TB ? g.B + (size_t)col * g.ldb + 8 * q : g.B + (size_t)(8 * q) * g.ldb + col;
#pragma unroll
            for (int ks = 0; ks < (pass ? 4 : KS); ++ks) {
                asm volatile("" : "+v"(wp));          // (opaque: hipcc would form every address up front and spill them)
                float v[8];
                if (TB) {
                    const float4 v0 = *reinterpret_cast<const float4*>(wp + 32 * ks), v1 = *reinterpret_cast<const float4*>(wp + 32 * ks + 4);
                    v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = wp[(size_t)(32 * ks + j) * g.ldb];
                }
                uint2 h0, m0, l0, h1, m1, l1;
                split4s(v[0], v[1], v[2], v[3], h0, m0, l0);
                split4s(v[4], v[5], v[6], v[7], h1, m1, l1);
                if (pass == 0) {
                    wh[cb][ks] = frag8s(h0.x, h0.y, h1.x, h1.y);
                    wm[cb][ks] = frag8s(m0.x, m0.y, m1.x, m1.y);
                    asm volatile("" : "+a"(wh[cb][ks]), "+a"(wm[cb][ks]) : : "memory");
                    if (KS > 4 && ks >= 4) wl_s[((w * 4 + cb) * (KS - 4) + (ks - 4)) * 64 + lane] = make_uint4(l0.x, l0.y, l1.x, l1.y);
                } else {
                    wlv[cb][ks] = frag8s(l0.x, l0.y, l1.x, l1.y);
                    asm volatile("" : "+v"(wlv[cb][ks]) : : "memory");
                }
            }
            if (pass == 0) {                           // the bias of the lane's four columns: the C operand of a chain's first MFMA
                const int c4 = PW * panel + WC * w + 16 * cb + 4 * q;
                f32x4 b = {0.f, 0.f, 0.f, 0.f};
                if (g.bias) { const float4 t = *reinterpret_cast<const float4*>(g.bias + c4); b += f32x4{t.x, t.y, t.z, t.w}; }
                if (g.bias2) { const float4 t = *reinterpret_cast<const float4*>(g.bias2 + c4); b += f32x4{t.x, t.y, t.z, t.w}; }
                bv[cb] = b;
            }
        }
    }

    // ---- per-lane addresses ----
    // staging: thread -> rows sr + RPP i (i = 0 .. NI-1), 32-byte chunk sc of the row's K floats
    const int sc = tid % CPR, sr = tid / CPR;
    const uint32_t voff_a = (uint32_t)sr * g.lda * 4u + sc * 32u;                       // + RPP i lda 4 (soffset) + 16
    // LDS writes: row rr = sr + RPP i, chunk c at rr * ROWB + 16 (c ^ (rr & 15))
    uint32_t wro[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) wro[i] = (uint32_t)(sr + RPP * i) * ROWB + 16u * (sc ^ ((sr + RPP * i) & 15));
    // fragment reads of a half: row n, chunk 4 ks + q at (4 ks + q) ^ n = 4 (ks ^ (n >> 2)) + (q ^ (n & 3)): one base per ks & 3
    uint32_t rdo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rdo[j] = n * ROWB + 64u * (j ^ (n >> 2)) + 16u * (q ^ (n & 3));      // + buffer + plane + 16 ROWB half + 256 (ks >> 2)
    const uint8_t* wlrd = sm6s + 2 * GS_BUF + w * (4 * (KS - 4) * 1024) + lane * 16;     // + 1024 ((KS - 4) cb + ks - 4)
    const uint32_t voff_c = (uint32_t)n * g.ldc * 4u + ((uint32_t)WC * w + 4u * q) * 4u;            // + 16 half ldc 4 (soffset) + 64 cb
    // mask words of the block: [column] within the panel's 256; written by lanes 0-15 (column 16 cb + lane), the others out of bounds
    const uint32_t voff_bo = lane < 16 ? ((uint32_t)WC * w + lane) * 4u : 0x7fff0000u;               // + 64 cb + 2 half
    const uint32_t voff_bi = ((uint32_t)WC * w + 4u * q) * 4u;                                       // + 64 cb: the words of the lane's four columns
    uint32_t bm[5];                       // EPI 2: lane masks - ballot (lane & 3), its low half for (lane >> 2) & 3 < 2 - and the shift
#pragma unroll
    for (int r = 0; r < 4; ++r) bm[r] = (lane & 3) == r ? 0xffffffffu : 0u;
    bm[4] = ((lane >> 2) & 3) < 2 ? 0xffffffffu : 0u;
    const uint32_t bsh = 16u * ((lane >> 2) & 1);
#pragma unroll
    for (int r = 0; r < 5; ++r) asm volatile("" : "+v"(bm[r]));

    auto blk_of = [&](int i) { return stream + i * nstream; };
    // (the clamp is a v_med3 on the vector ALU: readfirstlane brings the row count - and the resource built from it - back to SGPRs)
    auto rows_of = [&](int i) { const int r = g.M - 32 * blk_of(i); return __builtin_amdgcn_readfirstlane(i < 0 ? 0 : r < 0 ? 0 : r > 32 ? 32 : r); };
    auto rs_a = [&](int i) { return rsrc_s(g.A + (size_t)(i < 0 ? 0 : blk_of(i)) * 32 * g.lda, (uint32_t)rows_of(i) * g.lda * 4u); };
    auto rs_bits = [&](int i) {
        const uint32_t* base = EPI == 2 ? g.bits_out : g.bits_in;
        return rsrc_s(base + (size_t)(i < 0 ? 0 : blk_of(i)) * g.N + PW * panel, rows_of(i) > 0 ? (uint32_t)PW * 4u : 0u);
    };
    auto rs_c = [&](int i) { return rsrc_s(g.C + (size_t)(i < 0 ? 0 : blk_of(i)) * 32 * g.ldc + PW * panel, (uint32_t)rows_of(i) * g.ldc * 4u); };
    const int nb = __builtin_amdgcn_readfirstlane(stream < nblk ? (nblk - stream + nstream - 1) / nstream : 0);      // blocks of this workgroup

    // ---- state ----
    f32x4 acc[2][NCB], xs[2 * NI];
    u32x4 bw[NCB];                         // EPI 3: the mask words of the lane's columns, per column block
    bf16x8 bfr[2][3], lfr[NCB];
    uint4 sph, spm, spl;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2 * NCB; ++i) { acc[i / NCB][i % NCB] = z4; }
#pragma unroll
    for (int i = 0; i < 2 * NI; ++i) xs[i] = z4;
#pragma unroll
    for (int i = 0; i < 6; ++i) (&bfr[0][0])[i] = frag8s(0u, 0u, 0u, 0u);
#pragma unroll
    for (int i = 0; i < NCB; ++i) lfr[i] = frag8s(0u, 0u, 0u, 0u);
    sph = spm = spl = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) bw[cb] = u32x4{0u, 0u, 0u, 0u};
    auto load_x = [&](int i4, int part, __amdgpu_buffer_rsrc_t r) __attribute__((always_inline)) {
        xs[2 * i4 + part] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff_a + 16u * part, (uint32_t)(RPP * i4) * g.lda * 4u, 0));
    };
    auto split_x = [&](int i4, int part) __attribute__((always_inline)) {         // pieces of split4s on xs[2 i4], xs[2 i4 + 1] (residuals in place)
        f32x4 &a = xs[2 * i4], &b = xs[2 * i4 + 1];
        if (part == 0) { sph = make_uint4(pk2s(a[0], a[1]), pk2s(a[2], a[3]), pk2s(b[0], b[1]), pk2s(b[2], b[3])); asm("" : "+v"(sph.x), "+v"(sph.y), "+v"(sph.z), "+v"(sph.w)); }
        if (part == 1) { a[0] -= bfs_lo(sph.x); a[1] -= bfs_hi(sph.x); a[2] -= bfs_lo(sph.y); a[3] -= bfs_hi(sph.y); }
        if (part == 2) { b[0] -= bfs_lo(sph.z); b[1] -= bfs_hi(sph.z); b[2] -= bfs_lo(sph.w); b[3] -= bfs_hi(sph.w); }
        if (part == 3) { spm = make_uint4(pk2s(a[0], a[1]), pk2s(a[2], a[3]), pk2s(b[0], b[1]), pk2s(b[2], b[3])); asm("" : "+v"(spm.x), "+v"(spm.y), "+v"(spm.z), "+v"(spm.w)); }
        if (part == 4) { a[0] -= bfs_lo(spm.x); a[1] -= bfs_hi(spm.x); a[2] -= bfs_lo(spm.y); a[3] -= bfs_hi(spm.y); }
        if (part == 5) { b[0] -= bfs_lo(spm.z); b[1] -= bfs_hi(spm.z); b[2] -= bfs_lo(spm.w); b[3] -= bfs_hi(spm.w); }
        if (part == 6) spl = make_uint4(pk2s(a[0], a[1]), pk2s(a[2], a[3]), pk2s(b[0], b[1]), pk2s(b[2], b[3]));
    };
    auto write_x = [&](int buf, int i4, int pl) __attribute__((always_inline)) {
        *reinterpret_cast<uint4*>(sm6s + buf * GS_BUF + pl * GS_PLANE + wro[i4]) = pl == 0 ? sph : pl == 1 ? spm : spl;
    };
    // prologue: block 0 split into buffer 0, block 1 into the staging registers
    {
        const __amdgpu_buffer_rsrc_t r0 = rs_a(0), r1 = rs_a(1);
#pragma unroll
        for (int i4 = 0; i4 < NI; ++i4) { load_x(i4, 0, r0); load_x(i4, 1, r0); }
#pragma unroll
        for (int i4 = 0; i4 < NI; ++i4) {
#pragma unroll
            for (int part = 0; part < 7; ++part) split_x(i4, part);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) write_x(0, i4, pl);
            load_x(i4, 0, r1);
            load_x(i4, 1, r1);
        }
#pragma unroll
        for (int i = 0; i < 2 * NI; ++i) asm volatile("" : "+v"(xs[i]));       // consumed: no load is pending when the first block starts (see lstm6w.hip)
    }
    __syncthreads();

#define GAP_END __builtin_amdgcn_sched_barrier(0)
    // One tick = block i (buffer X = i & 1): its chain; the stores of half 1 of block i - 1 and, later, of half 0 of block i; the split of
    // block i + 1 (in the staging registers) into buffer 1 - X and the loads of block i + 2.
    auto tick = [&](auto XC, int i) __attribute__((always_inline)) {
        constexpr int X = decltype(XC)::value, Y = 1 - X;
        const __amdgpu_buffer_rsrc_t rc0 = rs_c(i), rc1 = rs_c(i - 1), ra2 = rs_a(i + 2);
        const __amdgpu_buffer_rsrc_t rb0 = EPI >= 2 ? rs_bits(i) : rc0, rb1 = EPI == 2 ? rs_bits(i - 1) : rc1;
        (void)rb0; (void)rb1;
        auto RB = [&](int half, int ks, int pl) __attribute__((always_inline)) {
            bfr[(KS * half + ks) & 1][pl] = *reinterpret_cast<const bf16x8*>(sm6s + X * GS_BUF + pl * GS_PLANE + half * 16 * ROWB + 256 * (ks >> 2) + rdo[ks & 3]);
        };
        auto RL = [&](int ks, int cb) __attribute__((always_inline)) { lfr[cb] = *reinterpret_cast<const bf16x8*>(wlrd + 1024 * ((KS - 4) * cb + ks - 4)); };
        auto MG = [&](int half, int ks, int p, int cb, int first) __attribute__((always_inline)) {
            const int bp = p == 2 ? 2 : (p == 0 || p == 4) ? 1 : 0;                  // streamed plane: m, h, l, h, m, h
            const bf16x8 xv = bfr[(KS * half + ks) & 1][bp];
            f32x4& d = acc[half][cb];
            if (p == 1) {                                                            // weight plane l: registers (k-steps 0-3) or the LDS fragment
                const bf16x8 lv = ks < 4 ? wlv[cb][ks < 4 ? ks : 0] : lfr[cb];
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(lv), "v"(xv));
            } else {
                const bf16x8 av = (p == 0 || p == 3) ? wm[cb][ks] : wh[cb][ks];
                const f32x4 seed = bv[cb];
                if (first) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "+v"(d) : "a"(av), "v"(xv), "v"(seed));     // (tied output: see lstm6w.hip)
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(av), "v"(xv));
            }
        };
        auto EP = [&](int half, int cb) __attribute__((always_inline)) {           // half 0: of this block; half 1: of the previous one
            f32x4 v = acc[half][cb];
            if (EPI == 3) {
                const uint32_t sh = 16 * half + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = ((bw[cb][r] >> sh) & 1u) ? v[r] * g.mask_scale : 0.f;
                // (the words of THIS tick's block, for its half 0 later in the tick and its half 1 at the start of the next one)
                if (half == 1) bw[cb] = __builtin_amdgcn_raw_buffer_load_b128(rb0, voff_bi + 64u * cb, 0, 0);
            }
            if (EPI == 1 || EPI == 2) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
            if (EPI == 2) {
                const uint64_t b0 = __builtin_amdgcn_ballot_w64(v[0] > 0.f), b1 = __builtin_amdgcn_ballot_w64(v[1] > 0.f),
                               b2 = __builtin_amdgcn_ballot_w64(v[2] > 0.f), b3 = __builtin_amdgcn_ballot_w64(v[3] > 0.f);
                // lane t < 16 stores column 16 cb + t = 4 q' + r' : bits 16 q' .. 16 q' + 15 of ballot r' - picked with lane masks (as
                // selects hipcc turns the choice into divergent branches, which cut the fenced body into pieces)
                const uint32_t lo = ((uint32_t)b0 & bm[0]) | ((uint32_t)b1 & bm[1]) | ((uint32_t)b2 & bm[2]) | ((uint32_t)b3 & bm[3]);
                const uint32_t hi = ((uint32_t)(b0 >> 32) & bm[0]) | ((uint32_t)(b1 >> 32) & bm[1]) | ((uint32_t)(b2 >> 32) & bm[2]) | ((uint32_t)(b3 >> 32) & bm[3]);
                const uint32_t piece = ((lo & bm[4]) | (hi & ~bm[4])) >> bsh;
                __builtin_amdgcn_raw_buffer_store_b16((short)piece, half ? rb1 : rb0, voff_bo + 64u * cb + 2u * half, 0, 0);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), half ? rc1 : rc0, voff_c + 64u * cb, (uint32_t)(16 * half) * g.ldc * 4u, 0);
        };
        auto SX = [&](int i4, int part) __attribute__((always_inline)) { split_x(i4, part); };
        auto LX = [&](int i4, int part) __attribute__((always_inline)) { load_x(i4, part, ra2); };
        auto WX = [&](int i4, int pl) __attribute__((always_inline)) { write_x(Y, i4, pl); };
        RB(0, 0, 0); RB(0, 0, 1); RB(0, 0, 2);
        GAP_END;
        if constexpr (KS == 8) {
#include "gemm6s_body.inc"
        } else if constexpr (NCB == 4) {
#include "gemm6s_k128_body.inc"
        } else {
#include "gemm6s_k128n128_body.inc"
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    const std::integral_constant<int, 0> H0;
    const std::integral_constant<int, 1> H1;
    // nb blocks + one more tick for the last block's second half: an even number of ticks, whatever is past the end reads zeros and
    // stores nothing (buffer bounds)
    for (int i = 0; i < nb + 1; i += 2) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int k4 = 0; k4 < KS; k4 += 4)
                asm volatile("" : "+a"(wh[cb][k4]), "+a"(wh[cb][k4 + 1]), "+a"(wh[cb][k4 + 2]), "+a"(wh[cb][k4 + 3]), "+a"(wm[cb][k4]), "+a"(wm[cb][k4 + 1]),
                             "+a"(wm[cb][k4 + 2]), "+a"(wm[cb][k4 + 3]));
        tick(H0, i);
        tick(H1, i + 1);
    }
#undef GAP_END
}

}  // namespace

bool rlt_gemm6s_ok(const Gemm6sArgs& g) {
    static const bool on = [] { const char* e = getenv("RLT_GEMM6S"); return !e || atoi(e) != 0; }();      // RLT_GEMM6S=0: the tiled kernels (A/B runs)
    return on && (g.K == 256 || g.K == 128) && (g.N % 256 == 0 || (g.K == 128 && g.N % 128 == 0 && !g.bits_out && !g.bits_in)) && g.N <= 256 * 256 && g.M >= 32 * 256 && g.lda % 4 == 0 && g.ldb % 4 == 0 && g.ldc % 4 == 0 &&
           rlt_aligned16(g.A) && rlt_aligned16(g.B) && rlt_aligned16(g.C) && (!g.bias || rlt_aligned16(g.bias)) &&
           (!g.bias2 || rlt_aligned16(g.bias2)) && (size_t)g.lda * 4 * 32 < (1u << 31) && (size_t)g.ldc * 4 * 32 < (1u << 31) &&
           !(g.bits_out && g.bits_in) && (!g.bits_out || rlt_aligned16(g.bits_out)) && (!g.bits_in || rlt_aligned16(g.bits_in));
}

int rlt_gemm6s_launch(const Gemm6sArgs& g, bool tb, bool relu, void* stream) {
    const bool narrow = g.N % 256 != 0;              // 128-column panels (K = 128)
    const int npanel = g.N / (narrow ? 128 : 256);
    const int nblk = (g.M + 31) / 32;
    int nstream = 256 / npanel;                      // one workgroup per CU: the panels x as many row streams as fill the chip
    if (nstream < 1) nstream = 1;
    if (nstream > nblk) nstream = nblk;
    const size_t lds = g.K == 256 ? gs_lds<8>() : gs_lds<4>();
    auto go = [&](auto kern) {
        const int rc = rlt_allow_lds(kern, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(npanel * nstream), dim3(256), lds, rlt_stream(stream), g);
        return 0;
    };
    if (g.bits_out && !relu) return -1;
    const int epi = g.bits_in ? 3 : g.bits_out ? 2 : relu ? 1 : 0;
    if (narrow) {
        if (g.K != 128 || epi >= 2) return -1;
        return tb ? (epi ? go(gemm6s_kernel<4, 2, true, 1>) : go(gemm6s_kernel<4, 2, true, 0>))
                  : (epi ? go(gemm6s_kernel<4, 2, false, 1>) : go(gemm6s_kernel<4, 2, false, 0>));
    }
#define GS_GO(KS_) (tb ? (epi == 3 ? go(gemm6s_kernel<KS_, 4, true, 3>) : epi == 2 ? go(gemm6s_kernel<KS_, 4, true, 2>) : epi == 1 ? go(gemm6s_kernel<KS_, 4, true, 1>) : go(gemm6s_kernel<KS_, 4, true, 0>)) \
                       : (epi == 3 ? go(gemm6s_kernel<KS_, 4, false, 3>) : epi == 2 ? go(gemm6s_kernel<KS_, 4, false, 2>) : epi == 1 ? go(gemm6s_kernel<KS_, 4, false, 1>) : go(gemm6s_kernel<KS_, 4, false, 0>)))
    return g.K == 256 ? GS_GO(8) : GS_GO(4);
#undef GS_GO
}

// fp32 dense contraction on the f32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32 products and
// accumulation, 64 FLOP/clk/SIMD = the chip's 157 TFLOP/s fp32 peak).  Serves every nn.Linear-shaped
// product of the hot path and its backward: in_proj / out_proj / FFN of the encoder layer
// (models/AttnCut.py:9), the LSTM input projections (models/AttnCut.py:8) and the weight gradients
// dW = dY^T X (split-K, deterministic slab reduction - no float atomics).
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 wavefronts as 2x2, each 64x64 = 2x2 MFMA
// tiles, 64 accumulator VGPRs), K step 16, operands staged in LDS k-major ([k][m], row 132 floats)
// so that the MFMA operand of lane l (row l&31, k = l>>5) is one conflict-free ds_read_b32; register
// prefetch of the next K tile overlaps the MFMAs of the current one.  blockIdx is remapped so that
// the tiles an XCD runs are contiguous (they share the A panel in that XCD's L2).
#include "common.h"
#include "gemm6s.h"
#include "split6.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128, LDT = 132;

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias; const float* bias2;
    int M, N, K, lda, ldb, ldc;
    int vecA, vecB;      // 16-byte vector loads allowed
    int flags;
    int kchunk;          // K range per z-slice (multiple of BK); gridDim.z slices
    float* slab;         // split-K partials [z][M*N] or null
    int tiles_m, tiles_n;
    const float* mask; int ldmask;   // epilogue: C = mask > 0 ? C : 0   (ReLU backward fused in dX)
    float* colsum;       // [M]: sum_k op(A)[m][k] (bias gradient from the dW product), TA only; or null
    float* cs_slab;      // split-K partials of colsum [z][M] or null
    float drop_p; uint32_t drop_thr, seed;   // epilogue dropout on the output (after ReLU), keep -> /(1-p)
    float mask_scale;    // with `mask`: kept elements are multiplied by this (1/(1-p) of the forward dropout)
    int slab_xcd;        // split-K launched as a 1-D grid with K slabs pinned to XCDs (see decode_block)
    // 1-bit masks, packed along ROWS: bit (row & 31) of word [(row >> 5) * ldbits + col], ldbits = N.  In the MFMA
    // accumulator layout a lane owns one column and 16 of the 32 rows of a block (the other 16 sit in lane ^ 32), so a
    // 32-row x 32-column block is ONE coalesced 128-byte word store / load per wavefront (packed along columns it took a
    // ballot and a one-lane store per accumulator register: 128 of each per wavefront and 256 x 256 tile)
    uint32_t* bits_out;  // with RLT_GEMM_RELU: bit = (C > 0)
    const uint32_t* bits_in;   // epilogue mask from such bits: C = bit ? C * mask_scale : 0
    int ldbits;
};

// (tile, K slab) of this workgroup.
//  * no split-K, or a split count that is not a multiple of 8: XCD-aware bijective remap of the flat tile id (the
//    tiles an XCD runs are contiguous and share operand panels in that XCD's L2); slab = blockIdx.z.
//  * split-K with nsplit % 8 == 0, launched as a 1-D grid of tiles x nsplit: under round-robin dispatch XCD c gets the
//    ids = c (mod 8); it is given the K slabs z = c (mod 8) and runs ALL tiles of a slab together, so every slab of A
//    and B is fetched from HBM by exactly one XCD and shared by that slab's tiles through its L2.  (With tiles spread
//    over XCDs instead, each XCD streams the whole small operand: 8x its bytes - measured 20 GB instead of 11 GB on
//    the FFN weight gradients.)
__device__ __forceinline__ void decode_block(const GemmArgs& g, int& bid, int& z) {
    const int nwg = g.tiles_m * g.tiles_n;
    const int id = blockIdx.x;
    if (g.slab_xcd) {
        const int xcd = id & 7, j = id >> 3, grp = j / nwg;
        z = grp * 8 + xcd;
        bid = j - grp * nwg;
    } else {
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, j = id >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        z = blockIdx.z;
    }
}

__device__ __forceinline__ float gemm_epilogue(const GemmArgs& g, float v, float bv, int row, int col, const float* dst) {
    v += bv;
    if (g.flags & RLT_GEMM_ACCUMULATE) v += *dst;
    if (g.flags & RLT_GEMM_RELU) v = fmaxf(v, 0.f);
    if (g.mask) v = (g.mask[(size_t)row * g.ldmask + col] > 0.f) ? v * g.mask_scale : 0.f;
    if (g.drop_p > 0.f) v = rlt_keep(g.seed, (uint32_t)row, (uint32_t)col, g.drop_thr) ? v * (1.f / (1.f - g.drop_p)) : 0.f;
    return v;
}

// KC = true : operand stored [MN][K] (K contiguous)   -> transposing LDS store
// KC = false: operand stored [K][MN] (MN contiguous)  -> direct LDS store
// FAST (host-checked: 16-byte aligned operand, K % 4 == 0, and MN % 4 == 0 for an MN-contiguous operand):
// branch-free - the address is clamped into the matrix, the load is unconditional and out-of-range
// lanes are zeroed with a select.  Guarded loads make hipcc branch around every load and drain vmcnt
// per element, which serialises the whole staging burst.
template <bool KC, int BK, bool FAST>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int mn0, int k0, int MN, int Kend,
                                          bool vec, int tid, float4 (&reg)[BK / 8]) {
#pragma unroll
    for (int i = 0; i < BK / 8; ++i) {
        const int idx = tid + 256 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FAST) {
            int r, c;          // r: row of the stored matrix, c: first of 4 contiguous columns
            bool ok;
            if (KC) { r = mn0 + idx / (BK / 4); c = k0 + 4 * (idx % (BK / 4)); ok = r < MN && c < Kend; r = min(r, MN - 1); c = min(c, Kend - 4); }
            else { r = k0 + (idx >> 5); c = mn0 + 4 * (idx & 31); ok = r < Kend && c < MN; r = min(r, Kend - 1); c = min(c, MN - 4); }
            // NOTE: the zero-select for out-of-range lanes is applied in store_tile, NOT here: consuming the
            // loaded value right away would put an s_waitcnt directly behind the load and expose its latency
            (void)ok;
            v = *reinterpret_cast<const float4*>(P + (size_t)r * ld + c);
        } else if (KC) {
            const int row = mn0 + idx / (BK / 4), kk = k0 + 4 * (idx % (BK / 4));
            if (row < MN) {
                const float* src = P + (size_t)row * ld + kk;
                if (vec && kk + 3 < Kend) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (kk + 0 < Kend) v.x = src[0];
                    if (kk + 1 < Kend) v.y = src[1];
                    if (kk + 2 < Kend) v.z = src[2];
                    if (kk + 3 < Kend) v.w = src[3];
                }
            }
        } else {
            const int kk = k0 + (idx >> 5), col = mn0 + 4 * (idx & 31);
            if (kk < Kend) {
                const float* src = P + (size_t)kk * ld + col;
                if (vec && col + 3 < MN) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (col + 0 < MN) v.x = src[0];
                    if (col + 1 < MN) v.y = src[1];
                    if (col + 2 < MN) v.z = src[2];
                    if (col + 3 < MN) v.w = src[3];
                }
            }
        }
        reg[i] = v;
    }
}

// FAST path: zero, in place, the lanes whose (clamped) load was out of range - called just before the registers
// are consumed (LDS store / bias-gradient side sum), i.e. as late as possible
template <bool KC, int BK>
__device__ __forceinline__ void mask_tile(float4 (&reg)[BK / 8], int tid, int mn0, int k0, int MN, int Kend) {
#pragma unroll
    for (int i = 0; i < BK / 8; ++i) {
        const int idx = tid + 256 * i;
        const bool ok = KC ? (mn0 + idx / (BK / 4) < MN && k0 + 4 * (idx % (BK / 4)) < Kend)
                           : (k0 + (idx >> 5) < Kend && mn0 + 4 * (idx & 31) < MN);
        reg[i] = make_float4(ok ? reg[i].x : 0.f, ok ? reg[i].y : 0.f, ok ? reg[i].z : 0.f, ok ? reg[i].w : 0.f);
    }
}

template <bool KC, int BK>
__device__ __forceinline__ void store_tile(float* __restrict__ T, int tid, const float4 (&reg)[BK / 8]) {
#pragma unroll
    for (int i = 0; i < BK / 8; ++i) {
        const int idx = tid + 256 * i;
        if (KC) {
            const int mn = idx / (BK / 4), k = 4 * (idx % (BK / 4));
            T[(k + 0) * LDT + mn] = reg[i].x;
            T[(k + 1) * LDT + mn] = reg[i].y;
            T[(k + 2) * LDT + mn] = reg[i].z;
            T[(k + 3) * LDT + mn] = reg[i].w;
        } else {
            const int k = idx >> 5, mn = 4 * (idx & 31);
            *reinterpret_cast<float4*>(&T[k * LDT + mn]) = reg[i];
        }
    }
}

// ---- shared epilogue: accumulator tiles -> C (or split-K slab), fused bias/ReLU/mask/dropout -------
// wavefront tile = 64 rows (2 MFMA blocks) x 32*NJ columns at (rbase, cbase); `interior`: the whole workgroup tile is
// inside C
template <bool V> struct BoolTag { static constexpr bool value = V; };
template <int NJ>
__device__ __forceinline__ void write_output_t(const GemmArgs& g, const f32x16 (&acc)[2][NJ], int rbase, int cbase,
                                               bool interior, int l31, int hh, int z) {
    const bool to_slab = g.slab != nullptr;
    float* out = to_slab ? g.slab + (size_t)z * g.M * g.N : g.C;
    const int ldo = to_slab ? g.N : g.ldc;
    // interior tiles with one of the common epilogues: the bounds and mode tests are hoisted out of the per-element
    // loop (tested per element they cost >1 ms of a K = 256 product that writes 10 GB)
    // Addressing of the interior paths: the row base out + (rb + 32 i + dr) * ldo + cb + 32 j is WAVE-UNIFORM (rb, cb are
    // made scalar with readfirstlane) and every lane adds the same 32-bit element offset 4 hh ldo + l31 to all of them, so
    // a store is `global_store_dword v_off, v_data, s[base]` with the bases stepped on the scalar unit.  With the
    // per-lane 64-bit address arithmetic the compiler generated before (rbase derived from threadIdx in VGPRs), the 128
    // stores of a wavefront cost ~5 VALU instructions and two address registers each and the epilogue spilled.
    const int rb = __builtin_amdgcn_readfirstlane(rbase), cb = __builtin_amdgcn_readfirstlane(cbase);
    const int loff = 4 * hh * ldo + l31;
    if (interior && g.drop_p <= 0.f) {
        auto tile = [&](auto f) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int col = cb + j * 32 + l31;
                float bv = 0.f;
                if (!to_slab) {
                    if (g.bias) bv += g.bias[col];
                    if (g.bias2) bv += g.bias2[col];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        float* base = out + (size_t)(rb + i * 32 + dr) * ldo + (cb + j * 32);       // scalar
                        base[loff] = f(acc[i][j][r] + bv, rb + i * 32 + dr + 4 * hh, col, base + loff);
                    }
                }
            }
        };
        const bool relu = g.flags & RLT_GEMM_RELU, accum = g.flags & RLT_GEMM_ACCUMULATE;
        if (to_slab || (!relu && !accum && !g.mask && !g.bits_in)) { tile([](float v, int, int, const float*) { return v; }); return; }
        if (relu && !accum && !g.mask && !g.bits_out && !g.bits_in) { tile([](float v, int, int, const float*) { return fmaxf(v, 0.f); }); return; }
        if (accum && !relu && !g.mask && !g.bits_in) { tile([](float v, int, int, const float* d) { return v + *d; }); return; }
        if (((g.bits_out && relu) || (g.bits_in && !relu)) && !accum && !g.mask && !to_slab) {
            // 1-bit masks (packed along rows): one word per lane and 32 x 32 block, written / read once
            const float sc = g.mask_scale;
            auto body = [&](auto wr_tag) {
            constexpr bool wr = decltype(wr_tag)::value;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int col = cb + j * 32 + l31;
                float bv = 0.f;
                if (g.bias) bv += g.bias[col];
                if (g.bias2) bv += g.bias2[col];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const size_t widx = (size_t)((rb + i * 32) >> 5) * g.ldbits + (cb + j * 32);      // scalar
                    uint32_t w = wr ? 0u : (g.bits_in + widx)[l31];
                    if (!wr) w >>= 4 * hh;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        float v = acc[i][j][r] + bv;
                        if (wr) {
                            v = fmaxf(v, 0.f);
                            w |= (v > 0.f ? 1u : 0u) << dr;
                        } else {
                            v = ((w >> dr) & 1u) ? v * sc : 0.f;
                        }
                        (out + (size_t)(rb + i * 32 + dr) * ldo + (cb + j * 32))[loff] = v;
                    }
                    if (wr) {
                        w <<= 4 * hh;
                        const uint32_t full = w | (uint32_t)__shfl_xor((int)w, 32, 64);
                        if (hh == 0) (g.bits_out + widx)[l31] = full;
                    }
                }
            }
            };
            if (g.bits_out) body(BoolTag<true>{}); else body(BoolTag<false>{});
            return;
        }
        if (g.mask && !relu && !accum && !g.bits_in && !g.bits_out) {
            const float* mk = g.mask; const int ldm = g.ldmask; const float sc = g.mask_scale;
            tile([=](float v, int row, int col, const float*) { return mk[(size_t)row * ldm + col] > 0.f ? v * sc : 0.f; });
            return;
        }
    }
    // interior tiles of the FFN hidden product in train mode: bias + ReLU + dropout (+ the 1-bit mask of what
    // survived both).  Row hashes of the two half-wave rows are wave-uniform (scalar unit); one column hash per lane
    if (interior && g.drop_p > 0.f && (g.flags & RLT_GEMM_RELU) && !(g.flags & RLT_GEMM_ACCUMULATE) && !g.mask && !g.bits_in &&
        !to_slab) {
        const float inv_keep = 1.f / (1.f - g.drop_p);
        const uint32_t thr = g.drop_thr, seed = g.seed;
        auto tile = [&](auto with_bits) {
            uint32_t hc[NJ];
            float bv[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int col = cb + j * 32 + l31;
                hc[j] = rlt_col_hash(seed, (uint32_t)col);
                bv[j] = 0.f;
                if (g.bias) bv[j] += g.bias[col];
                if (g.bias2) bv[j] += g.bias2[col];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                uint32_t w[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) w[j] = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row_lo = rb + i * 32 + (r & 3) + 8 * (r >> 2);
                    // readfirstlane keeps both hashes on the scalar unit (without it hipcc folds the select below into
                    // one per-lane hash of row_lo + 4*hh on the VALU: 64 hashes and as many live registers per lane)
                    const uint32_t h0 = __builtin_amdgcn_readfirstlane(rlt_row_hash(seed, (uint32_t)row_lo));
                    const uint32_t h1 = __builtin_amdgcn_readfirstlane(rlt_row_hash(seed, (uint32_t)(row_lo + 4)));
                    const uint32_t hr = hh ? h1 : h0;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        float v = fmaxf(acc[i][j][r] + bv[j], 0.f);
                        v = rlt_keep_rc(hr, hc[j], thr) ? v * inv_keep : 0.f;
                        if (decltype(with_bits)::value) w[j] |= (v > 0.f ? 1u : 0u) << acc_row(r, hh);
                        (out + (size_t)row_lo * ldo + (cb + j * 32))[loff] = v;
                    }
                }
                if (decltype(with_bits)::value) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const uint32_t full = w[j] | (uint32_t)__shfl_xor((int)w[j], 32, 64);
                        if (hh == 0) (g.bits_out + (size_t)((rb + i * 32) >> 5) * g.ldbits + (cb + j * 32))[l31] = full;
                    }
                }
            }
        };
        if (g.bits_out) tile(BoolTag<true>{}); else tile(BoolTag<false>{});
        return;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = cbase + j * 32 + l31;
        const bool col_ok = col < g.N;
        float bv = 0.f;
        if (!to_slab && col_ok) {
            if (g.bias) bv += g.bias[col];
            if (g.bias2) bv += g.bias2[col];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const size_t widx = (size_t)((rbase + i * 32) >> 5) * g.ldbits + col;
            const bool blk_ok = col_ok && rbase + i * 32 < g.M;
            const uint32_t win = (!to_slab && g.bits_in && blk_ok) ? g.bits_in[widx] : 0u;
            uint32_t wout = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + i * 32 + acc_row(r, hh);
                if (row >= g.M || !col_ok) continue;
                float v = acc[i][j][r];
                float* dst = out + (size_t)row * ldo + col;
                if (!to_slab) v = gemm_epilogue(g, v, bv, row, col, dst);
                if (!to_slab && g.bits_in) v = ((win >> acc_row(r, hh)) & 1u) ? v * g.mask_scale : 0.f;
                wout |= (v > 0.f ? 1u : 0u) << acc_row(r, hh);
                *dst = v;
            }
            if (!to_slab && g.bits_out) {          // rows beyond M keep a zero bit
                const uint32_t full = wout | (uint32_t)__shfl_xor((int)wout, 32, 64);
                if (hh == 0 && blk_ok) g.bits_out[widx] = full;
            }
        }
    }
}

__device__ __forceinline__ void write_output(const GemmArgs& g, const f32x16 (&acc)[2][2], int m0, int n0,
                                             int wm, int wn, int l31, int hh, int z) {
    write_output_t<2>(g, acc, m0 + wm * 64, n0 + wn * 64, m0 + BM <= g.M && n0 + BN <= g.N, l31, hh, z);
}

// 8 threads hold partial sums of the same 4 columns m: reduce through LDS.  KLOW = false: the 8 threads are
// tid, tid+32, ... (columns 4*(tid&31), the fp32 kernel's staging map); KLOW = true: tid = 8*mb + kb (columns 4*mb,
// the split-bf16 kernel's map, which keeps the k index in the low lane bits for conflict-free LDS stores)
template <bool KLOW>
__device__ __forceinline__ void finish_colsum(const GemmArgs& g, float4 csum, float* lds, int tid, int m0, int z) {
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(lds);
    red[tid] = csum;
    __syncthreads();
    if (tid < 32) {
        float4 t = red[KLOW ? 8 * tid : tid];
#pragma unroll
        for (int j = 1; j < 8; ++j) { const float4 o = red[KLOW ? 8 * tid + j : tid + 32 * j]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
        float* dst = (g.cs_slab ? g.cs_slab + (size_t)z * g.M : g.colsum) + m0 + 4 * tid;
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (m0 + 4 * tid + j < g.M) dst[j] = tv[j];
    }
}

template <bool TA, bool TB, int BK, int OCC, bool FAST>
__global__ __launch_bounds__(256, OCC) void gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float (*As)[BK * LDT] = reinterpret_cast<float (*)[BK * LDT]>(gsm);                    // [2][BK*LDT]
    float (*Bs)[BK * LDT] = reinterpret_cast<float (*)[BK * LDT]>(gsm + 2 * BK * LDT);     // [2][BK*LDT]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;

    int bid, zslab;
    decode_block(g, bid, zslab);
    const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = zslab * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[BK / 8], rb[BK / 8];
    // A operand: TA=0 -> stored [M][K] (K contiguous); B operand: TB=1 -> stored [N][K] (K contiguous)
    load_tile<!TA, BK, FAST>(g.A, g.lda, m0, kbeg, g.M, kend, g.vecA, tid, ra);
    load_tile<TB, BK, FAST>(g.B, g.ldb, n0, kbeg, g.N, kend, g.vecB, tid, rb);

    // bias gradient on the side: this thread's A elements are 4 consecutive m at fixed k rows
    const bool want_cs = TA && g.colsum != nullptr && tn == 0;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add_cs = [&]() {
#pragma unroll
        for (int i = 0; i < BK / 8; ++i) { csum.x += ra[i].x; csum.y += ra[i].y; csum.z += ra[i].z; csum.w += ra[i].w; }
    };
    const bool whole_mn = m0 + BM <= g.M && n0 + BN <= g.N;
    auto consume = [&](int k0, int buf) {        // mask (fast path), bias-gradient side sum, LDS store
        __builtin_amdgcn_sched_barrier(0);       // do not hoist the first use of the loaded registers above the MFMAs
        if (FAST && !(whole_mn && k0 + BK <= kend)) {       // (scalar test) a whole tile needs no zero-select
            mask_tile<!TA, BK>(ra, tid, m0, k0, g.M, kend);
            mask_tile<TB, BK>(rb, tid, n0, k0, g.N, kend);
        }
        if (want_cs) add_cs();
        store_tile<!TA, BK>(As[buf], tid, ra);
        store_tile<TB, BK>(Bs[buf], tid, rb);
    };
    consume(kbeg, 0);
    __syncthreads();

    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) {
            load_tile<!TA, BK, FAST>(g.A, g.lda, m0, k0 + BK, g.M, kend, g.vecA, tid, ra);
            load_tile<TB, BK, FAST>(g.B, g.ldb, n0, k0 + BK, g.N, kend, g.vecB, tid, rb);
        }
        const float* a_ = As[buf] + wm * 64 + l31;
        const float* b_ = Bs[buf] + wn * 64 + l31;
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int kk = (2 * ks + hh) * LDT;
            const float a0 = a_[kk], a1 = a_[kk + 32];
            const float b0 = b_[kk], b1 = b_[kk + 32];
            acc[0][0] = mfma32(a0, b0, acc[0][0]);
            acc[0][1] = mfma32(a0, b1, acc[0][1]);
            acc[1][0] = mfma32(a1, b0, acc[1][0]);
            acc[1][1] = mfma32(a1, b1, acc[1][1]);
        }
        if (more) consume(k0 + BK, buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    write_output(g, acc, m0, n0, wm, wn, l31, hh, zslab);
    if (want_cs) finish_colsum<false>(g, csum, gsm, tid, m0, zslab);
}

// ======================================================================================================
// Split-bf16 variant ("bf16x3"): every fp32 operand x is split on the fly into hi = bf16(x) and
// lo = bf16(x - hi); a product a*b is evaluated as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on the bf16 MFMA
// (v_mfma_f32_32x32x16_bf16, 16x the rate of the f32 MFMA) with fp32 accumulation.  16 mantissa bits
// per operand: relative error ~2^-16 per product (the dropped lo*lo term and the split residual),
// about 5x the throughput of the exact-fp32 kernel above at the same interface.
//
// Both operands sit in LDS k-contiguous ([mn][k], rows of 32 bf16 + 8 pad = 80 B) so that a lane's MFMA
// fragment (8 consecutive k of one row) is ONE ds_read_b128; K-contiguous global operands are split and
// stored with ds_write_b64, MN-contiguous ones through a 4x4 register transpose.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BK3 = 32, LDK3 = 40;                       // bf16 elements per LDS row
constexpr int TILE3 = 128 * LDK3;                        // one hi or lo tile (elements)

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    v2 t = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ void split4(float a, float b, float c, float d, uint2& hi, uint2& lo) {
    hi.x = pack_bf16x2(a, b);
    hi.y = pack_bf16x2(c, d);
    // opaque to the optimiser: otherwise hipcc re-derives the low element's bf16 with a second v_cvt_pk (x, 0) instead
    // of shifting the packed pair (one extra VALU instruction per two elements)
    asm("" : "+v"(hi.x), "+v"(hi.y));
    const float ah = __builtin_bit_cast(float, hi.x << 16), bh = __builtin_bit_cast(float, hi.x & 0xffff0000u);
    const float ch = __builtin_bit_cast(float, hi.y << 16), dh = __builtin_bit_cast(float, hi.y & 0xffff0000u);
    lo.x = pack_bf16x2(a - ah, b - bh);
    lo.y = pack_bf16x2(c - ch, d - dh);
}

// K-contiguous operand ([MN][K]): element idx of the 1024 float4 of a 128 x 32 tile -> (row, 16-byte k chunk).
// Bits: idx = [row>>3][kq>>1 (2 bits)][row&7 (3 bits)][kq&1]: a wavefront still covers 8 rows x 128 contiguous bytes
// of global memory, but the 16 lanes of a ds_write_b64 group hold 8 rows x 2 chunks, whose 4-dword windows at
// 20-dword row stride tile the 32 banks exactly (rows-major lanes, 2 rows x 8 chunks, overlap on 4 banks: a third of
// the LDS cycles of the NT kernel were bank conflicts)
__device__ __forceinline__ int kc_row(int idx) { return ((idx >> 1) & 7) | ((idx >> 6) << 3); }
__device__ __forceinline__ int kc_kq(int idx) { return (idx & 1) | (((idx >> 4) & 3) << 1); }

// registers of one staged operand tile: 4 float4 per thread
struct Stage3 { float4 v[4]; };

template <bool KC, bool FAST>
__device__ __forceinline__ void load3(const float* __restrict__ P, int ld, int mn0, int k0, int MN, int Kend, bool vec,
                                      int tid, Stage3& st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FAST) {     // branch-free: clamp, load, select (see load_tile)
            int r, c;
            bool ok;
            if (KC) { const int idx = tid + 256 * i; r = mn0 + kc_row(idx); c = k0 + 4 * kc_kq(idx); ok = r < MN && c < Kend; r = min(r, MN - 1); c = min(c, Kend - 4); }
            else { r = k0 + 4 * (tid & 7) + i; c = mn0 + 4 * (tid >> 3); ok = r < Kend && c < MN; r = min(r, Kend - 1); c = min(c, MN - 4); }
            (void)ok;          // the zero-select is applied by mask3 just before the registers are consumed
            v = *reinterpret_cast<const float4*>(P + (size_t)r * ld + c);
        } else if (KC) {       // [MN][K]: idx -> (row, 4 consecutive k)
            const int idx = tid + 256 * i;
            const int row = mn0 + kc_row(idx), kk = k0 + 4 * kc_kq(idx);
            if (row < MN) {
                const float* src = P + (size_t)row * ld + kk;
                if (vec && kk + 3 < Kend) v = *reinterpret_cast<const float4*>(src);
                else {
                    if (kk + 0 < Kend) v.x = src[0];
                    if (kk + 1 < Kend) v.y = src[1];
                    if (kk + 2 < Kend) v.z = src[2];
                    if (kk + 3 < Kend) v.w = src[3];
                }
            }
        } else {        // [K][MN]: thread owns a 4(k) x 4(mn) block, i = k row within the block; the k block is the LOW part
                        // of tid so that the 16 lanes of a ds_write_b64 group cover 8 k blocks x 2 rows = 32 distinct
                        // banks (with mn in the low bits the 16 rows are 80 dwords apart: an 8-way bank conflict)
            const int kk = k0 + 4 * (tid & 7) + i, col = mn0 + 4 * (tid >> 3);
            if (kk < Kend) {
                const float* src = P + (size_t)kk * ld + col;
                if (vec && col + 3 < MN) v = *reinterpret_cast<const float4*>(src);
                else {
                    if (col + 0 < MN) v.x = src[0];
                    if (col + 1 < MN) v.y = src[1];
                    if (col + 2 < MN) v.z = src[2];
                    if (col + 3 < MN) v.w = src[3];
                }
            }
        }
        st.v[i] = v;
    }
}

template <bool KC>
__device__ __forceinline__ void mask3(Stage3& st, int tid, int mn0, int k0, int MN, int Kend) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bool ok;
        if (KC) { const int idx = tid + 256 * i; ok = mn0 + kc_row(idx) < MN && k0 + 4 * kc_kq(idx) < Kend; }
        else ok = k0 + 4 * (tid & 7) + i < Kend && mn0 + 4 * (tid >> 3) < MN;
        st.v[i] = make_float4(ok ? st.v[i].x : 0.f, ok ? st.v[i].y : 0.f, ok ? st.v[i].z : 0.f, ok ? st.v[i].w : 0.f);
    }
}

template <bool KC>
__device__ __forceinline__ void store3(uint16_t* __restrict__ Thi, uint16_t* __restrict__ Tlo, int tid, const Stage3& st) {
    if (KC) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int mn = kc_row(idx), k = 4 * kc_kq(idx);
            uint2 hi, lo;
            split4(st.v[i].x, st.v[i].y, st.v[i].z, st.v[i].w, hi, lo);
            *reinterpret_cast<uint2*>(Thi + mn * LDK3 + k) = hi;
            *reinterpret_cast<uint2*>(Tlo + mn * LDK3 + k) = lo;
        }
    } else {
        const int k = 4 * (tid & 7), mn = 4 * (tid >> 3);
        const float* f0 = reinterpret_cast<const float*>(&st.v[0]);
        const float* f1 = reinterpret_cast<const float*>(&st.v[1]);
        const float* f2 = reinterpret_cast<const float*>(&st.v[2]);
        const float* f3 = reinterpret_cast<const float*>(&st.v[3]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint2 hi, lo;
            split4(f0[c], f1[c], f2[c], f3[c], hi, lo);
            *reinterpret_cast<uint2*>(Thi + (mn + c) * LDK3 + k) = hi;
            *reinterpret_cast<uint2*>(Tlo + (mn + c) * LDK3 + k) = lo;
        }
    }
}

template <bool TA, bool TB, bool FAST>
__global__ __launch_bounds__(256, 2) void gemm3_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    uint16_t* lds = reinterpret_cast<uint16_t*>(gsm);        // [buf][A_hi | A_lo | B_hi | B_lo][TILE3]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    int bid, zslab;
    decode_block(g, bid, zslab);
    const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = zslab * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // two register stages: while tile t is multiplied out of LDS, tile t+1 sits in registers and tile t+2 is
    // in flight, i.e. two K tiles of global loads are outstanding per thread (the kernel is bound by
    // bytes-in-flight x latency, not by the MFMA pipe)
    Stage3 a0, b0, a1, b1;
    const bool edge = m0 + BM > g.M || n0 + BN > g.N || ((kend - kbeg) & (BK3 - 1)) != 0;      // workgroup-uniform
    const bool want_cs = TA && g.colsum != nullptr && tn == 0;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int t, Stage3& sa, Stage3& sb) {
        const int k0 = kbeg + t * BK3;
        load3<!TA, FAST>(g.A, g.lda, m0, k0, g.M, kend, g.vecA, tid, sa);
        load3<TB, FAST>(g.B, g.ldb, n0, k0, g.N, kend, g.vecB, tid, sb);
    };
    // consume tile t's registers: mask (fast path), bias-gradient side sum, split + LDS store.  Nothing touches
    // the loaded values before this point, so the loads stay in flight across the MFMA block.
    auto stash = [&](int t, int buf, Stage3& sa, Stage3& sb) {
        // keep the compiler from hoisting the first use of the loaded registers (and with it their s_waitcnt)
        // above the MFMA block that is meant to hide the load latency
        __builtin_amdgcn_sched_barrier(0);
        const int k0 = kbeg + t * BK3;
        if (FAST && edge) {            // interior workgroups with whole K tiles load nothing out of range
            mask3<!TA>(sa, tid, m0, k0, g.M, kend);
            mask3<TB>(sb, tid, n0, k0, g.N, kend);
        }
        if (want_cs) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { csum.x += sa.v[i].x; csum.y += sa.v[i].y; csum.z += sa.v[i].z; csum.w += sa.v[i].w; }
        }
        uint16_t* nb = lds + buf * 4 * TILE3;
        store3<!TA>(nb + 0 * TILE3, nb + 1 * TILE3, tid, sa);
        store3<TB>(nb + 2 * TILE3, nb + 3 * TILE3, tid, sb);
    };
    auto multiply = [&](int buf) {
        const uint16_t* base = lds + buf * 4 * TILE3;
        const uint16_t* pa = base + (wm * 64 + l31) * LDK3 + 8 * hh;
        const uint16_t* pb = base + 2 * TILE3 + (wn * 64 + l31) * LDK3 + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * LDK3 + ks * 16);
                al[i] = *reinterpret_cast<const bf16x8*>(pa + TILE3 + i * 32 * LDK3 + ks * 16);
                bh[i] = *reinterpret_cast<const bf16x8*>(pb + i * 32 * LDK3 + ks * 16);
                bl[i] = *reinterpret_cast<const bf16x8*>(pb + TILE3 + i * 32 * LDK3 + ks * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    };
    const int nt = (kend - kbeg + BK3 - 1) / BK3;
    fetch(0, a0, b0);
    stash(0, 0, a0, b0);
    if (nt > 1) fetch(1, a0, b0);
    if (nt > 2) fetch(2, a1, b1);
    __syncthreads();
    for (int t = 0; t < nt; t += 2) {
        multiply(0);                                   // tile t;   regs0 = tile t+1, regs1 = tile t+2 (in flight)
        if (t + 1 < nt) stash(t + 1, 1, a0, b0);
        if (t + 3 < nt) fetch(t + 3, a0, b0);
        __syncthreads();
        if (t + 1 >= nt) break;
        multiply(1);                                   // tile t+1; regs1 = tile t+2, regs0 = tile t+3 (in flight)
        if (t + 2 < nt) stash(t + 2, 0, a1, b1);
        if (t + 4 < nt) fetch(t + 4, a1, b1);
        __syncthreads();
    }
    write_output(g, acc, m0, n0, wm, wn, l31, hh, zslab);
    if (want_cs) finish_colsum<true>(g, csum, gsm, tid, m0, zslab);
}

// ======================================================================================================
// 256 x 256 tile, 8 wavefronts (4 x 2, each 64 x 128: 128 accumulator VGPRs), one workgroup per CU.
// The 128 x 128 kernel above moves (128 + 128) * K * 4 bytes from L2 per 128 * 128 * K MACs; measured, every large
// product here drives the L2 at 9-10.7 TB/s of request bandwidth (TCC_REQ, profiles/r01_h_ablation_notes.md) - that,
// not the matrix pipe or HBM, is what the 128-wide tiles saturate.  A 256 x 256 tile halves the L2 bytes per MAC
// and cuts the LDS fragment reads per MFMA from 0.67 to 0.5.  Interior tiles and whole K tiles only (the host falls
// back to gemm3_kernel otherwise).  LDS: [buf][A_hi | A_lo | B_hi | B_lo], planes of 256 rows x 64 bytes (32 bf16),
// no padding: the four 16-byte chunks of a row are XOR-swizzled with (row >> 2) & 3, which makes the ds_read_b128
// fragment reads conflict-free for the instruction's lane groups; 2 x 64 KB.
#ifdef RLT_GEMM_ABL
#define RLT_GEMM_ABL_ RLT_GEMM_ABL
#else
#define RLT_GEMM_ABL_ 0
#endif
constexpr int BM2 = 256, BN2 = 256;
constexpr int PL2 = 256 * 32;
// physical LDS row of a logical tile row: bits 0 and 2 swapped, so that logical rows 4 apart (the two rows a 16-lane
// ds_write_b64 group covers, in both staging maps below) land in different 16-dword halves of the 32 store banks
__device__ __forceinline__ int prow2(int row) { return (row & ~5) | ((row & 1) << 2) | ((row >> 2) & 1); }
// byte-16 chunk swizzle on the PHYSICAL row: makes the ds_read_b128 fragment reads conflict-free
__device__ __forceinline__ int swz2(int prow, int chunk) { return chunk ^ ((prow >> 2) & 3); }
// K-contiguous staging map: element idx -> (row, 16-byte k group); 16 consecutive lanes = 8 k groups x rows {r, r+4}
__device__ __forceinline__ int kcb_row(int idx) { return (((idx >> 3) & 1) << 2) | ((idx >> 4) & 3) | ((idx >> 6) << 3); }

template <bool KC>
__device__ __forceinline__ void load3b(const float* __restrict__ P, int ld, int mn0, int k0, int tid, Stage3& st) {
    if (KC) {           // [MN][K]: element idx of the 2048 float4 of a 256 x 32 tile -> (row, 16-byte k group)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 512 * i;
            st.v[i] = *reinterpret_cast<const float4*>(P + (size_t)((mn0 + kcb_row(idx)) & ((RLT_GEMM_ABL_ & 128) ? 2047 : -1)) * ((RLT_GEMM_ABL_ & 64) ? 0 : ld) + k0 + 4 * (idx & 7));
        }
    } else {            // [K][MN]: thread owns a 4(k) x 4(mn) block, k block in the low lane bits
        const int kb = tid & 7, mb = tid >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            st.v[i] = *reinterpret_cast<const float4*>(P + (size_t)((k0 + 4 * kb + i) & ((RLT_GEMM_ABL_ & 128) ? 2047 : -1)) * ((RLT_GEMM_ABL_ & 64) ? 0 : ld) + mn0 + 4 * mb);
    }
}
// one quarter (`part` = 0..3) of the split + LDS store of a staged operand tile
template <bool KC>
__device__ __forceinline__ void store3b_part(uint16_t* __restrict__ Thi, uint16_t* __restrict__ Tlo, int tid, const Stage3& st, int part) {
    if (KC) {
        const int idx = tid + 512 * part;
        const int row = prow2(kcb_row(idx)), kq = idx & 7;
        const int off = row * 32 + 8 * swz2(row, kq >> 1) + 4 * (kq & 1);
        uint2 hi, lo;
        split4(st.v[part].x, st.v[part].y, st.v[part].z, st.v[part].w, hi, lo);
        *reinterpret_cast<uint2*>(Thi + off) = hi;
        *reinterpret_cast<uint2*>(Tlo + off) = lo;
    } else {
        const int kb = tid & 7, mb = tid >> 3;
        const float* f0 = reinterpret_cast<const float*>(&st.v[0]);
        const float* f1 = reinterpret_cast<const float*>(&st.v[1]);
        const float* f2 = reinterpret_cast<const float*>(&st.v[2]);
        const float* f3 = reinterpret_cast<const float*>(&st.v[3]);
        // offset of part 0; the other three follow from it by one xor-add (prow2 puts bit 0 of the part into row bit 2, which
        // is also the low bit of the row's swizzle key).  Kept opaque so that ONE offset register lives across the K loop:
        // hoisted as four, the fourth was spilled in the persistent NN kernel and its reload (a scratch load, in order
        // behind the tile prefetch) made every K tile wait for all loads in flight.
        int base = ((mb >> 1) * 8 + (mb & 1)) * 32 + 4 * (kb & 1) + 8 * ((kb >> 1) ^ (((mb >> 1) & 1) * 2));
        asm volatile("" : "+v"(base));
        const int off = (base ^ (8 * (part & 1))) + 128 * (part & 1) + 64 * (part >> 1);
        uint2 hi, lo;
        split4(f0[part], f1[part], f2[part], f3[part], hi, lo);
        *reinterpret_cast<uint2*>(Thi + off) = hi;
        *reinterpret_cast<uint2*>(Tlo + off) = lo;
    }
}
template <bool KC>
__device__ __forceinline__ void store3b(uint16_t* __restrict__ Thi, uint16_t* __restrict__ Tlo, int tid, const Stage3& st) {
#pragma unroll
    for (int part = 0; part < 4; ++part) store3b_part<KC>(Thi, Tlo, tid, st, part);
}

#if defined(RLT_STAMPS)
// timeline instrumentation (variant builds only): s_memtime at four points of one workgroup of gemm3b_kernel, per wavefront
__device__ unsigned long long rlt_gemm_stamp_buf[8 * 4];
#define RLT_GSTAMP(slot) do { if (blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0) \
    rlt_gemm_stamp_buf[(threadIdx.x >> 6) * 4 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define RLT_GSTAMP(slot) do {} while (0)
#endif

// PERSIST (no split-K, A stored [M][K], at least two K tiles, more tiles than workgroups): gridDim.x workgroups walk the
// tiles id, id + gridDim.x, ... as ONE stream of K tiles - the operand tiles of the next output tile are fetched and
// stashed during the last K tiles of the current one, and its epilogue stores go out while those loads are in flight.
// One workgroup per output tile paid the pipeline start-up (two exposed HBM latencies, ~6,000 cycles) and ran its first
// K tiles against cold loads for every tile: at K = 256 (8 K tiles) that was 57,000 cycles per tile for 24,600 of
// MFMA work (timeline stamps, profiles/r02_notes.md).
// timing-only ablation builds of the K loop (tools/build_variant.py ... -DRLT_GEMM_ABL=bits; results are wrong by design):
// 1 = no global loads / split / LDS stores inside the loop, 2 = no workgroup barrier per K tile, 4 = no fragment reads
// (the MFMAs run on the fragments of the first step), 8 = no output, 16 = no global loads inside the loop (the split + LDS
// stores run on stale registers), 32 = global loads, but no split / LDS stores (the registers are only consumed), 64 = every
// row of an operand tile aliases row 0 / k row 0 (same instructions, all loads served by the L1), 128 = operand rows taken
// modulo 2048 (distinct lines per instruction as in the product, but a 2-16 MB footprint that stays in the L2 / MALL)
#ifndef RLT_GEMM_ABL
#define RLT_GEMM_ABL 0
#endif
template <bool TA, bool TB, bool PERSIST = false>
__global__ __launch_bounds__(512) void gemm3b_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    uint16_t* lds = reinterpret_cast<uint16_t*>(gsm);
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    int bid, zslab;
    decode_block(g, bid, zslab);                     // g.tiles_* count 256-wide tiles here
    const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
    int m0 = tm * BM2, n0 = tn * BN2;
    const int kbeg = zslab * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Stage3 sa, sb;
    const bool want_cs = TA && g.colsum != nullptr && tn == 0;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int t) {
        const int k0 = kbeg + t * BK3;
        load3b<!TA>(g.A, g.lda, m0, k0, tid, sa);
        load3b<TB>(g.B, g.ldb, n0, k0, tid, sb);
    };
    auto stash = [&](int buf) {
        if (want_cs) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { csum.x += sa.v[i].x; csum.y += sa.v[i].y; csum.z += sa.v[i].z; csum.w += sa.v[i].w; }
        }
        uint16_t* nb = lds + buf * 4 * PL2;
        store3b<!TA>(nb, nb + PL2, tid, sa);
        store3b<TB>(nb + 2 * PL2, nb + 3 * PL2, tid, sb);
    };
    // fragment rows: row bases are multiples of 32, so the physical row is base + prow2(l31) and its swizzle key is fixed
    const int pl = prow2(l31);
    const int sw = (pl >> 2) & 3;
    // 8 steps (k-step ks, 32-column block j) of 6 MFMAs; the B fragments of the next step (and the A fragments of the
    // next k-step) are read while the current step multiplies.  The scheduling fences keep the compiler from hoisting
    // all 24 fragment reads to the top (96 more live VGPRs than the 256 available).
    // The split + LDS store of the NEXT tile (registers sa, sb) is spread over steps 4..7, two quarters per step, right
    // behind that step's MFMAs: the wavefront issues the VALU work while its MFMAs execute.  With the stash after the
    // whole multiply, the two wavefronts of a SIMD (same workgroup, in lock-step between the barriers) both sit in their
    // MFMA phase, then both in their VALU phase, and the two pipes take turns.  The global loads of tile t+2 follow the
    // last use of each staging register (A after step 5, B after step 7): a full iteration of latency cover.
    // tile of the stream after the current one (PERSIST)
    int vid = blockIdx.x, nm0 = m0, nn0 = n0;
    bool has_next = false;
    auto decode_next = [&]() {
        const int nwg = g.tiles_m * g.tiles_n;
        vid += gridDim.x;
        has_next = PERSIST && vid < nwg;
        if (has_next) {
            const int q = nwg >> 3, r = nwg & 7, xcd = vid & 7, jj = vid >> 3;          // decode_block's XCD-aware remap
            const int nb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + jj;
            const int ntm = nb / g.tiles_n;
            nm0 = ntm * BM2; nn0 = (nb - ntm * g.tiles_n) * BN2;
        }
    };
    auto multiply = [&](int buf, int t, int nt_) {
        const uint16_t* pa = lds + buf * 4 * PL2 + (wm * 64 + pl) * 32;
        const uint16_t* pb = lds + buf * 4 * PL2 + 2 * PL2 + (wn * 128 + pl) * 32;
        uint16_t* nb = lds + (buf ^ 1) * 4 * PL2;
        // the fetch of tile t+2 is unconditional (the last two iterations re-read the last tile into registers nobody
        // consumes): with the loads behind a branch hipcc merges the outstanding-load counts of both paths and waits
        // for the loads it has just issued (vmcnt(3..0) at every staging step instead of vmcnt(7..4)), which exposed
        // one full memory latency per 32-wide K step
        const bool do_stash = t + 1 < nt_ || has_next;
        // K tile t+2 of the stream: of this output tile, or K tile t+2-nt of the next one
        const bool wrap = PERSIST && has_next && t + 2 >= nt_;
        const int kf = kbeg + (wrap ? t + 2 - nt_ : min(t + 2, nt_ - 1)) * BK3;
        const int fm0 = wrap ? nm0 : m0, fn0 = wrap ? nn0 : n0;
        bf16x8 ah[2][2], al[2][2], bh[2], bl[2];
        auto load_a = [&](int ks, int slot) {
            const int co = 8 * ((2 * ks + hh) ^ sw);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[slot][i] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * 32 + co);
                al[slot][i] = *reinterpret_cast<const bf16x8*>(pa + PL2 + i * 32 * 32 + co);
            }
        };
        auto load_b = [&](int ks, int j, int slot) {
            const int co = 8 * ((2 * ks + hh) ^ sw);
            bh[slot] = *reinterpret_cast<const bf16x8*>(pb + j * 32 * 32 + co);
            bl[slot] = *reinterpret_cast<const bf16x8*>(pb + PL2 + j * 32 * 32 + co);
        };
        load_a(0, 0);
        load_b(0, 0, 0);
#pragma unroll
        for (int step = 0; step < 8; ++step) {
            const int ks = step >> 2, j = step & 3;
            if (!(RLT_GEMM_ABL & 4)) {
                if (step + 1 < 8) load_b((step + 1) >> 2, (step + 1) & 3, (step + 1) & 1);
                if (step == 2) load_a(1, 1);
            }
            constexpr int KS = (RLT_GEMM_ABL & 4) ? 0 : -1, BS = (RLT_GEMM_ABL & 4) ? 0 : -1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[KS < 0 ? ks : 0][i], bh[BS < 0 ? (step & 1) : 0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[KS < 0 ? ks : 0][i], bl[BS < 0 ? (step & 1) : 0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[KS < 0 ? ks : 0][i], bh[BS < 0 ? (step & 1) : 0], acc[i][j], 0, 0, 0);
            }
            if ((RLT_GEMM_ABL & 32) && step == 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    asm volatile("" :: "v"(sa.v[i].x), "v"(sa.v[i].y), "v"(sa.v[i].z), "v"(sa.v[i].w), "v"(sb.v[i].x), "v"(sb.v[i].y),
                                 "v"(sb.v[i].z), "v"(sb.v[i].w));
            }
            if (step >= 4 && do_stash && !(RLT_GEMM_ABL & (1 | 32))) {
                if (step == 4 && want_cs) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { csum.x += sa.v[i].x; csum.y += sa.v[i].y; csum.z += sa.v[i].z; csum.w += sa.v[i].w; }
                }
                if (step < 6) {
                    store3b_part<!TA>(nb, nb + PL2, tid, sa, 2 * (step - 4));
                    store3b_part<!TA>(nb, nb + PL2, tid, sa, 2 * (step - 4) + 1);
                } else {
                    store3b_part<TB>(nb + 2 * PL2, nb + 3 * PL2, tid, sb, 2 * (step - 6));
                    store3b_part<TB>(nb + 2 * PL2, nb + 3 * PL2, tid, sb, 2 * (step - 6) + 1);
                }
            }
            if (step == 5 && !(RLT_GEMM_ABL & (1 | 16))) load3b<!TA>(g.A, g.lda, fm0, kf, tid, sa);
            if (step == 7 && !(RLT_GEMM_ABL & (1 | 16))) load3b<TB>(g.B, g.ldb, fn0, kf, tid, sb);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int nt = (kend - kbeg) / BK3;
    RLT_GSTAMP(0);
    fetch(0);
    stash(0);
    fetch(min(1, nt - 1));                             // unconditional, like the fetches in multiply()
    __syncthreads();
    RLT_GSTAMP(1);
    if (PERSIST) decode_next();
    int u = 0;                                         // K tiles of the stream so far (LDS buffer parity)
    while (true) {
        for (int t = 0; t < nt; ++t, ++u) {
            multiply(u & 1, t, nt);                    // K tile t; stashes the next K tile of the stream (in registers), fetches the one after
            if (!(RLT_GEMM_ABL & 2)) __syncthreads();
        }
        RLT_GSTAMP(2);
        if (!(RLT_GEMM_ABL & 8) || acc[0][0][0] == 123.456f) write_output_t<4>(g, acc, m0 + wm * 64, n0 + wn * 128, true, l31, hh, zslab);
        RLT_GSTAMP(3);
        if (!PERSIST || !has_next) break;
        m0 = nm0; n0 = nn0;
        decode_next();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    if (want_cs) {                                     // 8 threads (tid = 8*mb + kb) hold partial sums of columns 4*mb..+3
        float4* red = reinterpret_cast<float4*>(gsm);
        red[tid] = csum;
        __syncthreads();
        if (tid < 64) {
            float4 t = red[8 * tid];
#pragma unroll
            for (int j = 1; j < 8; ++j) { const float4 o = red[8 * tid + j]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
            float* dst = (g.cs_slab ? g.cs_slab + (size_t)zslab * g.M : g.colsum) + m0 + 4 * tid;
            dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
        }
    }
}

// the 256-wide tile needs: split-bf16 mode, M % 256 == 0, N % 256 == 0, whole 32-wide K tiles per slab, the branch-free
// loader preconditions (16-byte aligned operands, leading dimensions multiples of 4)
bool gemm_big_ok(const GemmArgs& g, bool ta, bool tb) {
    static const int off = [] { const char* e = getenv("RLT_GEMM_NO_BIG"); return e ? atoi(e) : 0; }();
    if (off) return false;
    if (!g.vecA || !g.vecB) return false;
    if ((g.M % BM2) || (g.N % BN2) || (g.K % BK3) || (g.kchunk % BK3)) return false;
    (void)ta; (void)tb;
    return true;
}
template <bool TA, bool TB>
int launch_gemm3b(GemmArgs g, int ns, hipStream_t st) {
    const size_t shm = (size_t)2 * 4 * PL2 * sizeof(uint16_t);
    g.tiles_m = g.M / BM2; g.tiles_n = g.N / BN2;
    if constexpr (!TA) {
        // measured: NT -6 % at K = 256, -2 % at K = 2048.  NN (RLT_GEMM_PERSIST_NN=1): -7 % / -2 % alone with a bias epilogue
        // - once its K loop no longer reloaded a spilled LDS offset (store3b_part: a scratch load in order behind the tile
        // prefetch had cost +20 %) - but the training step, whose dX products accumulate in place, is 0.4 ms slower with it
        static const int persist_wgs = [] { const char* e = getenv("RLT_GEMM_PERSIST"); return e ? atoi(e) : 256; }();   // 0: off
        static const int persist_nn = [] { const char* e = getenv("RLT_GEMM_PERSIST_NN"); return e ? atoi(e) : 0; }();
        const long long tiles = (long long)g.tiles_m * g.tiles_n;
        const bool want = TB || (persist_nn && !g.bits_in);
        if (want && persist_wgs > 0 && ns == 1 && g.K / BK3 >= 2 && tiles > persist_wgs && persist_wgs % 8 == 0) {
            int rc = rlt_allow_lds(gemm3b_kernel<TA, TB, true>, shm);
            if (rc) return rc;
            hipLaunchKernelGGL((gemm3b_kernel<TA, TB, true>), dim3(persist_wgs), dim3(512), shm, st, g);
            return RLT_LAUNCH_RESULT();
        }
    }
    int rc = rlt_allow_lds(gemm3b_kernel<TA, TB>, shm);
    if (rc) return rc;
    dim3 grid(g.tiles_m * g.tiles_n * (g.slab_xcd ? ns : 1), 1, g.slab_xcd ? 1 : ns);
    hipLaunchKernelGGL((gemm3b_kernel<TA, TB>), grid, dim3(512), shm, st, g);
    return RLT_LAUNCH_RESULT();
}

// ======================================================================================================
// "bf16x6": fp32-FAITHFUL products on the bf16 matrix pipe.  Every fp32 operand is split EXACTLY into three bf16 values,
// x = h + m + l (h = bf16(x), m = bf16(x - h), l = x - h - m: 8 + 8 + 8 significand bits, the last residual is exactly
// representable), and a*b is evaluated as the six products  h*h' + h*m' + m*h' + h*l' + l*h' + m*m'  on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Each of them is exact in the matrix pipe (8 x 8 bits); what is dropped
// (m*l' + l*m' + l*l') is below 2^-23 |a b| in the worst case (|m| <= 2^-8 |x|, |l| <= 2^-16 |x|; over random operands the largest is
// 2^-24.3, the median 2^-29: tests/test_split6.py) - under one fp32 ulp of the product, of the size of the rounding of the fp32
// accumulation itself (2^-24 of the running sum) that the f32 MFMA kernel above pays as well.  So the result carries the
// full 24 operand bits, where bf16x3 carries 16: the mode is held to the EXACT-FP32 tolerances in the tests.  Six
// products cost 2500 / 6 = 417 TFLOP/s of fp32-level peak against 157.3 of the f32 MFMA.
// 256 x 128 tile, 8 wavefronts (4 x 2, each 64 x 64: 64 accumulator VGPRs), K tiles of 32; LDS: two buffers of
// [A_h | A_m | A_l] (256 rows) and [B_h | B_m | B_l] (128 rows), planes in gemm3b's layout (64-byte rows, permuted and
// chunk-swizzled), 2 x 72 KB.  With twice the matrix work per operand byte of the bf16x3 kernel the K loop is bound by
// the matrix pipe, not by the L2 -> CU path that bounds gemm3b.
constexpr int BM6 = 256, BN6 = 128;
constexpr int PLA6 = 256 * 32, PLB6 = 128 * 32;          // bf16 elements per plane
constexpr int BUF6 = 3 * PLA6 + 3 * PLB6;                // per LDS buffer

__device__ __forceinline__ void split4x3(float a, float b, float c, float d, uint2& hi, uint2& mid, uint2& lo) {
    hi.x = pack_bf16x2(a, b);
    hi.y = pack_bf16x2(c, d);
    asm("" : "+v"(hi.x), "+v"(hi.y));          // as in split4: keep the packed pair, do not re-convert
    const float ra = a - __builtin_bit_cast(float, hi.x << 16), rb = b - __builtin_bit_cast(float, hi.x & 0xffff0000u);
    const float rc = c - __builtin_bit_cast(float, hi.y << 16), rd = d - __builtin_bit_cast(float, hi.y & 0xffff0000u);
    mid.x = pack_bf16x2(ra, rb);
    mid.y = pack_bf16x2(rc, rd);
    asm("" : "+v"(mid.x), "+v"(mid.y));
    lo.x = pack_bf16x2(ra - __builtin_bit_cast(float, mid.x << 16), rb - __builtin_bit_cast(float, mid.x & 0xffff0000u));
    lo.y = pack_bf16x2(rc - __builtin_bit_cast(float, mid.y << 16), rd - __builtin_bit_cast(float, mid.y & 0xffff0000u));
}

// B operand tile: 128 (n) x 32 (k).  K-contiguous ([N][K]): 1024 float4, two per thread; N-contiguous ([K][N]): 4 x 4
// blocks like load3b, 256 threads (wavefronts 0..3) hold one each
struct Stage6B { float4 v[4]; };
template <bool KC>
__device__ __forceinline__ void load6b(const float* __restrict__ P, int ld, int n0, int k0, int tid, Stage6B& st) {
    if (KC) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 512 * i;
            st.v[i] = *reinterpret_cast<const float4*>(P + (size_t)(n0 + kcb_row(idx)) * ld + k0 + 4 * (idx & 7));
        }
    } else if (tid < 256) {
        const int kb = tid & 7, mb = tid >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            st.v[i] = *reinterpret_cast<const float4*>(P + (size_t)(k0 + 4 * kb + i) * ld + n0 + 4 * mb);
    }
}
// one part of the three-way split + LDS store of a staged tile; planes Th / Tm / Tl.  KC: part = which float4 of the thread;
// !KC: part = which of the 4 columns of the thread's 4 x 4 block
template <bool KC>
__device__ __forceinline__ void store6_part(uint16_t* __restrict__ Th, uint16_t* __restrict__ Tm, uint16_t* __restrict__ Tl,
                                            int tid, const float4 (&v)[4], int part) {
    uint2 hi, mid, lo;
    int off;
    if (KC) {
        const int idx = tid + 512 * part;
        const int row = prow2(kcb_row(idx)), kq = idx & 7;
        off = row * 32 + 8 * swz2(row, kq >> 1) + 4 * (kq & 1);
        split4x3(v[part].x, v[part].y, v[part].z, v[part].w, hi, mid, lo);
    } else {
        const int kb = tid & 7, mb = tid >> 3;
        const float* f0 = reinterpret_cast<const float*>(&v[0]);
        const float* f1 = reinterpret_cast<const float*>(&v[1]);
        const float* f2 = reinterpret_cast<const float*>(&v[2]);
        const float* f3 = reinterpret_cast<const float*>(&v[3]);
        int base = ((mb >> 1) * 8 + (mb & 1)) * 32 + 4 * (kb & 1) + 8 * ((kb >> 1) ^ (((mb >> 1) & 1) * 2));
        asm volatile("" : "+v"(base));
        off = (base ^ (8 * (part & 1))) + 128 * (part & 1) + 64 * (part >> 1);
        split4x3(f0[part], f1[part], f2[part], f3[part], hi, mid, lo);
    }
    *reinterpret_cast<uint2*>(Th + off) = hi;
    *reinterpret_cast<uint2*>(Tm + off) = mid;
    *reinterpret_cast<uint2*>(Tl + off) = lo;
}

template <bool TA, bool TB>
__global__ __launch_bounds__(512) void gemm6_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    uint16_t* lds = reinterpret_cast<uint16_t*>(gsm);
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    int bid, zslab;
    decode_block(g, bid, zslab);                     // g.tiles_* count 256 x 128 tiles here
    const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
    const int m0 = tm * BM6, n0 = tn * BN6;
    const int kbeg = zslab * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // two staging register sets: K tile t+1 (loaded one whole iteration ago) is split and stored to LDS during iteration
    // t while the loads of K tile t+2 go into the other set at the START of iteration t - a full iteration of cover for
    // the memory latency (with one set, loaded at the end of an iteration and consumed at the start of the next, every K
    // tile waited for its loads: 55 % of the matrix pace)
    Stage3 sa0, sa1;
    Stage6B sb0, sb1;
    constexpr bool AKC = !TA, BKC = TB;              // operand stored with K contiguous
    const bool b_active = BKC || tid < 256;          // wave-uniform
    const bool want_cs = TA && g.colsum != nullptr && tn == 0;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add_cs = [&](const Stage3& sa) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { csum.x += sa.v[i].x; csum.y += sa.v[i].y; csum.z += sa.v[i].z; csum.w += sa.v[i].w; }
    };
    auto planes_a = [&](int buf) { return lds + buf * BUF6; };
    auto planes_b = [&](int buf) { return lds + buf * BUF6 + 3 * PLA6; };
    auto stash_a = [&](int buf, int part, const Stage3& sa) {
        uint16_t* pa_ = planes_a(buf);
        store6_part<AKC>(pa_, pa_ + PLA6, pa_ + 2 * PLA6, tid, sa.v, part);
    };
    auto stash_b = [&](int buf, int part, const Stage6B& sb) {
        uint16_t* pb_ = planes_b(buf);
        if (b_active) store6_part<BKC>(pb_, pb_ + PLB6, pb_ + 2 * PLB6, tid, sb.v, part);
    };
    constexpr int BPARTS = BKC ? 2 : 4;
    const int pl = prow2(l31);
    const int sw = (pl >> 2) & 3;
    const int nt = (kend - kbeg) / BK3;

    // prologue: K tile 0 into LDS buffer 0, K tile 1 into staging set 0
    load3b<AKC>(g.A, g.lda, m0, kbeg, tid, sa0);
    load6b<BKC>(g.B, g.ldb, n0, kbeg, tid, sb0);
    if (want_cs) add_cs(sa0);
#pragma unroll
    for (int part = 0; part < 4; ++part) stash_a(0, part, sa0);
#pragma unroll
    for (int part = 0; part < BPARTS; ++part) stash_b(0, part, sb0);
    {
        const int k1 = kbeg + min(1, nt - 1) * BK3;
        load3b<AKC>(g.A, g.lda, m0, k1, tid, sa0);
        load6b<BKC>(g.B, g.ldb, n0, k1, tid, sb0);
    }
    __syncthreads();

    // iteration t: multiply K tile t (LDS buffer t & 1); split + store K tile t+1 from (sa_st, sb_st) into the other
    // buffer; fetch K tile t+2 into (sa_ld, sb_ld)
    auto iteration = [&](int t, const Stage3& sa_st, const Stage6B& sb_st, Stage3& sa_ld, Stage6B& sb_ld) {
        const int buf = t & 1;
        const uint16_t* pa = planes_a(buf) + (wm * 64 + pl) * 32;
        const uint16_t* pb = planes_b(buf) + (wn * 64 + pl) * 32;
        const bool do_stash = t + 1 < nt;
        const int kf = kbeg + min(t + 2, nt - 1) * BK3;          // unconditional fetch (see gemm3b_kernel)
        load3b<AKC>(g.A, g.lda, m0, kf, tid, sa_ld);
        load6b<BKC>(g.B, g.ldb, n0, kf, tid, sb_ld);
        bf16x8 a[2][2][3], b[2][3];                              // [slot][i][h|m|l], [slot][h|m|l]
        auto load_a = [&](int ks, int slot) {
            const int co = 8 * ((2 * ks + hh) ^ sw);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) a[slot][i][q] = *reinterpret_cast<const bf16x8*>(pa + q * PLA6 + i * 32 * 32 + co);
        };
        auto load_b = [&](int ks, int j, int slot) {
            const int co = 8 * ((2 * ks + hh) ^ sw);
#pragma unroll
            for (int q = 0; q < 3; ++q) b[slot][q] = *reinterpret_cast<const bf16x8*>(pb + q * PLB6 + j * 32 * 32 + co);
        };
        load_a(0, 0);
        load_b(0, 0, 0);
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            const int ks = step >> 1, j = step & 1;
            if (step + 1 < 4) load_b((step + 1) >> 1, (step + 1) & 1, (step + 1) & 1);
            if (step == 0) load_a(1, 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x16 c = acc[i][j];
                // smallest terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i][1], b[step & 1][1], c, 0, 0, 0);   // m m'
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i][2], b[step & 1][0], c, 0, 0, 0);   // l h'
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i][0], b[step & 1][2], c, 0, 0, 0);   // h l'
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i][1], b[step & 1][0], c, 0, 0, 0);   // m h'
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i][0], b[step & 1][1], c, 0, 0, 0);   // h m'
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i][0], b[step & 1][0], c, 0, 0, 0);   // h h'
                acc[i][j] = c;
            }
            // split + LDS store of the next K tile behind this step's MFMAs: A part `step`, B parts in steps 2 and 3
            if (do_stash) {
                if (step == 0 && want_cs) add_cs(sa_st);
                stash_a(buf ^ 1, step, sa_st);
                if (step >= 2) {
#pragma unroll
                    for (int q = 0; q < BPARTS / 2; ++q) stash_b(buf ^ 1, (step - 2) * (BPARTS / 2) + q, sb_st);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int t = 0; t < nt; t += 2) {
        iteration(t, sa0, sb0, sa1, sb1);
        __syncthreads();
        if (t + 1 < nt) {
            iteration(t + 1, sa1, sb1, sa0, sb0);
            __syncthreads();
        }
    }
    write_output_t<2>(g, acc, m0 + wm * 64, n0 + wn * 64, true, l31, hh, zslab);
    if (want_cs) {                                     // 8 threads (tid = 8*mb + kb) hold partial sums of columns 4*mb..+3
        float4* red = reinterpret_cast<float4*>(gsm);
        __syncthreads();
        red[tid] = csum;
        __syncthreads();
        if (tid < 64) {
            float4 t = red[8 * tid];
#pragma unroll
            for (int j = 1; j < 8; ++j) { const float4 o = red[8 * tid + j]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
            float* dst = (g.cs_slab ? g.cs_slab + (size_t)zslab * g.M : g.colsum) + m0 + 4 * tid;
            dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
        }
    }
}

// 256 x 256 tile of the same six-product scheme (N % 256 == 0).  The 256 x 128 kernel above asks the L2 for 384 cache
// lines per 48 MFMAs of a wavefront and runs at ~48 % of the matrix pace however far ahead its loads are issued (one or
// two staging register sets: same time) - like gemm3b it is bound by the number of lines a CU can have in flight, not by
// latency cover.  This tile asks for 512 lines per 96 MFMAs.  Six planes of gemm3b's layout (256 rows x 64 bytes each,
// 96 KB) leave no room for a second LDS buffer, so the K loop is: barrier, split + store K tile t from the staging
// registers, barrier, issue the loads of K tile t+1 (a whole multiply of cover), 96 MFMAs per wavefront.  The split is
// not hidden behind MFMAs (both wavefronts of a SIMD sit in it together): ~15 % of the loop, against the 2x it buys.
template <bool TA, bool TB>
__global__ __launch_bounds__(512) void gemm6b_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    uint16_t* lds = reinterpret_cast<uint16_t*>(gsm);          // [A_h | A_m | A_l | B_h | B_m | B_l], PL2 elements each
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    int bid, zslab;
    decode_block(g, bid, zslab);                     // g.tiles_* count 256 x 256 tiles here
    const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
    const int m0 = tm * BM2, n0 = tn * BN2;
    const int kbeg = zslab * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Stage3 sa, sb;
    constexpr bool AKC = !TA, BKC = TB;
    const bool want_cs = TA && g.colsum != nullptr && tn == 0;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    const int pl = prow2(l31);
    const int sw = (pl >> 2) & 3;
    const int nt = (kend - kbeg) / BK3;
    const uint16_t* pa = lds + (wm * 64 + pl) * 32;
    const uint16_t* pb = lds + 3 * PL2 + (wn * 128 + pl) * 32;

    load3b<AKC>(g.A, g.lda, m0, kbeg, tid, sa);
    load3b<BKC>(g.B, g.ldb, n0, kbeg, tid, sb);
    for (int t = 0; t < nt; ++t) {
        if (t > 0) __syncthreads();                  // every wavefront has read the fragments of K tile t-1
        if (want_cs) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { csum.x += sa.v[i].x; csum.y += sa.v[i].y; csum.z += sa.v[i].z; csum.w += sa.v[i].w; }
        }
#pragma unroll
        for (int part = 0; part < 4; ++part) {
            store6_part<AKC>(lds, lds + PL2, lds + 2 * PL2, tid, sa.v, part);
            store6_part<BKC>(lds + 3 * PL2, lds + 4 * PL2, lds + 5 * PL2, tid, sb.v, part);
        }
        __syncthreads();
        {                                            // unconditional (the last iteration re-reads the last K tile)
            const int kf = kbeg + min(t + 1, nt - 1) * BK3;
            load3b<AKC>(g.A, g.lda, m0, kf, tid, sa);
            load3b<BKC>(g.B, g.ldb, n0, kf, tid, sb);
        }
        bf16x8 a[2][3], b[2][3];                     // [i][h|m|l], [slot][h|m|l]
        auto load_a = [&](int ks) {
            const int co = 8 * ((2 * ks + hh) ^ sw);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) a[i][q] = *reinterpret_cast<const bf16x8*>(pa + q * PL2 + i * 32 * 32 + co);
        };
        auto load_b = [&](int ks, int j, int slot) {
            const int co = 8 * ((2 * ks + hh) ^ sw);
#pragma unroll
            for (int q = 0; q < 3; ++q) b[slot][q] = *reinterpret_cast<const bf16x8*>(pb + q * PL2 + j * 32 * 32 + co);
        };
        load_a(0);
        load_b(0, 0, 0);
#pragma unroll
        for (int step = 0; step < 8; ++step) {
            const int j = step & 3;
            if (step + 1 < 8) load_b((step + 1) >> 2, (step + 1) & 3, (step + 1) & 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x16 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[step & 1][1], c, 0, 0, 0);   // m m'
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[step & 1][0], c, 0, 0, 0);   // l h'
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[step & 1][2], c, 0, 0, 0);   // h l'
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[step & 1][0], c, 0, 0, 0);   // m h'
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[step & 1][1], c, 0, 0, 0);   // h m'
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[step & 1][0], c, 0, 0, 0);   // h h'
                acc[i][j] = c;
            }
            if (step == 3) load_a(1);                // the A fragments of k-step 1 (their registers are free now)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    write_output_t<4>(g, acc, m0 + wm * 64, n0 + wn * 128, true, l31, hh, zslab);
    if (want_cs) {                                   // 8 threads (tid = 8*mb + kb) hold partial sums of columns 4*mb..+3
        float4* red = reinterpret_cast<float4*>(gsm);
        __syncthreads();
        red[tid] = csum;
        __syncthreads();
        if (tid < 64) {
            float4 t = red[8 * tid];
#pragma unroll
            for (int j = 1; j < 8; ++j) { const float4 o = red[8 * tid + j]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
            float* dst = (g.cs_slab ? g.cs_slab + (size_t)zslab * g.M : g.colsum) + m0 + 4 * tid;
            dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
        }
    }
}
template <bool TA, bool TB>
int launch_gemm6b(GemmArgs g, int ns, hipStream_t st) {
    const size_t shm = (size_t)6 * PL2 * sizeof(uint16_t);
    g.tiles_m = g.M / BM2; g.tiles_n = g.N / BN2;
    int rc = rlt_allow_lds(gemm6b_kernel<TA, TB>, shm);
    if (rc) return rc;
    dim3 grid(g.tiles_m * g.tiles_n * (g.slab_xcd ? ns : 1), 1, g.slab_xcd ? 1 : ns);
    hipLaunchKernelGGL((gemm6b_kernel<TA, TB>), grid, dim3(512), shm, st, g);
    return RLT_LAUNCH_RESULT();
}

// ======================================================================================================
// gemm6c: the 256 x 256 six-product tile with the split of the next K tile BEHIND the MFMAs.
// gemm6b above holds one K tile of 32 as six 16 KB planes (96 KB: no room for a second copy) and alternates a multiply
// phase with a split phase in which both wavefronts of every SIMD do vector work only (~15-20 % of its K loop).  Here the
// unit is the MFMA k-step: a k-step buffer is [A_h | A_m | A_l | B_h | B_m | B_l] x 256 rows x 16 k = 48 KB, THREE of them
// rotate (144 KB), and the loop runs in slots of one k-step (48 MFMAs per wavefront, one barrier):
//   slot u multiplies k-step u out of buffer u % 3;
//   an ODD slot also splits the staged register tile (a K tile of 32 = the next two k-steps) into buffers (u+1) % 3 and
//   (u+2) % 3 - both free: their last readers ran before the barrier that opened this slot - piece by piece behind its MFMA
//   groups, then issues the global loads of the register tile after that one (consumed two slots later: one whole slot plus
//   of latency cover from a single staging set);
//   an EVEN slot only multiplies.
// Rows are 32 bytes (16 bf16): logical row r sits at physical row phys6(r) (the two low 2-bit fields swapped) and its two
// 16-byte chunks are XOR-swizzled with bit 3 of the physical row, which makes the ds_read_b128 fragment reads and the
// ds_write_b64 stores of BOTH staging maps conflict-free (derivation: DESIGN.md 4.2).
// Staging maps (every wave-level load covers whole 128-byte lines):
//   K-contiguous operand: the 8 lanes tid & 7 read the 8 float4 of one row's K tile; lanes 0-3 of a row group hold k-step 0
//   of the tile in their loads 0 / 2 and k-step 1 in their loads 1 / 3, lanes 4-7 the other way round (kq = (tid & 7) ^ 4 i):
//   one select per register pair at stash time puts k-step 0 in registers 0 / 2 for every lane (8 v_cndmask per float4 pair -
//   against half-line loads, which double the L2 requests the 256-wide tiles were built to save);
//   MN-contiguous operand: wavefronts 0-3 hold the 16 k rows of k-step 0 (a 4 x 4 block per thread), wavefronts 4-7 those of
//   k-step 1 (wavefronts w and w + 4 share a SIMD, so every SIMD splits the same amount).
// PERSIST (no split-K, A stored [M][K], more tiles than workgroups): the workgroups walk the output tiles as ONE stream of
// k-steps, the first K tile of the next output tile is staged in the last odd slot of the current one and its second K tile
// is in flight during the epilogue - at K = 256 (8 K tiles: five of the eight big products of the encoder layer) the
// prologue and the 256 KB epilogue of a tile were uncovered time.
constexpr int HB6 = 256 * 16;                  // bf16 elements of one plane of a k-step buffer
constexpr int KB6 = 6 * HB6;                   // one k-step buffer (48 KB)
__device__ __forceinline__ int phys6(int r) { return (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3); }

struct Stage6c { float4 v[4]; };
template <bool KC>
__device__ __forceinline__ void load6c(const float* __restrict__ P, int ld, int mn0, int k0, int tid, Stage6c& st) {
    if (KC) {
        const int t = tid >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pr = (t & 1) | ((i & 1) << 1) | ((t >> 1) << 2) | ((i >> 1) << 7);       // physical LDS row
            const int kq = (tid & 7) ^ ((i & 1) << 2);
            st.v[i] = *reinterpret_cast<const float4*>(P + (size_t)(mn0 + phys6(pr)) * ld + k0 + 4 * kq);
        }
    } else {
        const int half = tid >> 8, t2 = tid & 255, kb = t2 & 3, mb = t2 >> 2;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            st.v[i] = *reinterpret_cast<const float4*>(P + (size_t)(k0 + 16 * half + 4 * kb + i) * ld + mn0 + 4 * mb);
    }
}
// piece `part` (0..3) of the split + LDS store of a staged operand tile.  T0 / T1: the h plane of this operand in the
// buffers of the tile's k-step 0 / k-step 1 (the m and l planes follow at + HB6, + 2 HB6).
template <bool KC>
__device__ __forceinline__ void stash6c_part(uint16_t* __restrict__ T0, uint16_t* __restrict__ T1, int tid, const Stage6c& st, int part) {
    uint2 hi, mid, lo;
    uint16_t* dst;
    if (KC) {
        // part -> register pair (part >> 1), k-step (part & 1) of the tile
        const int hb = (tid >> 2) & 1, t = tid >> 3, j = 2 * (part >> 1) + (part & 1);       // register index after the swap
        const float4 a = st.v[2 * (part >> 1)], b = st.v[2 * (part >> 1) + 1];
        const bool second = ((part & 1) ^ hb) != 0;                                          // this lane's k-step `part & 1` sits in load j ^ hb
        const float4 x = make_float4(second ? b.x : a.x, second ? b.y : a.y, second ? b.z : a.z, second ? b.w : a.w);
        const int pr = (t & 1) | (((j ^ hb) & 1) << 1) | ((t >> 1) << 2) | ((j >> 1) << 7);
        const int kq4 = tid & 3;
        dst = ((part & 1) ? T1 : T0) + pr * 16 + 8 * ((kq4 >> 1) ^ ((pr >> 3) & 1)) + 4 * (kq4 & 1);
        split4x3(x.x, x.y, x.z, x.w, hi, mid, lo);
    } else {
        const int half = tid >> 8, t2 = tid & 255, kb = t2 & 3, mb = t2 >> 2;
        const float* f0 = reinterpret_cast<const float*>(&st.v[0]);
        const float* f1 = reinterpret_cast<const float*>(&st.v[1]);
        const float* f2 = reinterpret_cast<const float*>(&st.v[2]);
        const float* f3 = reinterpret_cast<const float*>(&st.v[3]);
        const int pr = ((4 * mb) & ~15) | (part << 2) | (mb & 3);                             // phys6(4 mb + part)
        dst = (half ? T1 : T0) + pr * 16 + 8 * ((kb >> 1) ^ ((pr >> 3) & 1)) + 4 * (kb & 1);
        split4x3(f0[part], f1[part], f2[part], f3[part], hi, mid, lo);
    }
    *reinterpret_cast<uint2*>(dst) = hi;
    *reinterpret_cast<uint2*>(dst + HB6) = mid;
    *reinterpret_cast<uint2*>(dst + 2 * HB6) = lo;
}

#if defined(RLT_STAMPS)
// timeline instrumentation (variant builds only, tools/bench_kernels.py g6c_stamps): s_memtime per wavefront at [slot start |
// last MFMA issued | barrier passed] of the first 40 slots of one workgroup
__device__ unsigned long long rlt_g6c_stamp_buf[8 * 40 * 3];
#define RLT_G6C_STAMP(slot_, k_) do { if (blockIdx.x == (gridDim.x / 2 | 1) && (threadIdx.x & 63) == 0 && (slot_) < 40) \
    rlt_g6c_stamp_buf[((threadIdx.x >> 6) * 40 + (slot_)) * 3 + (k_)] = __builtin_readcyclecounter(); } while (0)
#else
#define RLT_G6C_STAMP(slot_, k_) do {} while (0)
#endif
template <bool TA, bool TB, bool PERSIST>
__global__ __launch_bounds__(512) void gemm6c_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    uint16_t* lds = reinterpret_cast<uint16_t*>(gsm);          // [3][A_h | A_m | A_l | B_h | B_m | B_l]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    int bid, zslab;
    decode_block(g, bid, zslab);                     // g.tiles_* count 256 x 256 tiles here
    const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
    int m0 = tm * BM2, n0 = tn * BN2;
    const int kbeg = zslab * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nt = (kend - kbeg) / BK3;              // register tiles (K tiles of 32) per output tile; >= 2 (host-checked)

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Stage6c sa, sb;
    constexpr bool AKC = !TA, BKC = TB;
    const bool want_cs = TA && g.colsum != nullptr && tn == 0;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    const int pl = phys6(l31);
    const int csw = 8 * (hh ^ ((pl >> 3) & 1));      // element offset of this lane's 16-byte chunk inside its row
    const int arow = (wm * 64 + pl) * 16 + csw, brow = 3 * HB6 + (wn * 128 + pl) * 16 + csw;

    // tile of the stream after the current one (PERSIST): decode_block's XCD-aware remap of id + gridDim.x
    int vid = blockIdx.x, nm0 = m0, nn0 = n0;
    bool has_next = false;
    auto decode_next = [&]() {
        const int nwg = g.tiles_m * g.tiles_n;
        vid += gridDim.x;
        has_next = PERSIST && vid < nwg;
        if (has_next) {
            const int q = nwg >> 3, r = nwg & 7, xcd = vid & 7, jj = vid >> 3;
            const int nb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + jj;
            const int ntm = nb / g.tiles_n;
            nm0 = ntm * BM2; nn0 = (nb - ntm * g.tiles_n) * BN2;
        }
    };
    // register tile `ti` of the stream position: ti < nt -> this output tile; nt, nt + 1 -> the next one's tiles 0, 1 (or, at the
    // end of the stream, a harmless re-read of the last tile: the fetch is unconditional so that hipcc's vmcnt counts stay exact)
    auto fetch = [&](int ti) {
        const bool wrap = PERSIST && has_next && ti >= nt;
        const int k0 = kbeg + (wrap ? ti - nt : min(ti, nt - 1)) * BK3;
        load6c<AKC>(g.A, g.lda, wrap ? nm0 : m0, k0, tid, sa);
        load6c<BKC>(g.B, g.ldb, wrap ? nn0 : n0, k0, tid, sb);
    };
    auto add_cs = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) { csum.x += sa.v[i].x; csum.y += sa.v[i].y; csum.z += sa.v[i].z; csum.w += sa.v[i].w; }
    };

    // one slot: the 48 MFMAs of k-step buffer `buf`; STASH: pieces of the staged register tile go behind the MFMA groups into
    // buffers b1 (its k-step 0) and b2 (its k-step 1), then the loads of register tile `fetch_ti` are issued
    auto slot = [&](auto stash_tag, int buf, int b1, int b2, bool do_stash, int fetch_ti) {
        constexpr bool STASH = decltype(stash_tag)::value;
        const uint16_t* base = lds + buf * KB6;
        uint16_t* t1 = lds + b1 * KB6;
        uint16_t* t2 = lds + b2 * KB6;
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) a[i][q] = *reinterpret_cast<const bf16x8*>(base + arow + q * HB6 + i * 32 * 16);
        auto load_b = [&](int j, int s_) {
#pragma unroll
            for (int q = 0; q < 3; ++q) b[s_][q] = *reinterpret_cast<const bf16x8*>(base + brow + q * HB6 + j * 32 * 16);
        };
        load_b(0, 0);
        if (STASH && do_stash && want_cs) add_cs();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j + 1 < 4) load_b(j + 1, (j + 1) & 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x16 c = acc[i][j];
                // smallest terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j & 1][1], c, 0, 0, 0);   // m m'
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j & 1][0], c, 0, 0, 0);   // l h'
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j & 1][2], c, 0, 0, 0);   // h l'
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j & 1][0], c, 0, 0, 0);   // m h'
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j & 1][1], c, 0, 0, 0);   // h m'
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j & 1][0], c, 0, 0, 0);   // h h'
                acc[i][j] = c;
                if (STASH && do_stash) {               // piece 2 j + i of eight: A pieces 0..3, then B pieces 0..3
                    const int piece = 2 * j + i;
                    if (piece < 4) stash6c_part<AKC>(t1, t2, tid, sa, piece);
                    else stash6c_part<BKC>(t1 + 3 * HB6, t2 + 3 * HB6, tid, sb, piece - 4);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (STASH) fetch(fetch_ti);
    };

    // prologue: register tile 0 into buffers 0 and 1, register tile 1 in flight
    fetch(0);
    if (want_cs) add_cs();
#pragma unroll
    for (int part = 0; part < 4; ++part) {
        stash6c_part<AKC>(lds, lds + KB6, tid, sa, part);
        stash6c_part<BKC>(lds + 3 * HB6, lds + KB6 + 3 * HB6, tid, sb, part);
    }
    if (PERSIST) decode_next();
    fetch(1);
    __syncthreads();

    int u = 0;                                         // k-steps of the stream so far, modulo 3: the buffer of this slot
    int nslot = 0; (void)nslot;
    while (true) {
        for (int ti = 0; ti < nt; ++ti) {
            const int b0 = u, b1 = u == 2 ? 0 : u + 1, b2 = b1 == 2 ? 0 : b1 + 1;
            RLT_G6C_STAMP(nslot, 0);
            slot(BoolTag<false>{}, b0, 0, 0, false, 0);                                   // even slot: k-step 2 ti
            RLT_G6C_STAMP(nslot, 1);
            __syncthreads();
            RLT_G6C_STAMP(nslot, 2);
            ++nslot;
            // odd slot: k-step 2 ti + 1 out of b1; stages register tile ti + 1 into b2 and b0, fetches tile ti + 2
            RLT_G6C_STAMP(nslot, 0);
            slot(BoolTag<true>{}, b1, b2, b0, ti + 1 < nt || has_next, ti + 2);
            RLT_G6C_STAMP(nslot, 1);
            __syncthreads();
            RLT_G6C_STAMP(nslot, 2);
            ++nslot;
            u = b2;
        }
#ifdef RLT_G6C_NOSTORE      // timing-only ablation (tools/build_variant.py): no output (results are wrong by design)
        if (acc[0][0][0] == 123.456f)
#endif
        write_output_t<4>(g, acc, m0 + wm * 64, n0 + wn * 128, true, l31, hh, zslab);
        if (!PERSIST || !has_next) break;
        m0 = nm0; n0 = nn0;
        decode_next();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    if (want_cs) {          // threads 4 mb + kb and 256 + 4 mb + kb (kb < 4) hold partial sums of columns 4 mb .. 4 mb + 3
        float4* red = reinterpret_cast<float4*>(gsm);
        __syncthreads();
        red[tid] = csum;
        __syncthreads();
        if (tid < 64) {
            float4 t = red[4 * tid];
#pragma unroll
            for (int j = 1; j < 8; ++j) {
                const float4 o = red[(j >> 2) * 256 + 4 * tid + (j & 3)];
                t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
            }
            float* dst = (g.cs_slab ? g.cs_slab + (size_t)zslab * g.M : g.colsum) + m0 + 4 * tid;
            dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
        }
    }
}
template <bool TA, bool TB>
int launch_gemm6c(GemmArgs g, int ns, hipStream_t st) {
    const size_t shm = (size_t)3 * KB6 * sizeof(uint16_t);
    g.tiles_m = g.M / BM2; g.tiles_n = g.N / BN2;
    if constexpr (!TA) {
        static const int persist_wgs = [] { const char* e = getenv("RLT_GEMM6_PERSIST"); return e ? atoi(e) : 256; }();   // 0: off
        const long long tiles = (long long)g.tiles_m * g.tiles_n;
        if (persist_wgs > 0 && ns == 1 && tiles > persist_wgs && persist_wgs % 8 == 0) {
            int rc = rlt_allow_lds(gemm6c_kernel<TA, TB, true>, shm);
            if (rc) return rc;
            hipLaunchKernelGGL((gemm6c_kernel<TA, TB, true>), dim3(persist_wgs), dim3(512), shm, st, g);
            return RLT_LAUNCH_RESULT();
        }
    }
    int rc = rlt_allow_lds(gemm6c_kernel<TA, TB, false>, shm);
    if (rc) return rc;
    dim3 grid(g.tiles_m * g.tiles_n * (g.slab_xcd ? ns : 1), 1, g.slab_xcd ? 1 : ns);
    hipLaunchKernelGGL((gemm6c_kernel<TA, TB, false>), grid, dim3(512), shm, st, g);
    return RLT_LAUNCH_RESULT();
}

// ======================================================================================================
// gemm6e: the 256 x 256 six-product tile by ONE wavefront per SIMD (256 threads, 128 x 128 per wavefront: all 256 AGPRs are
// accumulators).  In gemm6c two wavefronts share a SIMD and the ~250 vector instructions a wavefront spends on the split of the
// next K tile are not absorbed by its partner's MFMAs (slot timeline in profiles/r04_notes.md: 3,800 cycles for a slot that only
// multiplies, 5,200 for one that also splits, 3,072 of MFMAs in both).  A bf16 MFMA hides ~4 plain vector instructions of the
// SAME wavefront when they sit right behind it (tools/micro/mfma_split.hip), so here a slot is one k-step of 16 = 16 groups of
// six MFMAs per wavefront, each MFMA followed by a gap that carries one part of the split of the k-step two slots ahead (its
// eight staged pieces: 4 of A, 4 of B per thread), a store + reload, or fragment reads - placed by tools/gen_gemm6e_slot.py,
// fenced so that hipcc keeps the order.  Three k-step buffers rotate as in gemm6c (same LDS layout, phys6 rows, swizzled
// 16-byte chunks); slot s multiplies buffer s % 3, writes k-step s + 2 into buffer (s + 2) % 3 (free since the barrier that
// opened the slot) and reloads the staging registers of its parity with k-step s + 4.  Loads are per k-step (64 bytes of a
// K-contiguous row per 4 lanes: half lines - the price of a uniform slot).
template <bool KC>
__device__ __forceinline__ size_t off6e(int tid, int ld) {           // this thread's element offset inside a k-step of an operand
    if (KC) return (size_t)phys6(tid >> 2) * ld + 4 * (tid & 3);     // row phys6(t + 64 i) = phys6(t) + 64 i, float4 kq
    return (size_t)(4 * (tid & 3)) * ld + 4 * (tid >> 2);            // k rows 4 kb + i, columns 4 mb ..
}
template <bool KC>
__device__ __forceinline__ void load6e(const float* __restrict__ P, int ld, int i, float4& v) {   // P: operand + uniform part + off6e
    v = *reinterpret_cast<const float4*>(P + (size_t)(KC ? 64 * i : i) * ld);
}
template <bool KC>
__device__ __forceinline__ int dst6e(int tid, int p) {               // bf16 element offset of piece p inside the operand's h plane
    const int x = tid & 3;
    const int pr = KC ? (tid >> 2) + 64 * p : ((((tid >> 2) * 4) & ~15) | (p << 2) | ((tid >> 2) & 3));
    return pr * 16 + 8 * ((x >> 1) ^ ((pr >> 3) & 1)) + 4 * (x & 1);
}
template <bool KC>
__device__ __forceinline__ void vals6e(const float4 (&v)[4], int p, float& a, float& b, float& c, float& d) {
    if (KC) { a = v[p].x; b = v[p].y; c = v[p].z; d = v[p].w; }
    else {
        const float* f0 = reinterpret_cast<const float*>(&v[0]);
        const float* f1 = reinterpret_cast<const float*>(&v[1]);
        const float* f2 = reinterpret_cast<const float*>(&v[2]);
        const float* f3 = reinterpret_cast<const float*>(&v[3]);
        a = f0[p]; b = f1[p]; c = f2[p]; d = f3[p];
    }
}

// epilogue of gemm6e for the plain / bias / bias + ReLU / accumulate / split-K-slab outputs (the host sends every other epilogue
// - 1-bit masks, float masks, dropout - to gemm6c): one loop nest, the accumulator registers read where they are stored
template <int NJ>
__device__ __forceinline__ void write_output_simple(const GemmArgs& g, const f32x16 (&acc)[2][NJ], int rbase, int cbase, int l31, int hh, int z) {
    const bool to_slab = g.slab != nullptr;
    float* out = to_slab ? g.slab + (size_t)z * g.M * g.N : g.C;
    const int ldo = to_slab ? g.N : g.ldc;
    const int rb = __builtin_amdgcn_readfirstlane(rbase), cb = __builtin_amdgcn_readfirstlane(cbase);
    const int loff = 4 * hh * ldo + l31;
    const bool relu = !to_slab && (g.flags & RLT_GEMM_RELU), accum = !to_slab && (g.flags & RLT_GEMM_ACCUMULATE);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = cb + j * 32 + l31;
        float bv = 0.f;
        if (!to_slab) {
            if (g.bias) bv += g.bias[col];
            if (g.bias2) bv += g.bias2[col];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                float* base = out + (size_t)(rb + i * 32 + dr) * ldo + (cb + j * 32);       // scalar
                float v = acc[i][j][r] + bv;
                if (accum) v += base[loff];
                if (relu) v = fmaxf(v, 0.f);
                base[loff] = v;
            }
        }
    }
}
inline bool gemm6e_epilogue_ok(const GemmArgs& g) {
    return !g.mask && !g.bits_in && !g.bits_out && g.drop_p <= 0.f;
}

template <bool TA, bool TB, bool PERSIST>
__global__ __launch_bounds__(256, 1) void gemm6e_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    uint16_t* lds = reinterpret_cast<uint16_t*>(gsm);          // [3][A_h | A_m | A_l | B_h | B_m | B_l]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    int bid, zslab;
    decode_block(g, bid, zslab);
    const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
    int m0 = tm * BM2, n0 = tn * BN2;
    const int kbeg = zslab * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nks = (kend - kbeg) / 16;              // k-steps per output tile: even, >= 4 (host-checked)

    f32x16 acc[2][2][4];                             // [row half of 64][A block in the half][B block]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i >> 1][i & 1][j][r] = 0.f;

    constexpr bool AKC = !TA, BKC = TB;
    const bool want_cs = TA && g.colsum != nullptr && tn == 0;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    const int pl = phys6(l31);
    const int csw = 8 * (hh ^ ((pl >> 3) & 1));
    const int arow = (wm * 128 + pl) * 16 + csw, brow = 3 * HB6 + (wn * 128 + pl) * 16 + csw;
    const size_t offA = off6e<AKC>(tid, g.lda), offB = off6e<BKC>(tid, g.ldb);
    int dA[4], dB[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) { dA[p] = dst6e<AKC>(tid, p); dB[p] = 3 * HB6 + dst6e<BKC>(tid, p); }

    int vid = blockIdx.x, nm0 = m0, nn0 = n0;
    bool has_next = false;
    auto decode_next = [&]() {
        const int nwg = g.tiles_m * g.tiles_n;
        vid += gridDim.x;
        has_next = PERSIST && vid < nwg;
        if (has_next) {
            const int q = nwg >> 3, r = nwg & 7, xcd = vid & 7, jj = vid >> 3;
            const int nb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + jj;
            const int ntm = nb / g.tiles_n;
            nm0 = ntm * BM2; nn0 = (nb - ntm * g.tiles_n) * BN2;
        }
    };
    // operand pointers of k-step `ks` of the stream: ks < nks -> this output tile; nks.. -> the next one's (or, at the end of the
    // stream, a harmless re-read of the last k-step: loads and stores of the slot body are unconditional)
    auto srcA = [&](int ks) {
        const bool wrap = PERSIST && has_next && ks >= nks;
        const int k0 = kbeg + 16 * (wrap ? ks - nks : min(ks, nks - 1)), mm = wrap ? nm0 : m0;
        return g.A + (AKC ? (size_t)mm * g.lda + k0 : (size_t)k0 * g.lda + mm) + offA;
    };
    auto srcB = [&](int ks) {
        const bool wrap = PERSIST && has_next && ks >= nks;
        const int k0 = kbeg + 16 * (wrap ? ks - nks : min(ks, nks - 1)), nn = wrap ? nn0 : n0;
        return g.B + (BKC ? (size_t)nn * g.ldb + k0 : (size_t)k0 * g.ldb + nn) + offB;
    };
    float4 ra[2][4], rb[2][4];                       // staged k-steps, by parity
    auto add_cs = [&](const float4 (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { csum.x += v[i].x; csum.y += v[i].y; csum.z += v[i].z; csum.w += v[i].w; }
    };
    auto stash_all = [&](uint16_t* wb, const float4 (&va)[4], const float4 (&vb)[4]) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float a, b, c, d;
            uint2 hi, mid, lo;
            vals6e<AKC>(va, p, a, b, c, d);
            split4x3(a, b, c, d, hi, mid, lo);
            *reinterpret_cast<uint2*>(wb + dA[p]) = hi;
            *reinterpret_cast<uint2*>(wb + dA[p] + HB6) = mid;
            *reinterpret_cast<uint2*>(wb + dA[p] + 2 * HB6) = lo;
            vals6e<BKC>(vb, p, a, b, c, d);
            split4x3(a, b, c, d, hi, mid, lo);
            *reinterpret_cast<uint2*>(wb + dB[p]) = hi;
            *reinterpret_cast<uint2*>(wb + dB[p] + HB6) = mid;
            *reinterpret_cast<uint2*>(wb + dB[p] + 2 * HB6) = lo;
        }
    };
    bf16x8 a[4][3], b[2][3];
    auto frag_a = [&](const uint16_t* base, int i, int q) { a[i][q] = *reinterpret_cast<const bf16x8*>(base + arow + q * HB6 + i * 32 * 16); };
    auto frag_b = [&](const uint16_t* base, int j, int q) { b[j & 1][q] = *reinterpret_cast<const bf16x8*>(base + brow + q * HB6 + j * 32 * 16); };

    // prologue: k-steps 0, 1 -> buffers 0, 1; k-steps 2, 3 in flight; fragments of A blocks 0, 1 and B block 0 of buffer 0
    {
        const float* pa = srcA(0); const float* pb = srcB(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { load6e<AKC>(pa, g.lda, i, ra[0][i]); load6e<BKC>(pb, g.ldb, i, rb[0][i]); }
        pa = srcA(1); pb = srcB(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) { load6e<AKC>(pa, g.lda, i, ra[1][i]); load6e<BKC>(pb, g.ldb, i, rb[1][i]); }
        if (want_cs) { add_cs(ra[0]); add_cs(ra[1]); }
        stash_all(lds, ra[0], rb[0]);
        stash_all(lds + KB6, ra[1], rb[1]);
        if (PERSIST) decode_next();
        pa = srcA(2); pb = srcB(2);
#pragma unroll
        for (int i = 0; i < 4; ++i) { load6e<AKC>(pa, g.lda, i, ra[0][i]); load6e<BKC>(pb, g.ldb, i, rb[0][i]); }
        pa = srcA(3); pb = srcB(3);
#pragma unroll
        for (int i = 0; i < 4; ++i) { load6e<AKC>(pa, g.lda, i, ra[1][i]); load6e<BKC>(pb, g.ldb, i, rb[1][i]); }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 3; ++q) { frag_a(lds, 0, q); frag_a(lds, 1, q); frag_b(lds, 0, q); }

    // one slot (parity PAR of its k-step index s): multiplies buffer `buf`, writes k-step s + 2 (staged set PAR) into `wbuf`,
    // reloads the set from (pa, pb) = k-step s + 4, reads the first fragments of the next slot out of `nbuf`
    auto slot = [&](auto par_tag, int buf, int wbuf, int nbuf, const float* pa, const float* pb) {
        constexpr int PAR = decltype(par_tag)::value;
        const uint16_t* base = lds + buf * KB6;
        const uint16_t* nbase = lds + nbuf * KB6;
        uint16_t* wb = lds + wbuf * KB6;
        Split6 su;
#define GAP_END __builtin_amdgcn_sched_barrier(0)
#define MM(i_, j_, k_) acc[(i_) >> 1][(i_) & 1][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16( \
            a[i_][(k_) == 0 || (k_) == 3 ? 1 : (k_) == 1 ? 2 : 0], b[(j_) & 1][(k_) == 0 || (k_) == 4 ? 1 : (k_) == 2 ? 2 : 0], acc[(i_) >> 1][(i_) & 1][j_], 0, 0, 0)
#define SP(p_, k_) do { float x0_, x1_, x2_, x3_; \
            if ((p_) < 4) vals6e<AKC>(ra[PAR], (p_) & 3, x0_, x1_, x2_, x3_); else vals6e<BKC>(rb[PAR], (p_) & 3, x0_, x1_, x2_, x3_); \
            split6_part(su, x0_, x1_, x2_, x3_, k_); } while (0)
#define ST(p_) do { const int d_ = (p_) < 4 ? dA[(p_) & 3] : dB[(p_) & 3]; \
            *reinterpret_cast<uint2*>(wb + d_) = su.hi; *reinterpret_cast<uint2*>(wb + d_ + HB6) = su.mid; \
            *reinterpret_cast<uint2*>(wb + d_ + 2 * HB6) = su.lo; \
            if ((p_) < 4) { if (AKC) load6e<AKC>(pa, g.lda, (p_) & 3, ra[PAR][(p_) & 3]); \
                            else if ((p_) == 3) { for (int i_ = 0; i_ < 4; ++i_) load6e<AKC>(pa, g.lda, i_, ra[PAR][i_]); } } \
            else { if (BKC) load6e<BKC>(pb, g.ldb, (p_) & 3, rb[PAR][(p_) & 3]); \
                   else if ((p_) == 7) { for (int i_ = 0; i_ < 4; ++i_) load6e<BKC>(pb, g.ldb, i_, rb[PAR][i_]); } } } while (0)
#define RA(i_, q_, n_) frag_a((n_) ? nbase : base, i_, q_)
#define RB(j_, q_, n_) frag_b((n_) ? nbase : base, j_, q_)
#include "gemm6e_slot.inc"
#undef GAP_END
#undef MM
#undef SP
#undef ST
#undef RA
#undef RB
    };

    int u = 0;                                       // buffer of the slot's k-step
    while (true) {
        for (int s = 0; s < nks; s += 2) {
            const int u1 = u == 2 ? 0 : u + 1, u2 = u1 == 2 ? 0 : u1 + 1;
            if (want_cs && s + 2 < nks) add_cs(ra[0]);
            slot(std::integral_constant<int, 0>{}, u, u2, u1, srcA(s + 4), srcB(s + 4));
            __syncthreads();
            if (want_cs && s + 3 < nks) add_cs(ra[1]);
            slot(std::integral_constant<int, 1>{}, u1, u, u2, srcA(s + 5), srcB(s + 5));
            __syncthreads();
            u = u2;
        }
        write_output_simple<4>(g, acc[0], m0 + wm * 128, n0 + wn * 128, l31, hh, zslab);
        write_output_simple<4>(g, acc[1], m0 + wm * 128 + 64, n0 + wn * 128, l31, hh, zslab);
        if (!PERSIST || !has_next) break;
        m0 = nm0; n0 = nn0;
        decode_next();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i >> 1][i & 1][j][r] = 0.f;
    }
    if (want_cs) {          // threads 4 mb + kb (kb < 4) hold partial sums of columns 4 mb .. 4 mb + 3
        float4* red = reinterpret_cast<float4*>(gsm);
        __syncthreads();
        red[tid] = csum;
        __syncthreads();
        if (tid < 64) {
            float4 t = red[4 * tid];
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                const float4 o = red[4 * tid + j];
                t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
            }
            float* dst = (g.cs_slab ? g.cs_slab + (size_t)zslab * g.M : g.colsum) + m0 + 4 * tid;
            dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
        }
    }
}
template <bool TA, bool TB>
int launch_gemm6e(GemmArgs g, int ns, hipStream_t st) {
    const size_t shm = (size_t)3 * KB6 * sizeof(uint16_t);
    g.tiles_m = g.M / BM2; g.tiles_n = g.N / BN2;
    if constexpr (!TA) {
        static const int persist_wgs = [] { const char* e = getenv("RLT_GEMM6_PERSIST"); return e ? atoi(e) : 256; }();   // 0: off
        const long long tiles = (long long)g.tiles_m * g.tiles_n;
        if (persist_wgs > 0 && ns == 1 && tiles > persist_wgs && persist_wgs % 8 == 0) {
            int rc = rlt_allow_lds(gemm6e_kernel<TA, TB, true>, shm);
            if (rc) return rc;
            hipLaunchKernelGGL((gemm6e_kernel<TA, TB, true>), dim3(persist_wgs), dim3(256), shm, st, g);
            return RLT_LAUNCH_RESULT();
        }
    }
    int rc = rlt_allow_lds(gemm6e_kernel<TA, TB, false>, shm);
    if (rc) return rc;
    dim3 grid(g.tiles_m * g.tiles_n * (g.slab_xcd ? ns : 1), 1, g.slab_xcd ? 1 : ns);
    hipLaunchKernelGGL((gemm6e_kernel<TA, TB, false>), grid, dim3(256), shm, st, g);
    return RLT_LAUNCH_RESULT();
}

// the bf16x6 tile needs M % 256 == 0, N % 128 == 0, whole 32-wide K tiles per slab and the branch-free loader
// preconditions; other shapes of that mode run on the exact f32 MFMA kernel (more exact still)
bool gemm6_ok(const GemmArgs& g) {
    if (!g.vecA || !g.vecB) return false;
    return !((g.M % BM6) || (g.N % BN6) || (g.K % BK3) || (g.kchunk % BK3));
}
template <bool TA, bool TB>
int launch_gemm6(GemmArgs g, int ns, hipStream_t st) {
    const size_t shm = (size_t)2 * BUF6 * sizeof(uint16_t);
    g.tiles_m = g.M / BM6; g.tiles_n = g.N / BN6;
    int rc = rlt_allow_lds(gemm6_kernel<TA, TB>, shm);
    if (rc) return rc;
    dim3 grid(g.tiles_m * g.tiles_n * (g.slab_xcd ? ns : 1), 1, g.slab_xcd ? 1 : ns);
    hipLaunchKernelGGL((gemm6_kernel<TA, TB>), grid, dim3(512), shm, st, g);
    return RLT_LAUNCH_RESULT();
}

template <bool TA, bool TB, bool FAST>
int launch_gemm3_v(const GemmArgs& g, dim3 grid, hipStream_t st) {
    const size_t shm = (size_t)2 * 4 * TILE3 * sizeof(uint16_t);
    int rc = rlt_allow_lds(gemm3_kernel<TA, TB, FAST>, shm);
    if (rc) return rc;
    hipLaunchKernelGGL((gemm3_kernel<TA, TB, FAST>), grid, dim3(256), shm, st, g);
    return RLT_LAUNCH_RESULT();
}
bool gemm_fast_ok(const GemmArgs& g, bool ta, bool tb);
template <bool TA, bool TB>
int launch_gemm3(const GemmArgs& g, dim3 grid, hipStream_t st) {
    if (gemm_fast_ok(g, TA, TB)) return launch_gemm3_v<TA, TB, true>(g, grid, st);
    return launch_gemm3_v<TA, TB, false>(g, grid, st);
}

// C = sum_z slab[z] (+bias, relu, accumulate); fixed order => deterministic.  Four independent partial sums per
// element keep four slab loads in flight (the z loop is otherwise one dependent chain of L2/HBM latencies).
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g, int nsplit) {
    const size_t mn = (size_t)g.M * g.N;
    const size_t stride = (size_t)gridDim.x * 256;
    if ((g.N & 3) == 0) {                     // rows hold whole float4s: a vector never straddles two rows
        const float4* slab4 = reinterpret_cast<const float4*>(g.slab);
        const size_t mn4 = mn / 4;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < mn4; i += stride) {
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
            int z = 0;
            for (; z + 3 < nsplit; z += 4) {
                a0 = f4add(a0, slab4[(size_t)z * mn4 + i]);
                a1 = f4add(a1, slab4[(size_t)(z + 1) * mn4 + i]);
                a2 = f4add(a2, slab4[(size_t)(z + 2) * mn4 + i]);
                a3 = f4add(a3, slab4[(size_t)(z + 3) * mn4 + i]);
            }
            for (; z < nsplit; ++z) a0 = f4add(a0, slab4[(size_t)z * mn4 + i]);
            const float4 v4 = f4add(f4add(a0, a1), f4add(a2, a3));
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
            const int row = (int)((4 * i) / g.N), col0 = (int)(4 * i - (size_t)row * g.N);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = col0 + e;
                float bv = 0.f;
                if (g.bias) bv += g.bias[col];
                if (g.bias2) bv += g.bias2[col];
                float* dst = g.C + (size_t)row * g.ldc + col;
                *dst = gemm_epilogue(g, v[e], bv, row, col, dst);
            }
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < mn; i += stride) {
            float v = 0.f;
            for (int z = 0; z < nsplit; ++z) v += g.slab[(size_t)z * mn + i];
            const int row = (int)(i / g.N), col = (int)(i - (size_t)row * g.N);
            float bv = 0.f;
            if (g.bias) bv += g.bias[col];
            if (g.bias2) bv += g.bias2[col];
            float* dst = g.C + (size_t)row * g.ldc + col;
            *dst = gemm_epilogue(g, v, bv, row, col, dst);
        }
    }
    if (g.cs_slab)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)g.M; i += stride) {
            float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
            int z = 0;
            for (; z + 3 < nsplit; z += 4) {
                c0 += g.cs_slab[(size_t)z * g.M + i];
                c1 += g.cs_slab[(size_t)(z + 1) * g.M + i];
                c2 += g.cs_slab[(size_t)(z + 2) * g.M + i];
                c3 += g.cs_slab[(size_t)(z + 3) * g.M + i];
            }
            for (; z < nsplit; ++z) c0 += g.cs_slab[(size_t)z * g.M + i];
            g.colsum[i] = (c0 + c1) + (c2 + c3);
        }
}

template <bool TA, bool TB, int BK, int OCC, bool FAST>
int launch_gemm_v(const GemmArgs& g, dim3 grid, hipStream_t st) {
    const size_t shm = (size_t)4 * BK * LDT * sizeof(float);
    int rc = rlt_allow_lds(gemm_kernel<TA, TB, BK, OCC, FAST>, shm);
    if (rc) return rc;
    hipLaunchKernelGGL((gemm_kernel<TA, TB, BK, OCC, FAST>), grid, dim3(256), shm, st, g);
    return RLT_LAUNCH_RESULT();
}
// the branch-free loaders need: aligned operands, K % 4 == 0, K-chunks >= 4, MN % 4 == 0 for MN-contiguous operands
bool gemm_fast_ok(const GemmArgs& g, bool ta, bool tb) {
    static const int off = [] { const char* e = getenv("RLT_GEMM_NOFAST"); return e ? atoi(e) : 0; }();
    if (off) return false;
    if (!g.vecA || !g.vecB || (g.K & 3) || g.K < 4) return false;
    if (ta && ((g.M & 3) || g.M < 4)) return false;       // A stored [K][M]
    if (!tb && ((g.N & 3) || g.N < 4)) return false;      // B stored [K][N]
    return true;
}
template <bool TA, bool TB, int BK>
int launch_gemm(const GemmArgs& g, dim3 grid, hipStream_t st) {
    if (gemm_fast_ok(g, TA, TB)) return launch_gemm_v<TA, TB, BK, 4, true>(g, grid, st);
    return launch_gemm_v<TA, TB, BK, 4, false>(g, grid, st);
}

// 0 = exact fp32 MFMA (parity mode), 1 = split-bf16 (bf16x3), 2 = fp32-faithful six-product split (bf16x6);
// RLT_GEMM_MODE overrides the library-wide mode
int gemm_mode() {
    static const int forced = [] {
        const char* e = getenv("RLT_GEMM_MODE");
        if (!e) return -1;
        if (!strcmp(e, "bf16x6") || !strcmp(e, "2")) return 2;
        return (!strcmp(e, "bf16x3") || !strcmp(e, "1")) ? 1 : 0;
    }();
    return forced >= 0 ? forced : rlt_precision();
}

int choose_split(int M, int N, int K) {
    const long long tiles = (long long)rlt_cdiv(M, BM) * rlt_cdiv(N, BN);
    // (K from 1024: the reference's own batch sizes - 32 / 63 lists x 300 positions = 9,600 / 18,900 rows - leave the K = 2048 products
    //  of the encoder with 75-150 output tiles: unsplit, a third of the chip ran 2048-long loops, 160-200 us per product at batch 32)
    static const int kmin = [] { const char* e = getenv("RLT_GEMM_SPLIT_KMIN"); return e ? atoi(e) : 1024; }();
    if (tiles >= 256 || K < kmin) return 1;
    static const int target = [] { const char* e = getenv("RLT_GEMM_SPLIT_TARGET"); return e ? atoi(e) : 1024; }();
    long long want = target / tiles > 0 ? target / tiles : 1;     // workgroups in flight: a whole number of waves of the grid
    const long long per = K >= 4096 ? 512 : 256;             // keep >= 512 of K per slice (>= 256 below K = 4096)
    const long long maxs = K / per > 0 ? K / per : 1;
    if (want > maxs) want = maxs;
    if (want > 256) want = 256;
    if (want >= 8) want = want / 8 * 8;       // whole groups of 8 slabs: one slab per XCD and round (decode_block)
    return (int)(want < 1 ? 1 : want);
}

#if defined(RLT_STAMPS)
extern "C" int rlt_debug_g6c_stamps(unsigned long long* host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rlt_g6c_stamp_buf), n * sizeof(unsigned long long));
}
extern "C" int rlt_debug_gemm_stamps(unsigned long long* host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rlt_gemm_stamp_buf), n * sizeof(unsigned long long));
}
#endif

// ---- column sums (bias gradients) ------------------------------------------------------------------
constexpr int CS_ROWS_PER_WG = 512;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, int ldx, int T, int N,
                                                             float* __restrict__ partial) {
    // block = (column block of 256, row chunk); thread = one column, 4 row phases via... keep simple:
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rphase = threadIdx.x >> 6;
    const int r0 = blockIdx.y * CS_ROWS_PER_WG, r1 = min(T, r0 + CS_ROWS_PER_WG);
    float acc = 0.f;
    if (col < N)
        for (int r = r0 + rphase; r < r1; r += 4) acc += X[(size_t)r * ldx + col];
    __shared__ float sm[4][64];
    sm[rphase][threadIdx.x & 63] = acc;
    __syncthreads();
    if (threadIdx.x < 64 && col < N)
        partial[(size_t)blockIdx.y * N + col] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void segment_colsum_kernel(const float* __restrict__ X, int ldx, int R, int N,
                                                             float* __restrict__ out, int ldo, int accumulate) {
    const int g = blockIdx.x;
    const int lane = threadIdx.x & 63, rphase = threadIdx.x >> 6;
    __shared__ float sm[4][64];
    for (int c0 = 0; c0 < N; c0 += 64) {
        const int col = c0 + lane;
        float acc = 0.f;
        if (col < N)
            for (int r = rphase; r < R; r += 4) acc += X[((size_t)g * R + r) * ldx + col];
        sm[rphase][lane] = acc;
        __syncthreads();
        if (threadIdx.x < 64 && col < N) {
            const float v = sm[0][lane] + sm[1][lane] + sm[2][lane] + sm[3][lane];
            float* dst = out + (size_t)g * ldo + col;
            *dst = accumulate ? *dst + v : v;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(float* __restrict__ dX, const float* __restrict__ Y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        if (!(Y[i] > 0.f)) dX[i] = 0.f;
}
__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, const float* __restrict__ s, size_t n) {
    const float f = s[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] *= f;
}

// ---- narrow weight gradient: dW[M][I] = A^T X, db[M] = column sums of A, for I <= 3 ------------------------
// The LSTM layer-0 input weights (I = 3 features): as a GEMM this is a 128x128-tile product with N = 3, one pass
// over A (the 5 GB gate-gradient stash) per direction at ~2 TB/s.  Here one streaming pass covers all M columns of
// both directions: a 256-thread workgroup owns whole rows (thread = 4 consecutive columns, 16-byte loads), a chunk of
// rows, and 16 register accumulators; fixed-order two-level reduction (bitwise reproducible).
constexpr int NDW_CHUNKS = 2048;
__global__ __launch_bounds__(256) void narrow_dw_partial_kernel(const float* __restrict__ A, int lda, const float* __restrict__ X,
                                                                int ldx, int I, int T, int M, int rows_per_wg,
                                                                float* __restrict__ partial) {
    const int m4 = blockIdx.y * 256 + threadIdx.x;          // float4 column index
    const int t0 = blockIdx.x * rows_per_wg, t1 = min(T, t0 + rows_per_wg);
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, ab = a0;
    if (4 * m4 < M) {
        const float* ap = A + 4 * m4;
        for (int t = t0; t < t1; ++t) {
            const float4 v = *reinterpret_cast<const float4*>(ap + (size_t)t * lda);
            const float* xr = X + (size_t)t * ldx;
            const float x0 = xr[0], x1 = I > 1 ? xr[1] : 0.f, x2 = I > 2 ? xr[2] : 0.f;
            a0.x += v.x * x0; a0.y += v.y * x0; a0.z += v.z * x0; a0.w += v.w * x0;
            a1.x += v.x * x1; a1.y += v.y * x1; a1.z += v.z * x1; a1.w += v.w * x1;
            a2.x += v.x * x2; a2.y += v.y * x2; a2.z += v.z * x2; a2.w += v.w * x2;
            ab.x += v.x; ab.y += v.y; ab.z += v.z; ab.w += v.w;
        }
        float* pr = partial + (size_t)blockIdx.x * 4 * M + 4 * m4;   // [chunk][f][m], f = 3 holds the column sums
        *reinterpret_cast<float4*>(pr) = a0;
        *reinterpret_cast<float4*>(pr + M) = a1;
        *reinterpret_cast<float4*>(pr + 2 * M) = a2;
        *reinterpret_cast<float4*>(pr + 3 * M) = ab;
    }
}
__global__ __launch_bounds__(256) void narrow_dw_scatter_kernel(const float* __restrict__ sums, int M, int I,
                                                                float* __restrict__ dW, float* __restrict__ db) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    for (int f = 0; f < I; ++f) dW[(size_t)m * I + f] = sums[(size_t)f * M + m];
    if (db) db[m] = sums[(size_t)3 * M + m];
}

int ew_grid(size_t n) { size_t g = (n + 1023) / 1024; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

}  // namespace

extern "C" {

size_t rlt_gemm_workspace(int ta, int tb, int M, int N, int K) {
    (void)ta; (void)tb;
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int ns = choose_split(M, N, K);
    return ns > 1 ? ((size_t)ns * M * N + (size_t)ns * M) * sizeof(float) : 0;
}

int rlt_gemm(int ta, int tb, int M, int N, int K,
             const float* A, int lda, const float* B, int ldb, float* C, int ldc,
             const float* bias, const float* bias2, int flags,
             void* ws, size_t ws_bytes, int precision, void* stream) {
    RLT_PREC_SCOPE(precision);
    return rlt_gemm_ex(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, bias2, flags, nullptr, 0, 1.0f, nullptr,
                       0.0f, 0u, ws, ws_bytes, RLT_PRECISION_DEFAULT, stream);
}

static int gemm_run(int ta, int tb, int M, int N, int K,
                    const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                    const float* bias, const float* bias2, int flags,
                    const float* relu_mask, int ldmask, float mask_scale, float* colsum_a,
                    float drop_p, uint32_t seed, uint32_t* bits_out, const uint32_t* bits_in,
                    void* ws, size_t ws_bytes, void* stream) {
    RLT_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0);
    RLT_CHECK_ARG(lda >= (ta ? M : K) && ldb >= (tb ? K : N) && ldc >= N);
    RLT_CHECK_ARG(!relu_mask || ldmask >= N);
    RLT_CHECK_ARG(!colsum_a || ta);          // the side column sum needs A stored [K,M]
    GemmArgs g;
    RLT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
    g.mask = relu_mask; g.ldmask = ldmask; g.colsum = colsum_a; g.cs_slab = nullptr;
    g.bits_out = bits_out; g.bits_in = bits_in; g.ldbits = N;
    g.mask_scale = mask_scale; g.drop_p = drop_p; g.drop_thr = rlt_drop_threshold(drop_p); g.seed = seed;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.bias2 = bias2;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.vecA = (lda % 4 == 0) && rlt_aligned16(A);
    g.vecB = (ldb % 4 == 0) && rlt_aligned16(B);
    g.flags = flags;
    g.tiles_m = rlt_cdiv(M, BM); g.tiles_n = rlt_cdiv(N, BN);
    int ns = (bits_out || bits_in) ? 1 : choose_split(M, N, K);       // the bit epilogues live in the GEMM kernel proper
    if (ns > 1 && (!ws || ws_bytes < ((size_t)ns * M * N + (size_t)ns * M) * sizeof(float))) {
        if (ws == nullptr && ws_bytes == 0) ns = 1; else return RLT_E_WORKSPACE;
    }
    int kchunk = rlt_cdiv(rlt_cdiv(K, ns), 32) * 32;
    ns = rlt_cdiv(K, kchunk);
    g.kchunk = kchunk;
    static const bool no_slab_xcd = getenv("RLT_GEMM_NO_SLAB_XCD") != nullptr;
    g.slab_xcd = (ns > 1 && ns % 8 == 0 && !no_slab_xcd) ? 1 : 0;
    g.slab = ns > 1 ? (float*)ws : nullptr;
    g.cs_slab = (ns > 1 && colsum_a) ? (float*)ws + (size_t)ns * M * N : nullptr;
    hipStream_t st = rlt_stream(stream);
    dim3 grid(g.tiles_m * g.tiles_n * (g.slab_xcd ? ns : 1), 1, g.slab_xcd ? 1 : ns);
    int rc = 0;
    // K = 256 with the long dimension in M (the Linear layers of the encoder, the LSTM input projection): the weights-stationary
    // streaming kernel (csrc/gemm6s.hip) - no K loop, nothing re-split per tile
    if (gemm_mode() == 2 && !ta && ns == 1 && !relu_mask && !colsum_a && drop_p == 0.f && !(flags & RLT_GEMM_ACCUMULATE) &&
        (!bits_in || (!bias && !bias2 && !(flags & RLT_GEMM_RELU)))) {
        const Gemm6sArgs s6{A, B, C, bias, bias2, M, N, K, lda, ldb, ldc, bits_out, bits_in, mask_scale};
        if (rlt_gemm6s_ok(s6)) {
            rc = rlt_gemm6s_launch(s6, tb != 0, (flags & RLT_GEMM_RELU) != 0, stream);
            return rc ? rc : RLT_LAUNCH_RESULT();
        }
    }
    static const bool x6_small_only = getenv("RLT_GEMM6_SMALL") != nullptr;      // A/B switch: 256 x 128 tiles everywhere
    static const bool x6_no_c = [] { const char* e = getenv("RLT_GEMM6C"); return e && atoi(e) == 0; }();   // A/B switch: RLT_GEMM6C=0 -> gemm6b
    if (gemm_mode() == 2 && gemm6_ok(g) && g.N % BN2 == 0 && !x6_small_only && !x6_no_c && g.kchunk / BK3 >= 2 &&
        (min(g.K, g.kchunk) / BK3) >= 2 && (g.K % g.kchunk == 0 || (g.K % g.kchunk) / BK3 >= 2)) {
        // (every K slab holds at least two K tiles of 32: the k-step pipeline stages one register tile ahead)
        static const bool x6_e = [] { const char* e = getenv("RLT_GEMM6E"); return !e || atoi(e) != 0; }();     // RLT_GEMM6E=0 -> gemm6c
        // gemm6e (one wavefront per SIMD) where the K loop decides - the weight-gradient products (A stored [K][M], K = the 1.2 M
        // rows: 5.66 against 6.38 ms for 2048 x 256 x 1,228,800) and K >= 1024 (ffn1 dX 5.92 against 6.34 ms, ffn2 forward 5.90 against
        // 6.04) - and the epilogue is one of its plain forms; the K = 256 products (prologue / epilogue bound: 7.02 against 6.70 ms
        // for 1,228,800 x 2048 x 256) and the mask / dropout epilogues stay on gemm6c.  RLT_GEMM6E_ALL=1: every shape (A/B switch)
        static const bool x6_e_all = [] { const char* e = getenv("RLT_GEMM6E_ALL"); return e && atoi(e) != 0; }();
        if (x6_e && gemm6e_epilogue_ok(g) && (ta || g.K >= 1024 || x6_e_all)) {
            if (!ta && tb) rc = launch_gemm6e<false, true>(g, ns, st);
            else if (!ta && !tb) rc = launch_gemm6e<false, false>(g, ns, st);
            else if (!tb) rc = launch_gemm6e<true, false>(g, ns, st);
            else rc = launch_gemm6e<true, true>(g, ns, st);
        } else if (!ta && tb) rc = launch_gemm6c<false, true>(g, ns, st);
        else if (!ta && !tb) rc = launch_gemm6c<false, false>(g, ns, st);
        else if (ta && !tb) rc = launch_gemm6c<true, false>(g, ns, st);
        else rc = launch_gemm6c<true, true>(g, ns, st);
    } else if (gemm_mode() == 2 && gemm6_ok(g) && g.N % BN2 == 0 && !x6_small_only) {
        if (!ta && tb) rc = launch_gemm6b<false, true>(g, ns, st);
        else if (!ta && !tb) rc = launch_gemm6b<false, false>(g, ns, st);
        else if (ta && !tb) rc = launch_gemm6b<true, false>(g, ns, st);
        else rc = launch_gemm6b<true, true>(g, ns, st);
    } else if (gemm_mode() == 2 && gemm6_ok(g)) {
        if (!ta && tb) rc = launch_gemm6<false, true>(g, ns, st);
        else if (!ta && !tb) rc = launch_gemm6<false, false>(g, ns, st);
        else if (ta && !tb) rc = launch_gemm6<true, false>(g, ns, st);
        else rc = launch_gemm6<true, true>(g, ns, st);
    } else if (gemm_mode() == 1 && gemm_big_ok(g, ta, tb)) {
        if (!ta && tb) rc = launch_gemm3b<false, true>(g, ns, st);
        else if (!ta && !tb) rc = launch_gemm3b<false, false>(g, ns, st);
        else if (ta && !tb) rc = launch_gemm3b<true, false>(g, ns, st);
        else rc = launch_gemm3b<true, true>(g, ns, st);
    } else if (gemm_mode() == 1) {
        if (!ta && tb) rc = launch_gemm3<false, true>(g, grid, st);
        else if (!ta && !tb) rc = launch_gemm3<false, false>(g, grid, st);
        else if (ta && !tb) rc = launch_gemm3<true, false>(g, grid, st);
        else rc = launch_gemm3<true, true>(g, grid, st);
    } else {
        if (!ta && tb) rc = launch_gemm<false, true, 16>(g, grid, st);
        else if (!ta && !tb) rc = launch_gemm<false, false, 16>(g, grid, st);
        else if (ta && !tb) rc = launch_gemm<true, false, 16>(g, grid, st);
        else rc = launch_gemm<true, true, 16>(g, grid, st);
    }
    if (rc) return rc;
    if (ns > 1)
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ew_grid((size_t)M * N)), dim3(256), 0, st, g, ns);
    return RLT_LAUNCH_RESULT();
}

int rlt_gemm_ex(int ta, int tb, int M, int N, int K,
                const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                const float* bias, const float* bias2, int flags,
                const float* relu_mask, int ldmask, float mask_scale, float* colsum_a,
                float drop_p, uint32_t seed,
                void* ws, size_t ws_bytes, int precision, void* stream) {
    RLT_PREC_SCOPE(precision);
    return gemm_run(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, bias2, flags, relu_mask, ldmask, mask_scale, colsum_a,
                    drop_p, seed, nullptr, nullptr, ws, ws_bytes, stream);
}

size_t rlt_gemm_bits_words(int M, int N) { return M > 0 && N > 0 ? (size_t)rlt_cdiv(M, 32) * N : 0; }

int rlt_gemm_bits(int ta, int tb, int M, int N, int K,
                  const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                  const float* bias, int flags, float drop_p, uint32_t seed,
                  uint32_t* relu_bits_out, const uint32_t* mask_bits_in, float mask_scale,
                  int precision, void* stream) {
    RLT_PREC_SCOPE(precision);
    RLT_CHECK_ARG((relu_bits_out != nullptr) != (mask_bits_in != nullptr));
    RLT_CHECK_ARG(!relu_bits_out || (flags & RLT_GEMM_RELU));
    RLT_CHECK_ARG(drop_p == 0.f || relu_bits_out);
    RLT_CHECK_SHAPE(N % 32 == 0);
    return gemm_run(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, nullptr, flags, nullptr, 0, mask_scale, nullptr,
                    drop_p, seed, relu_bits_out, mask_bits_in, nullptr, 0, stream);
}

size_t rlt_colsum_workspace(int T, int N) {
    if (T <= 0 || N <= 0) return 0;
    return (size_t)rlt_cdiv(T, CS_ROWS_PER_WG) * N * sizeof(float);
}

int rlt_colsum(const float* X, int ldx, int T, int N, float* out, int accumulate,
               void* ws, size_t ws_bytes, void* stream) {
    RLT_CHECK_ARG(X && out && ws && T > 0 && N > 0 && ldx >= N);
    if (ws_bytes < rlt_colsum_workspace(T, N)) return RLT_E_WORKSPACE;
    const int nchunk = rlt_cdiv(T, CS_ROWS_PER_WG);
    hipStream_t st = rlt_stream(stream);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(rlt_cdiv(N, 64), nchunk), dim3(256), 0, st, X, ldx, T, N, (float*)ws);
    hipLaunchKernelGGL(rlt_rows_reduce_kernel, dim3(rlt_cdiv(N, 16)), dim3(256), 0, st, (const float*)ws, nchunk, N, N, N, out, out, accumulate);
    return RLT_LAUNCH_RESULT();
}

int rlt_segment_colsum(const float* X, int ldx, int G, int R, int N, float* out, int ldo,
                       int accumulate, void* stream) {
    RLT_CHECK_ARG(X && out && G > 0 && R > 0 && N > 0 && ldx >= N && ldo >= N);
    hipLaunchKernelGGL(segment_colsum_kernel, dim3(G), dim3(256), 0, rlt_stream(stream), X, ldx, R, N, out, ldo, accumulate);
    return RLT_LAUNCH_RESULT();
}

size_t rlt_narrow_dw_workspace(int T, int M) {
    if (T <= 0 || M <= 0) return 0;
    const int rows = rlt_cdiv(T, NDW_CHUNKS);
    const int nchunk = rlt_cdiv(T, rows);
    return ((size_t)nchunk * 4 * M + (size_t)4 * M) * sizeof(float);
}

int rlt_narrow_dw(const float* A, int lda, const float* X, int ldx, int I, int T, int M,
                  float* dW, float* db, void* ws, size_t ws_bytes, void* stream) {
    RLT_CHECK_ARG(A && X && dW && ws && T > 0 && M > 0 && lda >= M && ldx >= I);
    RLT_CHECK_SHAPE(I >= 1 && I <= 3 && M % 4 == 0 && lda % 4 == 0);
    if (!rlt_aligned16(A)) return RLT_E_ALIGN;
    if (ws_bytes < rlt_narrow_dw_workspace(T, M)) return RLT_E_WORKSPACE;
    const int rows = rlt_cdiv(T, NDW_CHUNKS);
    const int nchunk = rlt_cdiv(T, rows);
    float* partial = (float*)ws;
    float* sums = partial + (size_t)nchunk * 4 * M;
    hipStream_t st = rlt_stream(stream);
    hipLaunchKernelGGL(narrow_dw_partial_kernel, dim3(nchunk, rlt_cdiv(M, 1024)), dim3(256), 0, st, A, lda, X, ldx, I, T, M, rows, partial);
    hipLaunchKernelGGL(rlt_rows_reduce_kernel, dim3(rlt_cdiv(4 * M, 16)), dim3(256), 0, st, (const float*)partial, nchunk, 4 * M,
                       4 * M, 4 * M, sums, sums, 0);
    hipLaunchKernelGGL(narrow_dw_scatter_kernel, dim3(rlt_cdiv(M, 256)), dim3(256), 0, st, (const float*)sums, M, I, dW, db);
    return RLT_LAUNCH_RESULT();
}

int rlt_relu_bwd(float* dX, const float* Y, size_t n, void* stream) {
    RLT_CHECK_ARG(dX && Y && n > 0);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, rlt_stream(stream), dX, Y, n);
    return RLT_LAUNCH_RESULT();
}

int rlt_scale(float* x, const float* scale, size_t n, void* stream) {
    RLT_CHECK_ARG(x && scale && n > 0);
    hipLaunchKernelGGL(scale_kernel, dim3(ew_grid(n)), dim3(256), 0, rlt_stream(stream), x, scale, n);
    return RLT_LAUNCH_RESULT();
}

}  // extern "C"

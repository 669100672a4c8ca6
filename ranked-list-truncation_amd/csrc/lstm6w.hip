// BiLSTM recurrences of the fp32-faithful mode ("bf16x6") with ONE wavefront per SIMD (row M2 of SURVEY.md section 8a; nn.LSTM
// call sites models/AttnCut.py:8,17, models/MtAttnCut.py:8,22, models/MMOECut.py:63,88).
//
// The recurrence is serial in time, so a launch costs S steps x the latency of a step; rounds 3-4 ran a step as "all wavefronts
// multiply, barrier, all wavefronts do the gate arithmetic, barrier": the matrix pipe idles through the vector phase and the
// vector ALU through the MFMA phase (8.2-11.3 us per step against 2.9 us of MFMAs).  Here a workgroup of four wavefronts (one per
// SIMD, 512 registers each) owns 32 lists as two INDEPENDENT 16-list halves and alternates ticks:
//     tick (X, Y):  MFMA chain of half X's step   ||   gate arithmetic, stores and the h exchange of half Y's step
// in ONE instruction stream - every v_mfma_f32_16x16x32_bf16 is followed by a fenced gap with ~8 cycles of half Y's vector work
// (tools/gen_lstm6w_body.py writes the tick body), which the MFMA hides (profiles/r05_notes.md).  One barrier per tick.
//
// Forward, wavefront w: hidden units 32w .. 32w+31 and their four gates = eight 16-row blocks rb = 4 ub + gate (unit block ub);
// A operand = W_hh rows (row l&15 of the block, k = 32 ks + 8 (l>>4) + j), B = h_{t-1} (k, list l&15), C: lane holds the four
// gates of units 4 (l>>4) + r of list l&15 in acc[4 ub + gate][r] - the cell update is lane-local.  W_hh's h and m planes are the
// wavefront's 256 AGPRs (64 fragments); the l plane - one of the six products - sits in LDS in fragment order (128 KB, each
// wavefront reads back what it wrote); h_{t-1} of a half: three planes x 16 lists x 272 B (16 B of padding: conflict-free
// ds_read_b128 with the k-step as an immediate offset).  Pre-activations of the next step are fetched one tick pair ahead into
// registers; loads and stores are buffer instructions with the row count as bound, so lists beyond B are dropped by the hardware
// (no exec-masked branches: hipcc keeps exact vmcnt counts).  Layer 0 (XIN): x W_ih^T + b_ih + b_hh is ONE more MFMA per row
// block - the 24 (plane of W_ih, plane of x) pairs of the six products of the four values (x0, x1, x2, 1) fill the 32 k slots.
#include "common.h"
#include "lstm_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int HID = 128;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int W6_WL = 4 * 8 * 4 * 64 * 16;       // l plane of W_hh in fragment order: [wavefront][row block][k-step][lane] x 16 B
constexpr int W6_ROW = 272;                      // bytes per list row of an h plane (128 bf16 + 16 B of padding)
constexpr int W6_PLANE = 16 * W6_ROW;
constexpr int W6_HALF = 3 * W6_PLANE;
constexpr size_t W6_LDS = (size_t)W6_WL + 2 * W6_HALF;
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ uint32_t pk2w(float a, float b) {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    v2 t = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// exact three-way split of four values: x = h + m + l, 8 + 8 + 8 significand bits
__device__ __forceinline__ void split4w(float a, float b, float c, float d, uint2& hi, uint2& mid, uint2& lo) {
    hi.x = pk2w(a, b);
    hi.y = pk2w(c, d);
    asm("" : "+v"(hi.x), "+v"(hi.y));          // keep the packed pair, do not re-convert
    const float ra = a - bf_lo(hi.x), rb = b - bf_hi(hi.x), rc = c - bf_lo(hi.y), rd = d - bf_hi(hi.y);
    mid.x = pk2w(ra, rb);
    mid.y = pk2w(rc, rd);
    asm("" : "+v"(mid.x), "+v"(mid.y));
    lo.x = pk2w(ra - bf_lo(mid.x), rb - bf_hi(mid.x));
    lo.y = pk2w(rc - bf_lo(mid.y), rd - bf_hi(mid.y));
}
__device__ __forceinline__ bf16x8 frag8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return __builtin_bit_cast(bf16x8, make_uint4(a, b, c, d));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

enum { TICK_FULL = 0, TICK_CHAIN = 1, TICK_ELEM = 2 };
#ifndef RLT_W6_STAGGER
#define RLT_W6_STAGGER 0    // experiment: wavefront w starts a tick 16 * w * RLT_W6_STAGGER cycles late (memory / LDS instructions of the four out of step)
#endif
#ifndef RLT_W6_ABL
#define RLT_W6_ABL 0        // timing ablations (tools/build_variant.py; wrong results): 1 no stores, 2 no pre-activation loads, 4 no MFMAs, 8 no exp / rcp
#endif

// SINGLE (batches that give every CU at most one 16-list half: B <= 16 x 128 per direction): a workgroup owns ONE half and a step is
// chain tick, then element-wise tick of the SAME half (nothing to overlap, but a step is one chain long instead of two: the latency
// of the recurrence halves - the reference's batch sizes, hyper_parameter_*.conf batch_size = 63 / 64).
template <bool XIN, bool SINGLE>
__global__ __launch_bounds__(256, 1) void bilstm6w_fwd_kernel(float* __restrict__ gates, const float* __restrict__ w_hh_f,
                                                              const float* __restrict__ w_hh_r, int S, int B,
                                                              float* __restrict__ h_out, float* __restrict__ c_out, RltXIn xi) {
    extern __shared__ __attribute__((aligned(16))) uint8_t sm6w[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 15, q = lane >> 4;
    const int dir = blockIdx.y, b0 = blockIdx.x * (SINGLE ? 16 : 32);
    uint4* wl_s = reinterpret_cast<uint4*>(sm6w);
    uint8_t* hx = sm6w + W6_WL;

    // ---- stationary operands ----
    bf16x8 wh[8][4], wm[8][4], wx[8];
    {
        const float* whh = dir ? w_hh_r : w_hh_f;
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) {
            const int row = (rb & 3) * HID + 32 * w + 16 * (rb >> 2) + n;
            const float* wp = whh + (size_t)row * HID + 8 * q;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float4 v0 = *reinterpret_cast<const float4*>(wp + 32 * ks);
                const float4 v1 = *reinterpret_cast<const float4*>(wp + 32 * ks + 4);
                uint2 h0, m0, l0, h1, m1, l1;
                split4w(v0.x, v0.y, v0.z, v0.w, h0, m0, l0);
                split4w(v1.x, v1.y, v1.z, v1.w, h1, m1, l1);
                wh[rb][ks] = frag8(h0.x, h0.y, h1.x, h1.y);
                wm[rb][ks] = frag8(m0.x, m0.y, m1.x, m1.y);
                wl_s[((w * 8 + rb) * 4 + ks) * 64 + lane] = make_uint4(l0.x, l0.y, l1.x, l1.y);
                asm volatile("" : "+a"(wh[rb][ks]), "+a"(wm[rb][ks]));      // into their AGPRs at once: no MFMA of any tick sees a register move
            }
            if (XIN) {
                // k slots 8q + j: j 0-3 <-> (x0, x1, x2, 1) against plane {h, m, l, -}[q] of (W_ih[row][0..2], b_ih + b_hh);
                //                 j 4-7 <-> the same four against plane {h, h, m, -}[q]; the x operand pairs them with
                //                 {xh, xh, xh, -} / {xm, xl, xm, -}: all six products of the split (see XB below)
                const float* wr = xi.w_ih[dir] + (size_t)row * xi.I;
                uint2 h0, m0, l0;
                split4w(wr[0], xi.I > 1 ? wr[1] : 0.f, xi.I > 2 ? wr[2] : 0.f, xi.b_ih[dir][row] + xi.b_hh[dir][row], h0, m0, l0);
                const uint2 z = make_uint2(0u, 0u);
                const uint2 s0 = q == 0 ? h0 : q == 1 ? m0 : q == 2 ? l0 : z, s1 = q == 0 ? h0 : q == 1 ? h0 : q == 2 ? m0 : z;
                wx[rb] = frag8(s0.x, s0.y, s1.x, s1.y);
            } else {
                wx[rb] = frag8(0u, 0u, 0u, 0u);
            }
        }
    }
    for (int i = tid; i < 2 * W6_HALF / 16; i += 256) reinterpret_cast<uint4*>(hx)[i] = make_uint4(0u, 0u, 0u, 0u);      // h_{-1} = 0

    // ---- per-lane addresses (everything else is an immediate offset) ----
    const uint32_t voff_g[2] = {(uint32_t)(b0 + n) * 4096u + dir * 2048u + 128u * w + 16u * q,
                                (uint32_t)(b0 + 16 + n) * 4096u + dir * 2048u + 128u * w + 16u * q};      // gate rows: + 512 g + 64 ub
    const uint32_t voff_h[2] = {(uint32_t)(b0 + n) * 1024u + dir * 512u + 128u * w + 16u * q,
                                (uint32_t)(b0 + 16 + n) * 1024u + dir * 512u + 128u * w + 16u * q};       // h / c rows: + 64 ub
    const uint32_t voff_x[2] = {(uint32_t)(b0 + n) * 4u * xi.I, (uint32_t)(b0 + 16 + n) * 4u * xi.I};
    const uint8_t* hrd[2] = {hx + n * W6_ROW + 16 * q, hx + W6_HALF + n * W6_ROW + 16 * q};                // + plane + 64 ks
    uint8_t* hwr[2] = {hx + n * W6_ROW + 64 * w + 8 * q, hx + W6_HALF + n * W6_ROW + 64 * w + 8 * q};       // + plane + 32 ub
    const uint8_t* wlrd = sm6w + w * 32768 + lane * 16;                                                     // + 1024 (4 rb + ks)
    auto step_of = [&](int t) { return dir ? S - 1 - t : t; };
    auto rs_gates = [&](int t) { return rsrc_of(gates + (size_t)step_of(t) * B * (8 * HID), (uint32_t)B * 4096u); };
    auto rs_h = [&](int t) { return rsrc_of(h_out + (size_t)step_of(t) * B * (2 * HID), (uint32_t)B * 1024u); };
    auto rs_c = [&](int t) { return rsrc_of(c_out + (size_t)step_of(t) * B * (2 * HID), (uint32_t)B * 1024u); };
    auto rs_x = [&](int t) { return rsrc_of(xi.x + (size_t)step_of(t) * B * xi.I, (uint32_t)B * 4u * xi.I); };

    // ---- state ----
    f32x4 acc[2][8], gin[2][8], cst[2][2];
    bf16x8 bfr[2][3], lfr[4], xb[2];
    float xv[2][3];
    uint2 sp[3];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) { acc[hf][rb] = z4; gin[hf][rb] = z4; }
        cst[hf][0] = z4; cst[hf][1] = z4;
        xb[hf] = frag8(0u, 0u, 0u, 0u);
        xv[hf][0] = xv[hf][1] = xv[hf][2] = 0.f;
    }
    sp[0] = sp[1] = sp[2] = make_uint2(0u, 0u);
#pragma unroll
    for (int i = 0; i < 6; ++i) (&bfr[0][0])[i] = frag8(0u, 0u, 0u, 0u);
    // (all three x columns are fetched whatever I is - a row's neighbours or, past the end, the buffer bound's zero - and the
    //  columns beyond I replaced by zero with a select: no branch in the tick body)
    auto load_x3 = [&](int hf, __amdgpu_buffer_rsrc_t r) __attribute__((always_inline)) {
        const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff_x[hf], 0, 0));
        const float x1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff_x[hf] + 4, 0, 0));
        const float x2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff_x[hf] + 8, 0, 0));
        xv[hf][0] = x0;          // (raw: the selects sit at the use, a tick pair later - here they would wait for the loads at once)
        xv[hf][1] = x1;
        xv[hf][2] = x2;
    };
    auto load_x = [&](int hf, int t) __attribute__((always_inline)) { load_x3(hf, rs_x(t)); };
    auto make_xb = [&](int hf) __attribute__((always_inline)) {
        uint2 xh, xm, xl;
        split4w(xv[hf][0], xi.I > 1 ? xv[hf][1] : 0.f, xi.I > 2 ? xv[hf][2] : 0.f, 1.f, xh, xm, xl);
        const uint2 z = make_uint2(0u, 0u);
        const uint2 s0 = q < 3 ? xh : z, s1 = q == 1 ? xl : q == 3 ? z : xm;
        xb[hf] = frag8(s0.x, s0.y, s1.x, s1.y);
    };
    if (XIN) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            load_x(hf, 0);
            make_xb(hf);
            load_x(hf, S > 1 ? 1 : 0);
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) asm volatile("" : "+v"(xv[hf][0]), "+v"(xv[hf][1]), "+v"(xv[hf][2]));      // (consumed: see below)
    } else {
        const __amdgpu_buffer_rsrc_t r = rs_gates(0);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int rb = 0; rb < 8; ++rb)
                gin[hf][rb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff_g[hf] + 512 * (rb & 3) + 64 * (rb >> 2), 0, 0));
        // consumed here (an empty asm statement reads them): with these loads still pending at the first tick, hipcc carries the
        // prologue's picture of outstanding operations into the loop and every tick waits with the prologue's small vmcnt values
        // (7, 5, 3, 1 / 15, 13, 11, 9 measured in the ISA) - i.e. for stores issued a few hundred cycles earlier - instead of 39, 38, ...
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int rb = 0; rb < 8; ++rb) asm volatile("" : "+v"(gin[hf][rb]));
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) lfr[j] = *reinterpret_cast<const bf16x8*>(wlrd + 1024 * (4 * j));          // group 0: row blocks 0-3, k-step 0

#define GAP_END __builtin_amdgcn_sched_barrier(0)
    // One tick: chain of half X at step tX, element-wise part of half Y = 1 - X at step tY.  MODE compiles one side out (first / last tick).
    auto tick = [&](auto XC, auto MC, int tX, int tY) __attribute__((always_inline)) {
        constexpr int X = decltype(XC)::value, Y = 1 - X, MODE = decltype(MC)::value;
        constexpr bool CH = MODE != TICK_ELEM, EL = MODE != TICK_CHAIN;
        (void)tX;
        const int tn = tY + 1 < S ? tY + 1 : tY;
        const __amdgpu_buffer_rsrc_t rg = rs_gates(tY), rgn = rs_gates(tn), rh = rs_h(tY), rc = rs_c(tY), rxn = rs_x(tY + 2 < S ? tY + 2 : S - 1);
        (void)rgn; (void)rxn;
        // ---- chain side ----
        auto RB = [&](int ks, int pl) __attribute__((always_inline)) {
            if (CH) bfr[ks & 1][pl] = *reinterpret_cast<const bf16x8*>(hrd[X] + pl * W6_PLANE + 64 * ks);
        };
        auto RL = [&](int rb, int ks, int j) __attribute__((always_inline)) {
            if (CH) lfr[j] = *reinterpret_cast<const bf16x8*>(wlrd + 1024 * (4 * rb + ks));
        };
        auto MF = [&](int rb, int ks, int p, int first) __attribute__((always_inline)) {
            if (!CH || (RLT_W6_ABL & 4)) return;
            const int bp = p == 2 ? 2 : (p == 0 || p == 4) ? 1 : 0;              // operand plane: m, h, l, h, m, h
            const bf16x8 bv = bfr[ks & 1][bp];
            f32x4& d = acc[X][rb];
            if (p == 1) {
                const bf16x8 lv = lfr[rb & 3];
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(lv), "v"(bv));
            } else {
                const bf16x8 av = (p == 0 || p == 3) ? wm[rb][ks] : wh[rb][ks];  // W plane: m, l, h, m, h, h
                if (first) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(d) : "a"(av), "v"(bv));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(av), "v"(bv));
            }
        };
        auto MX = [&](int rb) __attribute__((always_inline)) {
            if (!CH) return;
            const bf16x8 av = wx[rb], bv = xb[X];
            f32x4& d = acc[X][rb];
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(av), "v"(bv));
        };
        // ---- element-wise side (in place on half Y's accumulators: i, f, g, o of block ub in acc[Y][4 ub + 0..3]) ----
        auto EA = [&](int u, int g) __attribute__((always_inline)) { if (EL) acc[Y][4 * u + g] += gin[Y][4 * u + g]; };
        auto LG = [&](int u, int g) __attribute__((always_inline)) {
            if (EL && !(RLT_W6_ABL & 2)) gin[Y][4 * u + g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rgn, voff_g[Y] + 512 * g + 64 * u, 0, 0));
        };
        auto ES = [&](int u, int g) __attribute__((always_inline)) { if (EL) acc[Y][4 * u + g] *= (g == 2 ? -2.f * LOG2E : -LOG2E); };
        auto EX = [&](int u, int g, int r) __attribute__((always_inline)) { if (EL && !(RLT_W6_ABL & 8)) acc[Y][4 * u + g][r] = rlt_exp2(acc[Y][4 * u + g][r]); };
        auto E1 = [&](int u, int g) __attribute__((always_inline)) { if (EL) acc[Y][4 * u + g] += 1.f; };
        auto ER = [&](int u, int g, int r) __attribute__((always_inline)) { if (EL && !(RLT_W6_ABL & 8)) acc[Y][4 * u + g][r] = __builtin_amdgcn_rcpf(acc[Y][4 * u + g][r]); };
        auto EG = [&](int u) __attribute__((always_inline)) { if (EL) acc[Y][4 * u + 2] = 2.f * acc[Y][4 * u + 2] - 1.f; };
        auto SG = [&](int u, int g) __attribute__((always_inline)) {
            if (EL && !(RLT_W6_ABL & 1)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[Y][4 * u + g]), rg, voff_g[Y] + 512 * g + 64 * u, 0, 0);
        };
        auto EC1 = [&](int u) __attribute__((always_inline)) { if (EL) cst[Y][u] *= acc[Y][4 * u + 1]; };
        auto EC2 = [&](int u) __attribute__((always_inline)) { if (EL) cst[Y][u] += acc[Y][4 * u] * acc[Y][4 * u + 2]; };
        auto SC = [&](int u) __attribute__((always_inline)) {
            if (EL && !(RLT_W6_ABL & 1)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, cst[Y][u]), rc, voff_h[Y] + 64 * u, 0, 0);
        };
        // tanh(c) in the registers of the i gate, h in those of the f gate, split residuals in those of the g gate
        auto ET = [&](int u) __attribute__((always_inline)) { if (EL) acc[Y][4 * u] = cst[Y][u] * (-2.f * LOG2E); };
        auto EXC = [&](int u, int r) __attribute__((always_inline)) { if (EL) acc[Y][4 * u][r] = rlt_exp2(acc[Y][4 * u][r]); };
        auto E1C = [&](int u) __attribute__((always_inline)) { if (EL) acc[Y][4 * u] += 1.f; };
        auto ERC = [&](int u, int r) __attribute__((always_inline)) { if (EL) acc[Y][4 * u][r] = __builtin_amdgcn_rcpf(acc[Y][4 * u][r]); };
        auto EH1 = [&](int u) __attribute__((always_inline)) { if (EL) acc[Y][4 * u] = 2.f * acc[Y][4 * u] - 1.f; };
        auto EH2 = [&](int u) __attribute__((always_inline)) { if (EL) acc[Y][4 * u + 1] = acc[Y][4 * u + 3] * acc[Y][4 * u]; };
        auto SH = [&](int u) __attribute__((always_inline)) {
            if (EL && !(RLT_W6_ABL & 1)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[Y][4 * u + 1]), rh, voff_h[Y] + 64 * u, 0, 0);
        };
        auto SP = [&](int u, int part) __attribute__((always_inline)) {
            if (!EL) return;
            const f32x4& hv = acc[Y][4 * u + 1];
            f32x4& rs = acc[Y][4 * u + 2];
            if (part == 0) { sp[0].x = pk2w(hv[0], hv[1]); sp[0].y = pk2w(hv[2], hv[3]); asm("" : "+v"(sp[0].x), "+v"(sp[0].y)); }
            if (part == 1) { rs[0] = hv[0] - bf_lo(sp[0].x); rs[1] = hv[1] - bf_hi(sp[0].x); }
            if (part == 2) { rs[2] = hv[2] - bf_lo(sp[0].y); rs[3] = hv[3] - bf_hi(sp[0].y); }
            if (part == 3) { sp[1].x = pk2w(rs[0], rs[1]); sp[1].y = pk2w(rs[2], rs[3]); asm("" : "+v"(sp[1].x), "+v"(sp[1].y)); }
            if (part == 4) { rs[0] -= bf_lo(sp[1].x); rs[1] -= bf_hi(sp[1].x); }
            if (part == 5) { rs[2] -= bf_lo(sp[1].y); rs[3] -= bf_hi(sp[1].y); }
            if (part == 6) { sp[2].x = pk2w(rs[0], rs[1]); sp[2].y = pk2w(rs[2], rs[3]); }
        };
        auto LW = [&](int u, int pl) __attribute__((always_inline)) {
            if (EL) *reinterpret_cast<uint2*>(hwr[Y] + pl * W6_PLANE + 32 * u) = sp[pl];
        };
        auto XB = [&](int part) __attribute__((always_inline)) {           // XIN: half Y's operand of its next chain; then the x row after that
            if (!EL || !XIN) return;
            if (part == 0) make_xb(Y);
            if (part == 1) load_x3(Y, rxn);
        };
        if (CH) { RB(0, 0); RB(0, 1); RB(0, 2); }
        GAP_END;
        // two orders of the element-wise work: both unit blocks in lock step (the two-half kernels: the 64-byte halves of a 128-byte
        // line are loaded / stored back to back - forward layer 1 at 4096 lists 3.65 -> 2.75 ms) or block after block (SINGLE: a step
        // is latency there, 3.36 against 3.67 us)
        if constexpr (XIN && SINGLE) {
#include "lstm6w_fwd_xin_seq_body.inc"
        } else if constexpr (XIN) {
#include "lstm6w_fwd_xin_body.inc"
        } else if constexpr (SINGLE) {
#include "lstm6w_fwd_seq_body.inc"
        } else {
#include "lstm6w_fwd_body.inc"
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int i = 0; i < w * RLT_W6_STAGGER; ++i) asm volatile("s_nop 15");
    };
    const std::integral_constant<int, 0> H0;
    const std::integral_constant<int, 1> H1;
    const std::integral_constant<int, TICK_FULL> MFULL;
    const std::integral_constant<int, TICK_CHAIN> MCHAIN;
    const std::integral_constant<int, TICK_ELEM> MELEM;

    if constexpr (SINGLE) {
        for (int t = 0; t < S; ++t) {
#pragma unroll
            for (int rb = 0; rb < 8; ++rb)
                asm volatile("" : "+a"(wh[rb][0]), "+a"(wh[rb][1]), "+a"(wh[rb][2]), "+a"(wh[rb][3]), "+a"(wm[rb][0]), "+a"(wm[rb][1]),
                             "+a"(wm[rb][2]), "+a"(wm[rb][3]));
            tick(H0, MCHAIN, t, t);                 // chain of half 0
            tick(H1, MELEM, t, t);                  // its element-wise step (Y = 0)
        }
        return;
    }
    tick(H0, MCHAIN, 0, 0);
    if (S > 1) {       // the first tick pair outside the loop: the loop is then entered with the picture of outstanding loads and stores
                       // its back edge carries, and hipcc's vmcnt counts in it are the steady-state ones (not the prologue's few)
        tick(H1, MFULL, 0, 0);
        tick(H0, MFULL, 1, 0);
    }
    for (int t = 1; t + 1 < S; ++t) {
#pragma unroll
        for (int rb = 0; rb < 8; ++rb)                // (their only uses are "a" operands: keep them in AGPRs across the back edge)
            asm volatile("" : "+a"(wh[rb][0]), "+a"(wh[rb][1]), "+a"(wh[rb][2]), "+a"(wh[rb][3]), "+a"(wm[rb][0]), "+a"(wm[rb][1]),
                         "+a"(wm[rb][2]), "+a"(wm[rb][3]));
        tick(H1, MFULL, t, t);
        tick(H0, MFULL, t + 1, t);
    }
    tick(H1, MFULL, S - 1, S - 1);
    tick(H0, MELEM, S - 1, S - 1);
#undef GAP_END
}


// ---------------------------------------------------------------------------------------------------------------------------
// Backward recurrence, same scheme.  dh_{t-1}[unit][list] = sum over the 512 gate rows of W_hh[row][unit] dA_t[row][list]: wavefront w
// owns OUTPUT units 32w .. 32w+31 (two 16-unit blocks) over the whole contraction - no partial sums cross wavefronts (rounds 1-4
// split the rows over 16 wavefronts and reduced 16 partial tiles through LDS, two barriers and 132 KB per step).  A operand =
// W_hh^T (row l&15 = output unit, k slot 32 ks + 8 (l>>4) + j <-> gate row of unit 8 ks + 2 (l>>4) + (j>>2), gate j&3), B = the
// three planes of dA_t, which every wavefront reads whole from LDS: [plane][list][512 slots] bf16, 16-byte chunks XOR-swizzled
// with the list (no padding: LDS is full).  W_hh^T: h and m planes = the 256 AGPRs, l plane half in VGPRs (k-steps 0-7) and half in
// LDS (k-steps 8-15, 64 KB); dA planes of the two halves 96 KB: 160 KB exactly.  C: lane holds units 4 (l>>4) + r of list l&15 -
// where the gate arithmetic of those units runs (dA of unit u, gates i f g o = slots 4u .. 4u+3: a lane's four units are 32
// contiguous bytes of a plane).  Inputs of a step (gates, c_{t-1}, dL/dh) are fetched ONE tick ahead into a 12 x 4 register
// pool that both halves share, each slot reloaded for the other half as soon as this half has consumed it.
constexpr int WB_PLANE = 16 * 1024;                 // one plane of dA of a half: 16 lists x 512 bf16
constexpr int WB_HALF = 3 * WB_PLANE;
constexpr int WB_WL = 4 * 2 * 8 * 64 * 16;          // l plane of W_hh^T, k-steps 8-15: [wavefront][block][k-step - 8][lane] x 16 B
constexpr size_t WB_LDS = (size_t)2 * WB_HALF + WB_WL;
static_assert(WB_LDS == 160 * 1024, "the backward recurrence fills LDS exactly");

template <bool SINGLE>
__global__ __launch_bounds__(256, 1) void bilstm6w_bwd_kernel(float* __restrict__ gates, const float* __restrict__ cst,
                                                              const float* __restrict__ w_hh_f, const float* __restrict__ w_hh_r,
                                                              const float* __restrict__ d_hout, int S, int B) {
    extern __shared__ __attribute__((aligned(16))) uint8_t sm6w[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 15, q = lane >> 4;
    const int dir = blockIdx.y, b0 = blockIdx.x * (SINGLE ? 16 : 32);
    uint8_t* dax = sm6w;                                                    // [half][plane][list][64 chunks]
    uint4* wl_s = reinterpret_cast<uint4*>(sm6w + 2 * WB_HALF);

    // ---- stationary operand: W_hh^T ----
    bf16x8 wh[2][16], wm[2][16], wlv[2][8];
    {
        const float* whh = dir ? w_hh_r : w_hh_f;
#pragma unroll
        for (int ub = 0; ub < 2; ++ub) {
            const float* wp = whh + 32 * w + 16 * ub + n + (size_t)(2 * q) * HID;      // column = output unit
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                asm volatile("" : "+v"(wp));              // (opaque: hipcc would form all 256 addresses up front and spill them)
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = wp[(size_t)((j & 3) * HID + 8 * ks + (j >> 2)) * HID];
                uint2 h0, m0, l0, h1, m1, l1;
                split4w(v[0], v[1], v[2], v[3], h0, m0, l0);
                split4w(v[4], v[5], v[6], v[7], h1, m1, l1);
                wh[ub][ks] = frag8(h0.x, h0.y, h1.x, h1.y);
                wm[ub][ks] = frag8(m0.x, m0.y, m1.x, m1.y);
                asm volatile("" : "+a"(wh[ub][ks]), "+a"(wm[ub][ks]) : : "memory");     // ("memory": the loads of later fragments stay behind)
                if (ks >= 8) wl_s[((w * 2 + ub) * 8 + (ks - 8)) * 64 + lane] = make_uint4(l0.x, l0.y, l1.x, l1.y);
            }
        }
        // the l fragments of k-steps 0-7 in a second pass, when the other planes are in their AGPRs (in one pass hipcc spills them
        // through scratch during the set-up - harmless, but with a scratch access on record it waits with vmcnt(0) in every tick)
#pragma unroll
        for (int ub = 0; ub < 2; ++ub) {
            const float* wp = whh + 32 * w + 16 * ub + n + (size_t)(2 * q) * HID;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                asm volatile("" : "+v"(wp));
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = wp[(size_t)((j & 3) * HID + 8 * ks + (j >> 2)) * HID];
                uint2 h0, m0, l0, h1, m1, l1;
                split4w(v[0], v[1], v[2], v[3], h0, m0, l0);
                split4w(v[4], v[5], v[6], v[7], h1, m1, l1);
                wlv[ub][ks] = frag8(l0.x, l0.y, l1.x, l1.y);
                asm volatile("" : "+v"(wlv[ub][ks]) : : "memory");
            }
        }
    }

    // ---- per-lane addresses ----
    const uint32_t voff_g[2] = {(uint32_t)(b0 + n) * 4096u + dir * 2048u + 128u * w + 16u * q,
                                (uint32_t)(b0 + 16 + n) * 4096u + dir * 2048u + 128u * w + 16u * q};      // gate rows: + 512 g + 64 ub
    const uint32_t voff_h[2] = {(uint32_t)(b0 + n) * 1024u + dir * 512u + 128u * w + 16u * q,
                                (uint32_t)(b0 + 16 + n) * 1024u + dir * 512u + 128u * w + 16u * q};       // c / dh rows: + 64 ub
    // reads of dA: chunk 4 ks + q of list n sits at (4 ks + q) ^ n = 4 (ks ^ (n >> 2)) + (q ^ (n & 3)): one base per ks & 3, + 256 (ks >> 2)
    const uint8_t* drd[2][4];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int j = 0; j < 4; ++j) drd[hf][j] = dax + hf * WB_HALF + n * 1024 + 64 * (j ^ (n >> 2)) + 16 * (q ^ (n & 3));
    // writes: a lane's block-ub units are chunks 16 w + 8 ub + 2 q (units r = 0, 1) and the next one (r = 2, 3)
    uint32_t dwr[2][2];                 // (byte offsets into LDS: a pointer that has been through an integer XOR becomes a FLAT access)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int ub = 0; ub < 2; ++ub) dwr[hf][ub] = hf * WB_HALF + n * 1024 + 16 * ((16 * w + 8 * ub + 2 * q) ^ n);
    const uint8_t* wlrd = sm6w + 2 * WB_HALF + w * 16384 + lane * 16;                                       // + 1024 (8 ub + ks - 8)
    // elem index tt = 0 .. S-1 walks time backwards: t = S-1-tt, position s = t (forward direction) or S-1-t (reverse)
    auto pos_of = [&](int tt) { return dir ? tt : S - 1 - tt; };
    auto rs_gates = [&](int tt) { return rsrc_of(gates + (size_t)pos_of(tt) * B * (8 * HID), (uint32_t)B * 4096u); };
    auto rs_dh = [&](int tt) { return rsrc_of(d_hout + (size_t)pos_of(tt) * B * (2 * HID), (uint32_t)B * 1024u); };
    auto rs_cprev = [&](int tt) {      // c_{t-1}: the position one step earlier in time; t = 0 has none (bound 0: reads return zero)
        const int sp = dir ? pos_of(tt) + 1 : pos_of(tt) - 1;
        const bool has = tt + 1 < S;
        return rsrc_of(cst + (size_t)(has ? sp : 0) * B * (2 * HID), has ? (uint32_t)B * 1024u : 0u);
    };

    // ---- state ----
    f32x4 in[12], ct[2][2], dc[2][2], acc[2][2], tcv, dhv, dcu, t0, t1, t2;
    bf16x8 bfr[2][3], lfr[2];
    uint32_t h4[4], m4[4], l4[4];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int ub = 0; ub < 2; ++ub) { dc[hf][ub] = z4; acc[hf][ub] = z4; }
    tcv = dhv = dcu = t0 = t1 = t2 = z4;
#pragma unroll
    for (int i = 0; i < 4; ++i) h4[i] = m4[i] = l4[i] = 0u;
#pragma unroll
    for (int i = 0; i < 6; ++i) (&bfr[0][0])[i] = frag8(0u, 0u, 0u, 0u);
    lfr[0] = lfr[1] = frag8(0u, 0u, 0u, 0u);
    {
        const __amdgpu_buffer_rsrc_t rc = rsrc_of(cst + (size_t)pos_of(0) * B * (2 * HID), (uint32_t)B * 1024u);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int ub = 0; ub < 2; ++ub)
                ct[hf][ub] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rc, voff_h[hf] + 64 * ub, 0, 0));
    }
    auto load_slot = [&](int k, int hf, __amdgpu_buffer_rsrc_t rg, __amdgpu_buffer_rsrc_t rcp, __amdgpu_buffer_rsrc_t rdh) __attribute__((always_inline)) {
        if (k < 8) in[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, voff_g[hf] + 512 * (k & 3) + 64 * (k >> 2), 0, 0));
        else if (k < 10) in[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rcp, voff_h[hf] + 64 * (k - 8), 0, 0));
        else in[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdh, voff_h[hf] + 64 * (k - 10), 0, 0));
    };
    {
        const __amdgpu_buffer_rsrc_t rg = rs_gates(0), rcp = rs_cprev(0), rdh = rs_dh(0);
#pragma unroll
        for (int k = 0; k < 12; ++k) load_slot(k, 0, rg, rcp, rdh);
    }
    // consumed here: no load is pending when the first tick starts (hipcc would carry the prologue's vmcnt picture into the loop)
#pragma unroll
    for (int k = 0; k < 12; ++k) asm volatile("" : "+v"(in[k]));
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) asm volatile("" : "+v"(ct[hf][0]), "+v"(ct[hf][1]));
    __syncthreads();

#define GAP_END __builtin_amdgcn_sched_barrier(0)
    // One tick: chain of half X on the dA its last element-wise tick left in LDS; element-wise step ttY of half Y = 1 - X, whose pool slots
    // are reloaded for half X's step ttN as they are consumed.
    auto tick = [&](auto XC, auto MC, int ttY, int ttN) __attribute__((always_inline)) {
        constexpr int X = decltype(XC)::value, Y = 1 - X, MODE = decltype(MC)::value;
        constexpr bool CH = MODE != TICK_ELEM, EL = MODE != TICK_CHAIN;
        const __amdgpu_buffer_rsrc_t rgY = rs_gates(ttY), rgN = rs_gates(ttN), rcpN = rs_cprev(ttN), rdhN = rs_dh(ttN);
        auto RB = [&](int ks, int pl) __attribute__((always_inline)) {
            if (CH) bfr[ks & 1][pl] = *reinterpret_cast<const bf16x8*>(drd[X][ks & 3] + pl * WB_PLANE + 256 * (ks >> 2));
        };
        auto RL = [&](int ub, int ks) __attribute__((always_inline)) {
            if (CH) lfr[ub] = *reinterpret_cast<const bf16x8*>(wlrd + 1024 * (8 * ub + ks - 8));
        };
        auto MB = [&](int ub, int ks, int p, int first) __attribute__((always_inline)) {
            if (!CH || (RLT_W6_ABL & 4)) return;
            const int bp = p == 2 ? 2 : (p == 0 || p == 4) ? 1 : 0;              // dA plane: m, h, l, h, m, h
            const bf16x8 bv = bfr[ks & 1][bp];
            f32x4& d = acc[X][ub];
            if (p == 1) {                                                        // W plane l: registers (k-steps 0-7) or the LDS fragment
                const bf16x8 lv = ks < 8 ? wlv[ub][ks < 8 ? ks : 0] : lfr[ub];
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(lv), "v"(bv));
            } else {
                const bf16x8 av = (p == 0 || p == 3) ? wm[ub][ks] : wh[ub][ks];
                // (the first product of a chain reads a zero C operand but keeps the accumulator TIED: as a plain output hipcc gives it
                //  fresh registers every tick and moves pool slots - still being loaded - out of their way, waiting with vmcnt(0))
                if (first) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "+v"(d) : "a"(av), "v"(bv));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(av), "v"(bv));
            }
        };
        // ---- element-wise side: i, f, g, o of block u in in[4u .. 4u+3], c_{t-1} in in[8+u], dL/dh in in[10+u] ----
        auto T1 = [&](int u) __attribute__((always_inline)) { if (EL) tcv = ct[Y][u] * (-2.f * LOG2E); };
        auto TX = [&](int u, int r) __attribute__((always_inline)) { if (EL) tcv[r] = rlt_exp2(tcv[r]); };
        auto T3 = [&](int u) __attribute__((always_inline)) { if (EL) tcv += 1.f; };
        auto TR = [&](int u, int r) __attribute__((always_inline)) { if (EL) tcv[r] = __builtin_amdgcn_rcpf(tcv[r]); };
        auto T5 = [&](int u) __attribute__((always_inline)) { if (EL) tcv = 2.f * tcv - 1.f; };                         // tanh(c_t)
        auto B1 = [&](int u) __attribute__((always_inline)) { if (EL) dhv = in[10 + u] + acc[Y][u]; };
        auto LDN = [&](int k) __attribute__((always_inline)) { if (EL && !(RLT_W6_ABL & 2)) load_slot(k, SINGLE ? Y : X, rgN, rcpN, rdhN); };
        auto B2 = [&](int u) __attribute__((always_inline)) { if (EL) t0 = dhv * in[4 * u + 3]; };
        auto B3 = [&](int u) __attribute__((always_inline)) { if (EL) { t1 = tcv * tcv; t1 = 1.f - t1; } };
        auto B5 = [&](int u) __attribute__((always_inline)) { if (EL) { t0 = t0 * t1; dcu = t0 + dc[Y][u]; } };
        auto B6 = [&](int u) __attribute__((always_inline)) { if (EL) t0 = dhv * tcv; };
        auto B7 = [&](int u) __attribute__((always_inline)) { if (EL) { t1 = 1.f - in[4 * u + 3]; t1 = t1 * in[4 * u + 3]; } };
        auto B8 = [&](int u) __attribute__((always_inline)) { if (EL) in[4 * u + 3] = t0 * t1; };                      // dA_o
        auto B9 = [&](int u) __attribute__((always_inline)) { if (EL) t2 = dcu * in[4 * u]; };
        auto B10 = [&](int u) __attribute__((always_inline)) { if (EL) { t1 = 1.f - in[4 * u]; t1 = t1 * in[4 * u]; } };
        auto B11 = [&](int u) __attribute__((always_inline)) { if (EL) { t0 = dcu * in[4 * u + 2]; in[4 * u] = t0 * t1; } };   // dA_i
        auto B12 = [&](int u) __attribute__((always_inline)) { if (EL) { t1 = in[4 * u + 2] * in[4 * u + 2]; t1 = 1.f - t1; } };
        auto B13 = [&](int u) __attribute__((always_inline)) { if (EL) in[4 * u + 2] = t2 * t1; };                     // dA_g
        auto B14 = [&](int u) __attribute__((always_inline)) { if (EL) t0 = dcu * in[8 + u]; };
        auto B15 = [&](int u) __attribute__((always_inline)) {          // c_{t-1} is the next step's c_t: four real moves HERE (as a plain copy
            if (!EL) return;                                            // hipcc renames the slot instead and shuffles at the loop head, behind vmcnt(0))
#pragma unroll
            for (int r = 0; r < 4; ++r) asm volatile("v_mov_b32 %0, %1" : "=v"(ct[Y][u][r]) : "v"(in[8 + u][r]));
        };
        auto B16 = [&](int u) __attribute__((always_inline)) { if (EL) { t1 = 1.f - in[4 * u + 1]; t1 = t1 * in[4 * u + 1]; } };
        auto B17 = [&](int u) __attribute__((always_inline)) { if (EL) dc[Y][u] = dcu * in[4 * u + 1]; };
        auto B18 = [&](int u) __attribute__((always_inline)) { if (EL) in[4 * u + 1] = t0 * t1; };                     // dA_f
        auto ST = [&](int u, int g) __attribute__((always_inline)) {
            if (EL && !(RLT_W6_ABL & 1)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, in[4 * u + g]), rgY, voff_g[Y] + 512 * g + 64 * u, 0, 0);
        };
        // split of the chunk of units r = 2 ch, 2 ch + 1 (slots i f g o | i f g o), residuals in place
        auto SP = [&](int u, int ch, int part) __attribute__((always_inline)) {
            if (!EL) return;
            f32x4 &vi = in[4 * u], &vf = in[4 * u + 1], &vg = in[4 * u + 2], &vo = in[4 * u + 3];
            const int r0 = 2 * ch, r1 = 2 * ch + 1;
            auto hi4 = [&](uint32_t (&o4)[4]) __attribute__((always_inline)) {
                o4[0] = pk2w(vi[r0], vf[r0]); o4[1] = pk2w(vg[r0], vo[r0]); o4[2] = pk2w(vi[r1], vf[r1]); o4[3] = pk2w(vg[r1], vo[r1]);
                asm("" : "+v"(o4[0]), "+v"(o4[1]), "+v"(o4[2]), "+v"(o4[3]));
            };
            auto res = [&](const uint32_t (&o4)[4], int r, int k) __attribute__((always_inline)) {
                vi[r] -= bf_lo(o4[k]); vf[r] -= bf_hi(o4[k]); vg[r] -= bf_lo(o4[k + 1]); vo[r] -= bf_hi(o4[k + 1]);
            };
            if (part == 0) hi4(h4);
            if (part == 1) res(h4, r0, 0);
            if (part == 2) res(h4, r1, 2);
            if (part == 3) hi4(m4);
            if (part == 4) res(m4, r0, 0);
            if (part == 5) res(m4, r1, 2);
            if (part == 6) hi4(l4);
        };
        auto LW = [&](int u, int ch, int pl) __attribute__((always_inline)) {
            if (!EL) return;
            const uint32_t (&o4)[4] = pl == 0 ? h4 : pl == 1 ? m4 : l4;
            *reinterpret_cast<uint4*>(dax + ((dwr[Y][u] ^ (ch ? 16u : 0u)) + pl * WB_PLANE)) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
        };
        if (CH) { RB(0, 0); RB(0, 1); RB(0, 2); }
        GAP_END;
#include "lstm6w_bwd_body.inc"
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int i = 0; i < w * RLT_W6_STAGGER; ++i) asm volatile("s_nop 15");
    };
    const std::integral_constant<int, 0> H0;
    const std::integral_constant<int, 1> H1;
    const std::integral_constant<int, TICK_FULL> MFULL;
    const std::integral_constant<int, TICK_ELEM> MELEM;

    const std::integral_constant<int, TICK_CHAIN> MCHAIN;
    if constexpr (SINGLE) {
        for (int tt = 0; tt < S; ++tt) {
#pragma unroll
            for (int ub = 0; ub < 2; ++ub)
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4)
                    asm volatile("" : "+a"(wh[ub][4 * k4]), "+a"(wh[ub][4 * k4 + 1]), "+a"(wh[ub][4 * k4 + 2]), "+a"(wh[ub][4 * k4 + 3]),
                                 "+a"(wm[ub][4 * k4]), "+a"(wm[ub][4 * k4 + 1]), "+a"(wm[ub][4 * k4 + 2]), "+a"(wm[ub][4 * k4 + 3]));
            tick(H1, MELEM, tt, tt + 1 < S ? tt + 1 : tt);      // element-wise step of half 0 (Y = 0), slots reloaded for its next step
            tick(H0, MCHAIN, tt, tt);                           // chain of half 0 (the one behind the last step is not needed: one tick of S)
        }
        return;
    }
    tick(H1, MELEM, 0, 0);                                  // element-wise step 0 of half 0; its slots reloaded for half 1's step 0
    if (S > 1) {                                            // (first tick pair outside the loop, as in the forward kernel)
        tick(H0, MFULL, 0, 1);                              // chain of half 0 | step 0 of half 1, slots -> half 0's step 1
        tick(H1, MFULL, 1, 1);                              // chain of half 1 | step 1 of half 0, slots -> half 1's step 1
    }
    for (int tt = 1; tt + 1 < S; ++tt) {
#pragma unroll
        for (int ub = 0; ub < 2; ++ub)
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
                asm volatile("" : "+a"(wh[ub][4 * k4]), "+a"(wh[ub][4 * k4 + 1]), "+a"(wh[ub][4 * k4 + 2]), "+a"(wh[ub][4 * k4 + 3]),
                             "+a"(wm[ub][4 * k4]), "+a"(wm[ub][4 * k4 + 1]), "+a"(wm[ub][4 * k4 + 2]), "+a"(wm[ub][4 * k4 + 3]));
        tick(H0, MFULL, tt, tt + 1);
        tick(H1, MFULL, tt + 1, tt + 1);
    }
    tick(H0, MELEM, S - 1, S - 1);                          // the last step of half 1 (reloads: the same rows once more, never used)
#undef GAP_END
}

}  // namespace

// one 16-list half per workgroup while that still gives every workgroup its own CU (256 CUs, two directions)
static bool w6_single(int B) {
    static const int on = [] { const char* e = getenv("RLT_LSTM6W_SINGLE"); return e ? atoi(e) : 1; }();      // 0: always two halves (A/B runs)
    return on && B <= 16 * 128;
}

int rlt_lstm6w_fwd(float* gates, const float* w_hh_fwd, const float* w_hh_rev, int S, int B, float* h_out, float* c_out,
                   const RltXIn& xi, void* stream) {
    const bool single = w6_single(B);
    const dim3 grid(rlt_cdiv(B, single ? 16 : 32), 2), block(256);
    hipStream_t st = rlt_stream(stream);
    auto go = [&](auto kern) {
        const int rc = rlt_allow_lds(kern, W6_LDS);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, grid, block, W6_LDS, st, gates, w_hh_fwd, w_hh_rev, S, B, h_out, c_out, xi);
        return 0;
    };
    if (xi.x) return single ? go(bilstm6w_fwd_kernel<true, true>) : go(bilstm6w_fwd_kernel<true, false>);
    return single ? go(bilstm6w_fwd_kernel<false, true>) : go(bilstm6w_fwd_kernel<false, false>);
}

int rlt_lstm6w_bwd(float* gates, const float* c, const float* w_hh_fwd, const float* w_hh_rev, const float* d_hout, int S, int B,
                   void* stream) {
    const bool single = w6_single(B);
    auto go = [&](auto kern) {
        const int rc = rlt_allow_lds(kern, WB_LDS);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(rlt_cdiv(B, single ? 16 : 32), 2), dim3(256), WB_LDS, rlt_stream(stream), gates, c, w_hh_fwd, w_hh_rev,
                           d_hout, S, B);
        return 0;
    };
    return single ? go(bilstm6w_bwd_kernel<true>) : go(bilstm6w_bwd_kernel<false>);
}

// Shared device/host helpers for librlt_hip.so (gfx950 only: 64-lane wavefronts, f32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/rlt_hip.h"

#define RLT_WAVE 64

#define RLT_CHECK_ARG(cond)   do { if (!(cond)) return RLT_E_ARG; } while (0)
#define RLT_CHECK_SHAPE(cond) do { if (!(cond)) return RLT_E_SHAPE; } while (0)
#define RLT_LAUNCH_RESULT()   ((int)hipGetLastError())

static inline hipStream_t rlt_stream(void* s) { return (hipStream_t)s; }
static inline bool rlt_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int rlt_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
__device__ __forceinline__ int rlt_cdiv_dev(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ bool rlt_aligned16_dev(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// kernels that use more than 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU) must opt in
template <typename K>
static inline int rlt_allow_lds(K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return 0;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// Precision mode of the MFMA contractions seen by the code of one entry-point call (api.hip).  Every entry point whose
// arithmetic or buffer layout depends on the mode takes an `int precision` argument and opens a scope with it first thing:
// RLT_PRECISION_FP32 / _BF16X3 / _BF16X6 select the mode for THAT call (and for the library calls it makes itself: they pass
// RLT_PRECISION_DEFAULT and inherit the scope); RLT_PRECISION_DEFAULT outside any scope reads the process default
// (rlt_set_precision, env RLT_PRECISION).  The scope is a thread-local value: calls on different threads - or one after the
// other on one thread - with different modes do not see each other, and nothing but rlt_set_precision writes shared state.
int rlt_precision();
struct RltPrecScope {
    int saved; bool set;
    explicit RltPrecScope(int p);
    ~RltPrecScope();
    RltPrecScope(const RltPrecScope&) = delete;
    RltPrecScope& operator=(const RltPrecScope&) = delete;
};
static inline bool rlt_precision_arg_ok(int p) {
    return p == RLT_PRECISION_DEFAULT || p == RLT_PRECISION_FP32 || p == RLT_PRECISION_BF16X3 || p == RLT_PRECISION_BF16X6;
}
// first statement of an entry point with a `precision` argument: a code outside the four above is RLT_E_ARG (workspace
// queries: 0 bytes)
#define RLT_PREC_SCOPE(p)    if (!rlt_precision_arg_ok(p)) return RLT_E_ARG; RltPrecScope rlt_prec_scope_(p)
#define RLT_PREC_SCOPE_SZ(p) if (!rlt_precision_arg_ok(p)) return 0;         RltPrecScope rlt_prec_scope_(p)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- wavefront (64 lanes) reductions / scans ------------------------------------------------
// raw v_exp_f32 (2^x, 1 ulp): unlike exp2f() no denormal-range rescue sequence (6 extra VALU ops); results
// below 2^-126 flush to zero, which every caller (softmax weights, sigmoid) tolerates
__device__ __forceinline__ float rlt_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// All of them must be called by a whole wavefront (wave-uniform control flow, 64 active lanes).
// The cross-lane steps are DPP operand modifiers of the VALU (row_shr 1/2/4/8 inside the rows of 16 lanes, then
// row_bcast:15 / row_bcast:31 across the rows): 6 dependent vector instructions per scan.  The `__shfl` forms they
// replace went through the LDS crossbar (`ds_bpermute_b32` + `s_waitcnt lgkmcnt` + select per step: six dependent LDS
// round trips per reduction, which bound the wave-per-list kernels by latency).
template <int CTRL, int ROW_MASK, typename T>
__device__ __forceinline__ T rlt_dpp(T old, T v) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- or 64-bit values");
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xf, false));
    } else {
        struct P { int lo, hi; };
        const P o = __builtin_bit_cast(P, old), s = __builtin_bit_cast(P, v);
        P r;
        r.lo = __builtin_amdgcn_update_dpp(o.lo, s.lo, CTRL, ROW_MASK, 0xf, false);
        r.hi = __builtin_amdgcn_update_dpp(o.hi, s.hi, CTRL, ROW_MASK, 0xf, false);
        return __builtin_bit_cast(T, r);
    }
}
template <typename T>
__device__ __forceinline__ T rlt_readlane(T v, int lane) {
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
    } else {
        struct P { int lo, hi; };
        P s = __builtin_bit_cast(P, v);
        s.lo = __builtin_amdgcn_readlane(s.lo, lane);
        s.hi = __builtin_amdgcn_readlane(s.hi, lane);
        return __builtin_bit_cast(T, s);
    }
}
// inclusive scan over the 64 lanes with the associative `op` whose identity is `id` (lanes without a source keep `id`)
template <typename T, typename Op>
__device__ __forceinline__ T wave_scan_op(T v, T id, Op op) {
    v = op(v, rlt_dpp<0x111, 0xf>(id, v));      // row_shr:1
    v = op(v, rlt_dpp<0x112, 0xf>(id, v));      // row_shr:2
    v = op(v, rlt_dpp<0x114, 0xf>(id, v));      // row_shr:4
    v = op(v, rlt_dpp<0x118, 0xf>(id, v));      // row_shr:8   -> inclusive inside each row of 16
    v = op(v, rlt_dpp<0x142, 0xa>(id, v));      // row_bcast:15 into rows 1 and 3
    v = op(v, rlt_dpp<0x143, 0xc>(id, v));      // row_bcast:31 into rows 2 and 3
    return v;
}
// sum over the 64 lanes, the same value (a scalar register) in every lane; fixed order
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
    return rlt_readlane(wave_scan_op(v, T(0), [](T a, T b) { return a + b; }), 63);
}
__device__ __forceinline__ float wave_max(float v) {
    return rlt_readlane(wave_scan_op(v, -INFINITY, [](float a, float b) { return fmaxf(a, b); }), 63);
}
// inclusive prefix sum across the 64 lanes
template <typename T>
__device__ __forceinline__ T wave_scan_incl(T v, int /*lane*/) {
    return wave_scan_op(v, T(0), [](T a, T b) { return a + b; });
}

// ---- f32 MFMA (exact fp32 products; 64 FLOP/clk/SIMD on gfx950) -------------------------------
// 32x32x2: lane l supplies A[i = l&31][k = l>>5], B[k = l>>5][j = l&31];
//          D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31] in register r of 16.
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// row of the 32x32 accumulator tile held in register r by a lane of half hh = lane>>5
__device__ __forceinline__ constexpr int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// ---- counter-based dropout RNG -----------------------------------------------------------------
// keep(seed, row, col) is a pure function of its arguments, so backward recomputes the forward mask instead of
// storing it.  Every dropout site draws one number per element of a 2-D array (rows x cols; for the attention
// probabilities: query x key per (position, head), seeded per pair).  A full avalanche hash per element costs more
// VALU time than the softmax it sits in, so every row and every column gets one strong 32-bit hash ("lowbias32"
// mixer, two rounds) - computed once per lane / once per tile / on the scalar unit - and the element's number is
// their product mod 2^32 (column hash forced odd, so a row's numbers stay distinct per column hash): one integer
// multiply + one compare per element.  Measured on 2048 x 2048 masks (p = 0.1 .. 0.5): mean, row-pair, column-pair,
// 2x2-rectangle and neighbour statistics indistinguishable from independent draws.
__host__ __device__ __forceinline__ uint32_t rlt_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t rlt_row_hash(uint32_t seed, uint32_t row) {
    return rlt_mix32(seed + rlt_mix32(row + 0x9E3779B9U));
}
__host__ __device__ __forceinline__ uint32_t rlt_col_hash(uint32_t seed, uint32_t col) {
    return rlt_mix32((seed ^ 0x85EBCA6BU) + rlt_mix32(col + 0x7F4A7C15U)) | 1u;
}
__host__ __device__ __forceinline__ bool rlt_keep_rc(uint32_t row_hash, uint32_t col_hash, uint32_t thr) {
    return row_hash * col_hash >= thr;       // (a 24-bit multiply - v_mul_u32_u24 - in its place measured no different: profiles/r06_notes.md)
}
// threshold for "drop": drop iff number < thr, thr = p * 2^32
__host__ __device__ __forceinline__ uint32_t rlt_drop_threshold(float p) {
    const double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}
// un-hoisted form (mask export, small kernels)
__host__ __device__ __forceinline__ bool rlt_keep(uint32_t seed, uint32_t row, uint32_t col, uint32_t thr) {
    return rlt_keep_rc(rlt_row_hash(seed, row), rlt_col_hash(seed, col), thr);
}

// ---- deterministic column sums of a tall partial matrix -------------------------------------------------
// out[col] (+)= sum_{r < R} partial[r * ld + col] for col < ncol; columns >= split go to out1[col - split].
// One 256-thread workgroup per 16 columns: 16 row lanes x 16 columns, four loads in flight per lane, then a
// fixed-order LDS reduction over the row lanes (bitwise reproducible).  Launch with grid = cdiv(ncol, 16).
static __global__ __launch_bounds__(256) void rlt_rows_reduce_kernel(const float* __restrict__ partial, int R, int ld, int ncol,
                                                                     int split, float* __restrict__ out0,
                                                                     float* __restrict__ out1, int accumulate) {
    __shared__ float red[16][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (col < ncol) {
        const float* p = partial + col;
        int r = ry;
        for (; r + 48 < R; r += 64) {
            a0 += p[(size_t)r * ld];
            a1 += p[(size_t)(r + 16) * ld];
            a2 += p[(size_t)(r + 32) * ld];
            a3 += p[(size_t)(r + 48) * ld];
        }
        for (; r < R; r += 16) a0 += p[(size_t)r * ld];
    }
    red[ry][cx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ry == 0 && col < ncol) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += red[i][cx];
        float* dst = col < split ? out0 + col : out1 + (col - split);
        *dst = accumulate ? *dst + acc : acc;
    }
}

// Shared between the exact-fp32 (attention.hip) and split-bf16 (attention3.hip) list-axis attention kernels.
#pragma once
#include "common.h"

struct AttnArgs {
    const float* qkv; const float* out; const float* dout; const float* lse; const float* delta;
    const void* img; const void* dimg;        // bf16x6 mode: pre-split three-plane tile images of Q / K / V and of dO (attention6.hip)
    float* o; float* lse_o; float* dqkv;
    int S, B, H;
    float scale;
    float drop_p; uint32_t drop_thr, seed;    // dropout on the attention probabilities (train mode)
    uint32_t* redo;                           // pipelined forward kernels (attention6n.hip, attention6h.hip): one flag word per workgroup, or null
};

// One LDS-DMA piece as inline assembly: M0 carries the LDS destination.  hipcc treats M0 as a reserved register (a
// clobber on it is rejected with a warning and ignored), so the statement saves and restores it: whatever the compiler
// keeps in M0 around the asm (its own global_load_lds builtins in a mixed instantiation, movrel / readlane lowerings)
// survives.  Two scalar moves per piece, ~5 pieces per wavefront and tile.
#define RLT_DMA_ASM(dst, src)                                                                                         \
    do {                                                                                                              \
        uint32_t m0_keep_;                                                                                            \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(m0_keep_) : "s"(dst), "v"(src) : "memory");                                              \
    } while (0)

namespace {

constexpr int KT = 64;      // rows per LDS tile
constexpr int QT = 128;     // rows owned by a workgroup (4 wavefronts x 32)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;


// per-(position, head) stream of the dropout RNG; keep(pair_seed, query, key)
__device__ __forceinline__ uint32_t pair_seed(uint32_t seed, int pair) { return rlt_mix32(seed ^ ((uint32_t)pair * 0x9E3779B9U)); }

// flat block id -> (position*head pair, row tile); all row tiles of a pair go to one XCD (they
// share that pair's K/V in the XCD's L2) when the pair count allows.
__device__ __forceinline__ void map_block(int bid, int npair, int ntile, int& pair, int& tile) {
    if ((npair & 7) == 0) {
        const int xcd = bid & 7, j = bid >> 3;
        pair = (j / ntile) * 8 + xcd;
        tile = j % ntile;
    } else {
        pair = bid / ntile;
        tile = bid % ntile;
    }
}

// store D^T accumulators (row = d, col = lane's row index) to global rows: dst + row*ld + d
template <int HD>
__device__ __forceinline__ void store_acc_T(float* __restrict__ dst_row, int hh, const f32x16 (&acc)[(HD + 31) / 32],
                                            float mul) {
    constexpr int DT = (HD + 31) / 32;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = dt * 32 + 8 * g + 4 * hh;
            if (d0 < HD) {
                float4 v;
                v.x = acc[dt][4 * g + 0] * mul; v.y = acc[dt][4 * g + 1] * mul;
                v.z = acc[dt][4 * g + 2] * mul; v.w = acc[dt][4 * g + 3] * mul;
                *reinterpret_cast<float4*>(dst_row + d0) = v;
            }
        }
}


}  // namespace

// fp32-faithful six-product path (bf16x6 mode, head dims 16 / 32 / 64), defined in attention6.hip: which = 0 forward,
// 1 dK/dV, 2 dQ
int rlt_attn6_run(int which, const AttnArgs& a, int HD, hipStream_t st);
// ... with pre-split tile images (which = 3: write the Q / K / V images from a.qkv into a.img; 4: the dO images from a.dout into
// a.dimg); nmat matrices of S*H pairs x ceil(B / 64) tiles
size_t rlt_attn6_images_bytes(int S, int B, int H, int HD, int nmat);
// ... at head dim 16 on the 16x16x32 MFMA (no padded head-dim axis, plane pairs in the d contraction, the split of P / dS on the
// matrix pipe), defined in attention6n.hip: which = 0 forward, 1 dK/dV, 2 dQ
int rlt_attn6n_run(int which, const AttnArgs& a, hipStream_t st);
// ... its pipelined forward kernel: K / V images in a two-block buffer (prepare_at: matrix `what` into block `slot` of a.img)
size_t rlt_attn6n_fwd_images_bytes(int S, int B, int H);
int rlt_attn6n_prepare_at(int what, int slot, const AttnArgs& a, hipStream_t st);
// ... its pipelined backward kernels (512 lists and more, no dropout) stage pre-split tile images + row seeds from a.img:
// rlt_attn6n_images_bytes bytes, written by rlt_attn6n_prepare (what = 0 Q, 1 K, 2 V, 3 dO, 4 seeds)
size_t rlt_attn6n_images_bytes(int S, int B, int H);
int rlt_attn6n_prepare(int what, const AttnArgs& a, hipStream_t st);
// ... at head dim 64 in the same form (attention6h.hip): the pipelined forward stages pre-split K / V tile images (blocks 0 / 1 of
// a.img, written by rlt_attn6h_prepare2) and leaves a flag word per workgroup (a.redo) for the fix-up launch of attention6.hip
size_t rlt_attn6h_fwd_images_bytes(int S, int B, int H);
int rlt_attn6h_prepare2(int what0, int slot0, int what1, int slot1, const AttnArgs& a, hipStream_t st);
int rlt_attn6h_run(int which, const AttnArgs& a, hipStream_t st);
// exact fp32 at head dim 16 on the 16x16x4 MFMA (no padded head-dim axis), defined in attention16.hip: same `which`
int rlt_attn16_run(int which, const AttnArgs& a, hipStream_t st);
// split-bf16 ("bf16x3") path, defined in attention3.hip
size_t rlt_attn3_images_bytes(int S, int B, int H, int HD, int nmat);
int rlt_attn3_run(int which, const AttnArgs& a, int HD, void* images, void* dimages, hipStream_t st);

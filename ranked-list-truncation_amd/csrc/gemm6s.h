// csrc/gemm6s.hip: the weights-stationary K = 256 products of the bf16x6 mode, called from gemm.hip's dispatch.
#pragma once
#include <stddef.h>
#include <stdint.h>

struct Gemm6sArgs {
    const float* A; const float* B; float* C;      // C[M x N] = A[M x 256] op(B) (+ bias + bias2)
    const float* bias; const float* bias2;         // (N) or null
    int M, N, K, lda, ldb, ldc;
    // the FFN mask pair of rlt_gemm_bits (include/rlt_hip.h): bit (row & 31) of word [(row >> 5) * N + col]
    uint32_t* bits_out;                            // with relu: bit = (C > 0)
    const uint32_t* bits_in;                       // C = bit ? C * mask_scale : 0
    float mask_scale;
};
// shape / alignment conditions: K == 256 with N % 256 == 0, or K == 128 with N % 256 == 0 (N % 128 == 0 without the mask epilogues); M >= 8192.
// Mask epilogue (bits_out): the rows past M of the last 32-row block are evaluated as relu(bias) > 0 and their bits land in the final
// mask word - no consumer reads them (rlt_gemm_bits indexes rows < M), but the word differs from the tiled kernels' (which write 0 there).
bool rlt_gemm6s_ok(const Gemm6sArgs& g);
int rlt_gemm6s_launch(const Gemm6sArgs& g, bool tb, bool relu, void* stream);      // 0 or a hip error code (bits_out needs relu; not both bit pointers)

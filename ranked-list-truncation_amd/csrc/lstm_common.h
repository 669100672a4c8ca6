// Shared between lstm.hip and lstm6w.hip: the fused input projection of narrow inputs (I <= 3, the reference's score / tf-idf /
// doc2vec features, models/AttnCut.py:8,17) and the internal launchers of the whole-weights recurrences.
#pragma once
#include <stddef.h>

struct RltXIn {
    const float* x;            // (S*B, I) position-major, or null: `gates` holds the pre-activations
    const float* w_ih[2];      // (512, I) per direction
    const float* b_ih[2];
    const float* b_hh[2];
    int I;
};

// lstm6w.hip: bf16x6 recurrences with one wavefront per SIMD; 0 or a hip error code
int rlt_lstm6w_fwd(float* gates, const float* w_hh_fwd, const float* w_hh_rev, int S, int B, float* h_out, float* c_out,
                   const RltXIn& xi, void* stream);
int rlt_lstm6w_bwd(float* gates, const float* c, const float* w_hh_fwd, const float* w_hh_rev, const float* d_hout, int S, int B,
                   void* stream);

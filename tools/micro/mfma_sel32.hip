// Diagnostic for the 32x32x16 form of the matrix-pipe residual (tools/micro/mfma_resid.hip: check32): WHICH entry of the packed B operand
// does the selection matrix subtract from each accumulator register?  X[row][col] = 4 * row + col / 8 + 1 (exact in bf16), so the residual
// X - SEL h(X) must be zero everywhere, and a non-zero register names the entry that was subtracted instead.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_sel32.hip -o /tmp/mfma_sel32
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t pk2(float a, float b) {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    const v2 t = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

__global__ void sel32(float* out, int variant) {
    const int lane = threadIdx.x & 63, col = lane & 31, hh = lane >> 5;
    f32x16 c;
    float v[16];
    for (int r = 0; r < 16; ++r) { v[r] = variant >= 3 ? (float)(col + 1) : (float)(4 * acc_row(r, hh)) + 1.0f; c[r] = v[r]; }
    if (variant >= 3) variant -= 3;
    uint32_t as_[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    {
        const int i = lane & 31, hp_ = lane >> 5;
        if (hp_ == ((i >> 2) & 1)) {
            const int s = i >> 4, j = 4 * ((i >> 3) & 1) + (i & 3);
            as_[s][j >> 1] = (j & 1) ? 0xBF800000u : 0x0000BF80u;
        }
    }
    const bf16x8 A0 = __builtin_bit_cast(bf16x8, make_uint4(as_[0][0], as_[0][1], as_[0][2], as_[0][3]));
    const bf16x8 A1 = __builtin_bit_cast(bf16x8, make_uint4(as_[1][0], as_[1][1], as_[1][2], as_[1][3]));
    uint32_t hp[8];
    for (int i = 0; i < 8; ++i) hp[i] = pk2(v[2 * i], v[2 * i + 1]);
    const bf16x8 B0 = __builtin_bit_cast(bf16x8, make_uint4(hp[0], hp[1], hp[2], hp[3]));
    const bf16x8 B1 = __builtin_bit_cast(bf16x8, make_uint4(hp[4], hp[5], hp[6], hp[7]));
    f32x16 d;
    if (variant == 0) {                 // as check32: chained, in place
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, c, 0, 0, 0);
        d = c;
    } else if (variant == 1) {          // only the first selection (rows 0..15 expected to vanish)
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, c, 0, 0, 0);
    } else {                            // only the second
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, c, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) out[((size_t)blockIdx.x * 64 + lane) * 16 + r] = d[r];
}

// random data: both residual levels of a 32 x 32 tile on the matrix pipe against the vector form, bit for bit
__device__ __forceinline__ float lo_f(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
__global__ void rand32(const float* __restrict__ x, unsigned long long* bad) {
    const int lane = threadIdx.x & 63;
    f32x16 c;
    float v[16], ref1[16], ref2[16];
    for (int r = 0; r < 16; ++r) { v[r] = x[((size_t)blockIdx.x * 64 + lane) * 16 + r]; c[r] = v[r]; }
    uint32_t as_[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    {
        const int i = lane & 31, hp_ = lane >> 5;
        if (hp_ == ((i >> 2) & 1)) {
            const int s = i >> 4, j = 4 * ((i >> 3) & 1) + (i & 3);
            as_[s][j >> 1] = (j & 1) ? 0xBF800000u : 0x0000BF80u;
        }
    }
    const bf16x8 A0 = __builtin_bit_cast(bf16x8, make_uint4(as_[0][0], as_[0][1], as_[0][2], as_[0][3]));
    const bf16x8 A1 = __builtin_bit_cast(bf16x8, make_uint4(as_[1][0], as_[1][1], as_[1][2], as_[1][3]));
    uint32_t hp[8], mp[8];
    for (int i = 0; i < 8; ++i) {
        hp[i] = pk2(v[2 * i], v[2 * i + 1]);
        ref1[2 * i] = v[2 * i] - lo_f(hp[i]); ref1[2 * i + 1] = v[2 * i + 1] - hi_f(hp[i]);
        const uint32_t m = pk2(ref1[2 * i], ref1[2 * i + 1]);
        ref2[2 * i] = ref1[2 * i] - lo_f(m); ref2[2 * i + 1] = ref1[2 * i + 1] - hi_f(m);
    }
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, __builtin_bit_cast(bf16x8, make_uint4(hp[0], hp[1], hp[2], hp[3])), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, __builtin_bit_cast(bf16x8, make_uint4(hp[4], hp[5], hp[6], hp[7])), c, 0, 0, 0);
    unsigned long long n1 = 0, n2 = 0;
    for (int r = 0; r < 16; ++r) { const float cr = c[r]; n1 += __builtin_bit_cast(uint32_t, cr) != __builtin_bit_cast(uint32_t, ref1[r]); }   // (bit_cast of a vector ELEMENT reads element 0: copy first)
    for (int i = 0; i < 8; ++i) mp[i] = pk2(c[2 * i], c[2 * i + 1]);
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, __builtin_bit_cast(bf16x8, make_uint4(mp[0], mp[1], mp[2], mp[3])), c, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, __builtin_bit_cast(bf16x8, make_uint4(mp[4], mp[5], mp[6], mp[7])), d, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { const float dr = d[r]; n2 += __builtin_bit_cast(uint32_t, dr) != __builtin_bit_cast(uint32_t, ref2[r]); }
    if (n1) atomicAdd(bad, n1);
    if (n2) atomicAdd(bad + 1, n2);
}

int main() {
    {
        const int NB = 2048;
        const size_t n = (size_t)NB * 64 * 16;
        float* hx = (float*)malloc(n * 4);
        uint32_t st = 12345u;
        for (size_t i = 0; i < n; ++i) {            // every exponent between 2^-40 and 2^40, both signs, random significands
            st = st * 1664525u + 1013904223u;
            const uint32_t e = 87u + (st >> 9) % 81u;
            st = st * 1664525u + 1013904223u;
            const uint32_t bits = (st & 0x80000000u) | (e << 23) | (st >> 9);
            memcpy(&hx[i], &bits, 4);
        }
        float* dx; unsigned long long* dbad; unsigned long long hb[2];
        hipMalloc(&dx, n * 4); hipMalloc(&dbad, 16);
        hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice); hipMemset(dbad, 0, 16);
        hipLaunchKernelGGL(rand32, dim3(NB), dim3(64), 0, 0, dx, dbad);
        hipMemcpy(hb, dbad, 16, hipMemcpyDeviceToHost);
        printf("random data, %zu values: level-1 mismatches %llu, level-2 mismatches %llu (matrix-pipe residual vs vector residual, bitwise)\n", n, hb[0], hb[1]);
    }
    float* dout;
    hipMalloc(&dout, 64 * 16 * 4);
    static float h[64 * 16];
    for (int variant = 0; variant < 4; ++variant) {
        hipLaunchKernelGGL(sel32, dim3(1), dim3(64), 0, 0, dout, variant);
        hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
        printf("variant %d (residual per register r; x = %s):\n", variant, variant >= 3 ? "col + 1" : "4 row + 1");
        for (int lane : {0, 1, 5, 32, 37}) {
            printf("  lane %2d:", lane);
            for (int r = 0; r < 16; ++r) printf(" %6.1f", h[lane * 16 + r]);
            printf("\n");
        }
        int nz = 0;
        for (int i = 0; i < 64 * 16; ++i) nz += h[i] != 0.f;
        printf("  non-zero residuals: %d of 1024\n", nz);
    }
    return 0;
}

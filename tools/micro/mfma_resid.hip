// MFMA-assisted three-way bf16 split of an fp32 ACCUMULATOR tile (gfx950).
//
// The six-product kernels split every fresh operand (P, dS) x = h + m + l with 5.5 vector instructions per value
// (cvt_pk, unpack, subtract, ... - split6.h).  The residual x - float(h) can be formed by the matrix pipe instead: the
// packed h pairs ARE the B operand of the next product already, so
//     R1 = X - SEL * H        (one MFMA per tile: C = X, A = a constant selection matrix of -1.0 entries, B = the packed h)
// leaves the exact residual in the accumulator layout, and the split costs 1.5 vector instructions per value (three
// v_cvt_pk_bf16_f32 per two values) + two MFMAs per tile and level.
//   16x16x32: two 16x16 tiles (8 values per lane) form ONE B operand; tile kb is selected by A_kb.
//   32x32x16: registers 8s..8s+7 of a 32x32 tile form the B operand of k-step s; two chained MFMAs per tile.
// This program checks that the MFMA residual is BIT-IDENTICAL to the vector-ALU residual (x - float(bf16(x)), exact in fp32)
// at both levels, over random values of every exponent, probabilities exp2(-u), denormals, zeros, signed values and the
// `worst-split` pattern (low 16 significand bits 0x7F40).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_resid.hip -o /tmp/mfma_resid
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t pk2(float a, float b) {
    // (the cast form: hipcc emits v_cvt_pk_bf16_f32 AND the wait states an MFMA needs behind a vector write of its operand;
    // as an asm statement the conversion is invisible to the hazard recognizer and the MFMA reads stale registers)
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    const v2 t = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ float lo_f(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// 16x16x32: lane (c = l & 15, g = l >> 4) holds tile kb rows 4g..4g+3 of column c in x[4 kb + r]
__global__ void check16(const float* __restrict__ x, uint32_t* __restrict__ out, unsigned long long* bad) {
    const int lane = threadIdx.x & 63;
    const float* xp = x + ((size_t)blockIdx.x * 64 + lane) * 8;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = xp[i];
    // selection operands
    uint32_t a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
    if (((lane & 15) >> 2) == (lane >> 4)) {
        const int j0 = lane & 3, j1 = 4 + (lane & 3);
        a0[j0 >> 1] = (j0 & 1) ? 0xBF800000u : 0x0000BF80u;
        a1[j1 >> 1] = (j1 & 1) ? 0xBF800000u : 0x0000BF80u;
    }
    const bf16x8 A0 = __builtin_bit_cast(bf16x8, make_uint4(a0[0], a0[1], a0[2], a0[3]));
    const bf16x8 A1 = __builtin_bit_cast(bf16x8, make_uint4(a1[0], a1[1], a1[2], a1[3]));
    float ref1[8], ref2[8];
    uint32_t hp[4], mp[4], lp[4], mpr[4], lpr[4];
    for (int i = 0; i < 4; ++i) hp[i] = pk2(v[2 * i], v[2 * i + 1]);
    for (int i = 0; i < 4; ++i) { ref1[2 * i] = v[2 * i] - lo_f(hp[i]); ref1[2 * i + 1] = v[2 * i + 1] - hi_f(hp[i]); }
    for (int i = 0; i < 4; ++i) mpr[i] = pk2(ref1[2 * i], ref1[2 * i + 1]);
    for (int i = 0; i < 4; ++i) { ref2[2 * i] = ref1[2 * i] - lo_f(mpr[i]); ref2[2 * i + 1] = ref1[2 * i + 1] - hi_f(mpr[i]); }
    for (int i = 0; i < 4; ++i) lpr[i] = pk2(ref2[2 * i], ref2[2 * i + 1]);
    // the matrix-pipe form
    f32x4 c0 = {v[0], v[1], v[2], v[3]}, c1 = {v[4], v[5], v[6], v[7]};
    bf16x8 Bh = __builtin_bit_cast(bf16x8, make_uint4(hp[0], hp[1], hp[2], hp[3]));
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, Bh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, Bh, c1, 0, 0, 0);
    mp[0] = pk2(c0[0], c0[1]); mp[1] = pk2(c0[2], c0[3]); mp[2] = pk2(c1[0], c1[1]); mp[3] = pk2(c1[2], c1[3]);
    bf16x8 Bm = __builtin_bit_cast(bf16x8, make_uint4(mp[0], mp[1], mp[2], mp[3]));
    f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, Bm, c0, 0, 0, 0);
    f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, Bm, c1, 0, 0, 0);
    lp[0] = pk2(d0[0], d0[1]); lp[1] = pk2(d0[2], d0[3]); lp[2] = pk2(d1[0], d1[1]); lp[3] = pk2(d1[2], d1[3]);
    int nbad = 0;
    for (int i = 0; i < 4; ++i) {
        const float got1[2] = {c0[i], c1[i]}, got2[2] = {d0[i], d1[i]};
        for (int kb = 0; kb < 2; ++kb) {
            if (__builtin_bit_cast(uint32_t, got1[kb]) != __builtin_bit_cast(uint32_t, ref1[4 * kb + i])) {
                ++nbad;
                if (atomicAdd(bad + 1, 1ull) == 0) { out[8] = __builtin_bit_cast(uint32_t, v[4 * kb + i]); out[9] = __builtin_bit_cast(uint32_t, got1[kb]); out[10] = __builtin_bit_cast(uint32_t, ref1[4 * kb + i]); out[11] = lane * 16 + 4 * kb + i; }
            }
            if (__builtin_bit_cast(uint32_t, got2[kb]) != __builtin_bit_cast(uint32_t, ref2[4 * kb + i])) {
                ++nbad;
                if (atomicAdd(bad + 2, 1ull) == 0) { out[12] = __builtin_bit_cast(uint32_t, ref1[4 * kb + i]); out[13] = __builtin_bit_cast(uint32_t, got2[kb]); out[14] = __builtin_bit_cast(uint32_t, ref2[4 * kb + i]); out[15] = lane * 16 + 4 * kb + i; }
            }
        }
        nbad += mp[i] != mpr[i];
        nbad += lp[i] != lpr[i];
    }
    // exactness of the split itself: h + m + l == x in fp32 arithmetic (sum smallest first)
    for (int i = 0; i < 4; ++i) {
        const float s0 = (lo_f(lp[i]) + lo_f(mp[i])) + lo_f(hp[i]), s1 = (hi_f(lp[i]) + hi_f(mp[i])) + hi_f(hp[i]);
        nbad += !(s0 == v[2 * i] || (isnan(s0) && isnan(v[2 * i])));
        nbad += !(s1 == v[2 * i + 1] || (isnan(s1) && isnan(v[2 * i + 1])));
    }
    if (nbad) {
        if (atomicAdd(bad, (unsigned long long)nbad) == 0) {
            out[0] = __builtin_bit_cast(uint32_t, v[0]); out[1] = __builtin_bit_cast(uint32_t, c0[0]); out[2] = __builtin_bit_cast(uint32_t, ref1[0]);
            out[3] = __builtin_bit_cast(uint32_t, d0[0]); out[4] = __builtin_bit_cast(uint32_t, ref2[0]); out[5] = (uint32_t)lane;
        }
    }
}

// 32x32x16: lane (c = l & 31, h = l >> 5) holds rows (r & 3) + 8 (r >> 2) + 4 h of column c in x[r], r < 16
__global__ void check32(const float* __restrict__ x, uint32_t* __restrict__ out, unsigned long long* bad) {
    const int lane = threadIdx.x & 63;
    const float* xp = x + ((size_t)blockIdx.x * 64 + lane) * 16;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = xp[i];
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
    float ref1[16], ref2[16];
    uint32_t hp[8], mp[8], lp[8], mpr[8], lpr[8];
    for (int i = 0; i < 8; ++i) hp[i] = pk2(v[2 * i], v[2 * i + 1]);
    for (int i = 0; i < 8; ++i) { ref1[2 * i] = v[2 * i] - lo_f(hp[i]); ref1[2 * i + 1] = v[2 * i + 1] - hi_f(hp[i]); }
    for (int i = 0; i < 8; ++i) mpr[i] = pk2(ref1[2 * i], ref1[2 * i + 1]);
    for (int i = 0; i < 8; ++i) { ref2[2 * i] = ref1[2 * i] - lo_f(mpr[i]); ref2[2 * i + 1] = ref1[2 * i + 1] - hi_f(mpr[i]); }
    for (int i = 0; i < 8; ++i) lpr[i] = pk2(ref2[2 * i], ref2[2 * i + 1]);
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = v[i];
    bf16x8 B0 = __builtin_bit_cast(bf16x8, make_uint4(hp[0], hp[1], hp[2], hp[3]));
    bf16x8 B1 = __builtin_bit_cast(bf16x8, make_uint4(hp[4], hp[5], hp[6], hp[7]));
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, c, 0, 0, 0);
    for (int i = 0; i < 8; ++i) mp[i] = pk2(c[2 * i], c[2 * i + 1]);
    B0 = __builtin_bit_cast(bf16x8, make_uint4(mp[0], mp[1], mp[2], mp[3]));
    B1 = __builtin_bit_cast(bf16x8, make_uint4(mp[4], mp[5], mp[6], mp[7]));
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, c, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, d, 0, 0, 0);
    for (int i = 0; i < 8; ++i) lp[i] = pk2(d[2 * i], d[2 * i + 1]);
    int nbad = 0;
    // (round 6: `__builtin_bit_cast(uint32_t, c[i])` on an ELEMENT of an ext-vector compiles to a read of element 0 - hipcc 7.2 -, which is what
    // made this check report "register 0 identical, registers 1..15 not" for two rounds; the elements are copied to scalars first)
    float cs[16], ds_[16];
    for (int i = 0; i < 16; ++i) { cs[i] = c[i]; ds_[i] = d[i]; }
    for (int i = 0; i < 16; ++i) {
        if (__builtin_bit_cast(uint32_t, cs[i]) != __builtin_bit_cast(uint32_t, ref1[i])) {
            ++nbad;
            if (atomicAdd(bad + 1, 1ull) == 0) { out[8] = __builtin_bit_cast(uint32_t, v[i]); out[9] = __builtin_bit_cast(uint32_t, cs[i]); out[10] = __builtin_bit_cast(uint32_t, ref1[i]); out[11] = lane * 16 + i; }
            atomicOr(out + 6, 1u << i);
            if (lane < 32) atomicOr(out + 7, 1u << lane);
        }
        if (__builtin_bit_cast(uint32_t, ds_[i]) != __builtin_bit_cast(uint32_t, ref2[i])) {
            ++nbad;
            if (atomicAdd(bad + 2, 1ull) == 0) { out[12] = __builtin_bit_cast(uint32_t, ref1[i]); out[13] = __builtin_bit_cast(uint32_t, ds_[i]); out[14] = __builtin_bit_cast(uint32_t, ref2[i]); out[15] = lane * 16 + i; }
        }
    }
    for (int i = 0; i < 8; ++i) { nbad += mp[i] != mpr[i]; nbad += lp[i] != lpr[i]; }
    if (nbad) {
        if (atomicAdd(bad, (unsigned long long)nbad) == 0) {
            out[0] = __builtin_bit_cast(uint32_t, v[0]); out[1] = __builtin_bit_cast(uint32_t, cs[0]); out[2] = __builtin_bit_cast(uint32_t, ref1[0]);
            out[3] = __builtin_bit_cast(uint32_t, ds_[0]); out[4] = __builtin_bit_cast(uint32_t, ref2[0]); out[5] = (uint32_t)lane;
        }
    }
}

// throughput: the split of a 32-key x 16-query pair of tiles (8 values per lane), vector form against matrix form, beside
// NM "payload" MFMAs per iteration (independent accumulators), W wavefronts per SIMD (launch: 256 * W threads, one block per CU)
template <int MODE, int NM>
__global__ __launch_bounds__(1024) void rate16(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    uint32_t a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
    if (((lane & 15) >> 2) == (lane >> 4)) {
        const int j0 = lane & 3, j1 = 4 + (lane & 3);
        a0[j0 >> 1] = (j0 & 1) ? 0xBF800000u : 0x0000BF80u;
        a1[j1 >> 1] = (j1 & 1) ? 0xBF800000u : 0x0000BF80u;
    }
    const bf16x8 A0 = __builtin_bit_cast(bf16x8, make_uint4(a0[0], a0[1], a0[2], a0[3]));
    const bf16x8 A1 = __builtin_bit_cast(bf16x8, make_uint4(a1[0], a1[1], a1[2], a1[3]));
    f32x4 acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * (1.f + 0.37f * i + 0.011f * threadIdx.x);
    bf16x8 pay = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u + lane, 0x3f813f80u, 0x3f823f80u, 0x3f833f80u));
    for (int it = 0; it < iters; ++it) {
        uint32_t hp[4], mp[4], lp[4];
        for (int i = 0; i < 4; ++i) hp[i] = pk2(v[2 * i], v[2 * i + 1]);
        if (MODE == 0) {          // vector split
            float r1[8], r2[8];
            for (int i = 0; i < 4; ++i) { r1[2 * i] = v[2 * i] - lo_f(hp[i]); r1[2 * i + 1] = v[2 * i + 1] - hi_f(hp[i]); }
            for (int i = 0; i < 4; ++i) mp[i] = pk2(r1[2 * i], r1[2 * i + 1]);
            for (int i = 0; i < 4; ++i) { r2[2 * i] = r1[2 * i] - lo_f(mp[i]); r2[2 * i + 1] = r1[2 * i + 1] - hi_f(mp[i]); }
            for (int i = 0; i < 4; ++i) lp[i] = pk2(r2[2 * i], r2[2 * i + 1]);
        } else {                  // matrix split
            f32x4 c0 = {v[0], v[1], v[2], v[3]}, c1 = {v[4], v[5], v[6], v[7]};
            bf16x8 Bh = __builtin_bit_cast(bf16x8, make_uint4(hp[0], hp[1], hp[2], hp[3]));
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, Bh, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, Bh, c1, 0, 0, 0);
            mp[0] = pk2(c0[0], c0[1]); mp[1] = pk2(c0[2], c0[3]); mp[2] = pk2(c1[0], c1[1]); mp[3] = pk2(c1[2], c1[3]);
            bf16x8 Bm = __builtin_bit_cast(bf16x8, make_uint4(mp[0], mp[1], mp[2], mp[3]));
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, Bm, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, Bm, c1, 0, 0, 0);
            lp[0] = pk2(c0[0], c0[1]); lp[1] = pk2(c0[2], c0[3]); lp[2] = pk2(c1[0], c1[1]); lp[3] = pk2(c1[2], c1[3]);
        }
        const bf16x8 Ph = __builtin_bit_cast(bf16x8, make_uint4(hp[0], hp[1], hp[2], hp[3]));
        const bf16x8 Pm = __builtin_bit_cast(bf16x8, make_uint4(mp[0], mp[1], mp[2], mp[3]));
        const bf16x8 Pl = __builtin_bit_cast(bf16x8, make_uint4(lp[0], lp[1], lp[2], lp[3]));
        // payload: the six products of one output tile with the fresh planes + NM - 6 products on stationary operands
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const bf16x8 b = i == 0 ? Pm : i == 1 ? Ph : i == 2 ? Pl : i == 3 ? Ph : i == 4 ? Pm : i == 5 ? Ph : pay;
            acc[i % 6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pay, b, acc[i % 6], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) v[i] = v[i] * 1.0001f + 0.5f;      // (2 more vector instructions per value: stands for exp / scale)
    }
    float s = 0.f;
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s + v[0];
}

static float frand(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xffffff) / 16777216.f; }

template <typename K>
static void time_rate(const char* name, K kern, int threads) {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 1024);
    const int iters = 20000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // per SIMD: threads / 256 wavefronts each doing `iters` iterations
    printf("%-28s %d wave/SIMD: %8.3f ms  -> %7.1f ns per (iteration x wavefront-on-SIMD)\n", name, threads / 256, ms,
           ms * 1e6 / iters / (threads / 256));
    hipFree(out);
}

int main() {
    const int NB = 4096;
    const size_t n16 = (size_t)NB * 64 * 8, n32 = (size_t)NB * 64 * 16;
    std::vector<float> h(n32);
    uint64_t s = 12345;
    const char* names[] = {"uniform exponents", "probabilities exp2(-u)", "worst-split 0x7F40", "denormals / tiny", "all-ones low bits"};
    float *dx; uint32_t* dout; unsigned long long* dbad;
    hipMalloc(&dx, n32 * 4); hipMalloc(&dout, 64); hipMalloc(&dbad, 32);
    int fail = 0;
    for (int cls = 0; cls < 5; ++cls) {
        for (size_t i = 0; i < n32; ++i) {
            float v;
            if (cls == 0) { const int e = (int)(frand(s) * 200) - 100; v = ldexpf(1.f + frand(s), e) * (frand(s) < 0.5f ? -1.f : 1.f); }
            else if (cls == 1) v = exp2f(-frand(s) * 40.f);
            else if (cls == 2) { uint32_t u = __builtin_bit_cast(uint32_t, 0.5f + frand(s)); u = (u & 0xffff0000u) | 0x7F40u; v = __builtin_bit_cast(float, u); }
            else if (cls == 3) { const int e = -149 + (int)(frand(s) * 40); v = ldexpf(1.f + frand(s), e); if (frand(s) < 0.05f) v = 0.f; }
            else { uint32_t u = __builtin_bit_cast(uint32_t, 1.f + frand(s)); u |= 0xffffu; v = __builtin_bit_cast(float, u) * (frand(s) < 0.5f ? -1.f : 1.f); }
            h[i] = v;
        }
        hipMemcpy(dx, h.data(), n32 * 4, hipMemcpyHostToDevice);
        for (int shape = 0; shape < 2; ++shape) {
            hipMemset(dbad, 0, 32); hipMemset(dout, 0, 64);
            if (shape == 0) hipLaunchKernelGGL(check16, dim3(NB), dim3(64), 0, 0, dx, dout, dbad);
            else hipLaunchKernelGGL(check32, dim3(NB), dim3(64), 0, 0, dx, dout, dbad);
            unsigned long long badv[4]; uint32_t o[16];
            hipMemcpy(badv, dbad, 32, hipMemcpyDeviceToHost);
            hipMemcpy(o, dout, 64, hipMemcpyDeviceToHost);
            const unsigned long long bad = badv[0];
            if (shape == 1 && bad) printf("   regs with level-1 mismatches %04x, lanes<32 %08x\n", o[6], o[7]);
            if (bad) printf("   level-1 mismatches %llu (x=%08x got=%08x ref=%08x at lane*16+idx %u); level-2 %llu (r1=%08x got=%08x ref=%08x at %u)\n",
                                          badv[1], o[8], o[9], o[10], o[11], badv[2], o[12], o[13], o[14], o[15]);
            printf("%-26s %s: %llu mismatches of %zu values", names[cls], shape == 0 ? "16x16x32" : "32x32x16", bad, shape == 0 ? n16 : n32);
            if (bad) { printf("  first: x=%08x r1=%08x ref=%08x r2=%08x ref=%08x lane %u", o[0], o[1], o[2], o[3], o[4], o[5]); fail = 1; }
            printf("\n");
        }
    }
    (void)n16;
    printf("%s\n", fail ? "MISMATCH" : "matrix-pipe residuals bit-identical to the vector form");
    for (int w = 1; w <= 4; ++w) {
        time_rate("vector split + 12 MFMA", rate16<0, 12>, 256 * w);
        time_rate("matrix split + 12 MFMA", rate16<1, 12>, 256 * w);
        time_rate("vector split +  6 MFMA", rate16<0, 6>, 256 * w);
        time_rate("matrix split +  6 MFMA", rate16<1, 6>, 256 * w);
    }
    return fail;
}

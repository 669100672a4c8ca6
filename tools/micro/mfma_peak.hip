// Calibration: what one MI355X sustains on back-to-back v_mfma_f32_32x32x16_bf16 (no memory traffic), for
// 1, 2 and 4 wavefronts per SIMD, with zero and with random operands.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
    const uint4 a4 = in[threadIdx.x], b4 = in[256 + threadIdx.x];
    bf16x8 a = __builtin_bit_cast(bf16x8, a4), b = __builtin_bit_cast(bf16x8, b4);
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 123.456f) out[0] = s;
}

int main() {
    uint4* in; float* out;
    hipMalloc(&in, 512 * 16); hipMalloc(&out, 4);
    uint32_t h[2048];
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 2048; ++i) {
            // random bf16 pairs of magnitude ~1 (exponent 0x3f8 region) or zeros
            uint32_t lo = 0x3f00 | (rand() & 0xff) | ((rand() & 1) << 15), hi = 0x3f00 | (rand() & 0xff) | ((rand() & 1) << 15);
            h[i] = mode ? (hi << 16 | lo) : 0u;
        }
        hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
            const int iters = 20000, grid = 256 * wgs_per_cu;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, in, out, 100);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)grid * 4 /*waves*/ * iters * 4.0 * 32 * 32 * 16 * 2;
            const double per_simd = (double)wgs_per_cu * iters * 4.0;     // MFMAs per SIMD
            printf("%s operands, %d wave(s)/SIMD: %.3f ms  %.1f TF/s  -> %.2f GHz if 32 cyc/MFMA\n", mode ? "random" : "zero  ",
                   wgs_per_cu, ms, flops / ms / 1e9, per_simd * 32 / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}

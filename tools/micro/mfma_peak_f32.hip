// Calibration: what one MI355X sustains on back-to-back v_mfma_f32_32x32x2_f32 (the exact-fp32 mode's instruction; no
// memory traffic) for 1, 2 and 4 wavefronts per SIMD, zero and random operands - the ceiling `roofline.frac` of the fp32
// kernels should be read against (nominal: 157.3 TF/s = 256 flop / clk / CU at 2.4 GHz).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const float a = in[threadIdx.x], b = in[256 + threadIdx.x];
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 123.456f) out[0] = s;
}

int main() {
    float* in; float* out;
    hipMalloc(&in, 512 * 4); hipMalloc(&out, 4);
    float h[512];
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 512; ++i) h[i] = mode ? (float)rand() / RAND_MAX - 0.5f : 0.f;
        hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
            const int iters = 10000, grid = 256 * wgs_per_cu;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, in, out, 100);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)grid * 4 /*waves*/ * iters * 4.0 * 32 * 32 * 2 * 2;
            const double per_simd = (double)wgs_per_cu * iters * 4.0;     // MFMAs per SIMD
            printf("%s operands, %d wave(s)/SIMD: %.3f ms  %.1f TF/s = %.3f of 157.3 -> %.2f GHz if 64 cyc/MFMA\n", mode ? "random" : "zero  ",
                   wgs_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3, per_simd * 64 / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}

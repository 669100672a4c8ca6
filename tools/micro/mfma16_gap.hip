// How many vector instructions hide behind a v_mfma_f32_16x16x32_bf16 (gfx950)?  Register-only loop: 16 MFMAs per iteration
// (four independent accumulator chains), each followed by a fenced gap of NV plain vector instructions (v_add_f32 on independent
// registers) and NE v_exp_f32; W wavefronts per SIMD.  Prints cycles per MFMA (s_memtime; 16 = the matrix pace).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma16_gap.hip -o tools/micro/mfma16_gap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NE, int NC>
__global__ __launch_bounds__(1024) void loop(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u + lane, 0x3f813f80u, 0x3f823f80u, 0x3f833f80u));
    bf16x8 b = __builtin_bit_cast(bf16x8, make_uint4(0x3f813f82u + lane, 0x3f813f83u, 0x3f803f80u, 0x3f833f81u));
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8], e[4];
    uint32_t cv[4];
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * (lane + i);
    for (int i = 0; i < 4; ++i) { e[i] = -0.5f * i; cv[i] = 0; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(m * NV + k) & 7]) : "v"(e[0]));
#pragma unroll
            for (int k = 0; k < NE; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(e[(m * NE + k) & 3]));
#pragma unroll
            for (int k = 0; k < NC; ++k) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(cv[(m * NC + k) & 3]) : "v"(v[k]), "v"(v[k + 4]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + e[i] + __builtin_bit_cast(float, cv[i]);
    for (int i = 0; i < 8; ++i) s += v[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NV, int NE, int NC>
static void run(float* out, unsigned long long* cyc) {
    const int iters = 4000;
    for (int w = 1; w <= 2; ++w) {
        hipLaunchKernelGGL((loop<NV, NE, NC>), dim3(256), dim3(256 * w), 0, 0, out, cyc, 100);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop<NV, NE, NC>), dim3(256), dim3(256 * w), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("NV %d NE %d NC %d, %d wave/SIMD: %6.1f cycles per MFMA-of-one-wave (%.1f per SIMD-MFMA), %.3f ms, clock %.2f GHz\n", NV, NE, NC, w,
               (double)c / (iters * 16.0), (double)c / (iters * 16.0 * w), ms, (double)c / (ms * 1e6));
    }
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&cyc, 8);
    run<0, 0, 0>(out, cyc); run<1, 0, 0>(out, cyc); run<2, 0, 0>(out, cyc); run<3, 0, 0>(out, cyc); run<4, 0, 0>(out, cyc);
    run<0, 1, 0>(out, cyc); run<1, 1, 0>(out, cyc); run<2, 1, 0>(out, cyc); run<0, 0, 1>(out, cyc); run<0, 0, 2>(out, cyc); run<1, 0, 1>(out, cyc); run<0, 1, 1>(out, cyc);
    return 0;
}

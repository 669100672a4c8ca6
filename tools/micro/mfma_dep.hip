// Does a v_mfma_f32_32x32x16_bf16 whose C operand is the result of the MFMA issued just before it cost more than one with
// an independent accumulator?  (The split-bf16 kernels issue their three products of a block back to back into one
// accumulator.)  DIST = number of accumulators used round-robin: 1 = every MFMA depends on its predecessor, 3x1 = groups
// of three dependent MFMAs per accumulator over four accumulators (the kernels' pattern), 2 / 4 = distance 2 / 4.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_dep.hip -o tools/micro/mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// the fence keeps the order written below (the machine scheduler otherwise spreads dependent MFMAs apart by itself)
#define MF(c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
    const uint4 a4 = in[threadIdx.x], b4 = in[256 + threadIdx.x];
    bf16x8 a = __builtin_bit_cast(bf16x8, a4), b = __builtin_bit_cast(bf16x8, b4);
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    for (int i = 0; i < iters; ++i) {          // 12 MFMAs per iteration in every mode
        if (MODE == 1) { MF(c0); MF(c0); MF(c0); MF(c0); MF(c0); MF(c0); MF(c0); MF(c0); MF(c0); MF(c0); MF(c0); MF(c0); }
        if (MODE == 2) { MF(c0); MF(c1); MF(c0); MF(c1); MF(c0); MF(c1); MF(c0); MF(c1); MF(c0); MF(c1); MF(c0); MF(c1); }
        if (MODE == 3) { MF(c0); MF(c0); MF(c0); MF(c1); MF(c1); MF(c1); MF(c2); MF(c2); MF(c2); MF(c3); MF(c3); MF(c3); }
        if (MODE == 4) { MF(c0); MF(c1); MF(c2); MF(c3); MF(c0); MF(c1); MF(c2); MF(c3); MF(c0); MF(c1); MF(c2); MF(c3); }
        if (MODE == 5) { MF(c0); MF(c1); MF(c0); MF(c1); MF(c0); MF(c1); MF(c2); MF(c3); MF(c2); MF(c3); MF(c2); MF(c3); }   // pairs interleaved
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 123.456f) out[0] = s;
}

template <int MODE>
void run(const uint4* in, float* out, const char* what) {
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        const int iters = 8000, grid = 256 * wgs_per_cu;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, in, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 * iters * 12.0 * 32 * 32 * 16 * 2;
        printf("%-34s %d wave(s)/SIMD: %.3f ms  %.1f TF/s\n", what, wgs_per_cu, ms, flops / ms / 1e9);
    }
}

int main() {
    uint4* in; float* out;
    (void)hipMalloc(&in, 512 * 16); (void)hipMalloc(&out, 4);
    uint32_t h[2048];
    for (int i = 0; i < 2048; ++i) {
        uint32_t lo = 0x3f00 | (rand() & 0xff) | ((rand() & 1) << 15), hi = 0x3f00 | (rand() & 0xff) | ((rand() & 1) << 15);
        h[i] = hi << 16 | lo;
    }
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<4>(in, out, "4 accumulators round-robin");
        run<1>(in, out, "1 accumulator (all dependent)");
        run<2>(in, out, "2 accumulators alternating");
        run<3>(in, out, "3 dependent per accumulator");
        run<5>(in, out, "pairs interleaved (distance 2)");
    }
    return 0;
}

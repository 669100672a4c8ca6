// Do vector ALU instructions overlap a partner wavefront's fp32 MFMAs on one SIMD of gfx950?
//
// 512-thread workgroups, one per CU: wavefronts 0-3 and 4-7 land pairwise on the CU's four SIMDs.  Role A = a loop of
// independent v_mfma_f32_16x16x4_f32 (or 32x32x2), role B = a loop of independent v_fma_f32.  Timed: A alone (B exits),
// B alone, both, and ONE wavefront per SIMD running the same MFMAs with NF fillers placed after each MFMA in its own stream.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// the same pairing with the bf16 MFMA (v_mfma_f32_32x32x16_bf16: 8 passes, 32 cycles) for comparison: role A = 8 independent MFMAs per
// iteration, role B = NV vector instructions per iteration (v_fma_f32, or v_exp_f32 when KIND = 1)
template <int KIND, int NV, int PAD = 0>
__global__ __launch_bounds__(512) void two_waves_bf16(const uint4* __restrict__ in, float* __restrict__ out, int iters, int mode) {
    const int wv = threadIdx.x >> 6;
    const bf16x8 a = __builtin_bit_cast(bf16x8, in[threadIdx.x & 127]), b = __builtin_bit_cast(bf16x8, in[128 + (threadIdx.x & 127)]);
    if (wv < 4) {
        if (!(mode & 1)) return;
        f32x16 c[2] = {};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[u & 1]) : "v"(a), "v"(b));
                // PAD: the wave steps back from the vector issue port while its MFMA runs (s_nop N = N + 1 idle cycles of this wave)
                // (s_nop N holds the wave for 4 (N + 1) cycles: measured, 80 cycles for s_nop 7 + s_nop 7 + s_nop 3)
                if (PAD == 1) asm volatile("s_nop 2");           // 12 cycles
                if (PAD == 2) asm volatile("s_nop 3");           // 16 cycles
                if (PAD == 3) asm volatile("s_nop 4");           // 20 cycles
                if (PAD == 4) asm volatile("s_nop 5");           // 24 cycles
                if (PAD == 5) asm volatile("s_nop 6");           // 28 cycles
            }
        }
        float s = 0.f;
        for (int u = 0; u < 2; ++u) for (int r = 0; r < 16; ++r) s += c[u][r];
        if (s == 123.456f) out[0] = s;
    } else {
        if (!(mode & 2)) return;
        float f[8];
        const float bb = __builtin_bit_cast(float, in[threadIdx.x & 127].x);
        for (int j = 0; j < 8; ++j) f[j] = bb + 0.001f * j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[u & 7]) : "v"(bb));
                else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f[u & 7]));
                else if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[u & 7]) : "v"(bb));
                else if (KIND == 3) {            // the mix of a split: convert, shift, and, two subtractions, one exp in eight
                    if ((u & 7) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(f[u & 7]));
                    else if ((u & 7) < 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[u & 7]) : "v"(bb));
                    else if ((u & 7) < 5) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(f[u & 7]));
                    else asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[u & 7]) : "v"(bb));
                }
                else if (KIND == 6) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(f[u & 7]));
                else if (KIND == 7) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[u & 7]) : "v"(bb));
                else if (KIND == 8) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(f[u & 7]));
                else if (KIND == 9) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[u & 7]) : "v"(bb));
                else if (KIND == 10) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[u & 7]) : "v"(bb));
                else if (KIND == 11) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[u & 7]) : "v"(bb));
                else if (KIND == 12) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[u & 7]));
                else if (KIND == 13) asm volatile("v_mov_b32 %0, %1" : "+v"(f[u & 7]) : "v"(bb));
                else if (KIND == 14) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(f[u & 7]) : "v"(bb));
                else if (KIND == 15) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[u & 7]) : "v"(bb));
                else if (KIND == 16) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<unsigned long long*>(&f[2 * (u & 3)])) : "v"(*reinterpret_cast<unsigned long long*>(&f[0])));
                else if (KIND == 4) asm volatile("ds_write_b64 %0, %1" :: "v"((threadIdx.x & 255) * 8 + 16384), "v"(*reinterpret_cast<unsigned long long*>(&f[0])) : "memory");
                else if (KIND == 5) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(*reinterpret_cast<uint4*>(&f[0])) : "v"((threadIdx.x & 255) * 16 + 16384) : "memory");
            }
        }
        float s = 0.f;
        for (int j = 0; j < 8; ++j) s += f[j];
        if (s == 123.456f) out[1] = s;
    }
}

// mode bit 0: MFMA wavefronts run; bit 1: VALU wavefronts run.  nm MFMAs / nv FMAs per loop iteration are compile-time.
template <int SHAPE, int KIND = 0, int PAD = 0>
__global__ __launch_bounds__(512) void two_waves(const float* __restrict__ in, float* __restrict__ out, int iters, int mode) {
    const int wv = threadIdx.x >> 6;
    const float a = in[threadIdx.x & 255], b = in[256 + (threadIdx.x & 255)];
    if (wv < 4) {
        if (!(mode & 1)) return;
        if (SHAPE == 16) {
            f32x4 c[4] = {};
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c[u & 3]) : "v"(a), "v"(b));
                    if (PAD == 1) asm volatile("s_nop 2");
                    if (PAD == 2) asm volatile("s_nop 4");
                }
            }
            float s = 0.f;
            for (int u = 0; u < 4; ++u) s += c[u][0] + c[u][1] + c[u][2] + c[u][3];
            if (s == 123.456f) out[0] = s;
        } else {
            f32x16 c[2] = {};
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c[u & 1]) : "v"(a), "v"(b));
                    if (PAD == 1) asm volatile("s_nop 5");
                    if (PAD == 2) asm volatile("s_nop 7\n\ts_nop 3");
                }
            }
            float s = 0.f;
            for (int u = 0; u < 2; ++u) for (int r = 0; r < 16; ++r) s += c[u][r];
            if (s == 123.456f) out[0] = s;
        }
    } else {
        if (!(mode & 2)) return;
        float f[8];
        for (int j = 0; j < 8; ++j) f[j] = a + 0.001f * j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[u & 7]) : "v"(b));
                else asm volatile("v_exp_f32 %0, %0" : "+v"(f[u & 7]));
            }
        }
        float s = 0.f;
        for (int j = 0; j < 8; ++j) s += f[j];
        if (s == 123.456f) out[1] = s;
    }
}

// one wavefront per SIMD: each 16x16x4 MFMA followed by NF independent v_fma_f32 in the same stream
template <int NF, int KIND = 0>
__global__ __launch_bounds__(256) void one_wave(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const float a = in[threadIdx.x & 255], b = in[256 + (threadIdx.x & 255)];
    f32x4 c[4] = {};
    float f[8];
    for (int j = 0; j < 8; ++j) f[j] = a + 0.001f * j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c[u & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[j & 7]) : "v"(b));
                else asm volatile("v_exp_f32 %0, %0" : "+v"(f[j & 7]));
            }
        }
    }
    float s = 0.f;
    for (int u = 0; u < 4; ++u) s += c[u][0] + c[u][1] + c[u][2] + c[u][3];
    for (int j = 0; j < 8; ++j) s += f[j];
    if (s == 123.456f) out[0] = s;
}

template <typename F>
static float time_ms(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* in; float* out;
    hipMalloc(&in, 4096); hipMalloc(&out, 16);
    float h[512];
    for (int i = 0; i < 512; ++i) h[i] = (float)(i % 97) / 97.f - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int shape = 16; shape <= 32; shape += 16) {
        float t[4];
        for (int mode = 1; mode <= 3; ++mode)
            t[mode] = shape == 16 ? time_ms([&] { hipLaunchKernelGGL((two_waves<16>), dim3(256), dim3(512), 0, 0, in, out, iters, mode); })
                                  : time_ms([&] { hipLaunchKernelGGL((two_waves<32>), dim3(256), dim3(512), 0, 0, in, out, iters, mode); });
        // per iteration: 512 matrix-pipe cycles (16 x 32 or 8 x 64), 64 FMAs = 256 issue cycles
        printf("shape %dx: MFMA wave alone %.3f ms, VALU wave alone %.3f ms, both on one SIMD %.3f ms (sum %.3f, max %.3f)\n",
               shape, t[1], t[2], t[3], t[1] + t[2], t[1] > t[2] ? t[1] : t[2]);
    }
    {
        float t[4];
#define FPAD(SH, P) \
        for (int mode = 1; mode <= 3; ++mode) \
            t[mode] = time_ms([&] { hipLaunchKernelGGL((two_waves<SH, 0, P>), dim3(256), dim3(512), 0, 0, in, out, iters, mode); }); \
        printf("fp32 shape %d, MFMA wave padded with s_nop (variant %d): MFMA alone %.3f ms, vector alone %.3f, both %.3f (sum %.3f)\n", SH, P, t[1], t[2], t[3], t[1] + t[2]);
        FPAD(16, 1) FPAD(16, 2) FPAD(32, 1) FPAD(32, 2)
    }
    {
        float t[4];
        for (int mode = 1; mode <= 3; ++mode)
            t[mode] = time_ms([&] { hipLaunchKernelGGL((two_waves<16, 1>), dim3(256), dim3(512), 0, 0, in, out, iters, mode); });
        printf("shape 16x, partner runs v_exp_f32: MFMA wave alone %.3f ms, exp wave alone %.3f ms, both %.3f ms (sum %.3f)\n", t[1], t[2], t[3],
               t[1] + t[2]);
        printf("one wave, 16x16x4 + NF v_exp_f32 per MFMA: NF=1 %.3f", time_ms([&] { hipLaunchKernelGGL((one_wave<1, 1>), dim3(256), dim3(256), 0, 0, in, out, iters); }));
        printf(", 2: %.3f", time_ms([&] { hipLaunchKernelGGL((one_wave<2, 1>), dim3(256), dim3(256), 0, 0, in, out, iters); }));
        printf(", 4: %.3f ms\n", time_ms([&] { hipLaunchKernelGGL((one_wave<4, 1>), dim3(256), dim3(256), 0, 0, in, out, iters); }));
    }
    {
        float t[4];
        for (int mode = 1; mode <= 3; ++mode)
            t[mode] = time_ms([&] { hipLaunchKernelGGL((two_waves_bf16<0, 32>), dim3(256), dim3(512), 0, 0, (const uint4*)in, out, iters, mode); });
        printf("bf16 32x32x16 (8 per iteration = 256 pipe cycles) + partner 32 v_fma (128 issue cycles): MFMA alone %.3f ms, vector alone %.3f, both %.3f (sum %.3f)\n",
               t[1], t[2], t[3], t[1] + t[2]);
        for (int mode = 1; mode <= 3; ++mode)
            t[mode] = time_ms([&] { hipLaunchKernelGGL((two_waves_bf16<0, 48>), dim3(256), dim3(512), 0, 0, (const uint4*)in, out, iters, mode); });
        printf("bf16 32x32x16 + partner 48 v_fma (192 issue cycles = the 24 free cycles of every MFMA): MFMA alone %.3f ms, vector alone %.3f, both %.3f (sum %.3f)\n",
               t[1], t[2], t[3], t[1] + t[2]);
        for (int mode = 1; mode <= 3; ++mode)
            t[mode] = time_ms([&] { hipLaunchKernelGGL((two_waves_bf16<0, 64>), dim3(256), dim3(512), 0, 0, (const uint4*)in, out, iters, mode); });
        printf("bf16 32x32x16 + partner 64 v_fma (256 issue cycles): MFMA alone %.3f ms, vector alone %.3f, both %.3f (sum %.3f)\n",
               t[1], t[2], t[3], t[1] + t[2]);
#define KINDRUN(K, NVV, NAME) \
        for (int mode = 1; mode <= 3; ++mode) \
            t[mode] = time_ms([&] { hipLaunchKernelGGL((two_waves_bf16<K, NVV, 1>), dim3(256), dim3(512), 32768, 0, (const uint4*)in, out, iters, mode); }); \
        printf("bf16 MFMA + s_nop 2, partner %d x %s per 8 MFMAs: MFMA alone %.3f ms, partner alone %.3f, both %.3f (sum %.3f)\n", NVV, NAME, t[1], t[2], t[3], t[1] + t[2]);
        KINDRUN(1, 16, "v_exp_f32") KINDRUN(2, 40, "v_cvt_pk_bf16_f32") KINDRUN(3, 40, "split mix") KINDRUN(4, 16, "ds_write_b64") KINDRUN(5, 8, "ds_read_b128 + wait")
        KINDRUN(6, 40, "v_lshlrev_b32") KINDRUN(7, 40, "v_sub_f32") KINDRUN(8, 40, "v_and_b32") KINDRUN(9, 40, "v_max_f32") KINDRUN(10, 40, "v_mul_f32")
        KINDRUN(11, 40, "v_add_f32") KINDRUN(12, 16, "v_rcp_f32") KINDRUN(13, 40, "v_mov_b32") KINDRUN(14, 16, "v_mul_lo_u32") KINDRUN(15, 40, "v_cndmask_b32") KINDRUN(16, 20, "v_pk_mul_f32")
#define PADRUN(P) \
        for (int mode = 1; mode <= 3; ++mode) \
            t[mode] = time_ms([&] { hipLaunchKernelGGL((two_waves_bf16<0, 40, P>), dim3(256), dim3(512), 0, 0, (const uint4*)in, out, iters, mode); }); \
        printf("bf16 32x32x16 each followed by pad %d + partner 40 v_fma: MFMA alone %.3f ms, vector alone %.3f, both %.3f (sum %.3f)\n", P, t[1], t[2], t[3], t[1] + t[2]);
        PADRUN(0) PADRUN(1) PADRUN(2) PADRUN(3) PADRUN(4) PADRUN(5)
        for (int mode = 1; mode <= 3; ++mode)
            t[mode] = time_ms([&] { hipLaunchKernelGGL((two_waves_bf16<1, 16>), dim3(256), dim3(512), 0, 0, (const uint4*)in, out, iters, mode); });
        printf("bf16 32x32x16 + partner 16 v_exp: MFMA alone %.3f ms, vector alone %.3f, both %.3f (sum %.3f)\n", t[1], t[2], t[3], t[1] + t[2]);
    }
    const float base = time_ms([&] { hipLaunchKernelGGL((one_wave<0>), dim3(256), dim3(256), 0, 0, in, out, iters); });
    printf("one wave, 16x16x4 + NF fillers per MFMA: NF=0 %.3f ms", base);
    printf(", 2: %.3f", time_ms([&] { hipLaunchKernelGGL((one_wave<2>), dim3(256), dim3(256), 0, 0, in, out, iters); }));
    printf(", 4: %.3f", time_ms([&] { hipLaunchKernelGGL((one_wave<4>), dim3(256), dim3(256), 0, 0, in, out, iters); }));
    printf(", 5: %.3f", time_ms([&] { hipLaunchKernelGGL((one_wave<5>), dim3(256), dim3(256), 0, 0, in, out, iters); }));
    printf(", 6: %.3f", time_ms([&] { hipLaunchKernelGGL((one_wave<6>), dim3(256), dim3(256), 0, 0, in, out, iters); }));
    printf(", 8: %.3f", time_ms([&] { hipLaunchKernelGGL((one_wave<8>), dim3(256), dim3(256), 0, 0, in, out, iters); }));
    printf(", 12: %.3f ms\n", time_ms([&] { hipLaunchKernelGGL((one_wave<12>), dim3(256), dim3(256), 0, 0, in, out, iters); }));
    return 0;
}

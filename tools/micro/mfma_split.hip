// What does the exact three-way bf16 split of four fp32 values (22 vector instructions: cvt_pk / shift / and / sub, 7 levels deep,
// 4 chains wide) cost in the shadows of six dependent v_mfma_f32_32x32x16_bf16 of the SAME wavefront (one wavefront per SIMD)?
//   mode 0: the six MFMAs alone;  1: split in level order, 4 per gap;  2: split as one block behind the six MFMAs;
//   3: two splits (44 instructions), level order, 8 per gap;  4: level order, 4 per gap, plus 4 ds_read_b128 per step;
//   5: as 1 with the fragment of the NEXT MFMA written by the split (true dependence of the MFMA's B operand on the gap's work)
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_split.hip -o tools/micro/mfma_split
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MF asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#define CVT(d, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define SHL(d, x) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(d) : "v"(x))
#define AND(d, x) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(d) : "v"(x))
#define SUB(d, x, y) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ in, float* __restrict__ out, unsigned long long* cyc, int iters) {
    __shared__ uint4 lds[1024];
    lds[threadIdx.x] = in[threadIdx.x & 511];
    lds[threadIdx.x + 256] = in[threadIdx.x & 511];
    __syncthreads();
    bf16x8 a = __builtin_bit_cast(bf16x8, in[threadIdx.x & 511]), b = __builtin_bit_cast(bf16x8, in[(threadIdx.x + 256) & 511]);
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    float x0 = 1.0f + 0.001f * threadIdx.x, x1 = x0 * 1.3f, x2 = x0 * 1.7f, x3 = x0 * 2.1f;
    float y0 = x0 * 0.3f, y1 = x0 * 0.7f, y2 = x0 * 0.9f, y3 = x0 * 1.1f;
    uint32_t h01, h23, m01, m23, l01 = 0, l23 = 0, t0_, t1_, t2_, t3_;
    float r0, r1, r2, r3;
    uint32_t H01, H23, M01, M23, L01 = 0, L23 = 0, T0, T1, T2, T3;
    float R0, R1, R2, R3;
    uint4 d0 = {}, d1 = {}, d2 = {}, d3 = {};
    const uint32_t lp = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4*)(lds + (threadIdx.x & 255));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { MF; MF; MF; MF; MF; MF; }
        if (MODE == 1 || MODE == 4 || MODE == 5) {
            if (MODE == 4) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(d0) : "v"(lp));
                asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(d1) : "v"(lp));
                asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(d2) : "v"(lp));
                asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(d3) : "v"(lp));
            }
            MF; CVT(h01, x0, x1); CVT(h23, x2, x3); SHL(t0_, h01); AND(t1_, h01);
            MF; SHL(t2_, h23); AND(t3_, h23); SUB(r0, x0, t0_); SUB(r1, x1, t1_);
            MF; SUB(r2, x2, t2_); SUB(r3, x3, t3_); CVT(m01, r0, r1); CVT(m23, r2, r3);
            MF; SHL(t0_, m01); AND(t1_, m01); SHL(t2_, m23); AND(t3_, m23);
            MF; SUB(r0, r0, t0_); SUB(r1, r1, t1_); SUB(r2, r2, t2_); SUB(r3, r3, t3_);
            MF; CVT(l01, r0, r1); CVT(l23, r2, r3);
            if (MODE == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (MODE == 5) { uint4 nb = make_uint4(h01, h23, m01 ^ l01, m23 ^ l23); b = __builtin_bit_cast(bf16x8, nb); }
        }
        if (MODE == 2) {
            MF; MF; MF; MF; MF; MF;
            CVT(h01, x0, x1); CVT(h23, x2, x3); SHL(t0_, h01); AND(t1_, h01);
            SHL(t2_, h23); AND(t3_, h23); SUB(r0, x0, t0_); SUB(r1, x1, t1_);
            SUB(r2, x2, t2_); SUB(r3, x3, t3_); CVT(m01, r0, r1); CVT(m23, r2, r3);
            SHL(t0_, m01); AND(t1_, m01); SHL(t2_, m23); AND(t3_, m23);
            SUB(r0, r0, t0_); SUB(r1, r1, t1_); SUB(r2, r2, t2_); SUB(r3, r3, t3_);
            CVT(l01, r0, r1); CVT(l23, r2, r3);
        }
        if (MODE == 6 || MODE == 7 || MODE == 8) {
            // the X step of the attention kernels: the split in level order, 4 per gap, one ds_read_b128 in each of the first four
            // gaps (mode 7: the MFMAs of the NEXT step take what was read: a wait before each; mode 8: reads only, no split)
#define RD(d, o) asm volatile("ds_read_b128 %0, %1 offset:" #o : "=v"(d) : "v"(lp))
#define MFA(x) { bf16x8 xa = __builtin_bit_cast(bf16x8, x); asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(xa), "v"(b)); }
            if (MODE == 7) { asm volatile("s_waitcnt lgkmcnt(3)"); MFA(d0); } else MF;
            RD(d0, 0); if (MODE != 8) { CVT(h01, x0, x1); CVT(h23, x2, x3); SHL(t0_, h01); AND(t1_, h01); }
            if (MODE == 7) { asm volatile("s_waitcnt lgkmcnt(3)"); MFA(d1); } else MF;
            RD(d1, 4096); if (MODE != 8) { SHL(t2_, h23); AND(t3_, h23); SUB(r0, x0, t0_); SUB(r1, x1, t1_); }
            if (MODE == 7) { asm volatile("s_waitcnt lgkmcnt(3)"); MFA(d2); } else MF;
            RD(d2, 8192); if (MODE != 8) { SUB(r2, x2, t2_); SUB(r3, x3, t3_); CVT(m01, r0, r1); CVT(m23, r2, r3); }
            if (MODE == 7) { asm volatile("s_waitcnt lgkmcnt(3)"); MFA(d3); } else MF;
            RD(d3, 12288); if (MODE != 8) { SHL(t0_, m01); AND(t1_, m01); SHL(t2_, m23); AND(t3_, m23); }
            MF; if (MODE != 8) { SUB(r0, r0, t0_); SUB(r1, r1, t1_); SUB(r2, r2, t2_); SUB(r3, r3, t3_); }
            MF; if (MODE != 8) { CVT(l01, r0, r1); CVT(l23, r2, r3); }
        }
        if (MODE == 9 || MODE == 10) {
            // the split with its subtractions as v_pk_add_f32 (two values per instruction): 18 instead of 22 instructions
            typedef float f2 __attribute__((ext_vector_type(2)));
#define PKSUB(d, x, y) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y))
            f2 xa = {x0, x1}, xb = {x2, x3}, ta, tb, ra, rb;
            uint32_t ua0, ua1, ub0, ub1;
            MF; CVT(h01, xa.x, xa.y); CVT(h23, xb.x, xb.y); SHL(ua0, h01); AND(ua1, h01);
            MF; SHL(ub0, h23); AND(ub1, h23); ta = f2{__builtin_bit_cast(float, ua0), __builtin_bit_cast(float, ua1)}; PKSUB(ra, xa, ta); tb = f2{__builtin_bit_cast(float, ub0), __builtin_bit_cast(float, ub1)}; PKSUB(rb, xb, tb);
            MF; CVT(m01, ra.x, ra.y); CVT(m23, rb.x, rb.y); SHL(ua0, m01); AND(ua1, m01);
            MF; SHL(ub0, m23); AND(ub1, m23); ta = f2{__builtin_bit_cast(float, ua0), __builtin_bit_cast(float, ua1)}; PKSUB(ra, ra, ta); tb = f2{__builtin_bit_cast(float, ub0), __builtin_bit_cast(float, ub1)}; PKSUB(rb, rb, tb);
            MF; CVT(l01, ra.x, ra.y); CVT(l23, rb.x, rb.y);
            if (MODE == 10) { SHL(ua0, l01); AND(ua1, l01); SHL(ub0, l23); AND(ub1, l23); }
            MF;
            if (MODE == 10) { CVT(L01, x0, x2); CVT(L23, x1, x3); SHL(T0, L01); AND(T1, L01); }
        }
        if (MODE == 3) {
            MF; CVT(h01, x0, x1); CVT(h23, x2, x3); SHL(t0_, h01); AND(t1_, h01); CVT(H01, y0, y1); CVT(H23, y2, y3); SHL(T0, H01); AND(T1, H01);
            MF; SHL(t2_, h23); AND(t3_, h23); SUB(r0, x0, t0_); SUB(r1, x1, t1_); SHL(T2, H23); AND(T3, H23); SUB(R0, y0, T0); SUB(R1, y1, T1);
            MF; SUB(r2, x2, t2_); SUB(r3, x3, t3_); CVT(m01, r0, r1); CVT(m23, r2, r3); SUB(R2, y2, T2); SUB(R3, y3, T3); CVT(M01, R0, R1); CVT(M23, R2, R3);
            MF; SHL(t0_, m01); AND(t1_, m01); SHL(t2_, m23); AND(t3_, m23); SHL(T0, M01); AND(T1, M01); SHL(T2, M23); AND(T3, M23);
            MF; SUB(r0, r0, t0_); SUB(r1, r1, t1_); SUB(r2, r2, t2_); SUB(r3, r3, t3_); SUB(R0, R0, T0); SUB(R1, R1, T1); SUB(R2, R2, T2); SUB(R3, R3, T3);
            MF; CVT(l01, r0, r1); CVT(l23, r2, r3); CVT(L01, R0, R1); CVT(L23, R2, R3);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c[r];
    s += __builtin_bit_cast(float, l01 ^ l23 ^ L01 ^ L23);
    if (MODE == 4 || MODE >= 6) { asm volatile("s_waitcnt lgkmcnt(0)"); s += __builtin_bit_cast(float, d0.x ^ d1.y ^ d2.z ^ d3.w); }
    if (s == 123.456f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const uint4* in, float* out, unsigned long long* cyc, const char* what) {
    const int iters = 4000;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, in, out, cyc, 200);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-70s %7.1f cycles per step of 6 MFMAs (192 = matrix pace)\n", what, (double)h / iters);
}

int main() {
    uint4* in; float* out; unsigned long long* cyc;
    hipMalloc(&in, 512 * 16); hipMalloc(&out, 4); hipMalloc(&cyc, 8);
    uint32_t h[2048];
    for (int i = 0; i < 2048; ++i) h[i] = (0x3f00u | (rand() & 0xff)) << 16 | (0x3f00u | (rand() & 0xff));
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>(in, out, cyc, "six dependent MFMAs");
    run<1>(in, out, cyc, "+ one split (22 instructions), level order, 4 per gap");
    run<2>(in, out, cyc, "+ one split as a block behind the MFMAs");
    run<3>(in, out, cyc, "+ two splits (44), level order, 8 per gap");
    run<4>(in, out, cyc, "+ one split, 4 per gap, + 4 ds_read_b128 and a wait per step");
    run<5>(in, out, cyc, "+ one split, 4 per gap, next step's B operand from the split");
    run<8>(in, out, cyc, "six MFMAs + one ds_read_b128 behind each of the first four");
    run<6>(in, out, cyc, "+ one split, 4 per gap, one ds_read_b128 in each of the first four gaps");
    run<7>(in, out, cyc, "  same, the next step's MFMAs take the data (wait before each)");
    run<9>(in, out, cyc, "+ one split with v_pk_add_f32 subtractions (18 instructions), <= 4 per gap");
    run<10>(in, out, cyc, "  same + 8 more plain instructions in the two light gaps");
    return 0;
}

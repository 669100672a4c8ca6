// Exact three-way bf16 split of four fp32 values, handed out in parts (bf16x6 kernels that fill the gaps behind their MFMAs
// themselves: attention6.hip, gemm.hip).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

// The split written one instruction per asm statement in LEVEL order (both conversions, the four unpacks, the four subtractions,
// ...), as six parts (4, 4, 4, 4, 4, 2 instructions) with its state, for kernels that hand the parts out one per MFMA gap: in
// that order no instruction sits next to the one it depends on, and the 22 instructions vanish in the shadows of six MFMAs of the
// same wavefront (tools/micro/mfma_split.hip: 200 cycles per step with and without; 304 as a block behind the MFMAs).  The wait
// state a v_cvt_pk_bf16_f32 needs before its result is used two instructions later is inserted by hipcc for these asm statements
// exactly as for its own code (`cvt; cvt; s_nop 0; use` in the ISA): no `s_nop` in the asm text.
#ifndef RLT_SPLIT6_NOP
#define RLT_SPLIT6_NOP ""
#endif
struct Split6 { uint2 hi, mid, lo; uint32_t t0, t1, t2, t3; float r0, r1, r2, r3; };
__device__ __forceinline__ void split6_part(Split6& u, float a, float b, float c, float d, int part) {
#ifdef RLT_SPLIT6_EMPTY        // diagnostic: the results exist for the compiler, no instruction is issued (wrong results)
    if (part == 5) asm volatile("" : "=v"(u.hi.x), "=v"(u.hi.y), "=v"(u.mid.x), "=v"(u.mid.y), "=v"(u.lo.x), "=v"(u.lo.y) : "v"(a), "v"(b), "v"(c), "v"(d));
    return;
#endif
    if (part == 0) {
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u.hi.x) : "v"(a), "v"(b));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" RLT_SPLIT6_NOP : "=v"(u.hi.y) : "v"(c), "v"(d));
        asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u.t0) : "v"(u.hi.x));
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u.t1) : "v"(u.hi.x));
    } else if (part == 1) {
        asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u.t2) : "v"(u.hi.y));
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u.t3) : "v"(u.hi.y));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(u.r0) : "v"(a), "v"(u.t0));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(u.r1) : "v"(b), "v"(u.t1));
    } else if (part == 2) {
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(u.r2) : "v"(c), "v"(u.t2));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(u.r3) : "v"(d), "v"(u.t3));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u.mid.x) : "v"(u.r0), "v"(u.r1));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" RLT_SPLIT6_NOP : "=v"(u.mid.y) : "v"(u.r2), "v"(u.r3));
    } else if (part == 3) {
        asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u.t0) : "v"(u.mid.x));
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u.t1) : "v"(u.mid.x));
        asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u.t2) : "v"(u.mid.y));
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u.t3) : "v"(u.mid.y));
    } else if (part == 4) {
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(u.r0) : "v"(u.r0), "v"(u.t0));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(u.r1) : "v"(u.r1), "v"(u.t1));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(u.r2) : "v"(u.r2), "v"(u.t2));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(u.r3) : "v"(u.r3), "v"(u.t3));
    } else {
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u.lo.x) : "v"(u.r0), "v"(u.r1));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u.lo.y) : "v"(u.r2), "v"(u.r3));
    }
}

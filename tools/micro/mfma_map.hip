// Discover the operand maps of v_mfma_f32_32x32x16_bf16 / 16x16x32 empirically: A one-hot at (lane la, element ja), B[lane][j] = a code;
// prints, for every (la, ja), which output (lane, reg) entries are non-zero and which B (lane, element) they picked up.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_map.hip -o tools/micro/mfma_map
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int SHAPE>
__global__ void probe(int la, int ja, int mode, float* out) {
    const int lane = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)((lane == la && j == ja) ? 1.f : 0.f);
        b[j] = (__bf16)(mode == 0 ? (float)(lane + 1) : (float)(j + 1));
    }
    if (SHAPE == 32) {
        f32x16 c;
        for (int i = 0; i < 16; ++i) c[i] = 0.f;
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
    } else {
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        for (int i = 0; i < 4; ++i) out[lane * 16 + i] = c[i];
    }
}
int main() {
    float *d; hipMalloc(&d, 64 * 16 * 4);
    static float h0[1024], h1[1024];
    for (int shape = 32; shape >= 16; shape -= 16) {
        printf("shape %dx%d\n", shape, shape);
        const int las[] = {0, 1, 5, 17, 33, 37, 63};
        for (int la : las) for (int ja = 0; ja < 8; ++ja) {
            for (int mode = 0; mode < 2; ++mode) {
                hipMemset(d, 0, 4096);
                if (shape == 32) hipLaunchKernelGGL(probe<32>, dim3(1), dim3(64), 0, 0, la, ja, mode, d);
                else hipLaunchKernelGGL(probe<16>, dim3(1), dim3(64), 0, 0, la, ja, mode, d);
                hipMemcpy(mode ? h1 : h0, d, 4096, hipMemcpyDeviceToHost);
            }
            // report the first two non-zero outputs and the count
            int cnt = 0; char buf[256]; int n = 0;
            for (int i = 0; i < 1024; ++i) if (h0[i] != 0.f) { if (cnt < 2) n += snprintf(buf + n, sizeof(buf) - n, " D(lane %d, reg %d) <- B(lane %d, elem %d)", i / 16, i % 16, (int)h0[i] - 1, (int)h1[i] - 1); ++cnt; }
            printf("A(lane %2d, elem %d): %d outputs;%s\n", la, ja, cnt, buf);
        }
    }
    return 0;
}

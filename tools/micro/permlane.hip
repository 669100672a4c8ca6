// What v_permlane16_swap does on gfx950 (the row exchange the head-dim-16 attention kernels rely on): prints, for lane
// 0, 16, 32, 48, the values both results hold.  a = lane, b = 100 + lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; unsigned h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 16) printf("lane %2d: r0 = %3u  r1 = %3u\n", l, h[l], h[64 + l]);
    return 0;
}

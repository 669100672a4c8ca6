// Micro-benchmark: achievable global->register bandwidth for GEMM-like access patterns (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// pattern 0: each WG streams its own contiguous 128 KB panel (1 KB per wave instruction), `reps` panels apart
// pattern 1: same panel, but read as 8 k-tiles of [128 rows x 128 B] at row stride 1 KB (the KC GEMM staging order)
// every panel is read by `share` consecutive WGs (n-tiles sharing an A panel)
template <int PATTERN>
__global__ __launch_bounds__(256) void rd(const float4* __restrict__ buf, size_t npanel, int share, float* sink, int xcd_remap) {
    int bid = blockIdx.x;
    if (xcd_remap) { const int nwg = gridDim.x, q = nwg >> 3; bid = (bid & 7) * q + (bid >> 3); }
    const size_t panel = (size_t)(bid / share) % npanel;
    const float4* p = buf + panel * (128 * 256 / 4);
    const int tid = threadIdx.x;
    float4 acc = make_float4(0, 0, 0, 0);
    if (PATTERN == 0) {
#pragma unroll 8
        for (int i = 0; i < 32; ++i) { float4 v = p[i * 256 + tid]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    } else {
        for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * i, row = idx >> 3, kq = idx & 7;
                float4 v = p[row * 64 + kt * 8 + kq];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

int main() {
    const size_t npanel = 9600;                 // 9600 x 128 KB = 1.26 GB (the A matrix of the FFN GEMM)
    float4* buf; float* sink;
    CK(hipMalloc(&buf, npanel * 128 * 1024)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0, npanel * 128 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int pattern = 0; pattern < 2; ++pattern)
        for (int share : {1, 16})
            for (int remap : {0, 1}) {
                const int grid = (int)(npanel * share);
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(a));
                    if (pattern == 0) hipLaunchKernelGGL(rd<0>, dim3(grid), dim3(256), 0, 0, buf, npanel, share, sink, remap);
                    else hipLaunchKernelGGL(rd<1>, dim3(grid), dim3(256), 0, 0, buf, npanel, share, sink, remap);
                    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                }
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                const double gb = (double)grid * 128 * 1024 / 1e9;
                printf("pattern %d share %2d remap %d: %8.3f ms  %7.2f TB/s (%.1f GB moved)\n", pattern, share, remap, ms, gb / ms, gb);
            }
    return 0;
}

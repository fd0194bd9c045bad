// How fast can ONE XCD (32 CUs) stream from HBM?  Decides whether a decode step confined to one XCD is viable.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void stream_read(const float4* src, size_t n4, int n_xcd, float* sink, unsigned* counters) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xF;
    if ((int)xcc >= n_xcd) return;
    __shared__ unsigned rank_s;
    if (threadIdx.x == 0) rank_s = atomicAdd(counters, 1u);
    __syncthreads();
    const size_t members = (size_t)n_xcd * 32 * 8;      // launched: 8 workgroups per CU slot
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)rank_s * 256 + threadIdx.x; i < n4; i += members * 256) {
        const float4 v = src[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == -1.f) sink[0] = acc.x;
}

int main() {
    const size_t bytes = 1ull << 30;     // 1 GiB: beyond the 256 MiB Infinity Cache
    float4* src; float* sink; unsigned* counters;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&counters, 4));
    CK(hipMemset(src, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int n_xcd : {1, 2, 4, 8}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(counters, 0, 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(stream_read, dim3(256 * 8), dim3(256), 0, 0, src, bytes / 16, n_xcd, sink, counters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%d XCD(s) reading 1 GiB from HBM: %.3f ms = %.2f TB/s\n", n_xcd, best, bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}

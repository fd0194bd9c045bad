// Row LayerNorm in fp32 (whisper/model.py:39-41; eps = 1e-5, biased variance), one wave per row.
// d <= 1280 for every Whisper size, so a row lives in <= 20 registers per lane: one HBM read,
// one write, two wavefront reductions.
#include "common.h"

namespace wlk {

constexpr int kLnMaxPerLane = 24;  // supports d <= 1536

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        long ldy, int rows, int d, PtrTable z, int batch) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (batch > 0) {
        x = table_at(z.in, blockIdx.y);
        y = table_at(z.out, blockIdx.y);
    }
    const float* xr = x + (long)row * ldx;
    float v[kLnMaxPerLane];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kLnMaxPerLane; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < d ? xr[c] : 0.f;
        sum += v[i];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float mean = sum / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < kLnMaxPerLane; ++i) {
        const int c = lane + 64 * i;
        const float t = c < d ? v[i] - mean : 0.f;
        sq += t * t;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
    const float rstd = 1.0f / sqrtf(sq / (float)d + 1e-5f);
    float* yr = y + (long)row * ldy;
#pragma unroll
    for (int i = 0; i < kLnMaxPerLane; ++i) {
        const int c = lane + 64 * i;
        if (c < d) yr[c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
    }
}

void launch_layernorm(const LaunchCtx& ctx, const float* x, long ldx, const float* gamma, const float* beta,
                      float* y, long ldy, int rows, int d, const char* tag) {
    if (rows <= 0) return;
    if (d > 64 * kLnMaxPerLane) throw std::invalid_argument("layernorm: d too large");
    KernelScope ks(ctx, tag);
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, ctx.stream, x, ldx, gamma, beta, y,
                       ldy, rows, d, PtrTable{}, 0);
    WLK_HIP(hipGetLastError());
}

void launch_layernorm_batched(const LaunchCtx& ctx, const PtrTable& z, int batch, long ldx, const float* gamma,
                              const float* beta, long ldy, int rows, int d, const char* tag) {
    if (rows <= 0 || batch <= 0) return;
    if (d > 64 * kLnMaxPerLane || batch > kMaxBatch) throw std::invalid_argument("layernorm: d or batch too large");
    KernelScope ks(ctx, tag);
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4, batch), dim3(256), 0, ctx.stream, (const float*)nullptr, ldx,
                       gamma, beta, (float*)nullptr, ldy, rows, d, z, batch);
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

// Row LayerNorm in fp32 (whisper/model.py:39-41; eps = 1e-5, biased variance), one wave per row.
// d <= 1280 for every Whisper size, so a row lives in <= 20 registers per lane: one HBM read,
// one write, two wavefront reductions.
#include "common.h"
#include "wave_ops.h"
#include "x3.h"

namespace wlk {

constexpr int kLnMaxPerLane = 24;  // supports d <= 1536

// NPL = elements per lane (>= ceil(d / 64)).  Round 4: every load of the kernel - the row AND gamma / beta - is requested
// before the first wait (the affine used to be fetched behind both reductions: a second memory round trip per launch, 13
// launches per encode), addresses are clamped instead of predicated (hipcc turns `c < d ? x[c] : 0` into one branch per
// load), and the two reductions are VALU butterflies (wave_ops.h).  Sums run in the same order over the same values: a
// lane's elements past d contributed exact zeros before and are simply not there now.
template <int NPL>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        long ldy, int rows, int d, PtrTable z, int batch) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (x == nullptr) {          // batched form (x is a preloaded argument; `batch` would be one more scalar round trip)
        x = table_at(z.in, blockIdx.y);
        y = table_at(z.out, blockIdx.y);
    }
    const float* xr = x + (long)row * ldx;
    float v[NPL], g[NPL], b[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int c = lane + 64 * i;
        const int cc = c < d ? c : 0;
        v[i] = xr[cc];
        g[i] = gamma[cc];
        b[i] = beta[cc];
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) { pin_loaded(g[i]); pin_loaded(b[i]); }   // see layernorm_x3_kernel
    __builtin_amdgcn_sched_barrier(0);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        v[i] = (lane + 64 * i) < d ? v[i] : 0.f;
        sum += v[i];
    }
    sum = wave_sum(sum);
    const float mean = sum / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int c = lane + 64 * i;
        const float t = c < d ? v[i] - mean : 0.f;
        sq += t * t;
    }
    sq = wave_sum(sq);
    const float rstd = 1.0f / sqrtf(sq / (float)d + 1e-5f);
    float* yr = y + (long)row * ldy;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int c = lane + 64 * i;
        if (c < d) yr[c] = (v[i] - mean) * rstd * g[i] + b[i];
    }
}

// LayerNorm whose result feeds launch_gemm_x3: the normalised row is written in the X3 format (x3.h).  The statistics
// are layernorm_kernel's (same lane-strided loads, same order, same butterflies: mean and rstd are bit-identical to it);
// the output pass walks the row in 8-element chunks - lane l owns chunks l, l + 64, ... - so that a chunk's three
// 16-byte units are one 48-byte store and a wave writes 3 KiB contiguously.  NPL as above; NCH = chunks per lane.
template <int NPL>
__global__ __launch_bounds__(256) void layernorm_x3_kernel(const float* __restrict__ x, long ldx,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, unsigned short* __restrict__ y3,
                                                           long ldy3, int rows, int d, PtrTable z) {
    constexpr int NCH = (NPL + 7) / 8;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (x == nullptr) {
        x = table_at(z.in, blockIdx.y);
        y3 = reinterpret_cast<unsigned short*>(table_at(z.out, blockIdx.y));
    }
    const float* xr = x + (long)row * ldx;
    const int n_chunks = d >> 3;
    float v[NPL];
    float4 xc[NCH][2], gc[NCH][2], bc[NCH][2];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int c = lane + 64 * i;
        v[i] = xr[c < d ? c : 0];
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int q = lane + 64 * j;
        const int at = (q < n_chunks ? q : 0) * 8;
        xc[j][0] = *reinterpret_cast<const float4*>(xr + at);
        xc[j][1] = *reinterpret_cast<const float4*>(xr + at + 4);
        gc[j][0] = *reinterpret_cast<const float4*>(gamma + at);
        gc[j][1] = *reinterpret_cast<const float4*>(gamma + at + 4);
        bc[j][0] = *reinterpret_cast<const float4*>(beta + at);
        bc[j][1] = *reinterpret_cast<const float4*>(beta + at + 4);
    }
    // the ISA without this: v[0], v[1] -> wait -> the other loads -> statistics -> gamma / beta chunks inside the store's
    // branch: three dependent memory round trips for one
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        pin_loaded(xc[j][0]); pin_loaded(xc[j][1]);
        pin_loaded(gc[j][0]); pin_loaded(gc[j][1]);
        pin_loaded(bc[j][0]); pin_loaded(bc[j][1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        v[i] = (lane + 64 * i) < d ? v[i] : 0.f;
        sum += v[i];
    }
    sum = wave_sum(sum);
    const float mean = sum / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const float t = (lane + 64 * i) < d ? v[i] - mean : 0.f;
        sq += t * t;
    }
    sq = wave_sum(sq);
    const float rstd = 1.0f / sqrtf(sq / (float)d + 1e-5f);
    unsigned short* yr = y3 + (long)row * 3 * ldy3;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int q = lane + 64 * j;
        // computed by every lane (a lane past the row holds chunk 0's values), only the store is conditional: operands that
        // are used inside a branch only get their LOADS sunk into it by hipcc - a second memory round trip behind the
        // statistics
        const float o[8] = {(xc[j][0].x - mean) * rstd * gc[j][0].x + bc[j][0].x, (xc[j][0].y - mean) * rstd * gc[j][0].y + bc[j][0].y,
                            (xc[j][0].z - mean) * rstd * gc[j][0].z + bc[j][0].z, (xc[j][0].w - mean) * rstd * gc[j][0].w + bc[j][0].w,
                            (xc[j][1].x - mean) * rstd * gc[j][1].x + bc[j][1].x, (xc[j][1].y - mean) * rstd * gc[j][1].y + bc[j][1].y,
                            (xc[j][1].z - mean) * rstd * gc[j][1].z + bc[j][1].z, (xc[j][1].w - mean) * rstd * gc[j][1].w + bc[j][1].w};
        if (q < n_chunks) x3_store_chunk(yr + (long)q * 24, o);
    }
}

template <typename... Args>
static void launch_ln_x3_by_width(int d, dim3 grid, hipStream_t stream, Args... args) {
    const int npl = (d + 63) / 64;
    if (npl <= 8) hipLaunchKernelGGL(layernorm_x3_kernel<8>, grid, dim3(256), 0, stream, args...);
    else if (npl <= 12) hipLaunchKernelGGL(layernorm_x3_kernel<12>, grid, dim3(256), 0, stream, args...);
    else if (npl <= 16) hipLaunchKernelGGL(layernorm_x3_kernel<16>, grid, dim3(256), 0, stream, args...);
    else if (npl <= 20) hipLaunchKernelGGL(layernorm_x3_kernel<20>, grid, dim3(256), 0, stream, args...);
    else hipLaunchKernelGGL(layernorm_x3_kernel<kLnMaxPerLane>, grid, dim3(256), 0, stream, args...);
}

template <typename... Args>
static void launch_ln_by_width(int d, dim3 grid, hipStream_t stream, Args... args) {
    const int npl = (d + 63) / 64;
    if (npl <= 8) hipLaunchKernelGGL(layernorm_kernel<8>, grid, dim3(256), 0, stream, args...);
    else if (npl <= 12) hipLaunchKernelGGL(layernorm_kernel<12>, grid, dim3(256), 0, stream, args...);
    else if (npl <= 16) hipLaunchKernelGGL(layernorm_kernel<16>, grid, dim3(256), 0, stream, args...);
    else if (npl <= 20) hipLaunchKernelGGL(layernorm_kernel<20>, grid, dim3(256), 0, stream, args...);
    else hipLaunchKernelGGL(layernorm_kernel<kLnMaxPerLane>, grid, dim3(256), 0, stream, args...);
}

void launch_layernorm(const LaunchCtx& ctx, const float* x, long ldx, const float* gamma, const float* beta,
                      float* y, long ldy, int rows, int d, const char* tag) {
    if (rows <= 0) return;
    if (d > 64 * kLnMaxPerLane) throw std::invalid_argument("layernorm: d too large");
    KernelScope ks(ctx, tag);
    launch_ln_by_width(d, dim3((rows + 3) / 4), ctx.stream, x, ldx, gamma, beta, y, ldy, rows, d, PtrTable{}, 0);
    WLK_HIP(hipGetLastError());
}

void launch_layernorm_batched(const LaunchCtx& ctx, const PtrTable& z, int batch, long ldx, const float* gamma,
                              const float* beta, long ldy, int rows, int d, const char* tag) {
    if (rows <= 0 || batch <= 0) return;
    if (d > 64 * kLnMaxPerLane || batch > kMaxBatch) throw std::invalid_argument("layernorm: d or batch too large");
    KernelScope ks(ctx, tag);
    launch_ln_by_width(d, dim3((rows + 3) / 4, batch), ctx.stream, (const float*)nullptr, ldx, gamma, beta, (float*)nullptr,
                       ldy, rows, d, z, batch);
    WLK_HIP(hipGetLastError());
}


void launch_layernorm_x3(const LaunchCtx& ctx, const float* x, long ldx, const float* gamma, const float* beta,
                         unsigned short* y3, long ldy3, int rows, int d, const char* tag) {
    if (rows <= 0) return;
    if (d > 64 * kLnMaxPerLane || d % 8 != 0 || ldx % 4 != 0 || ldy3 % 8 != 0) throw std::invalid_argument("layernorm (x3): unsupported width");
    KernelScope ks(ctx, tag);
    launch_ln_x3_by_width(d, dim3((rows + 3) / 4), ctx.stream, x, ldx, gamma, beta, y3, ldy3, rows, d, PtrTable{});
    WLK_HIP(hipGetLastError());
}

void launch_layernorm_x3_batched(const LaunchCtx& ctx, const PtrTable& z, int batch, long ldx, const float* gamma,
                                 const float* beta, long ldy3, int rows, int d, const char* tag) {
    if (rows <= 0 || batch <= 0) return;
    if (d > 64 * kLnMaxPerLane || d % 8 != 0 || ldx % 4 != 0 || ldy3 % 8 != 0 || batch > kMaxBatch)
        throw std::invalid_argument("layernorm (x3): unsupported width or batch");
    KernelScope ks(ctx, tag);
    launch_ln_x3_by_width(d, dim3((rows + 3) / 4, batch), ctx.stream, (const float*)nullptr, ldx, gamma, beta,
                          (unsigned short*)nullptr, ldy3, rows, d, z);
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

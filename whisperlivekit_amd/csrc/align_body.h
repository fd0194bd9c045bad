// The z-score half of the AlignAtt read-out (select.hip) as a device function two translation units share: select.hip's
// own kernels, and the vocabulary GEMV of a graph-replayed decode step (gemm_f32.hip: gemv_f32_kernel's side workgroups),
// where it runs beside the weight stream instead of in front of the step's last two launches.  ONE body: a row of z is
// the same bits whichever launch carried it.
#pragma once
#include "common.h"
#include "wave_ops.h"

namespace wlk {

__device__ __forceinline__ void align_zscore_body(AlignArgs a, int fblock, int al, int b) {
    // 64 frame columns x 4 row groups per workgroup: the window rows (up to 448 + 15) are walked by
    // four threads per column in parallel and folded through LDS - the loop is latency-bound, so
    // parallel rows matter more than anything else here
    __shared__ double red[4][64];
    const int fx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int f = fblock * 64 + fx;
    const bool ok = f < a.T;
    if (a.rows) {                // batched steps: row b is a session of its own (one beam)
        const StepRow sr = a.rows[b];
        a.prefill_rows = sr.prefill_rows;
        a.n_single = sr.n_single;
        a.newest_row = sr.newest_row;
    }
    const gcf_ptr base = to_global(a.rows ? a.rows[b].ring + ((long)al * a.ring_rows) * a.T + (ok ? f : 0)
                                          : a.ring + ((long)(al * a.n_beam + b) * a.ring_rows) * a.T + (ok ? f : 0));
    const int n = a.prefill_rows + a.n_single;
    auto row_of = [&](int i) { return i < a.prefill_rows ? i : a.single_base + (i - a.prefill_rows); };
    // Round 4: the thread's first kZKeep window rows (windows of up to 4 kZKeep = 96 rows: every step but those of very
    // long prompts) are requested together and kept for the second pass - one memory round trip instead of
    // 2 x ceil(rows / 16) dependent ones; the sums run over the same values in the same order.
    constexpr int kZKeep = 24;
    float w[kZKeep];
#pragma unroll
    for (int t = 0; t < kZKeep; ++t) {
        const int i = rg + 4 * t;
        w[t] = base[(long)row_of(i < n ? i : 0) * a.T];
    }
    const float newest = base[(long)a.newest_row * a.T];
    __builtin_amdgcn_sched_barrier(0);   // all of them in flight before the first is folded (hipcc interleaves otherwise)
    double sum = 0.0;
#pragma unroll
    for (int t = 0; t < kZKeep; ++t)
        if (rg + 4 * t < n) sum += (double)w[t];
#pragma unroll 4
    for (int i = rg + 4 * kZKeep; i < n; i += 4) sum += (double)base[(long)row_of(i) * a.T];
    red[rg][fx] = sum;
    __syncthreads();
    const double mean = (red[0][fx] + red[1][fx] + red[2][fx] + red[3][fx]) / n;
    __syncthreads();
    double sq = 0.0;
#pragma unroll
    for (int t = 0; t < kZKeep; ++t) {
        if (rg + 4 * t < n) {
            const double d = (double)w[t] - mean;
            sq += d * d;
        }
    }
#pragma unroll 4
    for (int i = rg + 4 * kZKeep; i < n; i += 4) {
        const double t = (double)base[(long)row_of(i) * a.T] - mean;
        sq += t * t;
    }
    red[rg][fx] = sq;
    __syncthreads();
    if (rg == 0 && ok) {
        const float stdv = (float)sqrt((red[0][fx] + red[1][fx] + red[2][fx] + red[3][fx]) / n);
        a.z[((long)b * a.n_align + al) * a.T + f] = (newest - (float)mean) / (stdv + 1e-8f);
    }
}

}  // namespace wlk

// Dynamic time warping over a [tokens, frames] cost matrix: the alignment step of the reference's word timestamps
// (whisper/timing.py:82-105 `dtw_cpu`; the reference's CUDA build runs the same recurrence as a Triton kernel,
// timing.py:108-138).  SURVEY 8(f) rank 4: the one kernel LocalAgreement's batch Whisper needs beyond the encoder /
// decoder kernels of the streaming path.
//
//   cost[i][j] = x[i-1][j-1] + min(cost[i-1][j-1], cost[i-1][j], cost[i][j-1]),   cost[0][0] = 0, borders = inf
//   trace[i][j] = 0 (diagonal) if c0 < c1 and c0 < c2, else 1 (up) if c1 < c0 and c1 < c2, else 2 (left)
//
// exactly dtw_cpu's strict comparisons (ties go left), one fp32 add per cell.  The recurrence is a wavefront: all cells
// of an anti-diagonal i + j = k are independent.  One workgroup, one thread per token row (N <= 1024, the text context
// is 448), N + M - 1 steps with a workgroup barrier each; the three live anti-diagonals sit in LDS, a row's x values
// are read one step ahead, the trace goes out frame-major so that a step's writes are contiguous.  Latency-bound by
// construction (~2000 dependent steps for a 30 s window): what matters is that nothing but the barrier is on the
// step's critical path.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/wlk_hip.h"
#include "common.h"
#include "internal.h"

namespace wlk {

constexpr int kDtwMaxRows = 1024;

__global__ __launch_bounds__(kDtwMaxRows) void dtw_wavefront_kernel(const float* __restrict__ x, int N, int M,
                                                                    signed char* __restrict__ trace_t /* [M+1][N+1] */) {
    __shared__ float diag[3][kDtwMaxRows + 1];   // cost of row i on the anti-diagonals k-2, k-1, k
    const int i = threadIdx.x + 1;               // token row 1..N of the padded recurrence
    for (int r = threadIdx.x; r < 3 * (kDtwMaxRows + 1); r += blockDim.x) (&diag[0][0])[r] = INFINITY;
    __syncthreads();
    if (threadIdx.x == 0) diag[0][0] = 0.f;      // cost[0][0]; diag[1] = anti-diagonal 1: cost[0][1] = cost[1][0] = inf
    __syncthreads();
    int b2 = 0, b1 = 1, b0 = 2;                  // buffers of k-2, k-1, k
    const bool mine = i <= N;
    const float* xr = x + (long)(mine ? i - 1 : 0) * M;
    float x_next = mine && M > 0 ? xr[0] : 0.f;  // x[i-1][j-1] of this row's next cell
    for (int k = 2; k <= N + M; ++k) {
        const int j = k - i;
        if (mine && j >= 1 && j <= M) {
            const float xv = x_next;
            if (j < M) x_next = xr[j];
            const float c0 = diag[b2][i - 1], c1 = diag[b1][i - 1], c2 = diag[b1][i];
            float c;
            signed char t;
            if (c0 < c1 && c0 < c2) { c = c0; t = 0; }
            else if (c1 < c0 && c1 < c2) { c = c1; t = 1; }
            else { c = c2; t = 2; }
            diag[b0][i] = xv + c;
            trace_t[(long)j * (N + 1) + i] = t;
        }
        if (threadIdx.x == 0) diag[b0][0] = INFINITY;   // cost[0][k]
        __syncthreads();
        const int t = b2; b2 = b1; b1 = b0; b0 = t;
    }
}

}  // namespace wlk

using namespace wlk;

extern "C" int wlk_dtw(int device, const float* x, int32_t n_rows, int32_t n_cols, int8_t* trace) {
    if (!x || !trace) return fail(WLK_ERR_ARG, "dtw: NULL argument");
    if (n_rows < 1 || n_cols < 1) return fail(WLK_ERR_ARG, "dtw: empty cost matrix");
    if (n_rows > kDtwMaxRows) return fail(WLK_ERR_CAPACITY, "dtw: more than 1024 rows");
    if ((long)n_rows * n_cols > (1L << 28)) return fail(WLK_ERR_CAPACITY, "dtw: cost matrix too large");
    return guarded([&]() {
        WLK_HIP(hipSetDevice(device));
        const size_t nx = (size_t)n_rows * n_cols, nt = (size_t)(n_rows + 1) * (n_cols + 1);
        float* xd = nullptr;
        signed char* td = nullptr;
        struct Free {
            float*& a;
            signed char*& b;
            ~Free() {
                if (a) (void)hipFree(a);
                if (b) (void)hipFree(b);
            }
        } guard{xd, td};
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&xd), nx * sizeof(float)));
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&td), nt));
        WLK_HIP(hipMemcpy(xd, x, nx * sizeof(float), hipMemcpyHostToDevice));
        WLK_HIP(hipMemset(td, 0xff, nt));   // -1 like dtw_cpu's untouched border cells
        const int threads = std::max(64, ((n_rows + 63) / 64) * 64);
        hipLaunchKernelGGL(dtw_wavefront_kernel, dim3(1), dim3(threads), 0, nullptr, xd, n_rows, n_cols, td);
        WLK_HIP(hipGetLastError());
        std::vector<signed char> tt(nt);
        WLK_HIP(hipMemcpy(tt.data(), td, nt, hipMemcpyDeviceToHost));
        // frame-major on the device, token-major for the caller: trace[i][j], (n_rows + 1) x (n_cols + 1)
        for (int i = 0; i <= n_rows; ++i)
            for (int j = 0; j <= n_cols; ++j) trace[(size_t)i * (n_cols + 1) + j] = tt[(size_t)j * (n_rows + 1) + i];
        return WLK_OK;
    });
}

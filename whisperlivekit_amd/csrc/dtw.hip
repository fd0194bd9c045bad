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
#include <algorithm>
#include <cstring>
#include <mutex>

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

// frame-major [M+1][N+1] -> token-major [N+1][M+1] (the layout dtw_cpu hands to backtrace), 32x32 tiles through LDS
__global__ __launch_bounds__(256) void dtw_transpose_kernel(const signed char* __restrict__ in, signed char* __restrict__ out,
                                                            int rows_in /* M+1 */, int cols_in /* N+1 */) {
    __shared__ signed char tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int r = ty; r < 32; r += 8)
        if (r0 + r < rows_in && c0 + tx < cols_in) tile[r][tx] = in[(long)(r0 + r) * cols_in + c0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (c0 + r < cols_in && r0 + tx < rows_in) out[(long)(c0 + r) * rows_in + r0 + tx] = tile[tx][r];
}

// Per-device workspace of wlk_dtw: device buffers and pinned staging that only ever grow, and a non-blocking stream of
// their own - a word-timestamp alignment neither allocates (hipFree synchronises the whole device, i.e. every
// streaming session's encode and decode streams) nor touches the NULL stream.  One alignment at a time per device.
struct DtwWorkspace {
    std::mutex mu;
    hipStream_t stream = nullptr;
    float* xd = nullptr;
    signed char *td = nullptr, *tt = nullptr;
    char* pinned = nullptr;
    size_t x_cap = 0, t_cap = 0, pin_cap = 0;
    void reserve(size_t nx, size_t nt) {
        if (!stream) WLK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        if (nx > x_cap) {
            if (xd) WLK_HIP(hipFree(xd));
            xd = nullptr; x_cap = 0;
            WLK_HIP(hipMalloc(reinterpret_cast<void**>(&xd), nx * sizeof(float)));
            x_cap = nx;
        }
        if (nt > t_cap) {
            if (td) WLK_HIP(hipFree(td));
            if (tt) WLK_HIP(hipFree(tt));
            td = tt = nullptr; t_cap = 0;
            WLK_HIP(hipMalloc(reinterpret_cast<void**>(&td), nt));
            WLK_HIP(hipMalloc(reinterpret_cast<void**>(&tt), nt));
            t_cap = nt;
        }
        const size_t pin = std::max(nx * sizeof(float), nt);
        if (pin > pin_cap) {
            if (pinned) WLK_HIP(hipHostFree(pinned));
            pinned = nullptr; pin_cap = 0;
            WLK_HIP(hipHostMalloc(reinterpret_cast<void**>(&pinned), pin, hipHostMallocDefault));
            pin_cap = pin;
        }
    }
};
constexpr int kDtwMaxDevices = 64;
static DtwWorkspace g_dtw_ws[kDtwMaxDevices];

// the recurrence + transposition on `stream`: x_dev [N][M] -> trace_dev token-major [(N+1)][(M+1)]; scratch_t [(M+1)(N+1)]
void launch_dtw(hipStream_t stream, const float* x_dev, int n_rows, int n_cols, signed char* scratch_t, signed char* trace_dev) {
    const size_t nt = (size_t)(n_rows + 1) * (n_cols + 1);
    WLK_HIP(hipMemsetAsync(scratch_t, 0xff, nt, stream));   // -1 like dtw_cpu's untouched border cells
    const int threads = std::max(64, ((n_rows + 63) / 64) * 64);
    hipLaunchKernelGGL(dtw_wavefront_kernel, dim3(1), dim3(threads), 0, stream, x_dev, n_rows, n_cols, scratch_t);
    hipLaunchKernelGGL(dtw_transpose_kernel, dim3((n_rows + 1 + 31) / 32, (n_cols + 1 + 31) / 32), dim3(256), 0, stream,
                       scratch_t, trace_dev, n_cols + 1, n_rows + 1);
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

using namespace wlk;

extern "C" int wlk_dtw(int device, const float* x, int32_t n_rows, int32_t n_cols, int8_t* trace) {
    if (!x || !trace) return fail(WLK_ERR_ARG, "dtw: NULL argument");
    if (n_rows < 1 || n_cols < 1) return fail(WLK_ERR_ARG, "dtw: empty cost matrix");
    if (n_rows > kDtwMaxRows) return fail(WLK_ERR_CAPACITY, "dtw: more than 1024 rows");
    if ((long)n_rows * n_cols > (1L << 28)) return fail(WLK_ERR_CAPACITY, "dtw: cost matrix too large");
    if (device < 0 || device >= kDtwMaxDevices) return fail(WLK_ERR_ARG, "dtw: device out of range");
    return guarded([&]() {
        WLK_HIP(hipSetDevice(device));
        const size_t nx = (size_t)n_rows * n_cols, nt = (size_t)(n_rows + 1) * (n_cols + 1);
        DtwWorkspace& ws = g_dtw_ws[device];
        std::lock_guard<std::mutex> lk(ws.mu);
        ws.reserve(nx, nt);
        std::memcpy(ws.pinned, x, nx * sizeof(float));
        WLK_HIP(hipMemcpyAsync(ws.xd, ws.pinned, nx * sizeof(float), hipMemcpyHostToDevice, ws.stream));
        launch_dtw(ws.stream, ws.xd, n_rows, n_cols, ws.td, ws.tt);
        // (the upload has been consumed by the time the kernels are done: the pinned block is reused for the way back)
        WLK_HIP(hipMemcpyAsync(ws.pinned, ws.tt, nt, hipMemcpyDeviceToHost, ws.stream));
        WLK_HIP(hipStreamSynchronize(ws.stream));
        std::memcpy(trace, ws.pinned, nt);
        return WLK_OK;
    });
}

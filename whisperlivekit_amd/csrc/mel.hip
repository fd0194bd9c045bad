// Log-mel front end of the streaming path (whisper/audio.py:110-157 as called from
// simul_whisper/simul_whisper.py:344-350 with padding = 30 s of zeros).
//
//   frame t covers padded samples [160 t - 200, 160 t + 200)  (torch.stft center=True, reflect)
//   X[k]   = sum_n x[n] w[n] e^{-2 pi i k n / 400},  k = 0..200          (periodic hann w)
//   mel[m] = sum_k F[m][k] |X[k]|^2 ;  L = log10(max(mel, 1e-10))
//   out    = (max(L, max(L over ALL frames) - 8) + 4) / 4, first 3000 frames
//
// Per-frame results before the normalisation (log10 of the clamped mel energies, the frame maximum) stay in the
// session between calls: a frame whose 400-sample window saw no new sample is bit for bit what it was, so a call
// computes only the frames the appended chunk touches (+ the two reflect-padded head frames after an eviction);
// the global maximum and the normalised output are redone every call by the finishing kernel.
//
// Only frames that can see a non-zero sample (t < ceil((N + 200) / 160)) are computed; every
// other frame of the 30 s zero pad is exactly log10(1e-10) = -10 and is filled by the finishing
// kernel.  The audio never leaves HBM: the session keeps the rolling buffer resident and this
// kernel gathers straight from it (coalesced 400-sample reads, reflection resolved in the index).
//
// The DFT is a direct 201 x 400 matrix-vector product per frame with an fp64 twiddle table held
// in LDS and fp64 accumulation: 0.16 MFLOP per frame, far below any roofline, and more accurate
// than the fp32 FFT it is compared against.  Output is written TIME-major with one zero row on
// each side, which is exactly the im2col-free A operand of the conv1 GEMM (gemm_f32.hip).
#include "common.h"

namespace wlk {

__global__ __launch_bounds__(256) void mel_frame_kernel(MelArgs a) {
    __shared__ double tw[kNFft];
    __shared__ float xw[kNFft];
    __shared__ float power[kNFreq + 3];
    __shared__ float red[256];

    const int t = (int)blockIdx.x < a.head ? (int)blockIdx.x : a.first + ((int)blockIdx.x - a.head);
    const int tid = threadIdx.x;
    for (int n = tid; n < kNFft; n += 256) {
        tw[n] = a.twiddle[n];
        int j = kHop * t - kNFft / 2 + n;
        if (j < 0) j = -j;  // reflect (edge sample not repeated)
        if (j >= a.n_padded) j = 2 * (a.n_padded - 1) - j;
        const float s = j < a.n_samples ? a.audio[j] : 0.f;
        xw[n] = s * a.window[n];
    }
    __syncthreads();

    if (tid < kNFreq) {
        double re = 0.0, im = 0.0;
        int idx = 0;  // (tid * n) mod 400
        // the two fma chains are the only loop-carried dependency: unrolled, the LDS reads of eight steps are in
        // flight together instead of one round trip per step (same fma order)
#pragma unroll 8
        for (int n = 0; n < kNFft; ++n) {
            const double x = (double)xw[n];
            int sidx = idx + 300;  // sin(theta) = cos(theta - pi/2) = tw[(idx - 100) mod 400]
            if (sidx >= kNFft) sidx -= kNFft;
            re = fma(x, tw[idx], re);
            im = fma(-x, tw[sidx], im);
            idx += tid;
            if (idx >= kNFft) idx -= kNFft;
        }
        const float mag = hypotf((float)re, (float)im);  // stft(...).abs()
        power[tid] = mag * mag;                          // ** 2
    }
    __syncthreads();

    float lm = -INFINITY;
    if (tid < a.n_mels) {
        const float* f = a.filters + (long)tid * kNFreq;
        float acc = 0.f;
        const int lo = a.filt_lo[tid], hi = a.filt_hi[tid];
        for (int k = lo; k < hi; ++k) acc = fmaf(f[k], power[k], acc);
        lm = log10f(fmaxf(acc, 1e-10f));
        a.logmel[(long)t * a.n_mels + tid] = lm;
    }
    red[tid] = lm;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    if (tid == 0) a.frame_max[t] = red[0];
}

__global__ __launch_bounds__(256) void mel_finish_kernel(MelArgs a) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    float m = a.n_total > a.n_active ? -10.0f : -INFINITY;  // silent frames are exactly -10
    for (int i = tid; i < a.n_active; i += 256) m = fmaxf(m, a.frame_max[i]);
    red[tid] = m;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    const float floor_v = red[0] - 8.0f;
    const long total = (long)kMelFrames * a.n_mels;
    for (long i = (long)blockIdx.x * 256 + tid; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i / a.n_mels);
        const float v = t < a.n_active ? a.logmel[i] : -10.0f;
        a.mel_t[i + a.n_mels] = (fmaxf(v, floor_v) + 4.0f) / 4.0f;  // +n_mels: row 0 is conv padding
    }
}

void launch_mel(const LaunchCtx& ctx, const MelArgs& a) {
    const int n_frames = a.head + (a.n_active - a.first);
    if (n_frames > 0) {
        KernelScope ks(ctx, "mel_frames");
        hipLaunchKernelGGL(mel_frame_kernel, dim3(n_frames), dim3(256), 0, ctx.stream, a);
        WLK_HIP(hipGetLastError());
    }
    {
        KernelScope ks(ctx, "mel_finish");
        hipLaunchKernelGGL(mel_finish_kernel, dim3(240), dim3(256), 0, ctx.stream, a);
        WLK_HIP(hipGetLastError());
    }
}

// ---- whole-file form ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mel_full_max_kernel(const float* __restrict__ frame_max, int n_active, int n_total,
                                                           float* __restrict__ out) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    float m = n_total > n_active ? -10.0f : -INFINITY;  // frames that only see padding are exactly log10(1e-10)
    for (int i = tid; i < n_active; i += 256) m = fmaxf(m, frame_max[i]);
    red[tid] = m;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    if (tid == 0) out[0] = red[0];
}

// 64 frames per workgroup: time-major rows in (coalesced), mel-major rows out (64-frame runs)
__global__ __launch_bounds__(256) void mel_full_finish_kernel(const float* __restrict__ logmel, const float* __restrict__ gmax,
                                                              int n_mels, int n_active, int n_total, float* __restrict__ out) {
    __shared__ float tile[64 * 129];
    const int t0 = blockIdx.x * 64, tid = threadIdx.x;
    const float floor_v = gmax[0] - 8.0f;
    const int ld = n_mels + 1;
    for (int i = tid; i < 64 * n_mels; i += 256) {
        const int r = i / n_mels, c = i - r * n_mels, t = t0 + r;
        tile[r * ld + c] = t < n_active ? logmel[(long)t * n_mels + c] : -10.0f;
    }
    __syncthreads();
    for (int i = tid; i < 64 * n_mels; i += 256) {
        const int c = i >> 6, r = i & 63, t = t0 + r;
        if (t < n_total) out[(long)c * n_total + t] = (fmaxf(tile[r * ld + c], floor_v) + 4.0f) / 4.0f;
    }
}

void launch_mel_full(const LaunchCtx& ctx, const MelArgs& a, float* max_scratch, float* out) {
    if (a.n_mels > 128) throw std::invalid_argument("log-mel: more than 128 mel bins");
    if (a.n_active > 0) {
        KernelScope ks(ctx, "mel_frames");
        hipLaunchKernelGGL(mel_frame_kernel, dim3(a.n_active), dim3(256), 0, ctx.stream, a);
        WLK_HIP(hipGetLastError());
    }
    if (a.n_total > 0) {
        KernelScope ks(ctx, "mel_finish");
        hipLaunchKernelGGL(mel_full_max_kernel, dim3(1), dim3(256), 0, ctx.stream, a.frame_max, a.n_active, a.n_total, max_scratch);
        hipLaunchKernelGGL(mel_full_finish_kernel, dim3((a.n_total + 63) / 64), dim3(256), 0, ctx.stream, a.logmel, max_scratch,
                           a.n_mels, a.n_active, a.n_total, out);
        WLK_HIP(hipGetLastError());
    }
}

// int16 PCM straight off the wire -> the fp32 sample buffer: np.frombuffer(int16).astype(float32) / 32768.0
// (convert_pcm_to_float, whisperlivekit/audio_processor.py:416-418); the division is an exact power-of-two scale.
__global__ void pcm16_to_float_kernel(const short* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i] * (1.0f / 32768.0f);
}

void launch_pcm16_to_float(const LaunchCtx& ctx, const short* in, float* out, int n) {
    if (n <= 0) return;
    KernelScope ks(ctx, "pcm16_to_float", 0.0, 6.0 * n);
    hipLaunchKernelGGL(pcm16_to_float_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx.stream, in, out, n);
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

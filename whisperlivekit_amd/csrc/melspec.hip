// Generic log-mel feature extractor for the diarization front end.
//
// The reference computes these features with NeMo's AudioToMelSpectrogramPreprocessor
// (whisperlivekit/diarization/sortformer_backend.py:181-188, 273-275: window 25 ms, stride 10 ms,
// n_fft 512, 128 mel bins, normalize "NA"); NeMo is a third-party dependency that is not in the
// reference tree (nemo-toolkit[asr] >=3,<4, pyproject.toml:80-83), so this kernel follows the
// published FilterbankFeatures algorithm: pre-emphasis y[n] = x[n] - 0.97 x[n-1], centred STFT with
// zero padding and a symmetric hann window of win_length samples centred in the n_fft frame,
// power spectrum, slaney mel filterbank, log(mel + 2^-24).  Parity against NeMo itself is UNPINNED
// (no NeMo here); the kernel is checked against oracle/sortformer_oracle.py.
//
// Same structure as mel.hip: one workgroup per frame, direct DFT with an fp64 twiddle table in LDS
// (n_fft <= 512), sparse filterbank rows, time-major output.
#include "common.h"

namespace wlk {

constexpr int kMaxFft = 512;

// 320 threads: n_fft / 2 + 1 = 257 frequency bins in ONE pass (with 256 threads bin 256 was a second 512-step fp64 loop of
// thread 0 alone - the whole launch waited for it: 51 us per 101 frames; round 6)
__global__ __launch_bounds__(320) void melspec_frame_kernel(MelSpecArgs a) {
    __shared__ double tw[kMaxFft];
    __shared__ float xw[kMaxFft];
    __shared__ float power[kMaxFft / 2 + 4];
    const int t = blockIdx.x;
    const int tid = threadIdx.x;
    const int n_freq = a.n_fft / 2 + 1;
    const int wlo = (a.n_fft - a.win_length) / 2;
    for (int n = tid; n < a.n_fft; n += blockDim.x) {
        tw[n] = a.twiddle[n];
        const int j = a.hop * t - a.n_fft / 2 + n;          // centre=True, constant (zero) padding
        float s = 0.f;
        if (j >= 0 && j < a.n_samples) {
            s = a.audio[j];
            if (a.preemph != 0.f && j > 0) s -= a.preemph * a.audio[j - 1];
        }
        const int wi = n - wlo;
        xw[n] = (wi >= 0 && wi < a.win_length) ? s * a.window[wi] : 0.f;
    }
    __syncthreads();
    for (int k = tid; k < n_freq; k += blockDim.x) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        const int quarter3 = (3 * a.n_fft) / 4;
        for (int n = 0; n < a.n_fft; ++n) {
            const double x = (double)xw[n];
            int sidx = idx + quarter3;                        // sin(theta) = cos(theta - pi/2)
            if (sidx >= a.n_fft) sidx -= a.n_fft;
            re = fma(x, tw[idx], re);
            im = fma(-x, tw[sidx], im);
            idx += k;
            if (idx >= a.n_fft) idx -= a.n_fft;
        }
        const float rf = (float)re, imf = (float)im;
        const float mag = sqrtf(rf * rf + imf * imf);
        power[k] = mag * mag;
    }
    __syncthreads();
    if (tid < a.n_mels) {
        const float* f = a.filters + (long)tid * n_freq;
        float acc = 0.f;
        for (int k = a.filt_lo[tid]; k < a.filt_hi[tid]; ++k) acc = fmaf(f[k], power[k], acc);
        a.out[(long)t * a.n_mels + tid] = logf(acc + a.log_guard);
    }
}

void launch_melspec(const LaunchCtx& ctx, const MelSpecArgs& a, int n_frames) {
    if (n_frames <= 0) return;
    if (a.n_fft > kMaxFft || a.n_fft % 4 != 0 || a.n_mels > 256) throw std::invalid_argument("melspec: unsupported size");
    KernelScope ks(ctx, "melspec_frames", 0.0, 4.0 * a.n_samples + 4.0 * n_frames * a.n_mels);
    hipLaunchKernelGGL(melspec_frame_kernel, dim3(n_frames), dim3(320), 0, ctx.stream, a);
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

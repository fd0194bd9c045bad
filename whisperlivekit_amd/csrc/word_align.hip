// Word timestamps of LocalAgreement's batch Whisper, device half: everything of the reference's
// `find_alignment` (whisperlivekit/whisper/timing.py:163-243) between the decoder and the warping path.
//
//   tokens = [sot sequence, <|notimestamps|>, text tokens, <|endoftext|>]          (:175-182)
//   ONE decoder pass over all of them against the session's encoder output         (:195-196)
//   token probabilities: softmax(logits[n_sot:, :eot])[i, text_token_i]             (:197-200)
//   cross-attention scores of the alignment heads, all token rows                   (:184-191, :205)
//   softmax over the first num_frames // 2 encoder positions, * qk_scale            (:206-207)
//   z-score over the TOKEN axis (population std, no epsilon)                        (:208-209)
//   median filter of width 7 along the frames (reflect padding)                     (:210)
//   mean over the heads, rows [n_sot, -1), negated                                  (:212-214)
//   dtw                                                                             (:214, dtw.hip)
//
// The decoder pass is the session's ordinary prefill (wlk_decode, first = 1) with two switches: cross-attention goes
// through the MFMA flash kernel whatever the row count (it leaves the alignment heads' RAW scores in the alignment
// window rows 0 .. P-1), and the in-place softmax over all 1500 positions that the streaming policy wants is skipped.
// What the host gets back: the cost matrix (optional), dtw's step codes and the token probabilities; walking the path
// back and cutting it into words is integer bookkeeping and stays in Python (whisperlivekit_amd/timing.py).
// The mel of a 30 s window can be handed in as the reference's callers have it (wlk_encode_mel): `transcribe()` slices
// it out of the whole file's log-mel (whisper/transcribe.py), find_alignment gets that segment.
#include <algorithm>
#include <cstring>
#include <vector>

#include "common.h"
#include "internal.h"

namespace wlk {

void launch_dtw(hipStream_t stream, const float* x_dev, int n_rows, int n_cols, signed char* scratch_t, signed char* trace_dev);

namespace {

__device__ __forceinline__ float wa_block_max(float v, float* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, red[w]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float wa_block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r += red[w];
    __syncthreads();
    return r;
}

// softmax(logits[row, :n_cols])[target[row]]: timing.py:197-199 (the columns from <|endoftext|> on are cut off BEFORE the softmax)
__global__ __launch_bounds__(256) void wa_token_prob_kernel(const float* __restrict__ logits, long ld, int n_cols,
                                                            const int* __restrict__ target, float* __restrict__ probs) {
    __shared__ float red[4];
    const float* x = logits + (long)blockIdx.x * ld;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n_cols; i += 256) mx = fmaxf(mx, x[i]);
    mx = wa_block_max(mx, red);
    float sum = 0.f;
    for (int i = threadIdx.x; i < n_cols; i += 256) sum += expf(x[i] - mx);
    sum = wa_block_sum(sum, red);
    if (threadIdx.x == 0) probs[blockIdx.x] = expf(x[target[blockIdx.x]] - mx) / sum;
}

// w[a][r][0 .. F) = softmax(qk_scale * raw[a][r][0 .. F)): one workgroup per (token row r, alignment rank a)
__global__ __launch_bounds__(256) void wa_softmax_kernel(const float* __restrict__ ring, int ring_rows, int T, int F, float qk_scale,
                                                         float* __restrict__ w, int P) {
    __shared__ float red[4];
    const int r = blockIdx.x, a = blockIdx.y;
    const float* x = ring + ((long)a * ring_rows + r) * T;
    float* y = w + ((long)a * P + r) * F;
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < F; j += 256) mx = fmaxf(mx, x[j] * qk_scale);
    mx = wa_block_max(mx, red);
    float sum = 0.f;
    for (int j = threadIdx.x; j < F; j += 256) {
        const float e = expf(x[j] * qk_scale - mx);
        y[j] = e;
        sum += e;
    }
    sum = wa_block_sum(sum, red);
    for (int j = threadIdx.x; j < F; j += 256) y[j] = y[j] / sum;
}

// z-score over the token axis, in place: per (head a, frame f) mean and population std over the P rows in fp64 (as the
// streaming read-out does: torch's CPU std_mean accumulates in double), z = (w - mean) / std - NO epsilon here
__global__ __launch_bounds__(256) void wa_zscore_kernel(float* __restrict__ w, int P, int F) {
    __shared__ double red[4][64];
    const int fx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int f = blockIdx.x * 64 + fx, a = blockIdx.y;
    const bool ok = f < F;
    float* base = w + (long)a * P * F + (ok ? f : 0);
    double sum = 0.0;
    for (int i = rg; i < P; i += 4) sum += (double)base[(long)i * F];
    red[rg][fx] = sum;
    __syncthreads();
    const double mean = (red[0][fx] + red[1][fx] + red[2][fx] + red[3][fx]) / P;
    __syncthreads();
    double sq = 0.0;
    for (int i = rg; i < P; i += 4) {
        const double t = (double)base[(long)i * F] - mean;
        sq += t * t;
    }
    red[rg][fx] = sq;
    __syncthreads();
    const float stdv = (float)sqrt((red[0][fx] + red[1][fx] + red[2][fx] + red[3][fx]) / P);
    if (ok)
        for (int i = rg; i < P; i += 4) base[(long)i * F] = (base[(long)i * F] - (float)mean) / stdv;
}

__device__ __forceinline__ float wa_median7(float v0, float v1, float v2, float v3, float v4, float v5, float v6) {
    float v[7] = {v0, v1, v2, v3, v4, v5, v6};
#pragma unroll
    for (int i = 1; i < 7; ++i) {
#pragma unroll
        for (int j = i; j > 0; --j) {
            const float lo = fminf(v[j - 1], v[j]);
            const float hi = fmaxf(v[j - 1], v[j]);
            v[j - 1] = lo;
            v[j] = hi;
        }
    }
    return v[3];
}

// cost[r - row0][f] = -(1 / n_align) sum_a median7(z[a][r][f-3 .. f+3], reflect padding): rows row0 .. row0 + N - 1.
// median_filter (timing.py:19-54) leaves rows no longer than the padding (F <= 3) untouched.
__global__ __launch_bounds__(256) void wa_cost_kernel(const float* __restrict__ z, int n_align, int P, int F, int row0,
                                                      float* __restrict__ cost) {
    const int r = row0 + blockIdx.x;
    for (int f = threadIdx.x; f < F; f += 256) {
        float acc = 0.f;
        for (int a = 0; a < n_align; ++a) {
            const float* x = z + ((long)a * P + r) * F;
            float m;
            if (F <= 3) {
                m = x[f];
            } else {
                float v[7];
#pragma unroll
                for (int t = 0; t < 7; ++t) {
                    int j = f + t - 3;
                    if (j < 0) j = -j;
                    if (j >= F) j = 2 * (F - 1) - j;
                    v[t] = x[j];
                }
                m = wa_median7(v[0], v[1], v[2], v[3], v[4], v[5], v[6]);
            }
            acc += m;
        }
        cost[(long)blockIdx.x * F + f] = -(acc / (float)n_align);
    }
}

}  // namespace
}  // namespace wlk

using namespace wlk;

extern "C" {

int wlk_encode_mel(wlk_session* s, const float* mel, int32_t n_frames) {
    if (!s || !mel) return fail(WLK_ERR_ARG, "NULL argument");
    if (n_frames != kMelFrames) return fail(WLK_ERR_ARG, "encode_mel: the encoder takes exactly 3000 frames");
    return guarded([&]() {
        wlk_model* m = s->m;
        const int nm = m->D.n_mels;
        WLK_HIP(hipSetDevice(m->device));
        // [n_mels][3000] (the reference's layout) -> the time-major rows 1 .. 3000 of mel_t (rows 0 and 3001 stay zero)
        std::vector<float> tm((size_t)kMelFrames * nm);
        for (int t = 0; t < kMelFrames; ++t)
            for (int c = 0; c < nm; ++c) tm[(size_t)t * nm + c] = mel[(size_t)c * kMelFrames + t];
        WLK_HIP(hipStreamSynchronize(s->stream));
        WLK_HIP(hipMemcpyAsync(s->mel_t + nm, tm.data(), tm.size() * sizeof(float), hipMemcpyHostToDevice, s->stream));
        WLK_HIP(hipStreamSynchronize(s->stream));          // `tm` is pageable: the copy has left it
        s->mel_cached_samples = 0;                         // the per-frame cache no longer describes mel_t
        std::vector<int> content;
        wlk_encode_group({s}, s->ctx(), content, /*mel_given=*/true);
        s->content_len = m->D.n_audio_ctx;
        return WLK_OK;
    });
}

int wlk_log_mel(wlk_session* s, const float* pcm, int64_t n_samples, int32_t padding, float* mel, uint64_t capacity_floats,
                int32_t* n_frames) {
    if (!s || !n_frames || (mel && n_samples > 0 && !pcm)) return fail(WLK_ERR_ARG, "NULL argument");
    if (n_samples < 0 || padding < 0 || n_samples + padding > 0x7fff0000LL) return fail(WLK_ERR_ARG, "log_mel: bad sample counts");
    const long n_padded = n_samples + padding;
    if (n_padded <= kNFft / 2) return fail(WLK_ERR_ARG, "log_mel: the padded signal must be longer than the 200-sample reflection");
    wlk_model* m = s->m;
    const int nm = m->D.n_mels;
    const int n_total = (int)(n_padded / kHop);                   // torch.stft frames minus the dropped last one
    *n_frames = n_total;
    if (!mel) return WLK_OK;                                      // size query
    if ((uint64_t)n_total * nm > capacity_floats) return fail(WLK_ERR_CAPACITY, "log_mel: output buffer too small");
    return guarded([&]() {
        WLK_HIP(hipSetDevice(m->device));
        // frames whose 400-sample window sees a non-zero sample; with fewer than 200 trailing zeros the reflection at the end
        // brings samples back in: then every frame is computed
        long n_active = n_samples > 0 ? (n_samples + kNFft / 2 + kHop - 1) / kHop : 0;
        if (padding < kNFft / 2 || n_active > n_total) n_active = n_samples > 0 ? n_total : 0;
        struct Scratch {
            float *audio = nullptr, *logmel = nullptr, *fmax = nullptr, *out = nullptr, *gmax = nullptr;
            ~Scratch() { for (float* p : {audio, logmel, fmax, out, gmax}) if (p) (void)hipFree(p); }
        } w;
        w.audio = dev_alloc<float>((size_t)std::max<long>(n_padded, 1));
        w.logmel = dev_alloc<float>((size_t)std::max<long>(n_active, 1) * nm);
        w.fmax = dev_alloc<float>((size_t)std::max<long>(n_active, 1));
        w.out = dev_alloc<float>((size_t)n_total * nm);
        w.gmax = dev_alloc<float>(1);
        WLK_HIP(hipStreamSynchronize(s->stream));
        if (n_samples > 0) WLK_HIP(hipMemcpyAsync(w.audio, pcm, (size_t)n_samples * sizeof(float), hipMemcpyHostToDevice, s->stream));
        if (padding > 0) WLK_HIP(hipMemsetAsync(w.audio + n_samples, 0, (size_t)padding * sizeof(float), s->stream));
        MelArgs ma;
        ma.audio = w.audio; ma.n_samples = (int)n_padded; ma.n_padded = (int)n_padded;
        ma.window = m->w("mel.window"); ma.twiddle = m->twiddle; ma.filters = m->w("mel.filters");
        ma.filt_lo = m->filt_lo; ma.filt_hi = m->filt_hi; ma.n_mels = nm;
        ma.logmel = w.logmel; ma.frame_max = w.fmax; ma.mel_t = nullptr;
        ma.n_active = (int)n_active; ma.n_total = n_total;
        launch_mel_full(s->ctx(), ma, w.gmax, w.out);
        WLK_HIP(hipMemcpyAsync(mel, w.out, (size_t)n_total * nm * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        WLK_HIP(hipStreamSynchronize(s->stream));
        return WLK_OK;
    });
}

int wlk_find_alignment(wlk_session* s, const int64_t* tokens, int32_t n_tokens, int32_t n_sot, int32_t eot, int32_t num_frames,
                       float qk_scale, float* cost_host, int8_t* trace_host, float* token_probs_host) {
    if (!s || !tokens || !trace_host || !token_probs_host) return fail(WLK_ERR_ARG, "NULL argument");
    if (s->beam != 1) return fail(WLK_ERR_ARG, "find_alignment needs a beam-1 session");
    if (!s->encoded) return fail(WLK_ERR_STATE, "find_alignment before an encode");
    wlk_model* m = s->m;
    const wlk_dims& D = m->D;
    const int P = n_tokens, n_text = P - n_sot - 2, N = n_text + 1, F = num_frames / 2;
    if (n_sot < 1 || n_text < 1) return fail(WLK_ERR_ARG, "find_alignment: need [sot sequence, notimestamps, >= 1 text token, eot]");
    if (P > D.n_text_ctx) return fail(WLK_ERR_CAPACITY, "find_alignment: more tokens than the text context");
    if (F < 1 || F > D.n_audio_ctx) return fail(WLK_ERR_ARG, "find_alignment: num_frames out of range");
    if (eot < 1 || eot > D.n_vocab) return fail(WLK_ERR_ARG, "find_alignment: eot out of range");
    if (m->n_align < 1) return fail(WLK_ERR_STATE, "find_alignment: the model has no alignment heads");
    for (int i = 0; i < n_text; ++i)
        if (tokens[n_sot + 1 + i] < 0 || tokens[n_sot + 1 + i] >= eot) return fail(WLK_ERR_ARG, "find_alignment: text token >= eot");
    // the decoder pass: the session's prefill, raw alignment scores left in window rows 0 .. P-1
    s->align_raw_scores = true;
    const int rc = wlk_decode(s, tokens, 1, P, 1, 0);
    s->align_raw_scores = false;
    if (rc != WLK_OK) return rc;
    return guarded([&]() {
        WLK_HIP(hipSetDevice(m->device));
        const LaunchCtx c = s->ctx();
        const int d = D.n_text_state, T = D.n_audio_ctx;
        // workspace (grown on demand, kept by the session): [hidden n_text*d | logits n_text*eot | w n_align*P*F | cost N*F |
        // probs n_text | targets n_text (int) | trace 2 * (N+1)(F+1) bytes]
        const size_t n_hid = (size_t)n_text * d, n_log = (size_t)n_text * eot, n_w = (size_t)m->n_align * P * F, n_cost = (size_t)N * F;
        const size_t n_trace = (size_t)(N + 1) * (F + 1);
        const size_t floats = n_hid + n_log + n_w + n_cost + 2 * (size_t)n_text + (2 * n_trace + 3) / 4 + 16;
        if (floats > s->wa_cap) {
            WLK_HIP(hipStreamSynchronize(s->stream));
            if (s->wa_buf) WLK_HIP(hipFree(s->wa_buf));
            s->wa_buf = nullptr; s->wa_cap = 0;
            s->wa_buf = dev_alloc<float>(floats);
            s->wa_cap = floats;
        }
        float* hid = s->wa_buf;
        float* logits = hid + n_hid;
        float* w = logits + n_log;
        float* cost = w + n_w;
        float* probs = cost + n_cost;
        int* target = reinterpret_cast<int*>(probs + n_text);
        signed char* trace_t = reinterpret_cast<signed char*>(target + n_text);
        signed char* trace = trace_t + n_trace;

        // token probabilities: final LayerNorm + vocabulary projection of the rows that predict the text tokens
        int* stage = static_cast<int*>(s->pinned);
        for (int i = 0; i < n_text; ++i) stage[i] = (int)tokens[n_sot + 1 + i];
        WLK_HIP(hipMemcpyAsync(target, stage, (size_t)n_text * sizeof(int), hipMemcpyHostToDevice, s->stream));
        launch_layernorm(c, s->dx + (size_t)n_sot * d, d, m->w_ln_w, m->w_ln_b, hid, d, n_text, d, "wa_ln_f");
        GemmArgs lg;
        lg.A = hid; lg.lda = d; lg.W = m->w_tok_emb; lg.C = logits; lg.ldc = eot; lg.M = n_text; lg.N = eot; lg.K = d;
        launch_linear(c, lg, "wa_logits");
        hipLaunchKernelGGL(wa_token_prob_kernel, dim3(n_text), dim3(256), 0, s->stream, logits, (long)eot, eot, target, probs);
        // attention -> cost matrix
        hipLaunchKernelGGL(wa_softmax_kernel, dim3(P, m->n_align), dim3(256), 0, s->stream, s->ring, s->ring_rows, T, F, qk_scale, w, P);
        hipLaunchKernelGGL(wa_zscore_kernel, dim3((F + 63) / 64, m->n_align), dim3(256), 0, s->stream, w, P, F);
        hipLaunchKernelGGL(wa_cost_kernel, dim3(N), dim3(256), 0, s->stream, w, m->n_align, P, F, n_sot, cost);
        WLK_HIP(hipGetLastError());
        launch_dtw(s->stream, cost, N, F, trace_t, trace);
        // results: three copies behind one synchronisation (host buffers are the caller's: pageable, so synchronous)
        WLK_HIP(hipStreamSynchronize(s->stream));
        copy_sync(trace_host, trace, n_trace, hipMemcpyDeviceToHost);
        copy_sync(token_probs_host, probs, (size_t)n_text * sizeof(float), hipMemcpyDeviceToHost);
        if (cost_host) copy_sync(cost_host, cost, n_cost * sizeof(float), hipMemcpyDeviceToHost);
        // the window no longer holds what the streaming read-out expects: the next decode of this session must be a prefill
        s->n_steps = 0;
        s->self_len = 0;
        return WLK_OK;
    });
}

}  // extern "C"

// Silero VAD gate on the GPU: the voice-activity controller that sits immediately before the
// simul_whisper path (whisperlivekit/audio_processor.py:1189-1190 -> FixedVADIterator -> the TorchScript model of
// whisperlivekit/silero_vad_models/silero_vad.jit, loaded by load_jit_vad, silero_vad_iterator.py:163-184).  The
// reference evaluates it on the CPU, one 512-sample window per call (~0.7-1 ms each, i.e. 11-16 ms of host time
// per 0.5 s chunk inside the event loop).  Here all windows of a chunk go through two launches:
//   vad_features_kernel  one workgroup per window (windows are independent up to the recurrent cell):
//                        [64 context | 512] samples -> reflect pad -> 256-point DFT basis, hop 128 (4 frames) ->
//                        magnitude [129,4] -> 4 x (Conv1d k3 + ReLU) -> [128] -> W_ih x + b_ih  (512 gate inputs)
//   vad_lstm_kernel      one 512-thread workgroup per stream walks the windows in order: each thread keeps its row
//                        of W_hh in registers (128 VGPRs), h/c live in LDS; LSTMCell -> ReLU -> Conv1d(128,1) ->
//                        sigmoid = speech probability of the window.
// Network structure and state handling restated from the archive's code objects; pinned by reference-generated
// probabilities (tests/golden/vad_cases.npz).  fp32 throughout, like the reference.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/wlk_hip.h"
#include "common.h"

namespace wlk {

constexpr int kVadCtx = 64, kVadWin = 512, kVadX = kVadCtx + kVadWin, kVadPad = 64;
constexpr int kVadNfft = 256, kVadHop = 128, kVadBins = 129, kVadFrames = 4;
constexpr int kVadC0 = 128, kVadC1 = 64, kVadC2 = 64, kVadC3 = 128, kVadHid = 128, kVadGates = 4 * kVadHid;

struct VadSlot {
    const char* name;
    size_t numel;
};
// packed layout (all "T" tensors are stored input-major so that consecutive threads read consecutive floats)
static const VadSlot kVadSlots[] = {
    {"stft.basisT", (size_t)kVadNfft * 2 * kVadBins},        // [k][f]   from forward_basis_buffer [258,1,256]
    {"enc0.wT", (size_t)kVadBins * 3 * kVadC0}, {"enc0.b", kVadC0},   // [(ci*3+tap)][co]
    {"enc1.wT", (size_t)kVadC0 * 3 * kVadC1}, {"enc1.b", kVadC1},
    {"enc2.wT", (size_t)kVadC1 * 3 * kVadC2}, {"enc2.b", kVadC2},
    {"enc3.wT", (size_t)kVadC2 * 3 * kVadC3}, {"enc3.b", kVadC3},
    {"rnn.wihT", (size_t)kVadC3 * kVadGates}, {"rnn.whhT", (size_t)kVadHid * kVadGates},   // [k][gate]
    {"rnn.bih", kVadGates}, {"rnn.bhh", kVadGates},
    {"dec.w", kVadHid}, {"dec.b", 1},
};
constexpr int kVadNumSlots = sizeof(kVadSlots) / sizeof(kVadSlots[0]);

static size_t vad_offset(int slot) {
    size_t off = 0;
    for (int i = 0; i < slot; ++i) off += (kVadSlots[i].numel + 63) / 64 * 64;
    return off;
}
static size_t vad_total_floats() { return vad_offset(kVadNumSlots); }

struct VadWeights {
    const float *basisT, *w0, *b0, *w1, *b1, *w2, *b2, *w3, *b3, *wihT, *whhT, *bih, *bhh, *decw, *decb;
};

__global__ __launch_bounds__(256) void vad_features_kernel(VadWeights W, const float* __restrict__ audio,
                                                           float* __restrict__ gi) {
    __shared__ float xs[kVadX + kVadPad];
    __shared__ float spec[2 * kVadBins * kVadFrames];
    __shared__ float mag[kVadBins][kVadFrames + 2];      // one zero column either side: Conv1d padding 1
    __shared__ float a0[kVadC0][kVadFrames + 2];
    __shared__ float a1[kVadC1][2 + 2];
    __shared__ float a2[kVadC2];
    __shared__ float feat[kVadC3];
    const int tid = threadIdx.x, w = blockIdx.x;
    const float* x1 = audio + (size_t)w * kVadWin;         // this window's [context | samples]
    for (int i = tid; i < kVadX; i += 256) xs[i] = x1[i];
    if (tid < kVadPad) xs[kVadX + tid] = x1[kVadX - 2 - tid];   // ReflectionPad1d((0, 64))
    __syncthreads();
    for (int f = tid; f < 2 * kVadBins; f += 256) {
        float acc[kVadFrames] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < kVadNfft; ++k) {
            const float b = W.basisT[k * (2 * kVadBins) + f];
#pragma unroll
            for (int t = 0; t < kVadFrames; ++t) acc[t] = fmaf(b, xs[t * kVadHop + k], acc[t]);
        }
#pragma unroll
        for (int t = 0; t < kVadFrames; ++t) spec[f * kVadFrames + t] = acc[t];
    }
    __syncthreads();
    for (int i = tid; i < kVadBins * (kVadFrames + 2); i += 256) {
        const int f = i / (kVadFrames + 2), c = i - f * (kVadFrames + 2);
        float v = 0.f;
        if (c >= 1 && c <= kVadFrames) {
            const float re = spec[f * kVadFrames + c - 1], im = spec[(kVadBins + f) * kVadFrames + c - 1];
            v = sqrtf(re * re + im * im);
        }
        mag[f][c] = v;
    }
    __syncthreads();
    {   // encoder.0: 129 -> 128, stride 1; thread = (channel, frame pair)
        const int co = tid & 127, t0 = (tid >> 7) * 2;
        float s0 = 0.f, s1 = 0.f;
        for (int ci = 0; ci < kVadBins; ++ci)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float wv = W.w0[(ci * 3 + k) * kVadC0 + co];
                s0 = fmaf(wv, mag[ci][t0 + k], s0);
                s1 = fmaf(wv, mag[ci][t0 + 1 + k], s1);
            }
        a0[co][t0 + 1] = fmaxf(s0 + W.b0[co], 0.f);
        a0[co][t0 + 2] = fmaxf(s1 + W.b0[co], 0.f);
        if (tid < kVadC0) { a0[tid][0] = 0.f; a0[tid][kVadFrames + 1] = 0.f; }
    }
    __syncthreads();
    if (tid < 2 * kVadC1) {   // encoder.1: 128 -> 64, stride 2 -> 2 frames
        const int co = tid & 63, t = tid >> 6;
        float s = 0.f;
        for (int ci = 0; ci < kVadC0; ++ci)
#pragma unroll
            for (int k = 0; k < 3; ++k) s = fmaf(W.w1[(ci * 3 + k) * kVadC1 + co], a0[ci][2 * t + k], s);
        a1[co][t + 1] = fmaxf(s + W.b1[co], 0.f);
        if (t == 0) { a1[co][0] = 0.f; a1[co][3] = 0.f; }
    }
    __syncthreads();
    if (tid < kVadC2) {       // encoder.2: 64 -> 64, stride 2 -> 1 frame
        float s = 0.f;
        for (int ci = 0; ci < kVadC1; ++ci)
#pragma unroll
            for (int k = 0; k < 3; ++k) s = fmaf(W.w2[(ci * 3 + k) * kVadC2 + tid], a1[ci][k], s);
        a2[tid] = fmaxf(s + W.b2[tid], 0.f);
    }
    __syncthreads();
    if (tid < kVadC3) {       // encoder.3: 64 -> 128 on a single frame: only the centre tap sees data
        float s = 0.f;
        for (int ci = 0; ci < kVadC2; ++ci) s = fmaf(W.w3[(ci * 3 + 1) * kVadC3 + tid], a2[ci], s);
        feat[tid] = fmaxf(s + W.b3[tid], 0.f);
    }
    __syncthreads();
    for (int g = tid; g < kVadGates; g += 256) {   // LSTMCell input half: W_ih x + b_ih
        float s = 0.f;
        for (int k = 0; k < kVadC3; ++k) s = fmaf(W.wihT[k * kVadGates + g], feat[k], s);
        gi[(size_t)w * kVadGates + g] = s + W.bih[g];
    }
}

__global__ __launch_bounds__(512) void vad_lstm_kernel(VadWeights W, const float* __restrict__ gi, float* __restrict__ state,
                                                       float* __restrict__ probs, float* __restrict__ audio,
                                                       int n_windows) {
    __shared__ float hs[kVadHid], cs[kVadHid], gates[kVadGates], red[2];
    const int tid = threadIdx.x, lane = tid & 63;
    float whh[kVadHid];
#pragma unroll
    for (int k = 0; k < kVadHid; ++k) whh[k] = W.whhT[k * kVadGates + tid];
    const float bhh = W.bhh[tid];
    if (tid < kVadHid) { hs[tid] = state[tid]; cs[tid] = state[kVadHid + tid]; }
    __syncthreads();
    for (int w = 0; w < n_windows; ++w) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < kVadHid; ++k) acc = fmaf(whh[k], hs[k], acc);
        gates[tid] = gi[(size_t)w * kVadGates + tid] + (acc + bhh);
        __syncthreads();
        if (tid < kVadHid) {
            const float ig = 1.0f / (1.0f + expf(-gates[tid]));
            const float fg = 1.0f / (1.0f + expf(-gates[kVadHid + tid]));
            const float gg = tanhf(gates[2 * kVadHid + tid]);
            const float og = 1.0f / (1.0f + expf(-gates[3 * kVadHid + tid]));
            const float c = fg * cs[tid] + ig * gg;
            const float h = og * tanhf(c);
            cs[tid] = c;
            hs[tid] = h;
            float part = fmaxf(h, 0.f) * W.decw[tid];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
            if (lane == 0) red[tid >> 6] = part;
        }
        __syncthreads();
        if (tid == 0) probs[w] = 1.0f / (1.0f + expf(-((red[0] + red[1]) + W.decb[0])));
    }
    if (tid < kVadHid) { state[tid] = hs[tid]; state[kVadHid + tid] = cs[tid]; }
    if (tid < kVadCtx) audio[tid] = audio[(size_t)n_windows * kVadWin + tid];   // next call's context
}

}  // namespace wlk

using namespace wlk;

struct wlk_vad {
    int device = 0;
    float* arena = nullptr;
    VadWeights W{};
};
struct wlk_vad_stream {
    wlk_vad* m = nullptr;
    int max_windows = 0;
    hipStream_t stream = nullptr;
    float *audio = nullptr, *gi = nullptr, *state = nullptr, *probs = nullptr;
    float* pinned = nullptr;   // [max_windows * 512 samples | max_windows probs]
};

namespace {
int vad_fail(int code, const std::string& msg) {
    set_last_error(msg);
    return code;
}
template <typename F>
int vad_guarded(F&& f) {
    try {
        return f();
    } catch (const HipError& e) {
        return vad_fail(WLK_ERR_HIP, e.what());
    } catch (const std::exception& e) {
        return vad_fail(WLK_ERR_STATE, e.what());
    }
}
}  // namespace

extern "C" {

int wlk_vad_weights_floats(uint64_t* n_floats) {
    if (!n_floats) return vad_fail(WLK_ERR_ARG, "NULL argument");
    *n_floats = vad_total_floats();
    return WLK_OK;
}

int wlk_vad_tensor_lookup(const char* packed_name, uint64_t* offset_floats, uint64_t* numel) {
    if (!packed_name) return vad_fail(WLK_ERR_ARG, "NULL argument");
    for (int i = 0; i < kVadNumSlots; ++i)
        if (!strcmp(kVadSlots[i].name, packed_name)) {
            if (offset_floats) *offset_floats = vad_offset(i);
            if (numel) *numel = kVadSlots[i].numel;
            return WLK_OK;
        }
    return vad_fail(WLK_ERR_ARG, std::string("unknown packed tensor ") + packed_name);
}

int wlk_vad_tensor_name(int index, const char** name) {
    if (!name || index < 0 || index >= kVadNumSlots) return vad_fail(WLK_ERR_ARG, "tensor index out of range");
    *name = kVadSlots[index].name;
    return WLK_OK;
}

int wlk_vad_create(int device, const float* packed_host, uint64_t n_floats, wlk_vad** out) {
    if (!packed_host || !out) return vad_fail(WLK_ERR_ARG, "NULL argument");
    if (n_floats != vad_total_floats()) return vad_fail(WLK_ERR_ARG, "packed weight buffer has the wrong size");
    return vad_guarded([&]() {
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
            return vad_fail(WLK_ERR_HIP, "no HIP device: the VAD backend has no CPU fallback");
        WLK_HIP(hipSetDevice(device));
        auto m = std::make_unique<wlk_vad>();
        m->device = device;
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&m->arena), n_floats * sizeof(float)));
        copy_sync(m->arena, packed_host, n_floats * sizeof(float), hipMemcpyHostToDevice);
        const float* a = m->arena;
        auto at = [&](int i) { return a + vad_offset(i); };
        m->W = VadWeights{at(0), at(1), at(2), at(3), at(4), at(5), at(6), at(7), at(8), at(9), at(10), at(11), at(12),
                          at(13), at(14)};
        *out = m.release();
        return WLK_OK;
    });
}

int wlk_vad_destroy(wlk_vad* m) {
    if (!m) return WLK_OK;
    (void)hipSetDevice(m->device);
    if (m->arena) (void)hipFree(m->arena);
    delete m;
    return WLK_OK;
}

int wlk_vad_stream_create(wlk_vad* m, int max_windows, wlk_vad_stream** out) {
    if (!m || !out) return vad_fail(WLK_ERR_ARG, "NULL argument");
    if (max_windows < 1 || max_windows > 4096) return vad_fail(WLK_ERR_ARG, "max_windows out of range");
    return vad_guarded([&]() {
        WLK_HIP(hipSetDevice(m->device));
        auto s = std::make_unique<wlk_vad_stream>();
        s->m = m;
        s->max_windows = max_windows;
        WLK_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        const size_t n_audio = kVadCtx + (size_t)max_windows * kVadWin;
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&s->audio), n_audio * sizeof(float)));
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&s->gi), (size_t)max_windows * kVadGates * sizeof(float)));
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&s->state), 2 * kVadHid * sizeof(float)));
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&s->probs), (size_t)max_windows * sizeof(float)));
        WLK_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->pinned), ((size_t)max_windows * (kVadWin + 1)) * sizeof(float),
                              hipHostMallocDefault));
        WLK_HIP(hipMemsetAsync(s->audio, 0, n_audio * sizeof(float), s->stream));
        WLK_HIP(hipMemsetAsync(s->state, 0, 2 * kVadHid * sizeof(float), s->stream));
        WLK_HIP(hipStreamSynchronize(s->stream));
        *out = s.release();
        return WLK_OK;
    });
}

int wlk_vad_stream_reset(wlk_vad_stream* s) {
    if (!s) return vad_fail(WLK_ERR_ARG, "stream is NULL");
    return vad_guarded([&]() {
        WLK_HIP(hipSetDevice(s->m->device));
        WLK_HIP(hipMemsetAsync(s->audio, 0, kVadCtx * sizeof(float), s->stream));
        WLK_HIP(hipMemsetAsync(s->state, 0, 2 * kVadHid * sizeof(float), s->stream));
        WLK_HIP(hipStreamSynchronize(s->stream));
        return WLK_OK;
    });
}

int wlk_vad_stream_run(wlk_vad_stream* s, const float* pcm_host, int n_windows, float* probs_host) {
    if (!s || !pcm_host || !probs_host) return vad_fail(WLK_ERR_ARG, "NULL argument");
    if (n_windows < 1 || n_windows > s->max_windows) return vad_fail(WLK_ERR_CAPACITY, "window count out of range");
    return vad_guarded([&]() {
        WLK_HIP(hipSetDevice(s->m->device));
        const size_t n = (size_t)n_windows * kVadWin;
        std::memcpy(s->pinned, pcm_host, n * sizeof(float));
        WLK_HIP(hipMemcpyAsync(s->audio + kVadCtx, s->pinned, n * sizeof(float), hipMemcpyHostToDevice, s->stream));
        hipLaunchKernelGGL(vad_features_kernel, dim3(n_windows), dim3(256), 0, s->stream, s->m->W, s->audio, s->gi);
        WLK_HIP(hipGetLastError());
        hipLaunchKernelGGL(vad_lstm_kernel, dim3(1), dim3(512), 0, s->stream, s->m->W, s->gi, s->state, s->probs, s->audio,
                           n_windows);
        WLK_HIP(hipGetLastError());
        float* out = s->pinned + (size_t)s->max_windows * kVadWin;
        WLK_HIP(hipMemcpyAsync(out, s->probs, (size_t)n_windows * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        WLK_HIP(hipStreamSynchronize(s->stream));
        std::memcpy(probs_host, out, (size_t)n_windows * sizeof(float));
        return WLK_OK;
    });
}

int wlk_vad_stream_state(wlk_vad_stream* s, float* h_host, float* c_host) {
    if (!s || !h_host || !c_host) return vad_fail(WLK_ERR_ARG, "NULL argument");
    return vad_guarded([&]() {
        WLK_HIP(hipSetDevice(s->m->device));
        WLK_HIP(hipStreamSynchronize(s->stream));
        copy_sync(h_host, s->state, kVadHid * sizeof(float), hipMemcpyDeviceToHost);
        copy_sync(c_host, s->state + kVadHid, kVadHid * sizeof(float), hipMemcpyDeviceToHost);
        return WLK_OK;
    });
}

int wlk_vad_stream_destroy(wlk_vad_stream* s) {
    if (!s) return WLK_OK;
    (void)hipSetDevice(s->m->device);
    (void)hipStreamSynchronize(s->stream);
    for (float* p : {s->audio, s->gi, s->state, s->probs})
        if (p) (void)hipFree(p);
    if (s->pinned) (void)hipHostFree(s->pinned);
    (void)hipStreamDestroy(s->stream);
    delete s;
    return WLK_OK;
}

}  // extern "C"

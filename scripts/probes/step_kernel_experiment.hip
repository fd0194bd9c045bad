// EXPERIMENT, NOT BUILT INTO libwlk_hip.so (kept as groundwork; see DESIGN.md "Things tried and reverted").
// v2, measured on MI355X (base.en): BIT-IDENTICAL to the one-kernel-per-op path on 5 golden streams, but slower:
// 419 us per step with 128 workgroups (471 with 64, 609 with 32) against ~330 us.  v1's wrong results came from
// phase buffers sharing a 128-byte cache line (a reader of buffer A cached the neighbouring, not yet written, head of
// buffer B) - every buffer is padded to 256 bytes now.  Why it is still slower: a cross-XCD hand-off through memory
// is a chain of write-through drain (s_waitcnt) -> atomic arrive -> poll -> fabric read of the activation row ->
// epilogue operands, about 8.5 us per phase under real load (the isolated probe measured 2.9), i.e. no cheaper
// than a kernel boundary; and at 485 VGPRs a workgroup owns its CU, so the kernel would also starve other streams.
// Next idea: keep one step inside ONE XCD (32 CUs, hand-offs through its shared L2).  Needs the StepLayer / StepArgs
// declarations and the api.hip hook of commit history to compile.
// Persistent decode-step kernel (beam 1, one fed token): every decoder layer of TextDecoder.forward
// (whisper/model.py:279-332 via AlignAtt._get_logits_and_cross_attn, simul_whisper.py:357-368) in ONE launch.
//
// A decode step is a chain of ~50 dependent matrix-vector products and two small attentions over a 512..1280-float
// activation; as separate kernels every link costs a ~5.5 us kernel boundary (cold L2, dispatch) whatever its
// work, and the step is 330 us of which ~40 us is memory time.  Here 128 workgroups stay resident and hand the
// activation from phase to phase through a fence-free grid barrier (scripts/probes/barrier_probe.hip: 2.9 us):
//   * every phase writes its result with agent-scope relaxed atomic stores (global_store sc1, write-through) into
//     a buffer that NO workgroup has read earlier in this launch, so no L1 / per-XCD L2 can hold a stale line and
//     the consumers use plain loads - no release/acquire fence (those write back / invalidate whole caches);
//   * arrive = s_waitcnt vmcnt(0) + one relaxed agent atomicAdd; wait = bounded relaxed polling (a timeout sets
//     an error word the host checks; no spin is unbounded);
//   * "never read earlier" holds per 128-byte cache line, so every phase buffer is padded to 256 bytes;
//   * weights do not depend on activations: each wave loads the first K-chunk of the rows it will compute in the
//     NEXT phase (and that phase's LayerNorm affine, bias and residual operands) BEFORE it waits at the barrier, so
//     after the barrier only the activation row itself is a memory round trip;
//   * the two barrier words alternate between launches (a device-side launch counter picks one; the other is
//     re-zeroed), so the captured hipGraph node needs no memset node in front of it.
// Arithmetic is the existing kernels' (gemv1_f32_kernel, decoder_self_attention_kernel, cross_split_kernel,
// cross_merge_kernel) in the same order, so logits and alignment rows are bit-identical to the multi-kernel path
// (tests/test_gpu_parity.py::test_hipgraph_replay_equals_eager_launches).
#include "common.h"

namespace wlk {

constexpr int kStepSplit = 8;       // key splits of the cross-attention (== kCrossSplit of decoder.hip)
constexpr int kStepUnroll = 12;

__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void step_barrier(unsigned* word, unsigned target, int* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) {   // ~0.5 s: a workgroup never became resident
                *err = 1;
                break;
            }
        }
    }
    __syncthreads();
}

struct GemvPhase {
    const float* W = nullptr;       // [N][K]
    const float* bias = nullptr;
    int N = 0, K = 0;
    const float* x = nullptr;       // activation row [K]; nullptr = merged cross-attention output (pm / pl / po)
    const float* ln_g = nullptr;    // fused pre-LayerNorm
    const float* ln_b = nullptr;
    int flags = 0;
    float scale = 1.f;
    int scale_cols = 0;
    const float* R = nullptr;       // residual row
    float* out = nullptr;
    bool out_plain = false;         // last phase: consumed by the NEXT kernel, plain stores
    float* kc = nullptr;            // fused KV-cache append (columns [kv_d, 3 kv_d))
    float* vc = nullptr;
    int kv_d = 0;
    long kv_at = 0;
    const float* pm = nullptr;      // [H][kStepSplit] partial softmax states of the cross-attention
    const float* pl = nullptr;
    const float* po = nullptr;      // [H][kStepSplit][64]
};

// One float4 of the activation row at float4 index c (dims 4c .. 4c+3)
__device__ __forceinline__ float4 phase_x4(const GemvPhase& p, int c) {
    if (p.x) return *reinterpret_cast<const float4*>(p.x + 4 * c);
    // cross_merge_kernel's arithmetic for the head that owns these dims
    const int head = (4 * c) >> 6, dd = (4 * c) & 63;
    const long base = (long)head * kStepSplit;
    float M = p.pm[base];
#pragma unroll
    for (int s = 1; s < kStepSplit; ++s) M = fmaxf(M, p.pm[base + s]);
    float L = 0.f;
    float f[kStepSplit];
#pragma unroll
    for (int s = 0; s < kStepSplit; ++s) {
        f[s] = expf(p.pm[base + s] - M);
        L += p.pl[base + s] * f[s];
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < kStepSplit; ++s) {
        const float4 o = *reinterpret_cast<const float4*>(p.po + (base + s) * 64 + dd);
        acc.x += o.x * f[s]; acc.y += o.y * f[s]; acc.z += o.z * f[s]; acc.w += o.w * f[s];
    }
    return make_float4(acc.x / L, acc.y / L, acc.z / L, acc.w / L);
}

constexpr int kCH = 4;   // float4 chunks of K in flight per lane

// What a wave can fetch for its first row group of a GEMV phase before the activations exist
struct Pref {
    float4 w[kCH][4];
    float4 ga[kCH], be[kCH];
};

__device__ __forceinline__ int phase_rpw(const GemvPhase& p, int n_waves) { return p.N > n_waves ? 4 : 1; }

template <int RPW>
__device__ __forceinline__ void gemv_prefetch_rows(const GemvPhase& p, int n_base, int lane, Pref& pf) {
    const int K4 = p.K >> 2;
#pragma unroll
    for (int u = 0; u < kCH; ++u) {
        const int c = lane + 64 * u;
        const int cc = (c < K4 ? c : 0) * 4;
#pragma unroll
        for (int r = 0; r < RPW; ++r)
            pf.w[u][r] = *reinterpret_cast<const float4*>(p.W + (long)min(n_base + r, p.N - 1) * p.K + cc);
        if (p.ln_g) {
            pf.ga[u] = *reinterpret_cast<const float4*>(p.ln_g + cc);
            pf.be[u] = *reinterpret_cast<const float4*>(p.ln_b + cc);
        }
    }
}

__device__ __forceinline__ void gemv_prefetch(const GemvPhase& p, int wid, int n_waves, int lane, Pref& pf) {
    if (phase_rpw(p, n_waves) == 4) {
        if (wid * 4 < p.N) gemv_prefetch_rows<4>(p, wid * 4, lane, pf);
    } else {
        if (wid < p.N) gemv_prefetch_rows<1>(p, wid, lane, pf);
    }
}

// rows [n_base, n_base + RPW) of y = epilogue(W . LN(x) + b) by ONE wave; same reduction / fmaf order as
// gemv1_f32_kernel.  `pf` holds chunk 0 of W (and of the LayerNorm affine) when `use_pf`.
template <int RPW>
__device__ __forceinline__ void gemv_rows(const GemvPhase& p, int n_base, int lane, float mean, float rstd, const Pref& pf,
                                          bool use_pf) {
    const int K4 = p.K >> 2;
    float acc[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) acc[r] = 0.f;
    for (int c0 = 0; c0 < K4; c0 += 64 * kCH) {
        float4 w[kCH][RPW], x[kCH];
        const bool first = use_pf && c0 == 0;
#pragma unroll
        for (int u = 0; u < kCH; ++u) {
            const int c = c0 + lane + 64 * u;
            const bool ok = c < K4;
            const int cc = (ok ? c : 0) * 4;
#pragma unroll
            for (int r = 0; r < RPW; ++r)
                w[u][r] = first ? pf.w[u][r] : *reinterpret_cast<const float4*>(p.W + (long)min(n_base + r, p.N - 1) * p.K + cc);
            x[u] = ok ? phase_x4(p, c) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.ln_g) {
                const float4 ga = first ? pf.ga[u] : *reinterpret_cast<const float4*>(p.ln_g + cc);
                const float4 be = first ? pf.be[u] : *reinterpret_cast<const float4*>(p.ln_b + cc);
                x[u].x = (x[u].x - mean) * rstd * ga.x + be.x;
                x[u].y = (x[u].y - mean) * rstd * ga.y + be.y;
                x[u].z = (x[u].z - mean) * rstd * ga.z + be.z;
                x[u].w = (x[u].w - mean) * rstd * ga.w + be.w;
            }
        }
#pragma unroll
        for (int u = 0; u < kCH; ++u) {
            if (c0 + lane + 64 * u < K4) {
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    acc[r] = fmaf(w[u][r].x, x[u].x, acc[r]);
                    acc[r] = fmaf(w[u][r].y, x[u].y, acc[r]);
                    acc[r] = fmaf(w[u][r].z, x[u].z, acc[r]);
                    acc[r] = fmaf(w[u][r].w, x[u].w, acc[r]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc[r] += __shfl_xor(acc[r], off, 64);
    if (lane < RPW) {
        const int n = n_base + lane;
        if (n < p.N) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < RPW; ++r)
                if (r == lane) v = acc[r];
            if (p.bias) v += p.bias[n];
            if ((p.flags & kGemmScaleCols) && n < p.scale_cols) v *= p.scale;
            if (p.flags & kGemmGelu) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
            if (p.flags & kGemmResidual) v += p.R[n];
            if (p.out_plain) p.out[n] = v;
            else st_sc1(p.out + n, v);
            if (p.kc && n >= p.kv_d) {
                if (n < 2 * p.kv_d) st_sc1(p.kc + p.kv_at + n - p.kv_d, v);
                else st_sc1(p.vc + p.kv_at + n - 2 * p.kv_d, v);
            }
        }
    }
}

__device__ __forceinline__ void gemv_phase(const GemvPhase& p, int wid, int n_waves, int lane, const Pref& pf) {
    float mean = 0.f, rstd = 0.f;
    if (p.ln_g) {   // layernorm_kernel's statistics order: lane-strided scalar partial sums, xor fold
        constexpr int kMax = 24;
        float v[kMax];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < kMax; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < p.K ? p.x[c] : 0.f;
            sum += v[i];
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
        mean = sum / (float)p.K;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < kMax; ++i) {
            const float t = (lane + 64 * i) < p.K ? v[i] - mean : 0.f;
            sq += t * t;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
        rstd = 1.0f / sqrtf(sq / (float)p.K + 1e-5f);
    }
    if (phase_rpw(p, n_waves) == 4) {
        for (int n0 = wid * 4, it = 0; n0 < p.N; n0 += n_waves * 4, ++it) gemv_rows<4>(p, n0, lane, mean, rstd, pf, it == 0);
    } else {
        for (int n0 = wid, it = 0; n0 < p.N; n0 += n_waves, ++it) gemv_rows<1>(p, n0, lane, mean, rstd, pf, it == 0);
    }
}

struct StepShared {
    float qs[64];
    float sc[448 + 64];
    float red[8];
    float part[16 * 64];
};

// decoder_self_attention_kernel for (row 0, head): q / newest k, v come from this launch (sc1-written, first read here)
__device__ __attribute__((noinline)) void self_attention_head(StepShared& sh, const float* qkv, const float* kc, const float* vc, float* out, int head,
                                    int d, int n_keys) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 15, kq = lane >> 4;
    if (tid < 64) sh.qs[tid] = qkv[head * 64 + tid];
    __syncthreads();
    const float4 q4 = reinterpret_cast<const float4*>(sh.qs)[sub];
    const float* kb = kc + head * 64 + sub * 4;
    const float* vb = vc + head * 64 + sub * 4;
    float mx = -INFINITY;
    for (int base = wave * 4; base < n_keys; base += 16 * 8) {
        float4 kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = base + 16 * u + kq;
            const bool ok = j < n_keys;
            const float4 t = *reinterpret_cast<const float4*>(kb + (long)(ok ? j : 0) * d);
            kk[u] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = base + 16 * u + kq;
            float acc = 0.f;
            acc = fmaf(q4.x, kk[u].x, acc);
            acc = fmaf(q4.y, kk[u].y, acc);
            acc = fmaf(q4.z, kk[u].z, acc);
            acc = fmaf(q4.w, kk[u].w, acc);
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            acc += __shfl_xor(acc, 4, 64);
            acc += __shfl_xor(acc, 8, 64);
            if (j < n_keys) {
                if (sub == 0) sh.sc[j] = acc;
                mx = fmaxf(mx, acc);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) sh.red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sh.red[0], sh.red[1]), fmaxf(sh.red[2], sh.red[3]));
    float sum = 0.f;
    for (int j = tid; j < n_keys; j += 256) {
        const float e = expf(sh.sc[j] - mx);
        sh.sc[j] = e;
        sum += e;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) sh.red[4 + wave] = sum;
    __syncthreads();
    sum = (sh.red[4] + sh.red[5]) + (sh.red[6] + sh.red[7]);
    for (int j = tid; j < n_keys; j += 256) sh.sc[j] = sh.sc[j] / sum;
    __syncthreads();
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = wave * 4; base < n_keys; base += 16 * 8) {
        float4 vv[8];
        float ww[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = base + 16 * u + kq;
            const bool ok = j < n_keys;
            vv[u] = *reinterpret_cast<const float4*>(vb + (long)(ok ? j : 0) * d);
            ww[u] = ok ? sh.sc[j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            o.x = fmaf(ww[u], vv[u].x, o.x);
            o.y = fmaf(ww[u], vv[u].y, o.y);
            o.z = fmaf(ww[u], vv[u].z, o.z);
            o.w = fmaf(ww[u], vv[u].w, o.w);
        }
    }
    reinterpret_cast<float4*>(sh.part)[(wave * 4 + kq) * 16 + sub] = o;
    __syncthreads();
    if (tid < 64) {
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc += sh.part[s * 64 + tid];
        st_sc1(out + head * 64 + tid, acc);
    }
    __syncthreads();
}

// cross_split_kernel for (row 0, head, key split ks); raw scores are kept only for alignment heads
__device__ __attribute__((noinline)) void cross_split_part(StepShared& sh, const float* q, const float* kx, const float* vx, long ldkv, int T, int head,
                                 int ks, bool keep_scores, float* scores, float* pm, float* pl, float* po) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 15, kq = lane >> 4;
    const int chunk = (T + kStepSplit - 1) / kStepSplit;
    const int k_lo = ks * chunk, k_hi = min(T, k_lo + chunk);
    if (tid < 64) sh.qs[tid] = q[head * 64 + tid];
    __syncthreads();
    const float4 q4 = reinterpret_cast<const float4*>(sh.qs)[sub];
    const float* kb = kx + head * 64 + sub * 4;
    const float* vb = vx + head * 64 + sub * 4;
    float* srow = scores + (long)head * T;
    float4 kk[kStepUnroll];
#pragma unroll
    for (int u = 0; u < kStepUnroll; ++u) {
        const int j = k_lo + wave * 4 + 16 * u + kq;
        const bool ok = j < k_hi;
        const float4 t = *reinterpret_cast<const float4*>(kb + (long)(ok ? j : k_lo) * ldkv);
        kk[u] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < kStepUnroll; ++u) {
        const int j = k_lo + wave * 4 + 16 * u + kq;
        float acc = 0.f;
        acc = fmaf(q4.x, kk[u].x, acc);
        acc = fmaf(q4.y, kk[u].y, acc);
        acc = fmaf(q4.z, kk[u].z, acc);
        acc = fmaf(q4.w, kk[u].w, acc);
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        acc += __shfl_xor(acc, 8, 64);
        if (j < k_hi) {
            if (sub == 0) {
                sh.sc[j - k_lo] = acc;
                if (keep_scores) st_sc1(srow + j, acc);
            }
            mx = fmaxf(mx, acc);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) sh.red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sh.red[0], sh.red[1]), fmaxf(sh.red[2], sh.red[3]));
    float sum = 0.f;
    for (int j = tid; j < k_hi - k_lo; j += 256) {
        const float e = expf(sh.sc[j] - mx);
        sh.sc[j] = e;
        sum += e;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) sh.red[4 + wave] = sum;
    __syncthreads();
    sum = (sh.red[4] + sh.red[5]) + (sh.red[6] + sh.red[7]);
    float4 vv[kStepUnroll];
#pragma unroll
    for (int u = 0; u < kStepUnroll; ++u) {
        const int j = k_lo + wave * 4 + 16 * u + kq;
        vv[u] = *reinterpret_cast<const float4*>(vb + (long)(j < k_hi ? j : k_lo) * ldkv);
    }
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < kStepUnroll; ++u) {
        const int j = k_lo + wave * 4 + 16 * u + kq;
        const float w = j < k_hi ? sh.sc[j - k_lo] : 0.f;
        o.x = fmaf(w, vv[u].x, o.x);
        o.y = fmaf(w, vv[u].y, o.y);
        o.z = fmaf(w, vv[u].z, o.z);
        o.w = fmaf(w, vv[u].w, o.w);
    }
    reinterpret_cast<float4*>(sh.part)[(wave * 4 + kq) * 16 + sub] = o;
    __syncthreads();
    const long slot = (long)head * kStepSplit + ks;
    if (tid < 64) {
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc += sh.part[s * 64 + tid];
        st_sc1(po + slot * 64 + tid, acc);
    }
    if (tid == 0) {
        st_sc1(pm + slot, mx);
        st_sc1(pl + slot, sum);
    }
    __syncthreads();
}

__device__ __forceinline__ size_t pad64(size_t n) { return (n + 63) / 64 * 64; }

__global__ __launch_bounds__(256) void decode_step_kernel(StepArgs a) {
    __shared__ __attribute__((aligned(16))) StepShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x, b = blockIdx.x;
    const int wid = wave * G + b, n_waves = G * 4;     // consecutive row groups land on different workgroups
    const int d = a.d, H = a.n_head, T = a.T;
    // which barrier word this launch uses; the other one is made ready for the next launch
    const unsigned launch_no = a.bar[2];
    unsigned* word = a.bar + (launch_no & 1u);
    if (b == 0 && tid == 0) a.bar[(launch_no & 1u) ^ 1u] = 0u;
    unsigned phase = 0;
    const int offset = *a.offset_p;
    const int n_keys = offset + 1;
    Pref pf;

    auto make_qkv = [&](const StepLayer& L, const float* x_in, float* qkv) {
        GemvPhase p;
        p.W = L.qkvw; p.bias = L.qkvb; p.N = 3 * d; p.K = d; p.x = x_in; p.ln_g = L.ln1w; p.ln_b = L.ln1b;
        p.flags = kGemmScaleCols; p.scale = a.scale; p.scale_cols = 2 * d; p.out = qkv;
        p.kc = L.kc; p.vc = L.vc; p.kv_d = d; p.kv_at = (long)offset * d;
        return p;
    };

    // phase 0: token + position embedding (embed_kernel); the first layer's qkv rows are fetched meanwhile
    float* x_in = a.scratch;
    float* lay = a.scratch + a.emb_floats;
    {
        const float* e = a.tok_emb + (long)a.tokens[0] * d;
        const float* pe = a.pos_emb + (long)offset * d;
        for (int c = b * 256 + tid; c < d; c += G * 256) st_sc1(x_in + c, e[c] + pe[c]);
    }
    GemvPhase p_next = make_qkv(a.layers[0], x_in, lay);
    gemv_prefetch(p_next, wid, n_waves, lane, pf);
    step_barrier(word, ++phase * G, a.err);

    for (int l = 0; l < a.n_layer; ++l, lay += a.layer_floats) {
        const StepLayer& L = a.layers[l];
        float* qkv = lay;
        float* att = qkv + 3 * d;
        float* x1 = att + d;
        float* q = x1 + d;
        float* pm = q + d;
        float* pl = pm + pad64((size_t)H * kStepSplit);
        float* po = pl + pad64((size_t)H * kStepSplit);
        float* scores = po + (size_t)H * kStepSplit * 64;
        float* x2 = scores + pad64((size_t)H * T);
        float* mlp = x2 + d;
        float* x3 = mlp + 4 * d;

        gemv_phase(p_next, wid, n_waves, lane, pf);            // x -> [q | k | v] (+ caches)
        GemvPhase p_out;
        p_out.W = L.outw; p_out.bias = L.outb; p_out.N = d; p_out.K = d; p_out.x = att; p_out.flags = kGemmResidual;
        p_out.R = x_in; p_out.out = x1;
        step_barrier(word, ++phase * G, a.err);
        if (b < H) self_attention_head(sh, qkv, L.kc, L.vc, att, b, d, n_keys);
        gemv_prefetch(p_out, wid, n_waves, lane, pf);
        step_barrier(word, ++phase * G, a.err);
        gemv_phase(p_out, wid, n_waves, lane, pf);             // self-attention out projection + residual
        GemvPhase p_xq;
        p_xq.W = L.xqw; p_xq.bias = L.xqb; p_xq.N = d; p_xq.K = d; p_xq.x = x1; p_xq.ln_g = L.lnxw; p_xq.ln_b = L.lnxb;
        p_xq.flags = kGemmScaleCols; p_xq.scale = a.scale; p_xq.scale_cols = d; p_xq.out = q;
        gemv_prefetch(p_xq, wid, n_waves, lane, pf);
        step_barrier(word, ++phase * G, a.err);
        gemv_phase(p_xq, wid, n_waves, lane, pf);              // cross-attention query
        GemvPhase p_xo;   // merged cross-attention output -> out projection (+ residual); the merge is the operand load
        p_xo.W = L.xoutw; p_xo.bias = L.xoutb; p_xo.N = d; p_xo.K = d; p_xo.x = nullptr; p_xo.pm = pm; p_xo.pl = pl;
        p_xo.po = po; p_xo.flags = kGemmResidual; p_xo.R = x1; p_xo.out = x2;
        step_barrier(word, ++phase * G, a.err);
        for (int w = b; w < H * kStepSplit; w += G) {
            const int head = w / kStepSplit, ks = w - head * kStepSplit;
            const bool keep = L.head_rank && L.head_rank[head] >= 0;
            cross_split_part(sh, q, L.xk, L.xv, a.ldkv, T, head, ks, keep, scores, pm, pl, po);
        }
        gemv_prefetch(p_xo, wid, n_waves, lane, pf);
        step_barrier(word, ++phase * G, a.err);
        gemv_phase(p_xo, wid, n_waves, lane, pf);
        // side job of the last workgroups: softmax rows of this layer's alignment heads into the alignment window
        // (cross_merge_kernel's tail); consumed by later KERNELS, so plain stores
        if (L.head_rank) {
            int k = G - 1 - b, found = -1;
            for (int h = 0; h < H; ++h)
                if (L.head_rank[h] >= 0 && k-- == 0) found = h;
            if (found >= 0) {
                const long base = (long)found * kStepSplit;
                float M = pm[base];
#pragma unroll
                for (int s = 1; s < kStepSplit; ++s) M = fmaxf(M, pm[base + s]);
                float Lsum = 0.f;
#pragma unroll
                for (int s = 0; s < kStepSplit; ++s) Lsum += pl[base + s] * expf(pm[base + s] - M);
                float* dst = a.ring + (((long)L.head_rank[found] * a.n_beam + a.beam_of_row[0]) * a.ring_rows + a.ring_row[0]) * T;
                const float* srow = scores + (long)found * T;
                for (int j = tid; j < T; j += 256) dst[j] = expf(srow[j] - M) / Lsum;
            }
        }
        GemvPhase p_fc1;
        p_fc1.W = L.fc1w; p_fc1.bias = L.fc1b; p_fc1.N = 4 * d; p_fc1.K = d; p_fc1.x = x2; p_fc1.ln_g = L.ln2w;
        p_fc1.ln_b = L.ln2b; p_fc1.flags = kGemmGelu; p_fc1.out = mlp;
        gemv_prefetch(p_fc1, wid, n_waves, lane, pf);
        step_barrier(word, ++phase * G, a.err);
        gemv_phase(p_fc1, wid, n_waves, lane, pf);
        const bool last = l + 1 == a.n_layer;
        GemvPhase p_fc2;
        p_fc2.W = L.fc2w; p_fc2.bias = L.fc2b; p_fc2.N = d; p_fc2.K = 4 * d; p_fc2.x = mlp; p_fc2.flags = kGemmResidual;
        p_fc2.R = x2; p_fc2.out = last ? a.x_out : x3; p_fc2.out_plain = last;
        gemv_prefetch(p_fc2, wid, n_waves, lane, pf);
        step_barrier(word, ++phase * G, a.err);
        gemv_phase(p_fc2, wid, n_waves, lane, pf);
        if (!last) {
            p_next = make_qkv(a.layers[l + 1], x3, lay + a.layer_floats);
            gemv_prefetch(p_next, wid, n_waves, lane, pf);
            step_barrier(word, ++phase * G, a.err);
        }
        x_in = x3;
    }
    if (b == 0 && tid == 0) a.bar[2] = launch_no + 1u;
}

static size_t step_layer_floats(int d, int n_head, int T) {
    auto pad = [](size_t n) { return (n + 63) / 64 * 64; };
    return (size_t)3 * d + d + d + d + 2 * pad((size_t)n_head * kStepSplit) + (size_t)n_head * kStepSplit * 64 +
           pad((size_t)n_head * T) + d + 4 * (size_t)d + d;
}

size_t step_scratch_floats(int n_layer, int d, int n_head, int T) {
    return (size_t)d + (size_t)n_layer * step_layer_floats(d, n_head, T) + 64;
}

void launch_decode_step(const LaunchCtx& ctx, StepArgs a, int n_blocks) {
    if (a.d % 64 != 0 || a.d > 1536) throw std::invalid_argument("decode step kernel: unsupported width");
    if ((a.T + kStepSplit - 1) / kStepSplit > kStepUnroll * 16) throw std::invalid_argument("decode step kernel: T too large");
    a.emb_floats = a.d;
    a.layer_floats = step_layer_floats(a.d, a.n_head, a.T);
    // algorithmic bytes: every decoder-layer weight once + the cross K/V of every layer
    KernelScope ks(ctx, "dec_step_persistent", 2.0 * a.n_layer * 12.0 * a.d * a.d,
                   4.0 * a.n_layer * (12.0 * a.d * a.d + 2.0 * a.T * a.d));
    hipLaunchKernelGGL(decode_step_kernel, dim3(n_blocks), dim3(256), 0, ctx.stream, a);
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

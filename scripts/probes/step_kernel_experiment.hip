// EXPERIMENT, NOT BUILT INTO libwlk_hip.so (kept as groundwork; see DESIGN.md "Where the time goes").
// v3 (this file): the step lives on ONE XCD (32 member workgroups found by a run-time census of HW_REG_XCC_ID, plain
// stores through that XCD's L2, phase-table interpreter so the kernel needs 332 VGPRs and no scratch).  Measured on
// MI355X, base.en: BIT-IDENTICAL to the one-kernel-per-op path on 5 golden streams, 410 us per step against ~330 us
// (113 vs 135.5 audio-s/s in the same gpurun call); with 8 streams per GPU the census / barrier waits time out
// (workgroups that need a whole CU do not become resident next to other streams' kernels).
// Known flaw of this version: step_barrier's `s_waitcnt vmcnt(0)` (needed to drain the phase's stores) also waits for
// the NEXT phase's prefetched weight rows, so the ~2 us HBM latency of the prefetch is exposed before every barrier;
// the fix is arrive -> prefetch -> poll.  Even so the remaining per-phase chain (activation read, LayerNorm
// statistics, reduce, store, arrive) is ~6 us, i.e. what a kernel boundary costs.
// History: v1 (cross-XCD, sc1 stores) read stale data because phase buffers shared cache lines; v2 (padded buffers,
// weight prefetch) was bit-identical at 419 us.  Needs the StepLayer / StepPhase / StepArgs declarations and the
// api.hip hook (see git history) to compile.
// Persistent decode-step kernel (beam 1, one fed token): every decoder layer of TextDecoder.forward
// (whisper/model.py:279-332 via AlignAtt._get_logits_and_cross_attn, simul_whisper.py:357-368) in ONE launch that
// lives on ONE XCD.
//
// A decode step is a chain of ~50 dependent matrix-vector products and two small attentions over a 512..1024-float
// activation; as separate kernels every link costs a ~5.5 us kernel boundary whatever its work (330 us per step for
// base.en, of which the memory time of 112 MB is 85 us even through a single XCD's 1.3 TB/s port).  Here the 32
// workgroups that land on one XCD stay resident and hand the activation from phase to phase through that XCD's
// shared L2 and a grid barrier (scripts/probes/barrier_probe.hip, mode 2: 1.5 us per phase, 0 stale values):
//   * 256 workgroups are launched; each reads HW_REG_XCC_ID, reports, and leaves unless it is on the target XCD -
//     membership and ranks are COUNTED at run time, placement is never assumed;
//   * every phase writes its result with plain stores (the vector L1 is write-through, the line stays in the XCD's
//     L2) into a buffer that no member has read earlier in this launch - padded to 256 bytes, because "never read"
//     has to hold per cache line - so consumers use plain loads; no fences anywhere;
//   * arrive = s_waitcnt vmcnt(0) + one relaxed agent atomicAdd; wait = bounded relaxed polling (a timeout sets an
//     error word the host checks);
//   * weights do not depend on activations: each wave fetches the first rows of its NEXT phase (with the LayerNorm
//     affine, bias and residual operands) before it waits at the barrier, and inside a phase the next row group is
//     in flight while the current one is multiplied;
//   * barrier / census words come in two sets that alternate between launches (a device-side launch counter picks
//     one, the other is re-zeroed), so the captured hipGraph node needs no memset node.
// Arithmetic is the existing kernels' (gemv1_f32_kernel, decoder_self_attention_kernel, cross_split_kernel,
// cross_merge_kernel) in the same order: logits and alignment rows are bit-identical to the one-kernel-per-op path.
#include <vector>

#include "common.h"

namespace wlk {

constexpr int kStepSplit = 8;       // key splits of the cross-attention (== kCrossSplit of decoder.hip)
constexpr int kStepUnroll = 12;

// phase outputs: plain stores - write-through L1, the line stays in this XCD's L2 where every member reads it
__device__ __forceinline__ void st_sc1(float* p, float v) { *p = v; }

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

typedef StepPhase GemvPhase;   // the GEMV fields of a phase descriptor

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


// One work item of a GEMV phase = (row group, K chunk): what a wave can fetch before the activations exist
template <int kCH>   // float4 chunks of K in flight per lane: ceil(d / 256), so one item covers K = d
struct Pref {
    float4 w[kCH][4];
    float4 ga[kCH], be[kCH];
    float bias_v, res_v;
};

template <int RPW, int kCH>
__device__ __forceinline__ void load_item(const GemvPhase& p, int n_base, int c0, int lane, Pref<kCH>& pf) {
    const int K4 = p.K >> 2;
#pragma unroll
    for (int u = 0; u < kCH; ++u) {
        const int c = c0 + lane + 64 * u;
        const int cc = (c < K4 ? c : 0) * 4;
#pragma unroll
        for (int r = 0; r < RPW; ++r)
            pf.w[u][r] = *reinterpret_cast<const float4*>(p.W + (long)min(n_base + r, p.N - 1) * p.K + cc);
        if (p.ln_g) {
            pf.ga[u] = *reinterpret_cast<const float4*>(p.ln_g + cc);
            pf.be[u] = *reinterpret_cast<const float4*>(p.ln_b + cc);
        }
    }
    const int n = min(n_base + (lane < RPW ? lane : 0), p.N - 1);
    pf.bias_v = p.bias ? p.bias[n] : 0.f;
    pf.res_v = (p.flags & kGemmResidual) ? p.R[n] : 0.f;
}

__device__ __forceinline__ int phase_rpw(const GemvPhase& p, int n_waves) { return p.N > n_waves ? 4 : 1; }

template <int kCH>
__device__ __forceinline__ void gemv_prefetch(const GemvPhase& p, int wid, int n_waves, int lane, Pref<kCH>& pf) {
    if (phase_rpw(p, n_waves) == 4) {
        if (wid * 4 < p.N) load_item<4, kCH>(p, wid * 4, 0, lane, pf);
    } else {
        if (wid < p.N) load_item<1, kCH>(p, wid, 0, lane, pf);
    }
}

// All row groups of this wave; same reduction / fmaf order per row as gemv1_f32_kernel.  `pf` holds the first item.
template <int RPW, int kCH>
__device__ __forceinline__ void gemv_items(const GemvPhase& p, int wid, int n_waves, int lane, float mean, float rstd,
                                           Pref<kCH>& pf) {
    const int K4 = p.K >> 2;
    const int n_chunks = (K4 + 64 * kCH - 1) / (64 * kCH);
    const int n_groups = (p.N + RPW - 1) / RPW;
    float acc[RPW];
    int g = wid, ch = 0;
    bool have = g < n_groups;
    while (have) {
        int g2 = g, ch2 = ch + 1;
        if (ch2 == n_chunks) { ch2 = 0; g2 = g + n_waves; }
        const bool have2 = g2 < n_groups;
        Pref<kCH> nx;
        if (have2) load_item<RPW, kCH>(p, g2 * RPW, ch2 * 64 * kCH, lane, nx);   // next item in flight during this one
        if (ch == 0) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) acc[r] = 0.f;
        }
        const int c0 = ch * 64 * kCH;
        float4 x[kCH];
#pragma unroll
        for (int u = 0; u < kCH; ++u) {
            const int c = c0 + lane + 64 * u;
            const bool ok = c < K4;
            x[u] = ok ? phase_x4(p, c) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.ln_g) {
                x[u].x = (x[u].x - mean) * rstd * pf.ga[u].x + pf.be[u].x;
                x[u].y = (x[u].y - mean) * rstd * pf.ga[u].y + pf.be[u].y;
                x[u].z = (x[u].z - mean) * rstd * pf.ga[u].z + pf.be[u].z;
                x[u].w = (x[u].w - mean) * rstd * pf.ga[u].w + pf.be[u].w;
            }
        }
#pragma unroll
        for (int u = 0; u < kCH; ++u) {
            if (c0 + lane + 64 * u < K4) {
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    acc[r] = fmaf(pf.w[u][r].x, x[u].x, acc[r]);
                    acc[r] = fmaf(pf.w[u][r].y, x[u].y, acc[r]);
                    acc[r] = fmaf(pf.w[u][r].z, x[u].z, acc[r]);
                    acc[r] = fmaf(pf.w[u][r].w, x[u].w, acc[r]);
                }
            }
        }
        if (ch == n_chunks - 1) {
            float red[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                red[r] = acc[r];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) red[r] += __shfl_xor(red[r], off, 64);
            }
            const int n = g * RPW + lane;
            if (lane < RPW && n < p.N) {
                float v = 0.f;
#pragma unroll
                for (int r = 0; r < RPW; ++r)
                    if (r == lane) v = red[r];
                if (p.bias) v += pf.bias_v;
                if ((p.flags & kGemmScaleCols) && n < p.scale_cols) v *= p.scale;
                if (p.flags & kGemmGelu) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                if (p.flags & kGemmResidual) v += pf.res_v;
                st_sc1(p.out + n, v);
                if (p.kc && n >= p.kv_d) {
                    if (n < 2 * p.kv_d) st_sc1(p.kc + p.kv_at + n - p.kv_d, v);
                    else st_sc1(p.vc + p.kv_at + n - 2 * p.kv_d, v);
                }
            }
        }
        pf = nx;
        g = g2; ch = ch2; have = have2;
    }
}

template <int kCH>
__device__ __forceinline__ void gemv_phase(const GemvPhase& p, int wid, int n_waves, int lane, Pref<kCH>& pf) {
    float mean = 0.f, rstd = 0.f;
    if (p.ln_g) {   // layernorm_kernel's statistics order: lane-strided scalar partial sums, xor fold
        constexpr int kMax = 4 * kCH;   // K = d <= 256 kCH in every LayerNorm-fused phase
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
    if (phase_rpw(p, n_waves) == 4) gemv_items<4, kCH>(p, wid, n_waves, lane, mean, rstd, pf);
    else gemv_items<1, kCH>(p, wid, n_waves, lane, mean, rstd, pf);
}

struct StepShared {
    float qs[64];
    float sc[448 + 64];
    float red[8];
    float part[16 * 64];
};

// decoder_self_attention_kernel for (row 0, head): q / newest k, v come from this launch (sc1-written, first read here)
__device__ __forceinline__ void self_attention_head(StepShared& sh, const float* qkv, const float* kc, const float* vc, float* out, int head,
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

// cross_split_kernel for (row 0, head, key split ks) in two parts: the key rows can be fetched before q exists
__device__ __forceinline__ void cross_load_keys(float4 (&kk)[kStepUnroll], const float* kx, long ldkv, int T, int head, int ks,
                                                bool active) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 15, kq = lane >> 4;
    const int chunk = (T + kStepSplit - 1) / kStepSplit;
    const int k_lo = ks * chunk, k_hi = min(T, k_lo + chunk);
    const float* kb = kx + head * 64 + sub * 4;
#pragma unroll
    for (int u = 0; u < kStepUnroll; ++u) {
        const int j = k_lo + wave * 4 + 16 * u + kq;
        const bool ok = active && j < k_hi;
        const float4 t = *reinterpret_cast<const float4*>(kb + (long)(ok ? j : (active ? k_lo : 0)) * ldkv);
        kk[u] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// raw scores are kept only for alignment heads
__device__ __forceinline__ void cross_split_part(StepShared& sh, const float4 (&kk)[kStepUnroll], const float* q,
                                                           const float* vx, long ldkv, int T, int head, int ks,
                                                           bool keep_scores, float* scores, float* pm, float* pl, float* po) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 15, kq = lane >> 4;
    const int chunk = (T + kStepSplit - 1) / kStepSplit;
    const int k_lo = ks * chunk, k_hi = min(T, k_lo + chunk);
    if (tid < 64) sh.qs[tid] = q[head * 64 + tid];
    __syncthreads();
    const float4 q4 = reinterpret_cast<const float4*>(sh.qs)[sub];
    const float* vb = vx + head * 64 + sub * 4;
    float* srow = scores + (long)head * T;
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

// words per set: [0] barrier, [1] members on the target XCD, [2] workgroups that have reported, [3] rank tickets
constexpr int kWordSet = 4;

// The kernel is a small interpreter over the phase table the host built (build_step_table): ONE instance of each
// phase type's code, so the register allocation stays sane (a fully unrolled layer body needs > 512 VGPRs).
template <int kCH>
__global__ __launch_bounds__(256) void decode_step_kernel(StepArgs a) {
    __shared__ __attribute__((aligned(16))) StepShared sh;
    __shared__ unsigned census[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = a.d, H = a.n_head, T = a.T;
    const unsigned launch_no = a.bar[2 * kWordSet];
    unsigned* words = a.bar + (launch_no & 1u) * kWordSet;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const bool member = (int)(xcc & 0xF) == a.xcd;
    if (tid == 0) {
        if (member) census[1] = __hip_atomic_fetch_add(words + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (member) __hip_atomic_fetch_add(words + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(words + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!member) return;
    if (tid == 0) {   // membership is final once every launched workgroup has reported
        int spins = 0;
        while (__hip_atomic_load(words + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { *a.err = 1; break; }
        }
        census[0] = __hip_atomic_load(words + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int G = (int)census[0], b = (int)census[1];     // members of this launch, my rank among them
    if (b == 0 && tid == 0) {   // the other word set is the next launch's: make it ready
        unsigned* other = a.bar + ((launch_no & 1u) ^ 1u) * kWordSet;
        other[0] = 0u; other[1] = 0u; other[2] = 0u; other[3] = 0u;
    }
    const int wid = wave * G + b, n_waves = G * 4;         // consecutive row groups land on different workgroups
    unsigned phase = 0;
    const int offset = *a.offset_p;
    const int n_keys = offset + 1;
    const long kv_at = (long)offset * d;
    Pref<kCH> pf;

    // phase 0: token + position embedding (embed_kernel) into the first buffer
    {
        const float* e = a.tok_emb + (long)a.tokens[0] * d;
        const float* pe = a.pos_emb + (long)offset * d;
        for (int c = b * 256 + tid; c < d; c += G * 256) st_sc1(a.scratch + c, e[c] + pe[c]);
    }
    StepPhase cur = a.phases[0];
    cur.kv_at = kv_at;
    if (cur.kind == kPhaseGemv) gemv_prefetch(cur, wid, n_waves, lane, pf);
    for (int ph = 0; ph < a.n_phases; ++ph) {
        step_barrier(words, ++phase * G, a.err);
        if (cur.kind == kPhaseGemv) {
            gemv_phase(cur, wid, n_waves, lane, pf);
            if (cur.side_scores) {   // spare members: softmax rows of this layer's alignment heads (cross_merge_kernel's tail)
                for (int k0 = G - 1 - b; k0 < H; k0 += G) {
                    int k = k0, found = -1;
                    for (int h = 0; h < H; ++h)
                        if (cur.head_rank[h] >= 0 && k-- == 0) found = h;
                    if (found < 0) break;
                    const long base = (long)found * kStepSplit;
                    float M = cur.pm[base];
#pragma unroll
                    for (int s = 1; s < kStepSplit; ++s) M = fmaxf(M, cur.pm[base + s]);
                    float Lsum = 0.f;
#pragma unroll
                    for (int s = 0; s < kStepSplit; ++s) Lsum += cur.pl[base + s] * expf(cur.pm[base + s] - M);
                    float* dst = a.ring + (((long)cur.head_rank[found] * a.n_beam + a.beam_of_row[0]) * a.ring_rows + a.ring_row[0]) * T;
                    const float* srow = cur.side_scores + (long)found * T;
                    for (int j = tid; j < T; j += 256) dst[j] = expf(srow[j] - M) / Lsum;
                }
            }
        } else if (cur.kind == kPhaseSelfAttn) {
            for (int h = b; h < H; h += G) self_attention_head(sh, cur.x, cur.kc, cur.vc, cur.out, h, d, n_keys);
        } else {
            for (int w = b; w < H * kStepSplit; w += G) {
                const int head = w / kStepSplit, ks = w - head * kStepSplit;
                const bool keep = cur.head_rank && cur.head_rank[head] >= 0;
                float4 kk[kStepUnroll];
                cross_load_keys(kk, cur.xk, a.ldkv, T, head, ks, true);
                cross_split_part(sh, kk, cur.x, cur.xv, a.ldkv, T, head, ks, keep, cur.scores, cur.pm_w, cur.pl_w, cur.po_w);
            }
        }
        if (ph + 1 < a.n_phases) {
            cur = a.phases[ph + 1];
            cur.kv_at = kv_at;
            if (cur.kind == kPhaseGemv) gemv_prefetch(cur, wid, n_waves, lane, pf);
        }
    }
    if (b == 0 && tid == 0) a.bar[2 * kWordSet] = launch_no + 1u;
}

static size_t pad64h(size_t n) { return (n + 63) / 64 * 64; }

static size_t step_layer_floats(int d, int n_head, int T) {
    return (size_t)3 * d + d + d + d + 2 * pad64h((size_t)n_head * kStepSplit) + (size_t)n_head * kStepSplit * 64 +
           pad64h((size_t)n_head * T) + d + 4 * (size_t)d + d;
}

size_t step_scratch_floats(int n_layer, int d, int n_head, int T) {
    return (size_t)d + (size_t)n_layer * step_layer_floats(d, n_head, T) + 64;
}

// The phase table of one session: 8 phases per layer over the scratch layout above (all pointers are device pointers)
std::vector<StepPhase> build_step_table(const std::vector<StepLayer>& layers, int d, int n_head, int T, float scale,
                                        float* scratch, float* x_out) {
    std::vector<StepPhase> t;
    const float* x_in = scratch;
    float* lay = scratch + d;
    const int H = n_head;
    for (size_t l = 0; l < layers.size(); ++l, lay += step_layer_floats(d, n_head, T)) {
        const StepLayer& L = layers[l];
        float* qkv = lay;
        float* att = qkv + 3 * d;
        float* x1 = att + d;
        float* q = x1 + d;
        float* pm = q + d;
        float* pl = pm + pad64h((size_t)H * kStepSplit);
        float* po = pl + pad64h((size_t)H * kStepSplit);
        float* scores = po + (size_t)H * kStepSplit * 64;
        float* x2 = scores + pad64h((size_t)H * T);
        float* mlp = x2 + d;
        float* x3 = mlp + 4 * d;
        const bool last = l + 1 == layers.size();
        StepPhase p;
        p = StepPhase(); p.kind = kPhaseGemv;      // x -> [q | k | v] (+ caches)
        p.W = L.qkvw; p.bias = L.qkvb; p.N = 3 * d; p.K = d; p.x = x_in; p.ln_g = L.ln1w; p.ln_b = L.ln1b;
        p.flags = kGemmScaleCols; p.scale = scale; p.scale_cols = 2 * d; p.out = qkv; p.kc = L.kc; p.vc = L.vc; p.kv_d = d;
        t.push_back(p);
        p = StepPhase(); p.kind = kPhaseSelfAttn; p.x = qkv; p.kc = L.kc; p.vc = L.vc; p.out = att;
        t.push_back(p);
        p = StepPhase(); p.kind = kPhaseGemv;      // self-attention out projection + residual
        p.W = L.outw; p.bias = L.outb; p.N = d; p.K = d; p.x = att; p.flags = kGemmResidual; p.R = x_in; p.out = x1;
        t.push_back(p);
        p = StepPhase(); p.kind = kPhaseGemv;      // cross-attention query
        p.W = L.xqw; p.bias = L.xqb; p.N = d; p.K = d; p.x = x1; p.ln_g = L.lnxw; p.ln_b = L.lnxb;
        p.flags = kGemmScaleCols; p.scale = scale; p.scale_cols = d; p.out = q;
        t.push_back(p);
        p = StepPhase(); p.kind = kPhaseCross; p.x = q; p.xk = L.xk; p.xv = L.xv; p.head_rank = L.head_rank;
        p.scores = scores; p.pm_w = pm; p.pl_w = pl; p.po_w = po;
        t.push_back(p);
        p = StepPhase(); p.kind = kPhaseGemv;      // merged cross-attention output -> out projection + residual
        p.W = L.xoutw; p.bias = L.xoutb; p.N = d; p.K = d; p.x = nullptr; p.pm = pm; p.pl = pl; p.po = po;
        p.flags = kGemmResidual; p.R = x1; p.out = x2; p.head_rank = L.head_rank; p.side_scores = L.head_rank ? scores : nullptr;
        t.push_back(p);
        p = StepPhase(); p.kind = kPhaseGemv;
        p.W = L.fc1w; p.bias = L.fc1b; p.N = 4 * d; p.K = d; p.x = x2; p.ln_g = L.ln2w; p.ln_b = L.ln2b; p.flags = kGemmGelu;
        p.out = mlp;
        t.push_back(p);
        p = StepPhase(); p.kind = kPhaseGemv;
        p.W = L.fc2w; p.bias = L.fc2b; p.N = d; p.K = 4 * d; p.x = mlp; p.flags = kGemmResidual; p.R = x2;
        p.out = last ? x_out : x3;
        t.push_back(p);
        x_in = x3;
    }
    return t;
}

void launch_decode_step(const LaunchCtx& ctx, StepArgs a, int n_blocks) {
    if (a.d % 64 != 0 || a.d > 1024) throw std::invalid_argument("decode step kernel: unsupported width");
    if ((a.T + kStepSplit - 1) / kStepSplit > kStepUnroll * 16) throw std::invalid_argument("decode step kernel: T too large");
    // algorithmic bytes: every decoder-layer weight once + the cross K/V of every layer
    const double layers = a.n_phases / 8.0;
    KernelScope ks(ctx, "dec_step_persistent", 2.0 * layers * 12.0 * a.d * a.d,
                   4.0 * layers * (12.0 * a.d * a.d + 2.0 * a.T * a.d));
    const int ch = (a.d + 255) / 256;   // n_blocks: 8 x (members wanted)
    if (ch == 1) hipLaunchKernelGGL(decode_step_kernel<1>, dim3(n_blocks), dim3(256), 0, ctx.stream, a);
    else if (ch == 2) hipLaunchKernelGGL(decode_step_kernel<2>, dim3(n_blocks), dim3(256), 0, ctx.stream, a);
    else if (ch == 3) hipLaunchKernelGGL(decode_step_kernel<3>, dim3(n_blocks), dim3(256), 0, ctx.stream, a);
    else hipLaunchKernelGGL(decode_step_kernel<4>, dim3(n_blocks), dim3(256), 0, ctx.stream, a);
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

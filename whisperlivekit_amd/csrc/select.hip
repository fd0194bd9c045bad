// Token selection and AlignAtt read-out - the device half of one decode step's tail:
//   * logit adjustments (suppression = add -inf, DRY penalty = subtract)       a6
//   * log_softmax over the vocabulary + top-k                                   a7
//   * softmax probability of one token (no-speech check)                        a6
//   * AlignAtt: column mean/std over the windowed attention rows -> z-score of the newest row
//     -> width-7 median filter (reflect) -> mean over alignment heads -> arg-max over
//     [0, content_mel_len)                                                      a8
// Reference: whisper/decoding.py:317-338,427-432; simul_whisper/simul_whisper.py:370-437;
// whisper/timing.py:19-54.  Everything here is a small HBM/L2-bound reduction; wavefront
// shuffles do the folding and nothing is copied to the host except k+1 numbers per beam row.
#include <atomic>
#include <cstdlib>

#include "common.h"
#include "wave_ops.h"
#include "align_body.h"

namespace wlk {

constexpr int kSelThreads = 1024;
constexpr int kMaxTopK = 8;

__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, red[w]);
    return r;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += red[w];
    return r;
}

// log_softmax(x)[i] = (x_i - max) - log(sum exp(x - max)) and top-k, in two stages so the 51864-wide
// row is swept by kSelBlocks workgroups at once: stage 1 leaves per-slice (max, sum exp(x - max_slice),
// k best (value, index)); stage 2 folds the slices.  Ties resolve to the lowest index in both stages.
constexpr int kSelBlocks = 64;
struct SelPartial {
    float mx, sum;
    float v[kMaxTopK];
    int i[kMaxTopK];
};

struct TopkArgs {
    float* logits;
    int n_vocab, k, n_rows;
    SelPartial* parts;
    const int* adj_row;
    const int* adj_ids;
    const float* adj_deltas;
    int n_adj;
    float* top_vals;
    int* top_ids;
    StepHostOut host;   // single-session graph steps: n_adj comes from the device block, results also go to the host
    // first step of an infer: the no-speech probability of the sot rows rides in the second launch (n_rows more blocks)
    const float* ns_logits = nullptr;
    int ns_token = 0;
    float* ns_probs = nullptr;
};

__device__ __forceinline__ void topk_stage1_stream(float* __restrict__ logits, int n_vocab, int k,
                                                 SelPartial* __restrict__ parts, const int* __restrict__ adj_row,
                                                 const int* __restrict__ adj_ids, const float* __restrict__ adj_deltas,
                                                 int n_adj, int slice, int row) {
    __shared__ float red[16];
    __shared__ float cand_v[4];
    __shared__ int cand_i[4];
    __shared__ int taken[kMaxTopK];
    const int tid = threadIdx.x;
    const int per = (n_vocab + kSelBlocks - 1) / kSelBlocks;
    const int lo = slice * per;
    const int hi = min(n_vocab, lo + per);
    float* x = logits + (long)row * n_vocab;
    // logit adjustments that fall into this slice are applied here (each slice is owned by one
    // workgroup, so a workgroup barrier is all the ordering that is needed); ids are unique per row
    for (int i = tid; i < n_adj; i += 256) {
        const int id = adj_ids[i];
        const int r = adj_row[i];
        if (id >= lo && id < hi && (r < 0 || r == row)) x[id] += adj_deltas[i];
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int i = lo + tid; i < hi; i += 256) mx = fmaxf(mx, x[i]);
    mx = block_max(mx, red);
    float sum = 0.f;
    if (mx > -INFINITY)
        for (int i = lo + tid; i < hi; i += 256) sum += expf(x[i] - mx);
    sum = block_sum(sum, red);
    SelPartial* out = parts + (long)row * kSelBlocks + slice;
    if (tid == 0) { out->mx = mx; out->sum = sum; }
    for (int round = 0; round < k; ++round) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = lo + tid; i < hi; i += 256) {
            bool skip = false;
            for (int t = 0; t < round; ++t) skip |= (taken[t] == i);
            const float v = x[i];
            if (!skip && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
        }
        wave_argmax(bv, bi);
        __syncthreads();
        if ((tid & 63) == 0) { cand_v[tid >> 6] = bv; cand_i[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (cand_v[w] > bv || (cand_v[w] == bv && cand_i[w] < bi)) { bv = cand_v[w]; bi = cand_i[w]; }
            taken[round] = bi;
            out->v[round] = bv;
            out->i[round] = bi;
        }
        __syncthreads();
    }
}

// The same slice pass with the slice held in registers (slices of up to 1024 logits: every Whisper vocabulary): ONE read
// of the row instead of k + 2 passes over it (each pass of the streaming form is a memory round trip on a kernel that
// starts with a cold L2), the logit adjustments ride through LDS to the thread that owns the element, and the
// wave-level folds are VALU butterflies (wave_ops.h).  Max, exp-sum (same per-thread order: ascending index) and the
// index-tie-broken arg-max rounds give what the streaming form gives, bit for bit.
constexpr int kSelKeep = 4;
__device__ __forceinline__ void topk_stage1_body(float* __restrict__ logits, int n_vocab, int k,
                                                 SelPartial* __restrict__ parts, const int* __restrict__ adj_row,
                                                 const int* __restrict__ adj_ids, const float* __restrict__ adj_deltas,
                                                 int n_adj, int slice, int row) {
    const int per = (n_vocab + kSelBlocks - 1) / kSelBlocks;
    if (per > 256 * kSelKeep) {
        topk_stage1_stream(logits, n_vocab, k, parts, adj_row, adj_ids, adj_deltas, n_adj, slice, row);
        return;
    }
    __shared__ float red[16];
    __shared__ float cand_v[4];
    __shared__ int cand_i[4];
    __shared__ float adj_buf[256 * kSelKeep];
    const int tid = threadIdx.x;
    const int lo = slice * per;
    const int hi = min(n_vocab, lo + per);
    float* x = logits + (long)row * n_vocab;
    float xv[kSelKeep];
    // (an EMPTY trailing slice - 63 * ceil(V / 64) >= V, i.e. vocabularies below ~4000 tokens - has lo >= n_vocab: the
    // fallback element is clamped into the row so nothing is read behind it; its lanes all take the i >= hi path below)
    const int lo_safe = lo < n_vocab ? lo : n_vocab - 1;
#pragma unroll
    for (int j = 0; j < kSelKeep; ++j) {
        const int i = lo + tid + 256 * j;
        xv[j] = x[i < hi ? i : lo_safe];
        adj_buf[tid + 256 * j] = 0.f;
    }
    __syncthreads();
    // adjustments that fall into this slice (ids are unique per row): delivered to the owner of the element; a thread's
    // three operands are requested together (clamped index), not one dependent load after the other
    for (int i0 = 0; i0 < n_adj; i0 += 256) {
        const int i = i0 + tid, ic = i < n_adj ? i : 0;
        const int id = adj_ids[ic];
        const int r = adj_row[ic];
        const float dl = adj_deltas[ic];
        if (i < n_adj && id >= lo && id < hi && (r < 0 || r == row)) adj_buf[id - lo] = dl;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kSelKeep; ++j) {
        const int i = lo + tid + 256 * j;
        const float a = adj_buf[tid + 256 * j];
        if (a != 0.f && i < hi) {
            xv[j] += a;
            x[i] = xv[j];          // the row stays adjusted in memory, as wlk_select's contract says
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < kSelKeep; ++j)
        if (lo + tid + 256 * j < hi) mx = fmaxf(mx, xv[j]);
    mx = block_max(mx, red);
    float sum = 0.f;
    if (mx > -INFINITY) {
#pragma unroll
        for (int j = 0; j < kSelKeep; ++j)
            if (lo + tid + 256 * j < hi) sum += expf(xv[j] - mx);
    }
    sum = block_sum(sum, red);
    SelPartial* out = parts + (long)row * kSelBlocks + slice;
    if (tid == 0) { out->mx = mx; out->sum = sum; }
    int taken = -1;                      // round r skips what rounds < r took: every thread tracks the winners itself
    int taken_all[kMaxTopK];
#pragma unroll
    for (int t = 0; t < kMaxTopK; ++t) taken_all[t] = -1;
    for (int round = 0; round < k; ++round) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < kSelKeep; ++j) {
            const int i = lo + tid + 256 * j;
            if (i < hi) {
                bool skip = false;
#pragma unroll
                for (int t = 0; t < kMaxTopK; ++t) skip |= (t < round && taken_all[t] == i);
                const float v = xv[j];
                if (!skip && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
            }
        }
        wave_argmax(bv, bi);
        __syncthreads();
        if ((tid & 63) == 0) { cand_v[tid >> 6] = bv; cand_i[tid >> 6] = bi; }
        __syncthreads();
        bv = cand_v[0];
        bi = cand_i[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (cand_v[w] > bv || (cand_v[w] == bv && cand_i[w] < bi)) { bv = cand_v[w]; bi = cand_i[w]; }
        taken = bi;
#pragma unroll
        for (int t = 0; t < kMaxTopK; ++t)
            if (t == round) taken_all[t] = taken;
        if (tid == 0) {
            out->v[round] = bv;
            out->i[round] = bi;
        }
    }
}

__global__ __launch_bounds__(256) void topk_stage1_kernel(float* __restrict__ logits, int n_vocab, int k,
                                                          SelPartial* __restrict__ parts, const int* __restrict__ adj_row,
                                                          const int* __restrict__ adj_ids,
                                                          const float* __restrict__ adj_deltas, int n_adj) {
    topk_stage1_body(logits, n_vocab, k, parts, adj_row, adj_ids, adj_deltas, n_adj, blockIdx.x, blockIdx.y);
}

// results of a single-session graph step go straight into pinned host memory; the flag is stored after a system-scope
// fence, so a host that sees the flag sees the fields
__device__ __forceinline__ void publish_flag(unsigned* flag, unsigned seq) {
    __threadfence_system();
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void topk_stage2_body(const SelPartial* __restrict__ parts, int k,
                                                 float* __restrict__ top_vals, int* __restrict__ top_ids, int row,
                                                 const StepHostOut host = StepHostOut{}, unsigned seq = 0,
                                                 const bool publish = true) {
    const int lane = threadIdx.x;   // one lane per slice (the first wave of the workgroup)
    const SelPartial p = parts[(long)row * kSelBlocks + lane];
    float mx = wave_max(p.mx);
    asm volatile("" : "+v"(seq));   // requested with the partials, not in front of the flag store at the very end
    float sum = p.mx > -INFINITY ? p.sum * expf(p.mx - mx) : 0.f;
    sum = wave_sum(sum);
    const float lse = logf(sum);
    int head = 0;                   // next unconsumed candidate of this slice (its list is sorted)
    for (int round = 0; round < k; ++round) {
        float bv = head < k ? p.v[0] : -INFINITY;
        int bi = head < k ? p.i[0] : 0x7fffffff;
#pragma unroll
        for (int t = 1; t < kMaxTopK; ++t)
            if (t == head && t < k) { bv = p.v[t]; bi = p.i[t]; }
        if (head >= k) { bv = -INFINITY; bi = 0x7fffffff; }
        float wv = bv;
        int wi = bi;
        wave_argmax(wv, wi);
        if (wi == bi && bi != 0x7fffffff) ++head;
        if (lane == 0) {
            top_ids[row * k + round] = wi;
            top_vals[row * k + round] = (wv - mx) - lse;
            if (host.result && round < 2) {
                host.result[row].top_ids[round] = wi;
                host.result[row].top_vals[round] = (wv - mx) - lse;
            }
        }
    }
    if (publish && host.result && lane == 0) publish_flag(&host.result[row].flag_topk, seq);
}

__global__ __launch_bounds__(64) void topk_stage2_kernel(const SelPartial* __restrict__ parts, int k,
                                                         float* __restrict__ top_vals, int* __restrict__ top_ids) {
    topk_stage2_body(parts, k, top_vals, top_ids, blockIdx.x);
}

void launch_logsoftmax_topk(const LaunchCtx& ctx, float* logits, int n_vocab, int n_rows, int k,
                            float* top_vals, int* top_ids, void* scratch, const int* adj_row, const int* adj_ids,
                            const float* adj_deltas, int n_adj) {
    if (k < 1 || k > kMaxTopK) throw std::invalid_argument("top-k: k must be in [1, 8]");
    SelPartial* parts = static_cast<SelPartial*>(scratch);
    {
        KernelScope ks(ctx, "sel_topk_stage1", 0.0, 4.0 * 3.0 * n_rows * (double)n_vocab);
        hipLaunchKernelGGL(topk_stage1_kernel, dim3(kSelBlocks, n_rows), dim3(256), 0, ctx.stream, logits, n_vocab, k,
                           parts, adj_row, adj_ids, adj_deltas, n_adj);
        WLK_HIP(hipGetLastError());
    }
    {
        KernelScope ks(ctx, "sel_topk_stage2");
        hipLaunchKernelGGL(topk_stage2_kernel, dim3(n_rows), dim3(64), 0, ctx.stream, parts, k, top_vals, top_ids);
        WLK_HIP(hipGetLastError());
    }
}

size_t topk_scratch_bytes(int n_rows) { return sizeof(SelPartial) * kSelBlocks * (size_t)n_rows; }

// softmax(logits)[token] of one row by one 1024-thread workgroup.  The row is read ONCE (each thread keeps its strided
// elements in registers between the max pass and the sum pass); the order of every max / sum is that of the plain
// two-pass loops.
__device__ __forceinline__ void token_prob_body(const float* __restrict__ logits, int n_vocab, int token,
                                                float* __restrict__ probs, int row) {
    __shared__ float red[16];
    constexpr int kKeep = 64;   // elements per thread held in registers: rows up to 65 536 logits
    const int tid = threadIdx.x;
    const float* x = logits + (long)row * n_vocab;
    float mx = -INFINITY, sum = 0.f;
    if (n_vocab <= kKeep * kSelThreads) {
        float v[kKeep];
#pragma unroll
        for (int j = 0; j < kKeep; ++j) {
            const int i = tid + j * kSelThreads;
            v[j] = i < n_vocab ? x[i] : -INFINITY;
        }
#pragma unroll
        for (int j = 0; j < kKeep; ++j) mx = fmaxf(mx, v[j]);
        mx = block_max(mx, red);
#pragma unroll
        for (int j = 0; j < kKeep; ++j)
            if (tid + j * kSelThreads < n_vocab) sum += expf(v[j] - mx);
    } else {
        for (int i = tid; i < n_vocab; i += kSelThreads) mx = fmaxf(mx, x[i]);
        mx = block_max(mx, red);
        for (int i = tid; i < n_vocab; i += kSelThreads) sum += expf(x[i] - mx);
    }
    sum = block_sum(sum, red);
    if (tid == 0) probs[row] = expf(x[token] - mx) / sum;
}

__global__ __launch_bounds__(kSelThreads) void token_prob_kernel(const float* __restrict__ logits, int n_vocab,
                                                                 int token, float* __restrict__ probs) {
    token_prob_body(logits, n_vocab, token, probs, blockIdx.x);
}

void launch_token_prob(const LaunchCtx& ctx, const float* logits, int n_vocab, int n_rows, int token, float* probs) {
    KernelScope ks(ctx, "sel_token_prob");
    hipLaunchKernelGGL(token_prob_kernel, dim3(n_rows), dim3(kSelThreads), 0, ctx.stream, logits, n_vocab, token,
                       probs);
    WLK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// AlignAtt.  Window rows of alignment head `a`, beam `b` live at ring[(a*n_beam+b)*ring_rows + r]:
// rows [0, prefill_rows) belong to the first step of this `infer` (as long as it is among the
// last 16 steps), rows single_base + [0, n_single) are later single-token steps.
// Step 1 (this kernel): per frame column, mean and population std over the window rows in fp64
// (torch.std_mean(unbiased=False) accumulates in double on CPU), then the z-score of the newest row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void align_zscore_kernel(AlignArgs a) {
    align_zscore_body(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

__device__ __forceinline__ float median7(float v0, float v1, float v2, float v3, float v4, float v5, float v6) {
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

// Step 2: one workgroup per beam row - median filter each head's z row (reflect padding over the
// full T before the content_mel_len cut, timing.py:35), average the heads, first arg-max.
__global__ __launch_bounds__(256) void align_argmax_kernel(AlignArgs a) {
    __shared__ float bestv[256];
    __shared__ int besti[256];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    if (a.rows) a.content_len = a.rows[b].content_len;
    const float* zb = a.z + (long)b * a.n_align * a.T;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int f = tid; f < a.T; f += 256) {
        float acc = 0.f;
        for (int al = 0; al < a.n_align; ++al) {
            const float* z = zb + (long)al * a.T;
            float v[7];
#pragma unroll
            for (int o = -3; o <= 3; ++o) {
                int idx = f + o;
                if (idx < 0) idx = -idx;
                if (idx >= a.T) idx = 2 * (a.T - 1) - idx;
                v[o + 3] = z[idx];
            }
            acc += a.T > 3 ? median7(v[0], v[1], v[2], v[3], v[4], v[5], v[6]) : z[f];
        }
        const float m = acc / (float)a.n_align;
        a.attn_last[(long)b * a.T + f] = m;
        if (f < a.content_len && (m > bv || (m == bv && f < bi))) { bv = m; bi = f; }
    }
    bestv[tid] = bv;
    besti[tid] = bi;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) {
            const float ov = bestv[tid + s];
            const int oi = besti[tid + s];
            if (ov > bestv[tid] || (ov == bestv[tid] && oi < besti[tid])) { bestv[tid] = ov; besti[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) a.frames[b] = besti[0] == 0x7fffffff ? 0 : besti[0];
}

// Step 2 variant for the common case (z of all alignment heads fits LDS): 1024 threads stage z with
// coalesced loads once, then medians / head mean / arg-max run out of LDS.
__device__ __forceinline__ void align_argmax_lds_body(AlignArgs a, int b, const StepHostOut host = StepHostOut{}, unsigned seq = 0) {
    extern __shared__ __attribute__((aligned(16))) float zs[];   // [n_align][T]
    __shared__ float bestv[16];
    __shared__ int besti[16];
    const int tid = threadIdx.x;
    if (a.rows) a.content_len = a.rows[b].content_len;
    const float* zb = a.z + (long)b * a.n_align * a.T;
    for (int i = tid; i < a.n_align * a.T; i += 1024) zs[i] = zb[i];
    asm volatile("" : "+v"(seq));   // requested with the z rows, not in front of the flag store at the very end
    __syncthreads();
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int f = tid; f < a.T; f += 1024) {
        float acc = 0.f;
        for (int al = 0; al < a.n_align; ++al) {
            const float* z = zs + al * a.T;
            float v[7];
#pragma unroll
            for (int o = -3; o <= 3; ++o) {
                int idx = f + o;
                if (idx < 0) idx = -idx;
                if (idx >= a.T) idx = 2 * (a.T - 1) - idx;
                v[o + 3] = z[idx];
            }
            acc += a.T > 3 ? median7(v[0], v[1], v[2], v[3], v[4], v[5], v[6]) : z[f];
        }
        const float m = acc / (float)a.n_align;
        a.attn_last[(long)b * a.T + f] = m;
        if (f < a.content_len && (m > bv || (m == bv && f < bi))) { bv = m; bi = f; }
    }
    // (value, frame) arg-max with ties to the lowest frame: order-free, so sixteen VALU butterflies + one 16-entry fold
    // replace the ten-level LDS tree and its ten barriers
    wave_argmax(bv, bi);
    if ((tid & 63) == 0) { bestv[tid >> 6] = bv; besti[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (bestv[w] > bestv[0] || (bestv[w] == bestv[0] && besti[w] < besti[0])) { bestv[0] = bestv[w]; besti[0] = besti[w]; }
    }
    if (tid == 0) {
        const int frame = besti[0] == 0x7fffffff ? 0 : besti[0];
        a.frames[b] = frame;
        if (host.result) {
            host.result[b].frame = frame;
            publish_flag(&host.result[b].flag_align, seq);
        }
    }
}

__global__ __launch_bounds__(1024) void align_argmax_lds_kernel(AlignArgs a) { align_argmax_lds_body(a, blockIdx.x); }

// ---------------------------------------------------------------------------------------------
// The tail of a decode step in TWO launches instead of four: the slice pass of the top-k and the z-score of the
// alignment window are independent (logits vs cross-attention rows), and so are their second stages; each pair shares a
// launch, workgroups pick their role from the block index.  Same device functions, same arithmetic as the four
// separate kernels above (which remain for the configurations the fused form does not cover).
// ---------------------------------------------------------------------------------------------
// every field either role reads, requested by one s_load burst at kernel entry (see WLK_PIN_GEMM_ARGS)
#define WLK_PIN_SELECT_ARGS(t, a)                                                                                           \
    do {                                                                                                                    \
        asm volatile("" ::"s"((t).logits), "s"((t).n_vocab), "s"((t).k), "s"((t).n_rows), "s"((t).parts), "s"((t).adj_row), \
                     "s"((t).adj_ids), "s"((t).adj_deltas), "s"((t).n_adj), "s"((t).top_vals), "s"((t).top_ids),            \
                     "s"((t).host.result), "s"((t).host.n_adj), "s"((t).host.seq));                                         \
        asm volatile("" ::"s"((a).ring), "s"((a).n_align), "s"((a).n_beam), "s"((a).ring_rows), "s"((a).T),                 \
                     "s"((a).prefill_rows), "s"((a).n_single), "s"((a).newest_row), "s"((a).single_base),                   \
                     "s"((a).content_len), "s"((a).z), "s"((a).attn_last), "s"((a).frames), "s"((a).rows));                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
    } while (0)

__global__ __launch_bounds__(256) void select_stage1_kernel(TopkArgs t, AlignArgs a, int zf_blocks) {
    WLK_PIN_SELECT_ARGS(t, a);
    const int n_topk = kSelBlocks * t.n_rows;
    if ((int)blockIdx.x < n_topk) {
        const int n_adj = t.host.n_adj ? *t.host.n_adj : t.n_adj;
        topk_stage1_body(t.logits, t.n_vocab, t.k, t.parts, t.adj_row, t.adj_ids, t.adj_deltas, n_adj,
                         blockIdx.x % kSelBlocks, blockIdx.x / kSelBlocks);
    } else {
        const int i = blockIdx.x - n_topk;
        const int fblock = i % zf_blocks, rest = i / zf_blocks;
        align_zscore_body(a, fblock, rest % a.n_align, rest / a.n_align);
    }
}

__global__ __launch_bounds__(1024) void select_stage2_kernel(TopkArgs t, AlignArgs a) {
    WLK_PIN_SELECT_ARGS(t, a);
    // the step's sequence number (echoed in the result flags) is requested here, not in front of the flag store at the
    // very end of the step's last kernel - the host is waiting on exactly that store
    unsigned seq = 0;
    if (t.host.result) seq = *t.host.seq;     // (each role pins it behind its first batch of loads: one round trip with them)
    if ((int)blockIdx.x < a.n_beam) {
        align_argmax_lds_body(a, blockIdx.x, t.host, seq);
    } else if ((int)blockIdx.x < a.n_beam + t.n_rows) {
        if (threadIdx.x < 64) topk_stage2_body(t.parts, t.k, t.top_vals, t.top_ids, blockIdx.x - a.n_beam, t.host, seq);
    } else {
        token_prob_body(t.ns_logits, t.n_vocab, t.ns_token, t.ns_probs, blockIdx.x - a.n_beam - t.n_rows);
    }
}

// Graph-replayed single-session steps (round 6, `early_z`).  The z-score has already run beside the vocabulary projection
// (gemv_f32_kernel's side workgroups).  What the trace of the first form of this showed: the medians are VALU work - 1 500
// frames x 5 heads x 21 compare-exchanges - and ONE workgroup (one compute unit) needs 9-11 us for them however it is
// scheduled.  So the medians / head mean run as (T + 255) / 256 workgroups of the top-k SLICE pass's launch, one frame per
// thread, each leaving the (value, lowest frame) arg-max of its 256 frames; the step's last launch is one wave that folds the
// 64 slices of the top-k and the frame blocks of the read-out, stores both into the host's result block and raises both
// flags behind one system-scope fence.  Same per-frame expressions as align_argmax_lds_body, and an arg-max with ties to the
// lowest frame does not depend on how it is grouped: the step's numbers are those of the two launches above.
constexpr int kAlignPartFrames = 256;
__device__ __forceinline__ void align_median_part_body(AlignArgs a, int b, int blk, int n_blk) {
    __shared__ float bestv[4];
    __shared__ int besti[4];
    const int tid = threadIdx.x;
    if (a.rows) a.content_len = a.rows[b].content_len;
    const float* zb = a.z + (long)b * a.n_align * a.T;
    const int f = blk * kAlignPartFrames + tid;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if (f < a.T) {
        float acc = 0.f;
        for (int al = 0; al < a.n_align; ++al) {
            const float* z = zb + (long)al * a.T;
            float v[7];
#pragma unroll
            for (int o = -3; o <= 3; ++o) {
                int idx = f + o;
                if (idx < 0) idx = -idx;
                if (idx >= a.T) idx = 2 * (a.T - 1) - idx;
                v[o + 3] = z[idx];
            }
            acc += a.T > 3 ? median7(v[0], v[1], v[2], v[3], v[4], v[5], v[6]) : z[f];
        }
        const float m = acc / (float)a.n_align;
        a.attn_last[(long)b * a.T + f] = m;
        if (f < a.content_len && (m > bv || (m == bv && f < bi))) { bv = m; bi = f; }
    }
    wave_argmax(bv, bi);
    if ((tid & 63) == 0) { bestv[tid >> 6] = bv; besti[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (bestv[w] > bestv[0] || (bestv[w] == bestv[0] && besti[w] < besti[0])) { bestv[0] = bestv[w]; besti[0] = besti[w]; }
        float* slot = a.part + ((long)b * 64 + blk) * 2;
        slot[0] = bestv[0];
        slot[1] = __int_as_float(besti[0]);
    }
}

__global__ __launch_bounds__(256) void select_stage1_early_kernel(TopkArgs t, AlignArgs a, int n_blk) {
    WLK_PIN_SELECT_ARGS(t, a);
    const int n_topk = kSelBlocks * t.n_rows;
    if ((int)blockIdx.x < n_topk) {
        const int n_adj = t.host.n_adj ? *t.host.n_adj : t.n_adj;
        topk_stage1_body(t.logits, t.n_vocab, t.k, t.parts, t.adj_row, t.adj_ids, t.adj_deltas, n_adj,
                         blockIdx.x % kSelBlocks, blockIdx.x / kSelBlocks);
    } else {
        const int i = blockIdx.x - n_topk;
        align_median_part_body(a, i / n_blk, i % n_blk, n_blk);
    }
}

// one wave per row: lane = top-k slice, and lane = frame block of the read-out (n_blk <= 64)
__global__ __launch_bounds__(64) void select_stage2_early_kernel(TopkArgs t, AlignArgs a, int n_blk) {
    asm volatile("" ::"s"(t.parts), "s"(t.k), "s"(t.top_vals), "s"(t.top_ids), "s"(t.host.result), "s"(t.host.seq), "s"(a.part),
                 "s"(a.frames));
    __builtin_amdgcn_sched_barrier(0);
    const int row = blockIdx.x, lane = threadIdx.x;
    unsigned seq = 0;
    if (t.host.result) seq = *t.host.seq;
    const float* slot = a.part + ((long)row * 64 + (lane < n_blk ? lane : 0)) * 2;
    float bv = slot[0];
    int bi = __float_as_int(slot[1]);
    if (lane >= n_blk) { bv = -INFINITY; bi = 0x7fffffff; }
    topk_stage2_body(t.parts, t.k, t.top_vals, t.top_ids, row, t.host, seq, false);
    // a block whose frames all lie behind content_len (or are NaN) left (-inf, 0x7fffffff): it loses every comparison, as
    // its threads' candidates do in the one-workgroup form
    wave_argmax(bv, bi);
    if (lane == 0) {
        const int frame = bi == 0x7fffffff ? 0 : bi;
        a.frames[row] = frame;
        if (t.host.result) {
            t.host.result[row].frame = frame;
            __threadfence_system();
            __hip_atomic_store(&t.host.result[row].flag_topk, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&t.host.result[row].flag_align, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

bool select_early_z_enabled() {
    // read per call (a session asks once, when it captures its step graph): tests flip WLK_EARLY_Z inside one process
    const char* e = getenv("WLK_EARLY_Z");
    return !(e && e[0] == '0');
}

bool select_fused_applicable(int n_rows, int k, const AlignArgs& a) {
    static const bool enabled = [] {
        const char* e = getenv("WLK_SELECT_FUSED");
        return !(e && e[0] == '0');
    }();
    const size_t lds = (size_t)a.n_align * a.T * sizeof(float);
    return enabled && a.n_align > 0 && a.n_beam == n_rows && lds + 8192 + 1024 <= 150 * 1024 && k >= 1 && k <= kMaxTopK;
}

bool launch_select_fused(const LaunchCtx& ctx, float* logits, int n_vocab, int n_rows, int k, float* top_vals, int* top_ids,
                         void* scratch, const int* adj_row, const int* adj_ids, const float* adj_deltas, int n_adj,
                         const AlignArgs& a, const StepHostOut& host, const float* ns_logits, int ns_token, float* ns_probs,
                         bool early_z) {
    const size_t lds = (size_t)a.n_align * a.T * sizeof(float);
    if (!select_fused_applicable(n_rows, k, a)) return false;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    WLK_HIP(hipGetDevice(&dev));
    if (dev < 64 && !attr_set[dev].load(std::memory_order_acquire)) {
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(select_stage2_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_set[dev].store(true, std::memory_order_release);
    }
    TopkArgs t{logits, n_vocab, k, n_rows, static_cast<SelPartial*>(scratch), adj_row, adj_ids, adj_deltas, n_adj, top_vals, top_ids,
               host, ns_logits, ns_token, ns_probs};
    const int zf = (a.T + 63) / 64;
    if (early_z) {
        const int n_blk = (a.T + kAlignPartFrames - 1) / kAlignPartFrames;
        if (ns_logits || !a.part || n_blk > 64) throw std::invalid_argument("select: the early z-score form needs a.part, T <= 16384 and no no-speech block");
        {
            KernelScope ks(ctx, "sel_stage1", 0.0, 4.0 * 3.0 * n_rows * (double)n_vocab);
            hipLaunchKernelGGL(select_stage1_early_kernel, dim3(kSelBlocks * n_rows + n_blk * a.n_beam), dim3(256), 0, ctx.stream, t, a, n_blk);
            WLK_HIP(hipGetLastError());
        }
        {
            KernelScope ks(ctx, "sel_stage2");
            hipLaunchKernelGGL(select_stage2_early_kernel, dim3(n_rows), dim3(64), 0, ctx.stream, t, a, n_blk);
            WLK_HIP(hipGetLastError());
        }
        return true;
    }
    {
        KernelScope ks(ctx, "sel_stage1", 0.0, 4.0 * 3.0 * n_rows * (double)n_vocab);
        hipLaunchKernelGGL(select_stage1_kernel, dim3(kSelBlocks * n_rows + zf * a.n_align * a.n_beam), dim3(256), 0,
                           ctx.stream, t, a, zf);
        WLK_HIP(hipGetLastError());
    }
    {
        KernelScope ks(ctx, "sel_stage2");
        hipLaunchKernelGGL(select_stage2_kernel, dim3(a.n_beam + n_rows + (ns_logits ? n_rows : 0)), dim3(1024), lds,
                           ctx.stream, t, a);
        WLK_HIP(hipGetLastError());
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// The per-step logit rules of whisper's batch decoder (whisper/decoding.py:417-499: SuppressBlank, SuppressTokens,
// ApplyTimestampRules) and the greedy pick at temperature 0 (GreedyDecoder.update, :270-287) for ONE sequence, on the device:
// the host keeps what the rules need of the token history (three flags and a bound) and reads back the token and its
// log-probability - 8 bytes per step instead of the 207 KB logits row, and no 52k-element log_softmax on the host.
//   allowed(v) = not suppressed, not a blank at the first step, and (with timestamps) none of: <|notimestamps|>; a timestamp
//   behind a closed pair (mode 1); a text token behind an opening timestamp (mode 2); a timestamp below the monotonic bound;
//   at the first step a text token or a timestamp beyond max_initial.
//   Then, as the reference: if logsumexp(logprobs[timestamps]) > max(logprobs[text]) the text tokens go too.
//   token = argmax over what is left (lowest index on ties), log-probability = log_softmax over what is left at that
//   token = -log(sum exp(x - max)).
// One workgroup of 1024 threads, two strided passes over the row (51 elements per thread).
// ---------------------------------------------------------------------------------------------
namespace {
struct PickBest {
    float v;
    int i;
};
__device__ __forceinline__ PickBest pick_better(PickBest a, PickBest b) {      // larger value, lower index on ties
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ PickBest pick_wave_best(PickBest x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        PickBest o;
        o.v = __shfl_xor(x.v, off);
        o.i = __shfl_xor(x.i, off);
        x = pick_better(x, o);
    }
    return x;
}
}  // namespace

__global__ __launch_bounds__(1024) void rules_pick_kernel(const float* __restrict__ logits, int n_vocab,
                                                          const unsigned char* __restrict__ mask, PickRules p,
                                                          int* __restrict__ out_token, float* __restrict__ out_logprob) {
    __shared__ PickBest s_all[16], s_ts[16];
    __shared__ float s_text[16], s_sum_all[16], s_sum_ts[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tb = p.timestamp_begin;
    auto allowed = [&](int v) -> bool {
        const unsigned char m = mask[v];
        if ((m & 1) || (p.first_step && (m & 2))) return false;
        if (p.without_timestamps) return true;
        if (v == p.no_timestamps) return false;
        if (p.ts_mode == 1 && v >= tb) return false;
        if (p.ts_mode == 2 && v < p.eot) return false;
        if (v >= tb && v < p.ts_bound) return false;
        if (p.first_step && (v < tb || (p.max_initial >= 0 && v >= tb + p.max_initial + 1))) return false;
        return true;
    };
    PickBest all{-INFINITY, 0x7fffffff}, ts{-INFINITY, 0x7fffffff};
    float text = -INFINITY;
    for (int v = tid; v < n_vocab; v += 1024) {
        if (!allowed(v)) continue;
        const float x = logits[v];
        all = pick_better(all, PickBest{x, v});
        if (v >= tb) ts = pick_better(ts, PickBest{x, v});
        else text = fmaxf(text, x);
    }
    all = pick_wave_best(all);
    ts = pick_wave_best(ts);
    text = wave_max(text);
    if (lane == 0) {
        s_all[wave] = all;
        s_ts[wave] = ts;
        s_text[wave] = text;
    }
    __syncthreads();
    all = s_all[0];
    ts = s_ts[0];
    text = s_text[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) {
        all = pick_better(all, s_all[w]);
        ts = pick_better(ts, s_ts[w]);
        text = fmaxf(text, s_text[w]);
    }
    float sum_all = 0.f, sum_ts = 0.f;
    for (int v = tid; v < n_vocab; v += 1024) {
        if (!allowed(v)) continue;
        const float x = logits[v];
        sum_all += expf(x - all.v);
        if (v >= tb) sum_ts += expf(x - ts.v);
    }
    sum_all = wave_sum(sum_all);
    sum_ts = wave_sum(sum_ts);
    if (lane == 0) {
        s_sum_all[wave] = sum_all;
        s_sum_ts[wave] = sum_ts;
    }
    __syncthreads();
    if (tid == 0) {
        sum_all = 0.f;
        sum_ts = 0.f;
        for (int w = 0; w < 16; ++w) {
            sum_all += s_sum_all[w];
            sum_ts += s_sum_ts[w];
        }
        const float lse_all = all.v + logf(sum_all);
        // logsumexp(logprobs[tb:]) > logprobs[:tb].max(), both relative to the same log_softmax
        const bool ts_wins = !p.without_timestamps && ts.i != 0x7fffffff && (ts.v + logf(sum_ts)) - lse_all > text - lse_all;
        const PickBest pick = ts_wins ? ts : all;
        out_token[0] = pick.i == 0x7fffffff ? 0 : pick.i;
        out_logprob[0] = -logf(ts_wins ? sum_ts : sum_all);
    }
}

void launch_rules_pick(const LaunchCtx& ctx, const float* logits, int n_vocab, const unsigned char* mask, const PickRules& p,
                       int* out_token, float* out_logprob) {
    KernelScope ks(ctx, "rules_pick", 0.0, 8.0 * n_vocab);
    hipLaunchKernelGGL(rules_pick_kernel, dim3(1), dim3(1024), 0, ctx.stream, logits, n_vocab, mask, p, out_token, out_logprob);
    WLK_HIP(hipGetLastError());
}

void launch_alignatt_rows(const LaunchCtx& ctx, const AlignArgs& a0, const StepRow* rows) {
    AlignArgs a = a0;
    a.rows = rows;
    launch_alignatt(ctx, a);
}

void launch_alignatt(const LaunchCtx& ctx, const AlignArgs& a) {
    if (a.n_align <= 0) {
        WLK_HIP(hipMemsetAsync(a.frames, 0, sizeof(int) * a.n_beam, ctx.stream));
        WLK_HIP(hipMemsetAsync(a.attn_last, 0, sizeof(float) * a.n_beam * a.T, ctx.stream));
        return;
    }
    {
        const int n = a.rows ? kAlignWindow : a.prefill_rows + a.n_single;
        KernelScope ks(ctx, "align_zscore", 0.0, 4.0 * 2.0 * n * (double)a.n_align * a.T * a.n_beam);
        hipLaunchKernelGGL(align_zscore_kernel, dim3((a.T + 63) / 64, a.n_align, a.n_beam), dim3(256), 0,
                           ctx.stream, a);
        WLK_HIP(hipGetLastError());
    }
    const size_t lds = (size_t)a.n_align * a.T * sizeof(float);
    KernelScope ks(ctx, "align_argmax");
    if (lds + 8192 + 1024 <= 150 * 1024) {
        static std::atomic<bool> attr_set[64];   // launches come from several host threads
        int dev = 0;
        WLK_HIP(hipGetDevice(&dev));
        if (dev < 64 && !attr_set[dev].load(std::memory_order_acquire)) {
            WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(align_argmax_lds_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            attr_set[dev].store(true, std::memory_order_release);
        }
        hipLaunchKernelGGL(align_argmax_lds_kernel, dim3(a.n_beam), dim3(1024), lds, ctx.stream, a);
    } else {
        hipLaunchKernelGGL(align_argmax_kernel, dim3(a.n_beam), dim3(256), 0, ctx.stream, a);
    }
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

// Decoder-side kernels that are not plain linear layers (whisper/model.py:100-173, 279-332):
// token+position embedding, self-attention over the growing KV cache, and cross-attention over
// the 1500 encoder positions that ALSO produces what AlignAtt needs - the softmaxed attention
// row of every alignment head - without ever writing the [H, q, 1500] QK tensor the reference
// returns from each layer (model.py:170-173, simul_whisper.py:402-417).
//
// These are HBM/L2-bound byte movers (one pass over K and V per query row); the rules that
// matter are coalescing (16 lanes x float4 = one 256-byte head row per load), wavefront
// reductions for the softmax, and keeping scores in LDS.
#include <cstddef>

#include <cstdlib>

#include "common.h"
#include "wave_ops.h"

namespace wlk {

__global__ __launch_bounds__(128) void embed_kernel(const int* __restrict__ tokens,
                                                    const float* __restrict__ tok_emb,
                                                    const float* __restrict__ pos_emb, float* __restrict__ x,
                                                    int n_tok, const int* __restrict__ offset_p, int d) {
    const int row = blockIdx.x;             // row = beam * n_tok + p
    const int p = row % n_tok;
    const int offset = *offset_p;           // device scalar: keeps the launch replayable from a hipGraph
    const float* e = tok_emb + (long)tokens[row] * d;
    const float* pe = pos_emb + (long)(offset + p) * d;
    for (int c = threadIdx.x; c < d; c += 128) x[(long)row * d + c] = e[c] + pe[c];
}

void launch_embed(const LaunchCtx& ctx, const int* tokens, const float* tok_emb, const float* pos_emb,
                  float* x, int n_rows, int n_tok, const int* offset, int d) {
    KernelScope ks(ctx, "dec_embed");
    hipLaunchKernelGGL(embed_kernel, dim3(n_rows * n_tok), dim3(128), 0, ctx.stream, tokens, tok_emb, pos_emb, x,
                       n_tok, offset, d);
    WLK_HIP(hipGetLastError());
}

__global__ __launch_bounds__(128) void embed_rows_kernel(const StepRow* __restrict__ rows, const float* __restrict__ tok_emb,
                                                         const float* __restrict__ pos_emb, float* __restrict__ x, int d) {
    const int row = blockIdx.x;
    const StepRow sr = rows[row];
    const float* e = tok_emb + (long)sr.token * d;
    const float* pe = pos_emb + (long)sr.offset * d;
    for (int c = threadIdx.x; c < d; c += 128) x[(long)row * d + c] = e[c] + pe[c];
}

void launch_embed_rows(const LaunchCtx& ctx, const StepRow* rows, const float* tok_emb, const float* pos_emb, float* x,
                       int n_rows, int d) {
    KernelScope ks(ctx, "dec_embed");
    hipLaunchKernelGGL(embed_rows_kernel, dim3(n_rows), dim3(128), 0, ctx.stream, rows, tok_emb, pos_emb, x, d);
    WLK_HIP(hipGetLastError());
}

// First kernel of a single-session graph step: the step's inputs are read straight from the pinned host block (one
// uncached read per lane, all in flight together), spread into device memory for the kernels behind this one, and the
// token + position embedding of the fed token is written as in embed_kernel.
__global__ __launch_bounds__(256) void embed_step_kernel(const StepBlock* __restrict__ host_block, StepBlock* __restrict__ dev_block,
                                                         int* __restrict__ tokens_dev, int* __restrict__ ring_row,
                                                         int* __restrict__ beam_of_row, int* __restrict__ d_offset,
                                                         const float* __restrict__ tok_emb, const float* __restrict__ pos_emb,
                                                         float* __restrict__ x, int d) {
    constexpr int kWords = sizeof(StepBlock) / 4, kHead = sizeof(StepRow) / 4;
    static_assert(kWords <= 512 && sizeof(StepBlock) <= 2048, "two words of the step block per thread");
    __shared__ unsigned head[kHead];
    const int tid = threadIdx.x;
    const unsigned* src = reinterpret_cast<const unsigned*>(host_block);
    unsigned w0 = 0, w1 = 0;
    if (tid < kWords) w0 = __hip_atomic_load(src + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid + 256 < kWords) w1 = __hip_atomic_load(src + tid + 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid < kWords) reinterpret_cast<unsigned*>(dev_block)[tid] = w0;
    if (tid + 256 < kWords) reinterpret_cast<unsigned*>(dev_block)[tid + 256] = w1;
    if (tid < kHead) head[tid] = w0;
    __syncthreads();
    const StepRow* row = reinterpret_cast<const StepRow*>(head);
    const int token = row->token, offset = row->offset;
    if (tid == 0) {
        tokens_dev[0] = token;
        ring_row[0] = row->ring_row;
        beam_of_row[0] = 0;
        d_offset[0] = offset;
    }
    const float* e = tok_emb + (long)token * d;
    const float* pe = pos_emb + (long)offset * d;
    for (int c = tid; c < d; c += 256) x[c] = e[c] + pe[c];
}

void launch_embed_step(const LaunchCtx& ctx, const StepBlock* host_block, StepBlock* dev_block, int* tokens_dev,
                       int* ring_row, int* beam_of_row, int* d_offset, const float* tok_emb, const float* pos_emb, float* x,
                       int d) {
    KernelScope ks(ctx, "dec_embed");
    hipLaunchKernelGGL(embed_step_kernel, dim3(1), dim3(256), 0, ctx.stream, host_block, dev_block, tokens_dev, ring_row,
                       beam_of_row, d_offset, tok_emb, pos_emb, x, d);
    WLK_HIP(hipGetLastError());
}

// First kernel of a batched graph step: workgroup r embeds row r's token; together the workgroups copy the step's block
// (row table, adjustment count, logit adjustments) from host-coherent memory into its device copy.
__global__ __launch_bounds__(256) void embed_rows_step_kernel(const EngineBlock* __restrict__ host_block,
                                                              EngineBlock* __restrict__ dev_block,
                                                              const float* __restrict__ tok_emb,
                                                              const float* __restrict__ pos_emb, float* __restrict__ x, int d) {
    constexpr int kWords = sizeof(EngineBlock) / 4, kRowWords = sizeof(StepRow) / 4;
    constexpr int kRow0 = offsetof(EngineBlock, rows) / 4;
    __shared__ unsigned mine[kRowWords];
    const int tid = threadIdx.x, row = blockIdx.x, n_blocks = gridDim.x;
    const unsigned* src = reinterpret_cast<const unsigned*>(host_block);
    unsigned* dst = reinterpret_cast<unsigned*>(dev_block);
    if (tid < kRowWords) mine[tid] = __hip_atomic_load(src + kRow0 + row * kRowWords + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // the whole block in one round trip (12.5 KiB; every load is independent of the others)
    for (int w = row * 256 + tid; w < kWords; w += n_blocks * 256)
        dst[w] = __hip_atomic_load(src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    const StepRow* sr = reinterpret_cast<const StepRow*>(mine);
    const float* e = tok_emb + (long)sr->token * d;
    const float* pe = pos_emb + (long)sr->offset * d;
    for (int c = tid; c < d; c += 256) x[(long)row * d + c] = e[c] + pe[c];
}

void launch_embed_rows_step(const LaunchCtx& ctx, const EngineBlock* host_block, EngineBlock* dev_block, const float* tok_emb,
                            const float* pos_emb, float* x, int n_rows, int d) {
    KernelScope ks(ctx, "dec_embed");
    hipLaunchKernelGGL(embed_rows_step_kernel, dim3(n_rows), dim3(256), 0, ctx.stream, host_block, dev_block, tok_emb,
                       pos_emb, x, d);
    WLK_HIP(hipGetLastError());
}

// qkv rows are [q | k | v] (3d floats); append k and v of every row to the per-beam caches
__global__ __launch_bounds__(256) void kv_append_kernel(const float* __restrict__ qkv, float* __restrict__ kc,
                                                        float* __restrict__ vc, int n_tok,
                                                        const int* __restrict__ offset_p, int d, int ctx_len) {
    const int offset = *offset_p;
    const int row = blockIdx.x;
    const int b = row / n_tok, p = row - b * n_tok;
    const float* src = qkv + (long)row * 3 * d;
    const long dst = ((long)b * ctx_len + offset + p) * d;
    for (int c = threadIdx.x; c < d; c += 256) {
        kc[dst + c] = src[d + c];
        vc[dst + c] = src[2 * d + c];
    }
}

void launch_kv_append(const LaunchCtx& ctx, const float* qkv, float* kc, float* vc, int n_rows, int n_tok,
                      const int* offset, int d, int ctx_len) {
    KernelScope ks(ctx, "dec_kv_append");
    hipLaunchKernelGGL(kv_append_kernel, dim3(n_rows * n_tok), dim3(256), 0, ctx.stream, qkv, kc, vc, n_tok,
                       offset, d, ctx_len);
    WLK_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void kv_append_rows_kernel(const float* __restrict__ qkv, const StepRow* __restrict__ rows,
                                                             long layer_off, int d) {
    const int row = blockIdx.x;
    const StepRow sr = rows[row];
    const float* src = qkv + (long)row * 3 * d;
    const long dst = layer_off + (long)sr.offset * d;
    for (int c = threadIdx.x; c < d; c += 256) {
        sr.kcache[dst + c] = src[d + c];
        sr.vcache[dst + c] = src[2 * d + c];
    }
}

void launch_kv_append_rows(const LaunchCtx& ctx, const float* qkv, const StepRow* rows, long layer_off, int n_rows, int d) {
    KernelScope ks(ctx, "dec_kv_append");
    hipLaunchKernelGGL(kv_append_rows_kernel, dim3(n_rows), dim3(256), 0, ctx.stream, qkv, rows, layer_off, d);
    WLK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Self-attention over the growing KV cache, one 256-thread workgroup per (query row, head).
// Causal: row p of the fed block sees cache positions 0..offset+p (model.py:164-166 adds
// mask[:n_ctx,:n_ctx]; with a cache and one fed token that mask is the single 0, i.e. everything
// cached is visible).  Same data mapping as the cross-attention kernels below: a 16-lane group reads
// one 256-byte key/value row, a wave covers 4 rows per instruction, 8 instructions are in flight
// before the first dot product is folded.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decoder_self_attention_kernel(const float* __restrict__ qkv,
                                                                     const float* __restrict__ kc_in,
                                                                     const float* __restrict__ vc_in,
                                                                     const StepRow* __restrict__ step_rows,
                                                                     const int* __restrict__ offset_p, int d,
                                                                     int ctx_len, int n_tok, float* __restrict__ out,
                                                                     long layer_off, ProgressMark mark) {
    // Latency diet (round 4): the query slice comes straight from global memory (16 lanes x float4, no LDS stage and no
    // barrier in front of the keys), and the first 128 keys AND values - a whole decode step's cache in the common case -
    // are requested before anything is waited for; later chunks (long prompts) loop as before.  The key rows of the first
    // chunk do not even wait for the cache length (a device scalar): rows past it are allocated cache memory whose scores
    // are never stored.  The 16-lane dot folds and the wave reductions are VALU butterflies in the original order
    // (wave_ops.h); the leading arguments arrive preloaded in SGPRs.  Same arithmetic, same results.
    __shared__ float sc[448 + 64];
    __shared__ float red[8];
    __shared__ __attribute__((aligned(16))) float part[16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 15, kq = lane >> 4;
    const int row = blockIdx.x;
    const int head = blockIdx.y;
    const float4 q4 = *reinterpret_cast<const float4*>(qkv + (long)row * 3 * d + head * 64 + sub * 4);
    // (a step with the Infinity-Cache prefetcher beside it: the step's sequence number, requested here, stored with the layer
    // index behind the kernel's work - mall_prefetch_step_kernel paces itself on that word)
    const bool marks = mark.word != nullptr && (blockIdx.x | blockIdx.y) == 0 && tid == 0;
    unsigned mark_seq = 0;
    if (marks) mark_seq = *mark.seq;
    int offset, b, p;
    gcf_ptr kc, vc;
    if (step_rows) {             // batched steps: one fed token per row, every row has its own cache
        offset = step_rows[row].offset;
        kc = to_global(step_rows[row].kcache) + layer_off;
        vc = to_global(step_rows[row].vcache) + layer_off;
        b = 0;
        p = 0;
    } else {
        offset = *offset_p;      // requested here, needed only behind the key loads
        kc = to_global(kc_in);
        vc = to_global(vc_in);
        b = n_tok == 1 ? row : row / n_tok;      // (decode steps feed one token per row: no division on their path)
        p = row - b * n_tok;
    }
    const gcf_ptr kb = kc + (long)b * ctx_len * d + head * 64 + sub * 4;
    const gcf_ptr vb = vc + (long)b * ctx_len * d + head * 64 + sub * 4;
    float4 kk0[8], vv0[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int j = wave * 4 + 16 * u + kq;
        kk0[u] = ldg4(kb + (long)(j < ctx_len ? j : 0) * d);
    }
    __builtin_amdgcn_sched_barrier(0);   // the key loads go out before the wait for the cache length
    const int n_keys = offset + p + 1;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int j = wave * 4 + 16 * u + kq;
        vv0[u] = ldg4(vb + (long)(j < n_keys ? j : 0) * d);
    }

    float mx = -INFINITY;
    for (int base = wave * 4; base < n_keys; base += 16 * 8) {
        float4 kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = base + 16 * u + kq;
            const bool ok = j < n_keys;
            if (base == wave * 4) kk[u] = kk0[u];
            else kk[u] = ldg4(kb + (long)(ok ? j : 0) * d);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = base + 16 * u + kq;
            float acc = 0.f;
            acc = fmaf(q4.x, kk[u].x, acc);
            acc = fmaf(q4.y, kk[u].y, acc);
            acc = fmaf(q4.z, kk[u].z, acc);
            acc = fmaf(q4.w, kk[u].w, acc);
            acc = row16_sum_1248(acc);
            if (j < n_keys) {
                if (sub == 0) sc[j] = acc;
                mx = fmaxf(mx, acc);
            }
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < n_keys; j += 256) {
        const float e = expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    for (int j = tid; j < n_keys; j += 256) sc[j] = sc[j] / sum;
    __syncthreads();

    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = wave * 4; base < n_keys; base += 16 * 8) {
        float4 vv[8];
        float ww[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = base + 16 * u + kq;
            const bool ok = j < n_keys;
            if (base == wave * 4) vv[u] = vv0[u];
            else vv[u] = ldg4(vb + (long)(ok ? j : 0) * d);
            ww[u] = ok ? sc[j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            o.x = fmaf(ww[u], vv[u].x, o.x);
            o.y = fmaf(ww[u], vv[u].y, o.y);
            o.z = fmaf(ww[u], vv[u].z, o.z);
            o.w = fmaf(ww[u], vv[u].w, o.w);
        }
    }
    reinterpret_cast<float4*>(part)[(wave * 4 + kq) * 16 + sub] = o;
    __syncthreads();
    if (tid < 64) {
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc += part[s * 64 + tid];
        out[(long)row * d + head * 64 + tid] = acc;
    }
    if (marks)
        __hip_atomic_store(mark.word, ((unsigned long long)mark_seq << 32) | (unsigned)mark.layer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

void launch_decoder_self_attention(const LaunchCtx& ctx, const float* qkv, const float* kc, const float* vc,
                                   float* out, int n_rows, int n_tok, const int* offset, int d, int n_head,
                                   int ctx_len, const ProgressMark& mark) {
    if (ctx_len > 448 + 64) throw std::invalid_argument("self-attention: context too long");
    KernelScope ks(ctx, "dec_self_attention");
    hipLaunchKernelGGL(decoder_self_attention_kernel, dim3(n_rows * n_tok, n_head), dim3(256), 0, ctx.stream, qkv,
                       kc, vc, (const StepRow*)nullptr, offset, d, ctx_len, n_tok, out, 0L, mark);
    WLK_HIP(hipGetLastError());
}

void launch_decoder_self_attention_rows(const LaunchCtx& ctx, const float* qkv, const StepRow* rows, long layer_off,
                                        float* out, int n_rows, int d, int n_head, int ctx_len) {
    if (ctx_len > 448 + 64) throw std::invalid_argument("self-attention: context too long");
    KernelScope ks(ctx, "dec_self_attention");
    hipLaunchKernelGGL(decoder_self_attention_kernel, dim3(n_rows, n_head), dim3(256), 0, ctx.stream, qkv,
                       (const float*)nullptr, (const float*)nullptr, rows, (const int*)nullptr, d, ctx_len, 1, out,
                       layer_off, ProgressMark{});
    WLK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Cross-attention, one 256-thread workgroup per (query row, head).
//   pass 1: scores s_j = q . K_j for the T = 1500 encoder positions.  A 16-lane group reads one
//           256-byte key row (float4 per lane), a wave covers 4 keys per instruction, the dot is
//           folded with 4 xor-shuffles.
//   pass 2: softmax over all T positions (the reference softmaxes the full 1500 BEFORE trimming
//           to content_mel_len, simul_whisper.py:411,432) - wave reductions + one LDS exchange.
//           If (layer, head) is an alignment head the probabilities go to the session's
//           alignment window row (they are exactly F.softmax(qk) that _process_cross_attention
//           recomputes); in debug sessions the raw scores are exported too.
//   pass 3: out_d = sum_j w_j V_j[d] with the same 16-lane row mapping, partials merged in LDS.
// ---------------------------------------------------------------------------------------------
constexpr int kCrossMaxT = 1536;

__global__ __launch_bounds__(256) void decoder_cross_attention_kernel(CrossAttnArgs a) {
    __shared__ __attribute__((aligned(16))) float qs[64];
    __shared__ float sc[kCrossMaxT];
    __shared__ float red[8];
    __shared__ __attribute__((aligned(16))) float part[16 * 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int row = blockIdx.x;
    const int head = blockIdx.y;
    const int sub = lane & 15;   // float4 slot inside the 64-wide head
    const int kq = lane >> 4;    // which of the 4 keys of this wave-instruction
    if (tid < 64) qs[tid] = a.q[(long)row * a.d + head * 64 + tid];
    __syncthreads();
    const float4 q4 = reinterpret_cast<const float4*>(qs)[sub];
    const float* kb = a.k + head * 64 + sub * 4;
    const float* vb = a.v + head * 64 + sub * 4;

    // pass 1: 8 key-row loads (8 KiB per wave) in flight before the first dot product is folded
    float mx = -INFINITY;
    for (int base = wave * 4; base < a.T; base += 16 * 8) {
        float4 kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = base + 16 * u + kq;
            kk[u] = *reinterpret_cast<const float4*>(kb + (long)(j < a.T ? j : 0) * a.ldkv);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = base + 16 * u + kq;
            float acc = 0.f;
            acc = fmaf(q4.x, kk[u].x, acc);
            acc = fmaf(q4.y, kk[u].y, acc);
            acc = fmaf(q4.z, kk[u].z, acc);
            acc = fmaf(q4.w, kk[u].w, acc);
            acc = row16_sum_1248(acc);
            if (j < a.T) {
                if (sub == 0) sc[j] = acc;
                mx = fmaxf(mx, acc);
            }
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));

    if (a.qk_debug) {
        float* dst = a.qk_debug + ((long)row * a.n_head + head) * a.T;
        for (int j = tid; j < a.T; j += 256) dst[j] = sc[j];
    }

    // pass 2
    float sum = 0.f;
    for (int j = tid; j < a.T; j += 256) {
        const float e = expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    const int rank = a.head_rank ? a.head_rank[head] : -1;
    float* ring_dst = nullptr;
    if (rank >= 0) {
        ring_dst = a.ring + (((long)rank * a.n_beam + a.beam_of_row[row]) * a.ring_rows + a.ring_row[row]) * a.T;
    }
    for (int j = tid; j < a.T; j += 256) {
        const float w = sc[j] / sum;
        sc[j] = w;
        if (ring_dst) ring_dst[j] = w;
    }
    __syncthreads();

    // pass 3
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = wave * 4; base < a.T; base += 16 * 8) {
        float4 vv[8];
        float ww[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = base + 16 * u + kq;
            const bool ok = j < a.T;
            const float4 t = *reinterpret_cast<const float4*>(vb + (long)(ok ? j : 0) * a.ldkv);
            vv[u] = t;
            ww[u] = ok ? sc[j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            o.x = fmaf(ww[u], vv[u].x, o.x);
            o.y = fmaf(ww[u], vv[u].y, o.y);
            o.z = fmaf(ww[u], vv[u].z, o.z);
            o.w = fmaf(ww[u], vv[u].w, o.w);
        }
    }
    reinterpret_cast<float4*>(part)[(wave * 4 + kq) * 16 + sub] = o;
    __syncthreads();
    if (tid < 64) {
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc += part[s * 64 + tid];
        a.out[(long)row * a.d + head * 64 + tid] = acc;
    }
}

void launch_decoder_cross_attention(const LaunchCtx& ctx, const CrossAttnArgs& a) {
    if (a.T > kCrossMaxT) throw std::invalid_argument("cross-attention: T too large");
    // per (row, head): read K and V slices once (T x 64 floats each)
    KernelScope ks(ctx, "dec_cross_attention", 4.0 * a.rows * (double)a.T * a.d,
                   4.0 * 2.0 * a.rows * (double)a.T * a.d);
    hipLaunchKernelGGL(decoder_cross_attention_kernel, dim3(a.rows, a.n_head), dim3(256), 0, ctx.stream, a);
    WLK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Cross-attention for decode STEPS (<= 8 query rows).  A step has only rows x heads = 8 independent
// softmaxes, far too few workgroups to pull 6 MB of K/V per layer out of HBM / Infinity Cache at
// speed (every kernel starts with a cold L2 on this part).  So the 1500 keys of each (row, head) are
// split over kCrossSplit workgroups: each loads its ~188 keys with ALL its K rows in flight at once
// (12 x 1 KiB per wave), leaves raw scores, a local max, a local exp-sum and a local weighted V sum;
// cross_merge_kernel folds the slices and, for alignment heads, writes softmax = exp(s - M) / L of
// the full row into the alignment window.
// ---------------------------------------------------------------------------------------------
constexpr int kCrossSplit = 8;
static_assert(kCrossSplit == kCrossSplitWays, "gemv1's merged operand load assumes the same split count");
constexpr int kCrossUnroll = 12;   // key-row loads in flight per wave: covers ceil(1500/8)=188 keys / 16

// UB = float4 chunks of K per lane in the folded query projection (d <= 256 UB floats); 0 = q comes from memory
// Round 4 (latency diet): what the first loads need (K / V / q bases, row stride, T, d, head count) are leading scalar
// arguments - preloaded into SGPRs at wave start - instead of fields of a by-value CrossAttnArgs that three dependent
// s_load stages fetched before the first key row could be requested; the rest rides in CrossSplitRest.
struct CrossSplitRest {
    long kv_off;                // batched steps: keys / values of row r are step_rows[r].cross_kv + kv_off
    const float* xq_x;          // folded query projection (UB > 0), see CrossAttnArgs
    const float* xq_w;
    const float* xq_b;
    const float* xq_gamma;
    const float* xq_beta;
    float xq_scale;
};
template <int UB>
__global__ __launch_bounds__(256) void cross_split_kernel(const float* __restrict__ k_in, const float* __restrict__ v_in,
                                                          const float* __restrict__ q_in, long ldkv, int T, int d, int n_head,
                                                          const StepRow* __restrict__ step_rows,
                                                          float* __restrict__ scores, float* __restrict__ pm,
                                                          float* __restrict__ pl, float* __restrict__ po, CrossSplitRest a) {
    __shared__ __attribute__((aligned(16))) float qs[64];
    __shared__ float sc[kCrossUnroll * 16];
    __shared__ float red[8];
    __shared__ __attribute__((aligned(16))) float part[16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x, head = blockIdx.y, ks = blockIdx.z;
    const int sub = lane & 15, kq = lane >> 4;
    const int chunk = (T + kCrossSplit - 1) / kCrossSplit;
    const int k_lo = ks * chunk, k_hi = min(T, k_lo + chunk);
    const gcf_ptr kbase = to_global(step_rows ? step_rows[row].cross_kv + a.kv_off : k_in);
    const gcf_ptr kb = kbase + head * 64 + sub * 4;
    const gcf_ptr vb = (step_rows ? kbase + d : to_global(v_in)) + head * 64 + sub * 4;
    float* srow = scores + ((long)row * n_head + head) * T;

    // every lane fetches its float4 of the query itself (16 distinct addresses per wave): no LDS stage, no barrier; first
    // in the queue, because loads return in order and the query is what the first dot product waits for
    float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (UB == 0) q4 = *reinterpret_cast<const float4*>(q_in + (long)row * d + head * 64 + sub * 4);
    // the key rows do not depend on the query: all of them are requested before the query is derived / fetched
    // (addresses by pointer increments: one 64-bit multiply per lane instead of one per row)
    const long row_step = 16 * ldkv;
    const int j0 = k_lo + wave * 4 + kq;
    const gcf_ptr k_first = kb + (long)j0 * ldkv, k_safe = kb + (long)k_lo * ldkv;
    const long v_minus_k = vb - kb;
    float4 kk[kCrossUnroll];
#pragma unroll
    for (int u = 0; u < kCrossUnroll; ++u) {
        // (no zeroing of the rows past k_hi: their scores are never stored or folded, and a select on the loaded value
        // lets hipcc sink each load into its own branch - twelve predicated loads with a vmcnt(0) in between)
        kk[u] = ldg4(j0 + 16 * u < k_hi ? k_first + u * row_step : k_safe);
    }
    // ... and so are the value rows (round 4: they used to be requested behind the softmax - one more memory round trip
    // on a kernel that is nothing but round trips)
    float4 vv[kCrossUnroll];
#pragma unroll
    for (int u = 0; u < kCrossUnroll; ++u)
        vv[u] = ldg4((j0 + 16 * u < k_hi ? k_first + u * row_step : k_safe) + v_minus_k);
    if constexpr (UB != 0) {
        // q_h = scale * (Wq[64 head + i, :] . LN(x_row) + b), i = 0..63: wave w derives i = 16 w .. 16 w + 15.  Statement for
        // statement gemv1_f32_kernel's arithmetic (fused LayerNorm statistics over lane-strided scalars, lane-strided
        // float4 fmaf chains, butterfly folds), which is also what every row of the multi-row GEMV computes.
        // Round 4: the residual row, the LayerNorm affine, the biases and the first 32 / UB weight rows are requested
        // together with the keys and values above - ONE memory round trip in front of the scores instead of the separate
        // dec_lnx_xq launch (4 us) in front of this one.
        const int K = d, K4 = K >> 2;
        const float* xrow = a.xq_x + (long)row * K;
        constexpr int NPASS = 32 / UB;      // outputs whose weight rows are in flight together (32 float4 per lane)
        float4 x[UB], ga[UB], be[UB];
        float v[UB * 4];
        float4 w[NPASS][UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int c = lane + 64 * u;
            const int cc = (c < K4 ? c : 0) * 4;
            x[u] = *reinterpret_cast<const float4*>(xrow + cc);
            ga[u] = *reinterpret_cast<const float4*>(a.xq_gamma + cc);
            be[u] = *reinterpret_cast<const float4*>(a.xq_beta + cc);
        }
#pragma unroll
        for (int i = 0; i < UB * 4; ++i) {
            const int c = lane + 64 * i;
            v[i] = xrow[c < K ? c : 0];
        }
        const float bias_l = (a.xq_b ? a.xq_b + head * 64 + wave * 16 + (lane & 15) : xrow)[0];   // output (lane & 15)'s bias
        auto load_rows = [&](int i0) {
#pragma unroll
            for (int r = 0; r < NPASS; ++r) {
                const int n = head * 64 + wave * 16 + i0 + r;
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int c = lane + 64 * u;
                    w[r][u] = *reinterpret_cast<const float4*>(a.xq_w + (long)n * K + (c < K4 ? c : 0) * 4);
                }
            }
        };
        load_rows(0);
        {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < UB * 4; ++i) {
                v[i] = (lane + 64 * i) < K ? v[i] : 0.f;
                sum += v[i];
            }
            sum = wave_sum(sum);
            const float mean = sum / (float)K;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < UB * 4; ++i) {
                const float t = (lane + 64 * i) < K ? v[i] - mean : 0.f;
                sq += t * t;
            }
            sq = wave_sum(sq);
            const float rstd = 1.0f / sqrtf(sq / (float)K + 1e-5f);
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                x[u].x = (x[u].x - mean) * rstd * ga[u].x + be[u].x;
                x[u].y = (x[u].y - mean) * rstd * ga[u].y + be[u].y;
                x[u].z = (x[u].z - mean) * rstd * ga[u].z + be[u].z;
                x[u].w = (x[u].w - mean) * rstd * ga[u].w + be[u].w;
            }
        }
#pragma unroll
        for (int i0 = 0; i0 < 16; i0 += NPASS) {
            if (i0 > 0) load_rows(i0);
#pragma unroll
            for (int r = 0; r < NPASS; ++r) {
                float acc = 0.f;
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    if (lane + 64 * u < K4) {
                        acc = fmaf(w[r][u].x, x[u].x, acc);
                        acc = fmaf(w[r][u].y, x[u].y, acc);
                        acc = fmaf(w[r][u].z, x[u].z, acc);
                        acc = fmaf(w[r][u].w, x[u].w, acc);
                    }
                }
                acc = wave_sum(acc);
                const float bias_r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bias_l), i0 + r));
                if (lane == 0) {
                    float o = acc;
                    if (a.xq_b) o += bias_r;
                    o *= a.xq_scale;
                    qs[wave * 16 + i0 + r] = o;
                }
            }
        }
        __syncthreads();
        q4 = reinterpret_cast<const float4*>(qs)[sub];
    }
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < kCrossUnroll; ++u) {
        const int j = k_lo + wave * 4 + 16 * u + kq;
        float acc = 0.f;
        acc = fmaf(q4.x, kk[u].x, acc);
        acc = fmaf(q4.y, kk[u].y, acc);
        acc = fmaf(q4.z, kk[u].z, acc);
        acc = fmaf(q4.w, kk[u].w, acc);
        acc = row16_sum_1248(acc);
        if (j < k_hi) {
            if (sub == 0) { sc[j - k_lo] = acc; srow[j] = acc; }
            mx = fmaxf(mx, acc);
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < k_hi - k_lo; j += 256) {
        const float e = expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);

    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < kCrossUnroll; ++u) {
        const int j = k_lo + wave * 4 + 16 * u + kq;
        const float w = j < k_hi ? sc[j - k_lo] : 0.f;
        o.x = fmaf(w, vv[u].x, o.x);
        o.y = fmaf(w, vv[u].y, o.y);
        o.z = fmaf(w, vv[u].z, o.z);
        o.w = fmaf(w, vv[u].w, o.w);
    }
    reinterpret_cast<float4*>(part)[(wave * 4 + kq) * 16 + sub] = o;
    __syncthreads();
    const long slot = ((long)row * n_head + head) * kCrossSplit + ks;
    if (tid < 64) {
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc += part[s * 64 + tid];
        po[slot * 64 + tid] = acc;
    }
    if (tid == 0) { pm[slot] = mx; pl[slot] = sum; }
}

__global__ __launch_bounds__(256) void cross_merge_kernel(CrossAttnArgs a, const float* __restrict__ scores,
                                                          const float* __restrict__ pm, const float* __restrict__ pl,
                                                          const float* __restrict__ po) {
    const int tid = threadIdx.x;
    const int row = blockIdx.x, head = blockIdx.y;
    const long base = ((long)row * a.n_head + head) * kCrossSplit;
    float M = pm[base];
#pragma unroll
    for (int s = 1; s < kCrossSplit; ++s) M = fmaxf(M, pm[base + s]);
    float L = 0.f;
    float f[kCrossSplit];
#pragma unroll
    for (int s = 0; s < kCrossSplit; ++s) {
        f[s] = expf(pm[base + s] - M);
        L += pl[base + s] * f[s];
    }
    if (tid < 64) {
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < kCrossSplit; ++s) acc += po[(base + s) * 64 + tid] * f[s];
        a.out[(long)row * a.d + head * 64 + tid] = acc / L;
    }
    const int rank = a.head_rank ? a.head_rank[head] : -1;
    if (rank >= 0) {
        float* dst = a.step_rows
                         ? a.step_rows[row].ring + ((long)rank * a.ring_rows + a.step_rows[row].ring_row) * a.T
                         : a.ring + (((long)rank * a.n_beam + a.beam_of_row[row]) * a.ring_rows + a.ring_row[row]) * a.T;
        const float* srow = scores + ((long)row * a.n_head + head) * a.T;
        for (int j = tid; j < a.T; j += 256) dst[j] = expf(srow[j] - M) / L;
    }
}

void launch_decoder_cross_attention_split(const LaunchCtx& ctx, const CrossAttnArgs& a, float* scores, float* pm,
                                          float* pl, float* po, bool merge) {
    if ((a.T + kCrossSplit - 1) / kCrossSplit > kCrossUnroll * 16) throw std::invalid_argument("cross-attention: T too large");
    {
        KernelScope ks(ctx, "dec_cross_split", 4.0 * a.rows * (double)a.T * a.d, 4.0 * 2.0 * a.rows * (double)a.T * a.d);
        const dim3 grid(a.rows, a.n_head, kCrossSplit);
        const CrossSplitRest rest{a.kv_off, a.xq_x, a.xq_w, a.xq_b, a.xq_gamma, a.xq_beta, a.xq_scale};
#define WLK_CROSS_SPLIT(UBv)                                                                                              \
    hipLaunchKernelGGL(cross_split_kernel<UBv>, grid, dim3(256), 0, ctx.stream, a.k, a.v, a.q, a.ldkv, a.T, a.d, a.n_head, \
                       a.step_rows, scores, pm, pl, po, rest)
        if (a.xq_w) {
            if (a.d % 256 != 0 || a.d < 256 || a.d > 2048) throw std::invalid_argument("cross-attention: cannot fold the query projection");
            const int ub = a.d / 256;
            if (ub <= 2) WLK_CROSS_SPLIT(2);
            else if (ub <= 4) WLK_CROSS_SPLIT(4);
            else WLK_CROSS_SPLIT(8);
        } else {
            WLK_CROSS_SPLIT(0);
        }
#undef WLK_CROSS_SPLIT
        WLK_HIP(hipGetLastError());
    }
    if (merge) {   // beam-1 steps fold the merge into the out-projection GEMV (GemmArgs::mg_*)
        KernelScope ks(ctx, "dec_cross_merge");
        hipLaunchKernelGGL(cross_merge_kernel, dim3(a.rows, a.n_head), dim3(256), 0, ctx.stream, a, scores, pm, pl, po);
        WLK_HIP(hipGetLastError());
    }
}
// Round 3 measured this fold SLOWER than the separate dec_lnx_xq launch (141-142 vs 145-146 audio-s/s,
// profiles/r03_ab_xq_fold.txt): the folded prologue was a chain of memory round trips (residual row -> LayerNorm
// statistics -> affine -> four batches of weight rows) in front of the keys.  Round 4 requests all of it together with the
// keys and values (one round trip), and a dependent launch now costs ~3.9 us of a 200 us step: default ON, WLK_XQ_FOLD=0
// restores the separate launch (profiles/r04h_ab_xq_fold.txt).  All golden streams are bit-identical either way (the fold
// reproduces the GEMV's arithmetic).
// Where it pays: every (head, split) workgroup streams its head's 64 rows of Wq itself, n_head * 8 * 64 * d * 4 bytes per
// layer out of the L2s - 8 MB on base.en (neutral to +1 %), 52 MB on large-v3, where the split kernel went from 8.4 to
// 19 us per layer and a step from 2.32 to 2.52 ms (profiles/r04m_large_v3_folds.txt).  Default: d <= 512.
// WLK_XQ_FOLD=1 folds wherever the kernel can, =0 never.
bool cross_split_folds_query(int d) {
    static const int mode = [] {
        const char* e = getenv("WLK_XQ_FOLD");
        return !e ? -1 : (e[0] == '0' ? 0 : 1);
    }();
    if (mode == 0 || d % 256 != 0 || d < 256 || d > 2048) return false;
    return mode == 1 || d <= 512;
}
size_t cross_split_scratch_floats(int rows, int n_head, int T) {
    return (size_t)rows * n_head * ((size_t)T + kCrossSplit * (2 + 64));
}

// ---- Infinity-Cache prefetch of the decoder weights beside a single-token step ----------------------------------------------
// A single-row decode step of a model whose decoder is larger than the 256 MB memory-side cache streams every weight from
// HBM inside a chain of dependent launches: each GEMV waits for its first bytes at HBM latency and then has 3 - 5 us to
// move 7 - 26 MB (large-v3: 3.0 - 4.5 TB/s per launch, 0.31 of HBM over the step; the same launches on cache-resident weights
// run at 6.7 - 8.6 TB/s, scripts/mall_probe.py).  The weights do not depend on the token, so ONE kernel on a side stream
// reads layer l + 1's matrices while layer l's chain runs; nothing is kept - the point is that the lines are in the
// Infinity Cache when the GEMVs ask for them.  Pacing: the chain's self-attention launch of layer l stores (step sequence
// number, l) into a progress word (ProgressMark), the prefetcher's workgroups wait for it before they start on layer
// l + 1 - further ahead the cache (2.5 layers of large-v3) would lose the lines again before they are used.  A bounded
// wait: a chain that never comes ends the prefetcher, it cannot hang the GPU.  Reads only: results cannot change.
// (As a forked branch of the step's hipGraph the same reads made the replay 3 x slower - HIP 7.2 replays a graph with a
// parallel branch segment by segment - and crashed under GPU_MAX_HW_QUEUES=2: profiles/r06k_mall_ab_large-v3.txt.)
__global__ __launch_bounds__(256) void mall_prefetch_step_kernel(const TouchRanges* __restrict__ table, int n_layer, unsigned seq,
                                                                 const unsigned long long* __restrict__ progress, int lead,
                                                                 float* __restrict__ sink) {
    __shared__ int go;
    float acc = 0.f;
    const unsigned stride = gridDim.x * 256u;
#pragma unroll 1
    for (int l = 1; l < n_layer; ++l) {
        if (threadIdx.x == 0) {
            int ok = 0;
#pragma unroll 1
            for (int polls = 0; polls < 40000; ++polls) {            // ~1 us per poll: 40 ms at most
                const unsigned long long v = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(v >> 32) == seq && (int)(unsigned)v + lead >= l - 1) {
                    ok = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(127);
            }
            go = ok;
        }
        __syncthreads();
        const int ok = go;
        __syncthreads();
        if (!ok) break;
        const TouchRanges t = table[l];
#pragma unroll 1
        for (int r = 0; r < 6; ++r) {
            const float4* __restrict__ p = reinterpret_cast<const float4*>(t.p[r]);
            const unsigned n = t.n16[r];
            unsigned i = blockIdx.x * 256u + threadIdx.x;
#pragma unroll 1
            for (; i + 7u * stride < n; i += 8u * stride) {           // eight independent 16-byte loads per lane in flight
                const float4 a = p[i], b = p[i + stride], c = p[i + 2u * stride], e = p[i + 3u * stride];
                const float4 f = p[i + 4u * stride], g = p[i + 5u * stride], h = p[i + 6u * stride], k = p[i + 7u * stride];
                acc += ((a.x + b.x) + (c.x + e.x)) + ((f.x + g.x) + (h.x + k.x));
            }
            for (; i < n; i += stride) acc += p[i].x;
        }
    }
    if (acc == 1.2345678e-38f) *sink = acc;                       // never true in practice; keeps the loads alive
}

void launch_mall_prefetch_step(const LaunchCtx& ctx, const TouchRanges* table, int n_layer, unsigned seq,
                               const unsigned long long* progress, int lead, float* sink) {
    if (n_layer < 2) return;
    KernelScope ks(ctx, "dec_mall_prefetch");
    // 256 workgroups x 256 lanes x 4 loads of 16 bytes = 4 MB in flight: ~4 TB/s at 1 us of latency, beside the step's own chain
    hipLaunchKernelGGL(mall_prefetch_step_kernel, dim3(256), dim3(256), 0, ctx.stream, table, n_layer, seq, progress, lead, sink);
    WLK_HIP(hipGetLastError());
}

// beam reorder of the self-attention caches: dst[l][b] = src[l][source_rows[b]]
__global__ __launch_bounds__(256) void kv_gather_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                        const int* __restrict__ source_rows, int n_rows, int len,
                                                        int d, int ctx_len) {
    const int l = blockIdx.z, b = blockIdx.y, t = blockIdx.x;
    if (t >= len) return;
    const long layer = (long)l * n_rows * ctx_len * d;
    const float* s = src + layer + ((long)source_rows[b] * ctx_len + t) * d;
    float* o = dst + layer + ((long)b * ctx_len + t) * d;
    for (int c = threadIdx.x; c < d; c += 256) o[c] = s[c];
}

void launch_kv_gather(const LaunchCtx& ctx, const float* src, float* dst, const int* source_rows, int n_rows,
                      int len, int d, int ctx_len, int n_layer) {
    if (len <= 0) return;
    KernelScope ks(ctx, "dec_kv_gather");
    hipLaunchKernelGGL(kv_gather_kernel, dim3(len, n_rows, n_layer), dim3(256), 0, ctx.stream, src, dst,
                       source_rows, n_rows, len, d, ctx_len);
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

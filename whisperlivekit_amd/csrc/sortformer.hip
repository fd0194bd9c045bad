// Kernels of the streaming Sortformer diarizer network that are not plain GEMM / LayerNorm.
//
// The reference calls NeMo's SortformerEncLabelModel.forward_streaming_step
// (whisperlivekit/diarization/sortformer_backend.py:293-300); the network is a FastConformer encoder
// (dw_striding sub-sampling x8, 17 Conformer blocks with relative-position attention) followed by an 18-block
// post-LN Transformer and a sigmoid head.  NeMo is not part of the reference tree, so every kernel cites the
// NeMo module it restates; parity of this part is UNPINNED (oracle/sortformer_oracle.py says the same).
//
// Sequences here are short (<= 188 + 188 + ~26 frames) and the work per chunk is ~20 GFLOP of GEMM, so these
// helper kernels are written for low launch latency and coalesced channel-last access, not for peak rates.
#include <atomic>
#include <cstdlib>

#include "common.h"
#include "wave_ops.h"

namespace wlk {

// ---- stacked sessions -------------------------------------------------------------------------------------------
// A stacked step (sortformer_api.hip) lays the sessions' rows one after the other; the kernels that look along the time
// axis find their session in a by-value table.  Constant indices only: a runtime index into kernel arguments would put
// the whole argument struct into scratch (measured on the GEMMs in round 2), so lookups are select chains over <= 8 slots.
__device__ __forceinline__ int seg_find(const int (&start)[kSfMaxSegments], int n, int v) {     // largest s < n with start[s] <= v
    int s = 0;
#pragma unroll
    for (int i = 1; i < kSfMaxSegments; ++i)
        if (i < n && v >= start[i]) s = i;
    return s;
}
__device__ __forceinline__ int seg_pick(const int (&a)[kSfMaxSegments], int s) {
    int v = a[0];
#pragma unroll
    for (int i = 1; i < kSfMaxSegments; ++i) v = s == i ? a[i] : v;
    return v;
}

// ---- ConvSubsampling('dw_striding') ----------------------------------------------------------------
// conv0: Conv2d(1, C, 3, stride 2, padding 1) + ReLU.  One workgroup per output position, one thread per
// channel: the 9 input taps are wave-uniform, the output is channels-last so that the following pointwise
// convolutions are GEMMs over [positions, C].  Session s: input frames [in_start, in_start + in_len), output frames from
// out_start on.
__global__ __launch_bounds__(256) void sf_conv0_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ out, SfConvSegs sg,
                                                       int F, int F1, int C) {
    const int pos = blockIdx.x, tg = pos / F1, f1 = pos - tg * F1;
    const int s = seg_find(sg.out_start, sg.n, tg);
    const int t1 = tg - seg_pick(sg.out_start, s), T = seg_pick(sg.in_len, s);
    const float* src = in + (long)seg_pick(sg.in_start, s) * F;
    float x[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int tt = 2 * t1 - 1 + ky, ff = 2 * f1 - 1 + kx;
            x[ky * 3 + kx] = (tt >= 0 && tt < T && ff >= 0 && ff < F) ? src[(long)tt * F + ff] : 0.f;
        }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc = fmaf(x[k], w[c * 9 + k], acc);
        out[(long)pos * C + c] = fmaxf(acc + b[c], 0.f);
    }
}

void launch_sf_conv0(const LaunchCtx& ctx, const float* in, const float* w, const float* b, float* out, const SfConvSegs& sg,
                     int F, int C) {
    const int F1 = sf_sub_len(F);
    if (sg.out_total <= 0) return;
    KernelScope ks(ctx, "sf_conv0", 18.0 * sg.out_total * F1 * C, 4.0 * ((double)sg.in_total * F + (double)sg.out_total * F1 * C));
    hipLaunchKernelGGL(sf_conv0_kernel, dim3(sg.out_total * F1), dim3(C < 256 ? C : 256), 0, ctx.stream, in, w, b, out, sg, F, F1, C);
    WLK_HIP(hipGetLastError());
}

// depthwise Conv2d(C, C, 3, stride 2, padding 1, groups C) on channels-last data; weights tap-major [9][C]
__global__ __launch_bounds__(256) void sf_dwconv2d_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ out, SfConvSegs sg,
                                                          int Fi, int Fo, int C) {
    const int pos = blockIdx.x, tg = pos / Fo, fo = pos - tg * Fo;
    const int s = seg_find(sg.out_start, sg.n, tg);
    const int to = tg - seg_pick(sg.out_start, s), Ti = seg_pick(sg.in_len, s);
    const float* src = in + (long)seg_pick(sg.in_start, s) * Fi * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tt = 2 * to - 1 + ky, ff = 2 * fo - 1 + kx;
                const bool ok = tt >= 0 && tt < Ti && ff >= 0 && ff < Fi;
                const float x = ok ? src[((long)tt * Fi + ff) * C + c] : 0.f;
                acc = fmaf(x, w[(ky * 3 + kx) * C + c], acc);
            }
        out[(long)pos * C + c] = acc + b[c];
    }
}

void launch_sf_dwconv2d(const LaunchCtx& ctx, const float* in, const float* w, const float* b, float* out, const SfConvSegs& sg,
                        int Fi, int C) {
    const int Fo = sf_sub_len(Fi);
    if (sg.out_total <= 0) return;
    KernelScope ks(ctx, "sf_dwconv2d", 18.0 * sg.out_total * Fo * C, 4.0 * ((double)sg.in_total * Fi * C + (double)sg.out_total * Fo * C));
    hipLaunchKernelGGL(sf_dwconv2d_kernel, dim3(sg.out_total * Fo), dim3(C < 256 ? C : 256), 0, ctx.stream, in, w, b, out, sg, Fi,
                       Fo, C);
    WLK_HIP(hipGetLastError());
}

SfConvSegs sf_conv_segs_next(const SfConvSegs& prev) {      // the table of the next stride-2 stage: its input is prev's output
    SfConvSegs nx;
    nx.n = prev.n;
    int off = 0;
    for (int s = 0; s < prev.n; ++s) {
        const int len = (s + 1 < prev.n ? prev.out_start[s + 1] : prev.out_total) - prev.out_start[s];
        nx.in_start[s] = prev.out_start[s];
        nx.in_len[s] = len;
        nx.out_start[s] = off;
        off += sf_sub_len(len);
    }
    nx.in_total = prev.out_total;
    nx.out_total = off;
    return nx;
}

// x[row] = (the session's context row | its chunk row) * scale: the encoder input [speaker cache | FIFO | chunk] of every
// session of the step, assembled from the uploaded context rows (already at their stacked positions) and the stem's
// output (chunk rows of all sessions one after the other) - ConformerEncoder's xscaling rides along
__global__ __launch_bounds__(256) void sf_assemble_kernel(const float* __restrict__ ctx_rows, const float* __restrict__ chunk_rows,
                                                          float* __restrict__ x, SfSegments rows, SfSegments chunks, int d4,
                                                          float scale) {
    const int r = blockIdx.x;
    const int s = seg_find(rows.start, rows.n, r);
    const int t = r - seg_pick(rows.start, s), n_chunk = seg_pick(chunks.len, s), n_ctx = seg_pick(rows.len, s) - n_chunk;
    const float4* src = reinterpret_cast<const float4*>(t < n_ctx ? ctx_rows + (long)r * d4 * 4
                                                                  : chunk_rows + (long)(seg_pick(chunks.start, s) + t - n_ctx) * d4 * 4);
    float4* dst = reinterpret_cast<float4*>(x + (long)r * d4 * 4);
    for (int c = threadIdx.x; c < d4; c += blockDim.x) {
        const float4 v = src[c];
        dst[c] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
    }
}

void launch_sf_assemble(const LaunchCtx& ctx, const float* ctx_rows, const float* chunk_rows, float* x, const SfSegments& rows,
                        const SfSegments& chunks, int d, float scale) {
    const int R = rows.n > 0 ? rows.start[rows.n - 1] + rows.len[rows.n - 1] : 0;
    if (R <= 0) return;
    if (d % 4) throw std::invalid_argument("sortformer: model width must be a multiple of 4");
    KernelScope ks(ctx, "sf_assemble", 0.0, 8.0 * R * d);
    hipLaunchKernelGGL(sf_assemble_kernel, dim3(R), dim3(128), 0, ctx.stream, ctx_rows, chunk_rows, x, rows, chunks, d / 4, scale);
    WLK_HIP(hipGetLastError());
}

// ---- attention over one short sequence ---------------------------------------------------------------
// One wave per query row, four rows per workgroup, grid (ceil(T/4), heads).
//   Conformer (RelPositionMultiHeadAttention.forward): score[i][j] = ((q_i + u).k_j + (q_i + v).p[T-1-i+j]) / sqrt(dk)
//     - the rel_shift of matrix_bd is just that index map, so it is never materialised;
//   Transformer (NeMo MultiHeadAttention): score = q.k with q and k pre-scaled by dk^-1/4 in the projection GEMM.
// Scores: DHP/4 lanes share one key row (a coalesced 4*dh-byte read per row, 4 rows in flight per lane) and fold
// their partial dot products with xor shuffles; softmax through LDS; P.V with lanes over head dims (DHP-padded)
// and 64/DHP key sub-groups folded by shuffles.
template <int DHP>
__global__ __launch_bounds__(256) void sf_attention_kernel(SfAttnArgs a) {
    constexpr int LPK = DHP / 4;          // lanes per key row (one float4 each): 16 or 8
    constexpr int KPI = 64 / LPK;         // key rows per wave-instruction: 4 or 8
    constexpr int UNR = 4;                // row loads in flight per lane
    __shared__ __attribute__((aligned(16))) float qs[4][2][64];
    __shared__ float sc[4][kSfMaxFrames];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i_raw = blockIdx.x * 4 + wave, i = min(i_raw, a.T - 1), h = blockIdx.y;
    const int dh = a.dh;
    if (lane < dh) {
        const float q = a.q[(long)i * a.ldq + h * dh + lane];
        qs[wave][0][lane] = a.bias_u ? q + a.bias_u[h * dh + lane] : q;
        qs[wave][1][lane] = a.bias_v ? q + a.bias_v[h * dh + lane] : q;
    } else {
        qs[wave][0][lane] = 0.f;
        qs[wave][1][lane] = 0.f;
    }
    __syncthreads();
    // scores: LPK lanes share one key row (coalesced 4*dh-byte reads), partial dot products folded by xor shuffles
    const int sub = lane % LPK, kq = lane / LPK;
    const bool live4 = sub * 4 < dh;
    const float4 qu = reinterpret_cast<const float4*>(qs[wave][0])[sub];
    const float4 qv = reinterpret_cast<const float4*>(qs[wave][1])[sub];
    const float* kb = a.k + h * dh + (live4 ? sub * 4 : 0);
    const float* pb = a.pos ? a.pos + (long)(a.pos_row0 - i) * a.ldp + h * dh + (live4 ? sub * 4 : 0) : nullptr;
    float mx = -INFINITY;
    for (int j0 = 0; j0 < a.T; j0 += KPI * UNR) {
        float4 kk[UNR], pp[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int j = min(j0 + u * KPI + kq, a.T - 1);
            kk[u] = *reinterpret_cast<const float4*>(kb + (long)j * a.ldk);
            pp[u] = pb ? *reinterpret_cast<const float4*>(pb + (long)j * a.ldp) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int j = j0 + u * KPI + kq;
            float ac = 0.f;
            ac = fmaf(qu.x, kk[u].x, ac); ac = fmaf(qu.y, kk[u].y, ac); ac = fmaf(qu.z, kk[u].z, ac); ac = fmaf(qu.w, kk[u].w, ac);
            float bd = 0.f;
            bd = fmaf(qv.x, pp[u].x, bd); bd = fmaf(qv.y, pp[u].y, bd); bd = fmaf(qv.z, pp[u].z, bd); bd = fmaf(qv.w, pp[u].w, bd);
            float s = live4 ? ac + bd : 0.f;
#pragma unroll
            for (int off = LPK / 2; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
            s *= a.scale;
            if (j < a.T) {
                if (sub == 0) sc[wave][j] = s;
                mx = fmaxf(mx, s);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    __syncthreads();
    float sum = 0.f;
    for (int j = lane; j < a.T; j += 64) {
        const float e = expf(sc[wave][j] - mx);
        sc[wave][j] = e;
        sum += e;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float inv = 1.0f / sum;
    __syncthreads();
    constexpr int NG = 64 / DHP;
    const int d = lane % DHP, grp = lane / DHP;
    const bool live = d < dh;
    const float* vp = a.v + h * dh + (live ? d : 0);
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int j = grp;
    for (; j + 3 * NG < a.T; j += 4 * NG) {
        const float v0 = vp[(long)j * a.ldv], v1 = vp[(long)(j + NG) * a.ldv], v2 = vp[(long)(j + 2 * NG) * a.ldv],
                    v3 = vp[(long)(j + 3 * NG) * a.ldv];
        acc0 = fmaf(sc[wave][j], v0, acc0);
        acc1 = fmaf(sc[wave][j + NG], v1, acc1);
        acc2 = fmaf(sc[wave][j + 2 * NG], v2, acc2);
        acc3 = fmaf(sc[wave][j + 3 * NG], v3, acc3);
    }
    for (; j < a.T; j += NG) acc0 = fmaf(sc[wave][j], vp[(long)j * a.ldv], acc0);
    float acc = (acc0 + acc1) + (acc2 + acc3);
#pragma unroll
    for (int off = 32; off >= DHP; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (grp == 0 && live && i_raw < a.T) a.out[(long)i * a.ldo + h * dh + d] = acc * inv;
}

// ---- the same attention on the matrix cores (round 6) ------------------------------------------------------------
// The one-wave-per-query kernel above re-reads every key (and position) row once per query row: 2 T^2 dh floats per head
// out of L2 (T = 291: 350 MB per launch, 33 us of a 4.5 ms chunk x 35 launches = a quarter of the chunk).  Here a workgroup
// owns (segment, head, 16 queries) and its four waves split the KEY tiles, v_mfma_f32_16x16x4_f32 throughout:
//   1. G^T[r'][n] = pos[rbase + r'] . (q_n + v)  for r' in [0, T + 15), rbase = pos_row0 - i0 - 15: every relative-position
//      product the 16 queries can need, ONCE (the rel_shift of matrix_bd is then the index map j - n + 15 into that row);
//   2. S^T[j][n] = k_j . (q_n + u)  per 16-key tile (each lane owns a query column: acc[r] = key 4 (lane >> 4) + r);
//   3. softmax over s = (S + G shifted) * scale per query row, exponentials left un-normalised in LDS, 1 / sum aside;
//   4. O^T[d][n] += V^T[d][j] P^T[j][n] per key tile, the four waves' partial outputs added in wave order through LDS.
// K / pos / V fragments come straight from global memory (64 contiguous bytes per row and instruction), Q and P from LDS
// with row strides = 4 mod 32 floats (conflict-free b128 / b32 reads).  The d (and key) values of one MFMA step are
// {16 c + 4 g + e : g = 0..3} - a permutation of the dot product's terms, the same one in A and B.
// 291 frames, dh 64: 152 workgroups x ~930 MFMAs.
template <int DHP>
__global__ __launch_bounds__(256) void sf_attention_mfma_kernel(SfAttnArgs a) {
    constexpr int C = DHP / 16;                 // 16-wide d chunks (AC / G products) = 16-row d tiles (P.V)
    constexpr int QLD = DHP + 4;                // LDS row stride of the query tiles
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const int seg = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * 16;
    const int base = a.n_seg > 0 ? a.seg_start[seg] : 0, T = a.n_seg > 0 ? a.seg_T[seg] : a.T;
    if (i0 >= T) return;
    const int dh = a.dh;
    const int n_kt = (T + 15) >> 4, n_rt = (T + 15 + 15) >> 4;
    const int SP = n_kt * 16 + 4, GP = n_rt * 16 + 4;
    float* qu = smem;                           // [16][QLD]
    float* qv = qu + 16 * QLD;                  // [16][QLD]
    float* inv = qv + 16 * QLD;                 // [16]
    float* S = inv + 16;                        // [16][SP]
    float* G = S + 16 * SP;                     // [16][GP] (pos only); later the P.V merge buffer [3][C][4][64]
    const float* qbase = a.q + (long)base * a.ldq + h * dh;
    const float* kbase = a.k + (long)base * a.ldk + h * dh;
    const float* vbase = a.v + (long)base * a.ldv + h * dh;
    // Round 6, second form.  The first form's ISA: every load sat in its own exec-mask branch (`dlive[c] ? load : 0`, `if
    // (a.bias_u)`), hipcc joins such branches with s_waitcnt vmcnt(0), and each tile loop ran load -> wait -> 16 dependent MFMAs
    // with nothing in flight: fifteen exposed round trips per wave at 291 frames, ~15 of the launch's 20 us.  Now every load of
    // the kernel is unconditional (addresses clamped into the row / the head; what a dead chunk loads is multiplied by the query
    // fragments' zeros there, dead value rows are never stored), and each operand stream runs kSfRing tiles ahead in a rotating
    // register ring whose refills the compiler can count.  Same fragments into the same MFMAs in the same order: bit-identical.
    constexpr int kSfRing = 3;
    static_assert((16 * DHP) % 256 == 0, "whole passes of the staging loop");
    {
        constexpr int NQ = 16 * DHP / 256;
        const float* up = a.bias_u ? a.bias_u + h * dh : qbase;     // an absent bias reads the queries (always readable), then counts as 0
        const float* vp = a.bias_v ? a.bias_v + h * dh : qbase;
        float q[NQ], u[NQ], v[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int e = threadIdx.x + 256 * i, r = e / DHP, d = min(e - r * DHP, dh - 1);
            q[i] = qbase[(long)min(i0 + r, T - 1) * a.ldq + d];
            u[i] = up[d];
            v[i] = vp[d];
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) { pin_loaded(q[i]); pin_loaded(u[i]); pin_loaded(v[i]); }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int e = threadIdx.x + 256 * i, r = e / DHP, d = e - r * DHP;
            const bool live = d < dh;
            const float qq = live ? q[i] : 0.f, uu = live && a.bias_u ? u[i] : 0.f, vv = live && a.bias_v ? v[i] : 0.f;
            qu[r * QLD + d] = qq + uu;
            qv[r * QLD + d] = qq + vv;
        }
    }
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    // chunk c of a lane's fragment row lies inside the head iff 16 c + 4 g < dh; a dead chunk re-reads chunk 0 (valid memory)
    const int coff[4] = {0, 16 + 4 * g < dh ? 16 : 0, 32 + 4 * g < dh ? 32 : 0, 48 + 4 * g < dh ? 48 : 0};
    const int g4 = 4 * g < dh ? 4 * g : 0;
    auto load_rows = [&](const float* rows, long ld, int row, int hi, float4 (&x)[C]) {       // row index clamped into [0, hi]
        const float* p = rows + (long)min(max(row, 0), hi) * ld + g4;
#pragma unroll
        for (int c = 0; c < C; ++c) x[c] = *reinterpret_cast<const float4*>(p + coff[c]);
    };
    const int vcol[4] = {min(n, dh - 1), min(16 + n, dh - 1), min(32 + n, dh - 1), min(48 + n, dh - 1)};
    auto load_v = [&](int kt, float (&vv)[4][C]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* vr = vbase + (long)min(kt * 16 + 4 * g + e, T - 1) * a.ldv;
#pragma unroll
            for (int c = 0; c < C; ++c) vv[e][c] = vr[vcol[c]];
        }
    };
    // one 16-row tile of fragments times the query fragments (the MFMA order of the first form: c, then x y z w)
    auto dot_rows = [&](const float4 (&x)[C], const float4 (&b)[C]) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[c].x, b[c].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[c].y, b[c].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[c].z, b[c].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[c].w, b[c].w, acc, 0, 0, 0);
        }
        return acc;
    };
    const int my_kt = wave < n_kt ? (n_kt - wave + 3) / 4 : 0;          // this wave's key tiles kt = wave + 4 i
    const int my_rt = a.pos && wave < n_rt ? (n_rt - wave + 3) / 4 : 0;  // ... and relative-position tiles
    const int rbase = a.pos_row0 - i0 - 15;
    const float* pbase = a.pos ? a.pos + h * dh : kbase;                 // (no table: the ring below is primed with key rows and unused)
    const long ldp = a.pos ? a.ldp : a.ldk;
    const int phi = a.pos ? 2 * a.pos_row0 : T - 1;
    float4 px[kSfRing][C], kx[kSfRing][C];
#pragma unroll
    for (int d = 0; d < kSfRing; ++d) load_rows(pbase, ldp, rbase + (wave + 4 * d) * 16 + n, phi, px[d]);
#pragma unroll
    for (int d = 0; d < kSfRing; ++d) load_rows(kbase, a.ldk, (wave + 4 * d) * 16 + n, T - 1, kx[d]);
    __syncthreads();                            // the staged queries
    // B fragments of this lane's query: d = 16 c + 4 g + {0..3} (zeros past the head: what a dead chunk loaded counts for nothing)
    float4 bu[C], bv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        bu[c] = *reinterpret_cast<const float4*>(qu + n * QLD + 16 * c + 4 * g);
        bv[c] = *reinterpret_cast<const float4*>(qv + n * QLD + 16 * c + 4 * g);
    }
    // (ring slots are indexed statically - a rotating ring's register moves would read, i.e. wait for, the loads in flight;
    // the last group of a stream multiplies up to kSfRing - 1 clamped tiles nobody stores)
    for (int bi = 0; bi < my_rt; bi += kSfRing) {
#pragma unroll
        for (int d = 0; d < kSfRing; ++d) {
            const int i = bi + d, rt = wave + 4 * i;
            const f32x4 acc = dot_rows(px[d], bv);
            if (i < my_rt) *reinterpret_cast<float4*>(G + n * GP + rt * 16 + 4 * g) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            load_rows(pbase, ldp, rbase + (rt + 4 * kSfRing) * 16 + n, phi, px[d]);     // past the end: a clamped row nobody uses
        }
    }
    float vx[kSfRing][4][C];
#pragma unroll
    for (int d = 0; d < kSfRing; ++d) load_v(wave + 4 * d, vx[d]);
    for (int bi = 0; bi < my_kt; bi += kSfRing) {
#pragma unroll
        for (int d = 0; d < kSfRing; ++d) {
            const int i = bi + d, kt = wave + 4 * i;
            const f32x4 acc = dot_rows(kx[d], bu);
            if (i < my_kt) *reinterpret_cast<float4*>(S + n * SP + kt * 16 + 4 * g) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            load_rows(kbase, a.ldk, (kt + 4 * kSfRing) * 16 + n, T - 1, kx[d]);
        }
    }
    __syncthreads();
    // softmax of the four query rows of this wave
#pragma unroll 1
    for (int rq = 0; rq < 4; ++rq) {
        const int r = 4 * wave + rq;
        float* srow = S + r * SP;
        const float* grow = G + r * GP + 15 - r;
        float mx = -INFINITY;
        for (int j = lane; j < T; j += 64) {
            const float s = (a.pos ? srow[j] + grow[j] : srow[j]) * a.scale;
            srow[j] = s;
            mx = fmaxf(mx, s);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        float sum = 0.f;
        for (int j = lane; j < n_kt * 16; j += 64) {
            const float e = j < T ? expf(srow[j] - mx) : 0.f;
            srow[j] = e;
            sum += e;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
        if (lane == 0) inv[r] = 1.0f / sum;
    }
    __syncthreads();
    // O^T[d][n]: A = V^T (row m = d, k = key), B = P^T; one MFMA step takes keys {16 kt + 4 g + e}
    f32x4 o[C];
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int bi = 0; bi < my_kt; bi += kSfRing) {
#pragma unroll
        for (int d = 0; d < kSfRing; ++d) {
            const int i = bi + d, kt = wave + 4 * i;
            // (a tile past the wave's last one multiplies probabilities of 0: its S entries are not read - the index is clamped
            // to the wave's own first tile and the products are masked to zero)
            const int ktc = i < my_kt ? kt : wave;
            float4 p4 = *reinterpret_cast<const float4*>(S + n * SP + ktc * 16 + 4 * g);
            if (i >= my_kt) p4 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float pe[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < C; ++c) o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(vx[d][e][c], pe[e], o[c], 0, 0, 0);
            load_v(kt + 4 * kSfRing, vx[d]);
        }
    }
    float* red = G;                              // [3][C][4][64]: the scores' G is dead by now
    if (wave > 0) {
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((wave - 1) * C + c) * 4 + r) * 64 + lane] = o[c][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[c][r] += red[((w * C + c) * 4 + r) * 64 + lane];
    if (i0 + n < T) {
        const float sc = inv[n];
        float* orow = a.out + (long)(base + i0 + n) * a.ldo + h * dh;
#pragma unroll
        for (int c = 0; c < C; ++c)
            if (16 * c + 4 * g < dh)      // dh is a multiple of 4: the whole float4 is inside the head
                *reinterpret_cast<float4*>(orow + 16 * c + 4 * g) = make_float4(o[c][0] * sc, o[c][1] * sc, o[c][2] * sc, o[c][3] * sc);
    }
}

static size_t sf_attention_mfma_lds(int T, int dhp, bool pos) {
    const int n_kt = (T + 15) / 16, n_rt = (T + 30) / 16;
    const size_t g_floats = std::max<size_t>(pos ? 16 * (size_t)(n_rt * 16 + 4) : 0, 3 * (size_t)(dhp / 16) * 4 * 64);
    return (2 * 16 * (size_t)(dhp + 4) + 16 + 16 * (size_t)(n_kt * 16 + 4) + g_floats) * sizeof(float);
}

void launch_sf_attention(const LaunchCtx& ctx, const SfAttnArgs& a) {
    if (a.T <= 0) return;
    if (a.T > kSfMaxFrames) throw std::invalid_argument("sortformer attention: sequence longer than the LDS score buffer");
    if (a.dh % 4 != 0 || a.dh > 64 || a.dh < 4) throw std::invalid_argument("sortformer attention: unsupported head width");
    if (a.n_seg < 0 || a.n_seg > kSfMaxSegments) throw std::invalid_argument("sortformer attention: too many segments");
    double tsq = 0.0, tsum = 0.0;
    if (a.n_seg > 0) {
        for (int s = 0; s < a.n_seg; ++s) {
            if (a.seg_T[s] < 1 || a.seg_T[s] > a.T) throw std::invalid_argument("sortformer attention: bad segment length");
            tsq += (double)a.seg_T[s] * a.seg_T[s];
            tsum += a.seg_T[s];
        }
    } else {
        tsq = (double)a.T * a.T;
        tsum = a.T;
    }
    const double per = (a.pos ? 6.0 : 4.0) * tsq * a.dh * a.n_head;
    KernelScope ks(ctx, a.pos ? "sf_relpos_attention" : "sf_attention", per, 16.0 * tsum * a.n_head * a.dh);
    // WLK_SF_ATTN=valu: the round-1 kernel (one wave per query row) - A/B switch; it does not take stacked sessions
    static const bool valu = [] { const char* e = getenv("WLK_SF_ATTN"); return e && e[0] == 'v'; }();
    if (valu && a.n_seg == 0) {
        const dim3 grid((a.T + 3) / 4, a.n_head);
        if (a.dh <= 32) hipLaunchKernelGGL((sf_attention_kernel<32>), grid, dim3(256), 0, ctx.stream, a);
        else hipLaunchKernelGGL((sf_attention_kernel<64>), grid, dim3(256), 0, ctx.stream, a);
        WLK_HIP(hipGetLastError());
        return;
    }
    const int dhp = a.dh <= 32 ? 32 : 64;
    const size_t lds = sf_attention_mfma_lds(a.T, dhp, a.pos != nullptr);
    static std::atomic<uint64_t> configured{0};     // per device: more than 64 KiB of dynamic LDS has to be asked for
    int dev = 0;
    WLK_HIP(hipGetDevice(&dev));
    if (!(configured.load(std::memory_order_acquire) >> (dev & 63) & 1)) {
        const int cap = (int)sf_attention_mfma_lds(kSfMaxFrames, 64, true);
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_attention_mfma_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_attention_mfma_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        configured.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const dim3 grid((a.T + 15) / 16, a.n_head, std::max(a.n_seg, 1));
    if (dhp == 32) hipLaunchKernelGGL((sf_attention_mfma_kernel<32>), grid, dim3(256), lds, ctx.stream, a);
    else hipLaunchKernelGGL((sf_attention_mfma_kernel<64>), grid, dim3(256), lds, ctx.stream, a);
    WLK_HIP(hipGetLastError());
}

// ---- ConformerConvolution core: GLU -> depthwise Conv1d(k, same padding) -> BatchNorm1d(eval) -> Swish ----------
__global__ __launch_bounds__(256) void sf_glu_dwconv_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                            const float* __restrict__ b, const float* __restrict__ bn_mean,
                                                            const float* __restrict__ bn_invstd,
                                                            const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                            float* __restrict__ out, SfSegments rows, int d, int taps) {
    const int r = blockIdx.x, c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= d) return;
    const int s = seg_find(rows.start, rows.n, r);
    const int lo = seg_pick(rows.start, s), T = seg_pick(rows.len, s), t = r - lo;      // the taps stay inside the session
    const float* src = in + (long)lo * 2 * d;
    const int half = (taps - 1) / 2;
    float acc = 0.f;
    for (int k = 0; k < taps; ++k) {
        const int tt = t + k - half;
        if (tt < 0 || tt >= T) continue;
        const float x = src[(long)tt * 2 * d + c], gate = src[(long)tt * 2 * d + d + c];
        const float glu = x * (1.0f / (1.0f + expf(-gate)));
        acc = fmaf(glu, w[k * d + c], acc);
    }
    acc += b[c];
    const float y = (acc - bn_mean[c]) * bn_invstd[c] * bn_w[c] + bn_b[c];
    out[(long)r * d + c] = y / (1.0f + expf(-y));
}

void launch_sf_glu_dwconv(const LaunchCtx& ctx, const float* in, const float* w, const float* b, const float* bn_mean,
                          const float* bn_invstd, const float* bn_w, const float* bn_b, float* out, const SfSegments& rows, int d,
                          int taps) {
    const int R = rows.n > 0 ? rows.start[rows.n - 1] + rows.len[rows.n - 1] : 0;
    if (R <= 0) return;
    KernelScope ks(ctx, "sf_glu_dwconv", 2.0 * taps * R * d, 12.0 * R * d);
    hipLaunchKernelGGL(sf_glu_dwconv_kernel, dim3(R, (d + 255) / 256), dim3(256), 0, ctx.stream, in, w, b, bn_mean,
                       bn_invstd, bn_w, bn_b, out, rows, d, taps);
    WLK_HIP(hipGetLastError());
}

// ---- SortformerModules.forward_speaker_sigmoids (eval): relu -> Linear+relu -> Linear -> sigmoid ------------------
// One workgroup per frame, one THREAD per hidden unit: w1t is the first Linear's weight transposed at finalize ([k][n]), so the
// d threads read one coalesced row per k and the frame's relu(x) values come out of LDS as broadcasts - d fma per thread, no
// cross-lane reduction (round 1 gave every hidden unit a wave and folded it with shuffles: 68 us per launch at 293 frames;
// this form: the launch floor).  The n_spk outputs: one wave each, lane-strided products folded by shuffles.
__global__ __launch_bounds__(256) void sf_head_kernel(const float* __restrict__ x, const float* __restrict__ w1t,
                                                      const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, float* __restrict__ out, int d,
                                                      int n_spk) {
    extern __shared__ float sm[];   // [d] relu(x) | [d] hidden
    float* xs = sm;
    float* hs = sm + d;
    const int t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < d; c += blockDim.x) xs[c] = fmaxf(x[(long)t * d + c], 0.f);
    __syncthreads();
    for (int n = threadIdx.x; n < d; n += blockDim.x) {
        float acc = 0.f;
        for (int k = 0; k < d; ++k) acc = fmaf(xs[k], w1t[(long)k * d + n], acc);
        hs[n] = fmaxf(acc + b1[n], 0.f);
    }
    __syncthreads();
    for (int n = wave; n < n_spk; n += 4) {
        float acc = 0.f;
        for (int c = lane; c < d; c += 64) acc = fmaf(hs[c], w2[(long)n * d + c], acc);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) out[(long)t * n_spk + n] = 1.0f / (1.0f + expf(-(acc + b2[n])));
    }
}

void launch_sf_head(const LaunchCtx& ctx, const float* x, const float* w1t, const float* b1, const float* w2,
                    const float* b2, float* out, int T, int d, int n_spk) {
    if (T <= 0) return;
    KernelScope ks(ctx, "sf_head", 2.0 * T * d * (d + n_spk), 4.0 * T * (d + n_spk));
    hipLaunchKernelGGL(sf_head_kernel, dim3(T), dim3(256), 2 * d * sizeof(float), ctx.stream, x, w1t, b1, w2, b2, out, d,
                       n_spk);
    WLK_HIP(hipGetLastError());
}

// [rows][cols] -> [cols][rows] (finalize-time weight transposes)
__global__ void sf_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    dst[(long)c * rows + r] = src[i];
}
void launch_sf_transpose(const LaunchCtx& ctx, const float* src, float* dst, int rows, int cols) {
    const long n = (long)rows * cols;
    if (n <= 0) return;
    hipLaunchKernelGGL(sf_transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx.stream, src, dst, rows, cols);
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

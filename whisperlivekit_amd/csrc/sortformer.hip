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
#include "common.h"

namespace wlk {

// ---- ConvSubsampling('dw_striding') ----------------------------------------------------------------
// conv0: Conv2d(1, C, 3, stride 2, padding 1) + ReLU.  One workgroup per output position, one thread per
// channel: the 9 input taps are wave-uniform, the output is channels-last so that the following pointwise
// convolutions are GEMMs over [positions, C].
__global__ __launch_bounds__(256) void sf_conv0_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ out, int T,
                                                       int F, int F1, int C) {
    const int pos = blockIdx.x, t1 = pos / F1, f1 = pos - t1 * F1;
    float x[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int tt = 2 * t1 - 1 + ky, ff = 2 * f1 - 1 + kx;
            x[ky * 3 + kx] = (tt >= 0 && tt < T && ff >= 0 && ff < F) ? in[(long)tt * F + ff] : 0.f;
        }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc = fmaf(x[k], w[c * 9 + k], acc);
        out[(long)pos * C + c] = fmaxf(acc + b[c], 0.f);
    }
}

void launch_sf_conv0(const LaunchCtx& ctx, const float* in, const float* w, const float* b, float* out, int T, int F,
                     int C) {
    const int T1 = sf_sub_len(T), F1 = sf_sub_len(F);
    KernelScope ks(ctx, "sf_conv0", 18.0 * T1 * F1 * C, 4.0 * ((double)T * F + (double)T1 * F1 * C));
    hipLaunchKernelGGL(sf_conv0_kernel, dim3(T1 * F1), dim3(C < 256 ? C : 256), 0, ctx.stream, in, w, b, out, T, F, F1, C);
    WLK_HIP(hipGetLastError());
}

// depthwise Conv2d(C, C, 3, stride 2, padding 1, groups C) on channels-last data; weights tap-major [9][C]
__global__ __launch_bounds__(256) void sf_dwconv2d_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ out, int Ti,
                                                          int Fi, int Fo, int C) {
    const int pos = blockIdx.x, to = pos / Fo, fo = pos - to * Fo;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tt = 2 * to - 1 + ky, ff = 2 * fo - 1 + kx;
                const bool ok = tt >= 0 && tt < Ti && ff >= 0 && ff < Fi;
                const float x = ok ? in[((long)tt * Fi + ff) * C + c] : 0.f;
                acc = fmaf(x, w[(ky * 3 + kx) * C + c], acc);
            }
        out[(long)pos * C + c] = acc + b[c];
    }
}

void launch_sf_dwconv2d(const LaunchCtx& ctx, const float* in, const float* w, const float* b, float* out, int Ti,
                        int Fi, int C) {
    const int To = sf_sub_len(Ti), Fo = sf_sub_len(Fi);
    KernelScope ks(ctx, "sf_dwconv2d", 18.0 * To * Fo * C, 4.0 * ((double)Ti * Fi * C + (double)To * Fo * C));
    hipLaunchKernelGGL(sf_dwconv2d_kernel, dim3(To * Fo), dim3(C < 256 ? C : 256), 0, ctx.stream, in, w, b, out, Ti, Fi,
                       Fo, C);
    WLK_HIP(hipGetLastError());
}

__global__ void sf_scale_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long n, float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = src[i] * scale;
}

void launch_sf_scale_copy(const LaunchCtx& ctx, const float* src, float* dst, long n, float scale) {
    if (n <= 0) return;
    KernelScope ks(ctx, "sf_scale_copy", 0.0, 8.0 * n);
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(sf_scale_copy_kernel, dim3(blocks), dim3(256), 0, ctx.stream, src, dst, n, scale);
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

void launch_sf_attention(const LaunchCtx& ctx, const SfAttnArgs& a) {
    if (a.T <= 0) return;
    if (a.T > kSfMaxFrames) throw std::invalid_argument("sortformer attention: sequence longer than the LDS score buffer");
    if (a.dh % 4 != 0 || a.dh > 64 || a.dh < 4) throw std::invalid_argument("sortformer attention: unsupported head width");
    const double per = (a.pos ? 6.0 : 4.0) * a.T * a.T * a.dh * a.n_head;
    KernelScope ks(ctx, a.pos ? "sf_relpos_attention" : "sf_attention", per, 16.0 * a.T * a.n_head * a.dh);
    const dim3 grid((a.T + 3) / 4, a.n_head);
    if (a.dh <= 32) hipLaunchKernelGGL((sf_attention_kernel<32>), grid, dim3(256), 0, ctx.stream, a);
    else hipLaunchKernelGGL((sf_attention_kernel<64>), grid, dim3(256), 0, ctx.stream, a);
    WLK_HIP(hipGetLastError());
}

// ---- ConformerConvolution core: GLU -> depthwise Conv1d(k, same padding) -> BatchNorm1d(eval) -> Swish ----------
__global__ __launch_bounds__(256) void sf_glu_dwconv_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                            const float* __restrict__ b, const float* __restrict__ bn_mean,
                                                            const float* __restrict__ bn_invstd,
                                                            const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                            float* __restrict__ out, int T, int d, int taps) {
    const int t = blockIdx.x, c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= d) return;
    const int half = (taps - 1) / 2;
    float acc = 0.f;
    for (int k = 0; k < taps; ++k) {
        const int tt = t + k - half;
        if (tt < 0 || tt >= T) continue;
        const float x = in[(long)tt * 2 * d + c], gate = in[(long)tt * 2 * d + d + c];
        const float glu = x * (1.0f / (1.0f + expf(-gate)));
        acc = fmaf(glu, w[k * d + c], acc);
    }
    acc += b[c];
    const float y = (acc - bn_mean[c]) * bn_invstd[c] * bn_w[c] + bn_b[c];
    out[(long)t * d + c] = y / (1.0f + expf(-y));
}

void launch_sf_glu_dwconv(const LaunchCtx& ctx, const float* in, const float* w, const float* b, const float* bn_mean,
                          const float* bn_invstd, const float* bn_w, const float* bn_b, float* out, int T, int d,
                          int taps) {
    if (T <= 0) return;
    KernelScope ks(ctx, "sf_glu_dwconv", 2.0 * taps * T * d, 12.0 * T * d);
    hipLaunchKernelGGL(sf_glu_dwconv_kernel, dim3(T, (d + 255) / 256), dim3(256), 0, ctx.stream, in, w, b, bn_mean,
                       bn_invstd, bn_w, bn_b, out, T, d, taps);
    WLK_HIP(hipGetLastError());
}

// ---- SortformerModules.forward_speaker_sigmoids (eval): relu -> Linear+relu -> Linear -> sigmoid ------------------
// one workgroup per frame; the two small matrices stay in L2
__global__ __launch_bounds__(256) void sf_head_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, float* __restrict__ out, int d,
                                                      int n_spk) {
    extern __shared__ float sm[];   // [d] relu(x) | [d] hidden
    float* xs = sm;
    float* hs = sm + d;
    const int t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < d; c += blockDim.x) xs[c] = fmaxf(x[(long)t * d + c], 0.f);
    __syncthreads();
    for (int n = wave; n < d; n += 4) {            // a wave per hidden unit: coalesced weight rows
        float acc = 0.f;
        for (int c = lane; c < d; c += 64) acc = fmaf(xs[c], w1[(long)n * d + c], acc);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) hs[n] = fmaxf(acc + b1[n], 0.f);
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

void launch_sf_head(const LaunchCtx& ctx, const float* x, const float* w1, const float* b1, const float* w2,
                    const float* b2, float* out, int T, int d, int n_spk) {
    if (T <= 0) return;
    KernelScope ks(ctx, "sf_head", 2.0 * T * d * (d + n_spk), 4.0 * T * (d + n_spk));
    hipLaunchKernelGGL(sf_head_kernel, dim3(T), dim3(256), 2 * d * sizeof(float), ctx.stream, x, w1, b1, w2, b2, out, d,
                       n_spk);
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

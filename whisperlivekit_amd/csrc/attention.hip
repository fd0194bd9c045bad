// Encoder self-attention (whisper/model.py:148-173 with xa=None, mask=None): the reference
// materialises fp32 QK [H, 1500, 1500] per layer and throws it away (model.py:251); here the
// scores never leave registers.
//
// One 256-thread workgroup = one (head, 32-query tile).  Its four waves split the KEY axis:
// wave w takes key tiles w, w+4, w+8, ... (32 keys each) and keeps its own online-softmax state
// (running max m, running sum l, 64x32 output accumulator); the four partial states are merged
// through LDS at the end.  That gives 4 x 376 = 1504 independent MFMA streams for base.en
// (8 heads x 47 query tiles) - enough to cover the 1024 SIMDs of the chip, which 32-query tiles
// alone (376 waves) would not.
//
// Matrix core usage (v_mfma_f32_32x32x2_f32, exact fp32):
//   S^T[key][q]  = sum_d K[key][d] Q[q][d]          A = K tile, B = Q tile (both LDS, row stride 68:
//                                                    conflict-free ds_read_b128 of 4 consecutive d)
//   O^T[d][q]   += sum_key V[key][d] P^T[key][q]     A = V tile (LDS), B = P^T = the S^T accumulator
//                                                    registers themselves (C layout of step 1 is the
//                                                    B layout of step 2 up to a key permutation that the
//                                                    V fetch follows) - no shuffle, no LDS round trip.
// With S^T (keys on rows) every lane owns ONE query column, so the softmax row reductions are
// 15 in-lane ops + one cross-half swap, and the alpha rescale is a per-lane scalar.
#include "common.h"

namespace wlk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int QT = 32, KT = 32, NWAVE = 4, K_LD = 68, O_LD = 65;
constexpr int kAttnLdsFloats = NWAVE * KT * K_LD + NWAVE * KT * 64;  // K tiles + V tiles
constexpr int kAttnLdsTotal = kAttnLdsFloats + QT * K_LD;            // + Q tile

__global__ __launch_bounds__(256) void encoder_attention_kernel(const float* __restrict__ qkv,
                                                                float* __restrict__ out, int T, int d,
                                                                int n_head) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ks = lds;                            // [NWAVE][KT][K_LD]
    float* Vs = lds + NWAVE * KT * K_LD;        // [NWAVE][KT][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int head = blockIdx.x % n_head;       // head == XCD for 8 heads: K/V of a head stay in one L2
    const int q0 = (blockIdx.x / n_head) * QT;
    const long ld = 3L * d;
    const int half = lane >> 5;
    const int lq = lane & 31;

    // Q tile -> LDS once ([32][K_LD], same padded layout as the K tiles); the B fragments are
    // re-read per key tile (one ds_read_b128 per 4 MFMAs - the 64-cycle fp32 MFMA leaves the LDS idle)
    float* Qs = lds + kAttnLdsFloats;
    for (int i = tid; i < QT * 16; i += 256) {
        const int qr = i >> 4, c4 = i & 15;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 + qr < T) v = *reinterpret_cast<const float4*>(qkv + (long)(q0 + qr) * ld + head * 64 + c4 * 4);
        *reinterpret_cast<float4*>(&Qs[qr * K_LD + c4 * 4]) = v;
    }
    const float* Qw = Qs + lq * K_LD + half * 4;

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // staging map: per iteration 128 keys x 64 floats for K and for V = 2048 float4 each, 8 per thread
    const float* kbase = qkv + d + head * 64;
    const float* vbase = qkv + 2 * d + head * 64;
    float4 rk[8], rv[8];
    auto fetch = [&](int it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i;
            const int key = it * (NWAVE * KT) + (idx >> 4);
            const int c4 = idx & 15;
            const bool ok = key < T;
            const long off = (long)(ok ? key : 0) * ld + c4 * 4;   // branch-free: keeps rk/rv in registers
            const float4 k4 = *reinterpret_cast<const float4*>(kbase + off);
            const float4 v4 = *reinterpret_cast<const float4*>(vbase + off);
            rk[i] = ok ? k4 : make_float4(0.f, 0.f, 0.f, 0.f);
            rv[i] = ok ? v4 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i;
            const int kl = idx >> 4;  // 0..127: tile = kl >> 5, row = kl & 31
            const int c4 = idx & 15;
            *reinterpret_cast<float4*>(&Ks[(kl >> 5) * (KT * K_LD) + (kl & 31) * K_LD + c4 * 4]) = rk[i];
            *reinterpret_cast<float4*>(&Vs[kl * 64 + c4 * 4]) = rv[i];
        }
    };

    const int n_iter = (T + NWAVE * KT - 1) / (NWAVE * KT);
    fetch(0);
    const float* Kw = Ks + wave * (KT * K_LD) + lq * K_LD + half * 4;
    const float* Vw = Vs + wave * (KT * 64) + lq;

    for (int it = 0; it < n_iter; ++it) {
        __syncthreads();  // previous iteration's LDS reads are done
        stash();
        __syncthreads();
        if (it + 1 < n_iter) fetch(it + 1);

        const int key0 = it * (NWAVE * KT) + wave * KT;
        if (key0 < T) {  // wave-uniform
            f32x16 s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 k4 = *reinterpret_cast<const float4*>(Kw + g * 8);
                const float4 q4 = *reinterpret_cast<const float4*>(Qw + g * 8);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.x, q4.x, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.y, q4.y, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.z, q4.z, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.w, q4.w, s, 0, 0, 0);
            }
            // rows of s: key = key0 + (r&3) + 8*(r>>2) + 4*half ; column: query lq
            float mt = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (key >= T) s[r] = -INFINITY;
                mt = fmaxf(mt, s[r]);
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float m_new = fmaxf(m_run, mt);
            const float alpha = expf(m_run - m_new);
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = expf(s[r] - m_new);
                rs += s[r];
            }
            rs += __shfl_xor(rs, 32, 64);
            l_run = l_run * alpha + rs;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = (r & 3) + 8 * (r >> 2) + 4 * half;
                const float v0 = Vw[kk * 64];
                const float v1 = Vw[kk * 64 + 32];
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[r], o1, 0, 0, 0);
            }
        }
    }

    // merge the four key-split partial states through LDS (reusing the staging area)
    __syncthreads();
    float* Os = lds;                               // [NWAVE][QT][O_LD]
    float* Ms = lds + NWAVE * QT * O_LD;           // [NWAVE][QT]
    float* Ls = Ms + NWAVE * QT;                   // [NWAVE][QT]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dd = (r & 3) + 8 * (r >> 2) + 4 * half;
        Os[(wave * QT + lq) * O_LD + dd] = o0[r];
        Os[(wave * QT + lq) * O_LD + 32 + dd] = o1[r];
    }
    if (half == 0) {
        Ms[wave * QT + lq] = m_run;
        Ls[wave * QT + lq] = l_run;
    }
    __syncthreads();
    {
        const int dd = tid & 63;
        const int qg = tid >> 6;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = qg * 8 + i;
            const int qrow = q0 + q;
            float M = Ms[q];
#pragma unroll
            for (int w = 1; w < NWAVE; ++w) M = fmaxf(M, Ms[w * QT + q]);
            float L = 0.f, acc = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) {
                const float e = expf(Ms[w * QT + q] - M);
                L += e * Ls[w * QT + q];
                acc += e * Os[(w * QT + q) * O_LD + dd];
            }
            if (qrow < T) out[(long)qrow * d + head * 64 + dd] = acc / L;
        }
    }
}

void launch_encoder_attention(const LaunchCtx& ctx, const float* qkv, float* out, int T, int d, int n_head) {
    static bool attr_set = false;
    const size_t lds = kAttnLdsTotal * sizeof(float);
    if (!attr_set) {
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(encoder_attention_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int q_tiles = (T + QT - 1) / QT;
    // QK^T and PV: 2 * T*T*64 MACs per head each; reads q,k,v once, writes out
    KernelScope ks(ctx, "enc_attention", 4.0 * T * (double)T * d, 4.0 * 4.0 * T * d);
    hipLaunchKernelGGL(encoder_attention_kernel, dim3(q_tiles * n_head), dim3(256), lds, ctx.stream, qkv, out, T,
                       d, n_head);
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

// Where does the fp32 flash attention of the encoder lose its cycles?  The loop of flash_attention_kernel
// (whisperlivekit_amd/csrc/attention.hip; encoder form: no score dump, no key split) with pieces switched off:
//   bit 0: no global fetch / LDS stash (LDS keeps its initial contents)      bit 1: no barriers
//   bit 2: no QK^T MFMAs        bit 3: no softmax arithmetic (scores used as they are)      bit 4: no PV MFMAs
// Prints microseconds per launch for T = 1500, 8 heads (376 workgroups) and 20 heads (940 workgroups).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/probes/attn_ablation.hip -o scripts/probes/attn_ablation
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int QT = 32, KT = 32, NWAVE = 4, K_LD = 68, O_LD = 65;
constexpr int kLdsFloats = NWAVE * KT * K_LD + NWAVE * KT * 64;
constexpr int kLdsTotal = kLdsFloats + QT * K_LD;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ABL>
__global__ __launch_bounds__(256) void attn(const float* q, const float* k, const float* v, float* out, int T, int n_head,
                                            long ld, long ldo) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ks = lds;
    float* Vs = lds + NWAVE * KT * K_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = blockIdx.x % n_head, q0 = (blockIdx.x / n_head) * QT;
    const int half = lane >> 5, lq = lane & 31;
    float* Qs = lds + kLdsFloats;
    for (int i = tid; i < QT * 16; i += 256) {
        const int qr = i >> 4, c4 = i & 15;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 + qr < T) t = *reinterpret_cast<const float4*>(q + (long)(q0 + qr) * ld + head * 64 + c4 * 4);
        *reinterpret_cast<float4*>(&Qs[qr * K_LD + c4 * 4]) = t;
    }
    if (ABL & 1) for (int i = tid; i < kLdsFloats; i += 256) lds[i] = 0.001f * (i & 255);
    const float* Qw = Qs + lq * K_LD + half * 4;
    f32x16 o0, o1;
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    const float* kbase = k + head * 64;
    const float* vbase = v + head * 64;
    float4 rk[8], rv[8];
    auto fetch = [&](int it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i, key = it * 128 + (idx >> 4), c4 = idx & 15;
            const bool ok = key < T;
            const long off = (long)(ok ? key : 0) * ld + c4 * 4;
            const float4 k4 = *reinterpret_cast<const float4*>(kbase + off);
            const float4 v4 = *reinterpret_cast<const float4*>(vbase + off);
            rk[i] = ok ? k4 : make_float4(0.f, 0.f, 0.f, 0.f);
            rv[i] = ok ? v4 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i, kl = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<float4*>(&Ks[(kl >> 5) * (KT * K_LD) + (kl & 31) * K_LD + c4 * 4]) = rk[i];
            *reinterpret_cast<float4*>(&Vs[kl * 64 + c4 * 4]) = rv[i];
        }
    };
    const int n_iter = (T + 127) / 128;
    if (!(ABL & 1)) fetch(0);
    const float* Kw = Ks + wave * (KT * K_LD) + lq * K_LD + half * 4;
    const float* Vw = Vs + wave * (KT * 64) + lq;
    for (int it = 0; it < n_iter; ++it) {
        if (!(ABL & 2)) __syncthreads();
        if (!(ABL & 1)) stash();
        if (!(ABL & 2)) __syncthreads();
        if (!(ABL & 1) && it + 1 < n_iter) fetch(it + 1);
        const int key0 = it * 128 + wave * KT;
        if (key0 < T) {
            f32x16 s;
            for (int i = 0; i < 16; ++i) s[i] = 0.01f * i;
            if (!(ABL & 4)) {
                float4 k4 = *reinterpret_cast<const float4*>(Kw);
                float4 q4 = *reinterpret_cast<const float4*>(Qw);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.x, q4.x, s, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    float4 kn = k4, qn = q4;
                    if (g + 1 < 8) {
                        kn = *reinterpret_cast<const float4*>(Kw + (g + 1) * 8);
                        qn = *reinterpret_cast<const float4*>(Qw + (g + 1) * 8);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.y, q4.y, s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.z, q4.z, s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.w, q4.w, s, 0, 0, 0);
                    k4 = kn;
                    q4 = qn;
                }
            }
            if (!(ABL & 8)) {
                float mt = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (key >= T) s[r] = -INFINITY;
                    mt = fmaxf(mt, s[r]);
                }
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
                const float m_new = fmaxf(m_run, mt);
                const float alpha = __expf(m_run - m_new);
                float rs = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - m_new); rs += s[r]; }
                rs += __shfl_xor(rs, 32, 64);
                l_run = l_run * alpha + rs;
                m_run = m_new;
#pragma unroll
                for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
            }
            if (!(ABL & 16)) {
                float v0 = Vw[(4 * half) * 64], v1 = Vw[(4 * half) * 64 + 32];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v0n = v0, v1n = v1;
                    if (r + 1 < 16) {
                        const int kn = ((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * half;
                        v0n = Vw[kn * 64];
                        v1n = Vw[kn * 64 + 32];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[r], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[r], o1, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    v0 = v0n;
                    v1 = v1n;
                }
            } else {
                for (int i = 0; i < 16; ++i) o0[i] += s[i];
            }
        }
    }
    __syncthreads();
    float* Os = lds;
    float* Ms = lds + NWAVE * QT * O_LD;
    float* Ls = Ms + NWAVE * QT;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dd = (r & 3) + 8 * (r >> 2) + 4 * half;
        Os[(wave * QT + lq) * O_LD + dd] = o0[r];
        Os[(wave * QT + lq) * O_LD + 32 + dd] = o1[r];
    }
    if (half == 0) { Ms[wave * QT + lq] = m_run; Ls[wave * QT + lq] = l_run; }
    __syncthreads();
    const int dd = tid & 63, qg = tid >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int qq = qg * 8 + i, qrow = q0 + qq;
        float M = Ms[qq];
        for (int w = 1; w < NWAVE; ++w) M = fmaxf(M, Ms[w * QT + qq]);
        float L = 0.f, acc = 0.f;
        for (int w = 0; w < NWAVE; ++w) {
            const float e = expf(Ms[w * QT + qq] - M);
            L += e * Ls[w * QT + qq];
            acc += e * Os[(w * QT + qq) * O_LD + dd];
        }
        if (qrow < T) out[(long)qrow * ldo + head * 64 + dd] = acc / L;
    }
}

template <int ABL>
static int run(const char* what, const float* qkv, float* out, int T, int H, hipStream_t st) {
    const int d = 64 * H;
    const size_t lds = kLdsTotal * sizeof(float);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = ((T + QT - 1) / QT) * H;
    auto go = [&]() { hipLaunchKernelGGL(attn<ABL>, dim3(grid), dim3(256), lds, st, qkv, qkv + d, qkv + 2 * d, out, T, H, 3L * d, (long)d); };
    go();
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 20; ++i) go();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("H=%2d abl=%2d %-46s %8.2f us\n", H, ABL, what, 1e3f * ms / 20);
    return 0;
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int H : {8, 20}) {
        const int T = 1500, d = 64 * H;
        std::vector<float> h((size_t)T * 3 * d);
        unsigned seed = 1;
        for (auto& x : h) { seed = seed * 1664525u + 1013904223u; x = (((seed >> 8) & 0xffff) / 65536.0f - 0.5f) * 1.5f; }
        float *qkv, *out;
        CK(hipMalloc(&qkv, h.size() * 4));
        CK(hipMalloc(&out, (size_t)T * d * 4));
        CK(hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        run<0>("full kernel", qkv, out, T, H, st);
        run<1>("no global fetch / stash", qkv, out, T, H, st);
        run<3>("no fetch / stash / barriers", qkv, out, T, H, st);
        run<8>("no softmax arithmetic", qkv, out, T, H, st);
        run<11>("no fetch / stash / barriers / softmax", qkv, out, T, H, st);
        run<4>("no QK MFMAs", qkv, out, T, H, st);
        run<16>("no PV MFMAs", qkv, out, T, H, st);
        run<15>("PV MFMAs only", qkv, out, T, H, st);
        run<27>("QK MFMAs only", qkv, out, T, H, st);
        run<31>("nothing (prologue + epilogue)", qkv, out, T, H, st);
        (void)hipFree(qkv);
        (void)hipFree(out);
    }
    return 0;
}

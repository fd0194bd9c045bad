// Ablation of the 64-query encoder attention kernel (generated from whisperlivekit_amd/csrc/attention.hip by
// scripts/probes/gen_attn_q64_ablation.py): bit 0 no fetch/stash, bit 1 no barriers, bit 2 no QK^T MFMAs, bit 3 no
// softmax arithmetic, bit 4 no PV MFMAs.  Times the main kernel only (no merge kernel), k_splits = 4, T = 1500, 8 heads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int QT = 32, KT = 32, NWAVE = 4, K_LD = 68, O_LD = 65, kMaxBatch = 8;
struct PtrTable { const float* in[kMaxBatch]; float* out[kMaxBatch]; const float* res[kMaxBatch]; };
template <typename T> __device__ __forceinline__ T table_at(T const (&t)[kMaxBatch], unsigned z) { return t[0]; }
struct FlashArgs {
    const float* q = nullptr; long ldq = 0; const float* k = nullptr; const float* v = nullptr; long ldkv = 0; long kv_hs = 64;
    float* out = nullptr; long ldo = 0; int Tq = 0, Tk = 0, n_head = 0; int batch = 0; long z_k_off = 0, z_v_off = 0; PtrTable z;
    int k_splits = 1; float* part_o = nullptr; float* part_m = nullptr; float* part_l = nullptr;
};
constexpr int QT2 = 64;
constexpr int kAttn2Stage = 2 * KT * K_LD + 2 * KT * 64;
constexpr int kAttn2Merge = NWAVE * QT * O_LD + 2 * NWAVE * QT;
constexpr int kAttn2LdsFloats = kAttn2Stage > kAttn2Merge ? kAttn2Stage : kAttn2Merge;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int ABL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void q64(FlashArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const bool batched = a.batch > 0;
    const float* const aq = batched ? table_at(a.z.in, blockIdx.y) : a.q;
    const float* const ak = batched ? aq + a.z_k_off : a.k;
    const float* const av = batched ? aq + a.z_v_off : a.v;
    float* const aout = batched ? table_at(a.z.out, blockIdx.y) : a.out;
    float* const part = batched ? const_cast<float*>(table_at(a.z.res, blockIdx.y)) : a.part_o;
    const int T = a.Tk, n_head = a.n_head;
    float* Ks = lds;                          // [2][KT][K_LD]
    float* Vs = lds + 2 * KT * K_LD;          // [2][KT][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qs = wave & 1, kt = wave >> 1;
    const int head = blockIdx.x % n_head;
    const int q_tiles = (a.Tq + QT2 - 1) / QT2;
    const int qt_idx = (blockIdx.x / n_head) % q_tiles;
    const int ks = blockIdx.x / (n_head * q_tiles);
    const int q0 = qt_idx * QT2 + qs * QT;    // first query of this wave's sub-tile
    const long ld = a.ldkv;
    const int half = lane >> 5, lq = lane & 31;

    float4 qf[8];                             // Q[q0 + lq][8 g + 4 half .. +4]: B operand of QK^T group g
    {
        const int qr = min(q0 + lq, a.Tq - 1);
        const float* qp = aq + (long)qr * a.ldq + head * 64 + half * 4;
        const bool ok = q0 + lq < a.Tq;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(qp + g * 8);
            qf[g] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    const float* kbase = ak + head * a.kv_hs;
    const float* vbase = av + head * a.kv_hs;
    // staging map: a pair = 64 keys x 64 floats for K and for V = 1024 float4 each, 4 + 4 per thread
    float4 rk[4], rv[4];
    auto fetch = [&](int pair) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int key = pair * (2 * KT) + (idx >> 4);
            const int c4 = idx & 15;
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
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int kl = idx >> 4;          // 0..63: tile = kl >> 5, row = kl & 31
            const int c4 = idx & 15;
            *reinterpret_cast<float4*>(&Ks[(kl >> 5) * (KT * K_LD) + (kl & 31) * K_LD + c4 * 4]) = rk[i];
            *reinterpret_cast<float4*>(&Vs[kl * 64 + c4 * 4]) = rv[i];
        }
    };
    const int n_pairs = (T + 2 * KT - 1) / (2 * KT);
    const float* Kw = Ks + kt * (KT * K_LD) + lq * K_LD + half * 4;
    const float* Vw = Vs + kt * (KT * 64) + lq;
    if (ABL & 1) for (int i = tid; i < kAttn2Stage; i += 256) lds[i] = 0.001f * (i & 255);
    if (!(ABL & 1) && ks < n_pairs) fetch(ks);
    for (int pair = ks; pair < n_pairs; pair += a.k_splits) {
        if (!(ABL & 2)) __syncthreads();
        if (!(ABL & 1)) stash();
        if (!(ABL & 2)) __syncthreads();
        if (!(ABL & 1) && pair + a.k_splits < n_pairs) fetch(pair + a.k_splits);
        const int key0 = pair * (2 * KT) + kt * KT;
        if (key0 < T) {                       // wave-uniform
            f32x16 s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.01f * i;
            if (!(ABL & 4)) {
            float4 k4 = *reinterpret_cast<const float4*>(Kw);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.x, qf[g].x, s, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                float4 kn = k4;
                if (g + 1 < 8) kn = *reinterpret_cast<const float4*>(Kw + (g + 1) * 8);
                __builtin_amdgcn_sched_barrier(0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.y, qf[g].y, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.z, qf[g].z, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.w, qf[g].w, s, 0, 0, 0);
                k4 = kn;
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
            for (int r = 0; r < 16; ++r) {
                s[r] = __expf(s[r] - m_new);
                rs += s[r];
            }
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
            } else { for (int i = 0; i < 16; ++i) o0[i] += s[i]; }
        }
    }

    // merge the two key streams of each query sub-tile through LDS (overlaying the staging area)
    __syncthreads();
    float* Os = lds;                               // [NWAVE][QT][O_LD]
    float* Ms = lds + NWAVE * QT * O_LD;           // [NWAVE][QT]
    float* Ls = Ms + NWAVE * QT;
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
        const int qg = tid >> 6;                   // 16 queries per thread group
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int q = qg * 16 + i;             // 0..63 within the workgroup's tile
            const int sub = q >> 5, ql = q & 31;
            const int w0 = sub, w1 = sub + 2;      // the two key streams of this sub-tile
            const int qrow = qt_idx * QT2 + q;
            const float m0 = Ms[w0 * QT + ql], m1 = Ms[w1 * QT + ql];
            const float M = fmaxf(m0, m1);
            // a stream that saw no key keeps m = -inf, l = 0, o = 0 and must contribute nothing (exp(-inf - M) = 0,
            // but -inf - (-inf) would be NaN when BOTH streams are empty: then the row is outside this split's keys)
            const float e0 = m0 == -INFINITY ? 0.f : expf(m0 - M), e1 = m1 == -INFINITY ? 0.f : expf(m1 - M);
            const float L = e0 * Ls[w0 * QT + ql] + e1 * Ls[w1 * QT + ql];
            const float acc = e0 * Os[(w0 * QT + ql) * O_LD + dd] + e1 * Os[(w1 * QT + ql) * O_LD + dd];
            if (qrow < a.Tq) {
                if (a.k_splits == 1) {
                    aout[(long)qrow * a.ldo + head * 64 + dd] = acc / L;
                } else {
                    const long slot = ((long)qrow * n_head + head) * a.k_splits + ks;
                    float* pm = part + (size_t)a.Tq * n_head * a.k_splits * 64;
                    float* pl = pm + (size_t)a.Tq * n_head * a.k_splits;
                    part[slot * 64 + dd] = acc;
                    if (dd == 0) { pm[slot] = M; pl[slot] = L; }
                }
            }
        }
    }
}


template <int ABL>
static int run(const char* what, FlashArgs a, hipStream_t st) {
    const size_t lds = kAttn2LdsFloats * sizeof(float);
    const int grid = ((a.Tq + QT2 - 1) / QT2) * a.n_head * a.k_splits;
    auto go = [&]() { hipLaunchKernelGGL(q64<ABL>, dim3(grid), dim3(256), lds, st, a); };
    go();
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 20; ++i) go();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("ks=%d abl=%2d %-46s %8.2f us\n", a.k_splits, ABL, what, 1e3f * ms / 20);
    return 0;
}
int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int T = 1500, H = 8, d = 64 * H;
    std::vector<float> h((size_t)T * 3 * d);
    unsigned seed = 1;
    for (auto& x : h) { seed = seed * 1664525u + 1013904223u; x = (((seed >> 8) & 0xffff) / 65536.0f - 0.5f) * 1.5f; }
    float *qkv, *out, *part;
    CK(hipMalloc(&qkv, h.size() * 4));
    CK(hipMalloc(&out, (size_t)T * d * 4));
    CK(hipMalloc(&part, (size_t)T * H * 8 * 66 * 4));
    CK(hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int ks : {4, 1}) {
        FlashArgs a;
        a.q = qkv; a.ldq = 3L * d; a.k = qkv + d; a.v = qkv + 2 * d; a.ldkv = 3L * d; a.out = out; a.ldo = d;
        a.Tq = T; a.Tk = T; a.n_head = H; a.k_splits = ks; a.part_o = part;
        run<0>("full kernel", a, st);
        run<1>("no global fetch / stash", a, st);
        run<3>("no fetch / stash / barriers", a, st);
        run<8>("no softmax arithmetic", a, st);
        run<11>("no fetch / stash / barriers / softmax", a, st);
        run<15>("PV MFMAs only", a, st);
        run<27>("QK MFMAs only", a, st);
        run<31>("nothing (prologue + epilogue)", a, st);
    }
    return 0;
}

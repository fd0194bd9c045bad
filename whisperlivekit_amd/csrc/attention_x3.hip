// Encoder self-attention on the bf16 matrix cores at fp32 accuracy (x3.h): softmax(Q K^T) V per 64-wide head with Q, K and
// V as three bf16 planes each, six bf16 MFMAs per fp32 product, fp32 accumulation, fp32 softmax.  gfx950 only.
// Replaces whisper/model.py:148-173 (qkv_attention) for the encoder, as enc_attention_pw_kernel (attention.hip) does
// with fp32 MFMAs: 4.6 GFLOP per launch on base.en took 72 us there (0.40 of the fp32 matrix peak, 31 % of an encode).
//
// Operands come straight from the qkv projection's epilogue (gemm_x3.hip, X3GemmArgs::x3_out):
//   Q | K : X3 rows [T][2 d] (chunks run along the feature axis) - a lane's A / B fragment is one 16-byte unit;
//   V^T   : X3 rows [d][T padded to 16] - chunks run along the KEY axis, keys of every 16-group stored 0-3, 8-11, 4-7,
//           12-15: exactly the keys whose probabilities a lane holds after S^T = K Q^T, so the accumulator registers of
//           step 1 are (after the split into planes) the B operand of step 2 without any data movement between lanes.
// Workgroup = (head, 64 queries); eight waves: compute wave (qb, ks) owns queries 32 qb .. 32 qb + 31 and the key tiles
// t = ks (mod 2) (flash-style online softmax per wave, the two key streams of a query block merged at the end), four
// loader waves feed a ring of three slots of two 32-key tiles each (K tile 12 KiB + V^T tile 12 KiB per stream) by
// LDS-DMA - the wide GEMM's schedule: one workgroup barrier per pair of tiles.  Per tile and wave: 24 MFMAs for S^T,
// 24 for O^T += V^T P^T, a 16-value softmax per lane in between.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "wave_ops.h"
#include "x3.h"

namespace wlk {

namespace {
typedef float af32x16 __attribute__((ext_vector_type(16)));
typedef float af32x4 __attribute__((ext_vector_type(4)));

constexpr int AX_QT = 64, AX_KT = 32;
constexpr int AX_K_ROW = 384;                              // K tile row: 8 chunks x 3 planes x 16 bytes
constexpr int AX_V_ROW = 192;                              // V^T tile row: 4 stored chunks x 3 planes x 16 bytes
constexpr int AX_K_BYTES = AX_KT * AX_K_ROW;               // 12 288
constexpr int AX_V_BYTES = 64 * AX_V_ROW;                  // 12 288
constexpr int AX_TILE_BYTES = AX_K_BYTES + AX_V_BYTES;     // one stream's tile
constexpr int AX_SLOT_BYTES = 2 * AX_TILE_BYTES;           // both streams: 49 152
constexpr int AX_NB = 3, AX_DT = AX_NB - 1;
constexpr int AX_NPW = AX_SLOT_BYTES / 1024 / 4;           // 12 DMA pieces per loader wave and slot
constexpr size_t AX_LDS_BYTES = (size_t)AX_NB * AX_SLOT_BYTES;
constexpr int AX_PA[6] = {2, 0, 1, 1, 0, 0}, AX_PB[6] = {0, 2, 1, 0, 1, 0};   // plane products, small terms first

template <int N, int I = 0, typename F>
__device__ __forceinline__ void ax_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ax_static_for<N, I + 1>(f);
    }
}
__device__ __forceinline__ void ax_read12(af32x4 (&f)[12], const unsigned (&addr)[12], unsigned off) {
#pragma unroll
    for (int t = 0; t < 12; ++t) asm volatile("ds_read_b128 %0, %1" : "=v"(f[t]) : "v"(addr[t] + off));
}
__device__ __forceinline__ void ax_wait12(af32x4 (&f)[12]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]));
    asm volatile("" : "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]));
}
// eight fp32 values -> their three bf16 planes as MFMA operands
__device__ __forceinline__ void ax_split8(const float (&v)[8], bf16x8 (&out)[3]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const X3Triple t = x3_split(v[e]);
        out[0][e] = t.h;
        out[1][e] = t.m;
        out[2][e] = t.l;
    }
}
}  // namespace

struct AttnX3Args {
    const unsigned short* qk3;   // X3 rows [T][ldqk]: columns [0, d) = q (pre-scaled), [d, 2 d) = k (pre-scaled)
    long ldqk;                   // fp32-element units (2 d)
    long vt_off;                 // V^T = qk3 + vt_off (bf16 units): X3 rows [d][vt_ld], chunks along the keys
    long vt_ld;
    float* out;                  // [T][ldo] fp32
    long ldo;
    int T, d, n_head;
    int batch;
    PtrTable z;                  // batched encodes: qk3 = z.in[i], out = z.out[i]
};

__global__ __launch_bounds__(512) void enc_attention_x3_kernel(AttnX3Args a) {
    asm volatile("" ::"s"(a.qk3), "s"(a.ldqk), "s"(a.vt_off), "s"(a.vt_ld), "s"(a.out), "s"(a.ldo), "s"(a.T), "s"(a.d), "s"(a.n_head),
                 "s"(a.batch));
    __builtin_amdgcn_sched_barrier(0);
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool batched = a.batch > 0;
    const unsigned short* const qk3 = batched ? reinterpret_cast<const unsigned short*>(table_at(a.z.in, blockIdx.y)) : a.qk3;
    float* const aout = batched ? table_at(a.z.out, blockIdx.y) : a.out;
    const int T = a.T;
    const int head = blockIdx.x % a.n_head;          // head == XCD for 8 heads: a head's K / V stay in one L2
    const int q0 = (blockIdx.x / a.n_head) * AX_QT;
    const int n_tiles = (T + AX_KT - 1) / AX_KT;
    const int n_steps = (n_tiles + 1) / 2;            // pairs of key tiles: stream 0 takes tile 2 j, stream 1 tile 2 j + 1
    const long row_qk = 3 * a.ldqk * 2, row_vt = 3 * a.vt_ld * 2;    // bytes

    if (wave >= 4) {
        // ---- loader: piece j of a slot covers its bytes [1024 j, 1024 j + 1024): [stream 0: K | V^T][stream 1: K | V^T] ----
        const int lw = wave - 4;
        const char* src[AX_NPW];
        long adv[AX_NPW];         // bytes per pair step
        int clamp_keys[AX_NPW];   // K pieces: the key row this lane fetches relative to the tile's first key (-1: V^T piece)
        int stream_of[AX_NPW];
#pragma unroll
        for (int i = 0; i < AX_NPW; ++i) {
            const int j = lw + 4 * i;
            const int byte = 1024 * j + 16 * lane;
            const int stream = byte / AX_TILE_BYTES, in_tile = byte - stream * AX_TILE_BYTES;
            stream_of[i] = stream;
            if (in_tile < AX_K_BYTES) {
                const int row = in_tile / AX_K_ROW;
                const int unit = ((in_tile - row * AX_K_ROW) >> 4) ^ ((row >> 1) & 7);
                clamp_keys[i] = row;
                src[i] = reinterpret_cast<const char*>(qk3) + (long)(a.d / 8 + head * 8) * 48 + unit * 16;    // + key * row_qk
                adv[i] = 0;
            } else {
                const int off = in_tile - AX_K_BYTES;
                const int row = off / AX_V_ROW;
                const int unit = ((off - row * AX_V_ROW) >> 4) ^ ((row >> 2) & 3);
                clamp_keys[i] = -1;
                src[i] = reinterpret_cast<const char*>(qk3 + a.vt_off) + (long)(head * 64 + row) * row_vt + unit * 16 +
                         (long)stream * (AX_KT / 8) * 48;
                adv[i] = 2L * (AX_KT / 8) * 48;          // two tiles of 32 keys = 8 stored chunks further
            }
        }
        auto issue_step = [&](int step) {
            const int sc = min(step, n_steps - 1);         // the tail re-fetches the last pair into a free slot
            const int slot = step % AX_NB;
            ax_static_for<AX_NPW>([&](auto I) {
                constexpr int i = decltype(I)::value;
                const char* p;
                if (clamp_keys[i] >= 0) {
                    const int key = min((2 * sc + stream_of[i]) * AX_KT + clamp_keys[i], T - 1);   // rows past T: clamped, masked below
                    p = src[i] + (long)key * row_qk;
                } else {
                    // a second stream without a tile (odd tile count) reads the padding columns of V^T: inside the row
                    p = src[i] + (long)min(sc, (n_tiles - 1 - stream_of[i]) / 2) * adv[i];
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(lds + slot * AX_SLOT_BYTES + (lw + 4 * i) * 1024),
                                                 16, 0, 0);
            });
        };
        issue_step(0);
        issue_step(1);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int st = 0; st < n_steps; ++st) {
            issue_step(st + AX_DT);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        return;
    }

    // ---- compute waves ------------------------------------------------------------------------------------------------
    const int qb = wave & 1, ks = wave >> 1;
    const int r = lane & 31, hi = lane >> 5;
    // Q fragments (B operand of S^T = K Q^T): query q0 + 32 qb + r, feature chunk 2 s + hi, plane p
    bf16x8 qf[12];
    {
        const int q = min(q0 + 32 * qb + r, T - 1);
        const char* qp = reinterpret_cast<const char*>(qk3) + (long)q * row_qk + (long)(head * 8) * 48;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                qf[s * 3 + p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const x3_u32x4*>(qp + ((2 * s + hi) * 3 + p) * 16));
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned tile_base = lds_base + (unsigned)ks * AX_TILE_BYTES;
    unsigned k_addr[12], v_addr[2][6];
    {
        const unsigned kswz = (unsigned)((r >> 1) & 7), vswz = (unsigned)((r >> 2) & 3);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                k_addr[s * 3 + p] = tile_base + (unsigned)(r * AX_K_ROW) + ((unsigned)((2 * s + hi) * 3 + p) ^ kswz) * 16u;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p)      // d = 32 b + r (swizzle of row 32 b + r == that of row r), stored chunk 2 t + hi
                    v_addr[b][t * 3 + p] = tile_base + AX_K_BYTES + (unsigned)((32 * b + r) * AX_V_ROW) + ((unsigned)((2 * t + hi) * 3 + p) ^ vswz) * 16u;
    }
    af32x16 o[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[b][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    __builtin_amdgcn_s_barrier();              // pair 0 has landed
    for (int st = 0; st < n_steps; ++st) {
        const unsigned off = (unsigned)(st % AX_NB) * (unsigned)AX_SLOT_BYTES;
        const int tile = 2 * st + ks;
        const bool have = tile < n_tiles;      // wave-uniform
        if (have) {
            const int key0 = tile * AX_KT;
            af32x4 kf[12], vf[12];
            ax_read12(kf, k_addr, off);
            {
                unsigned va[12];
#pragma unroll
                for (int i = 0; i < 6; ++i) { va[i] = v_addr[0][i]; va[6 + i] = v_addr[1][i]; }
                ax_read12(vf, va, off);
            }
            ax_wait12(kf);
            // ---- S^T = K Q^T: two accumulator chains (even / odd k-steps), added afterwards
            af32x16 sa, sb;
#pragma unroll
            for (int i = 0; i < 16; ++i) { sa[i] = 0.f; sb[i] = 0.f; }
            ax_static_for<24>([&](auto X) {
                constexpr int x = decltype(X)::value;
                constexpr int t = x / 4, s = x % 4;      // product t of k-step s
                if constexpr (s % 2 == 0)
                    sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[s * 3 + AX_PA[t]]), qf[s * 3 + AX_PB[t]], sa, 0, 0, 0);
                else
                    sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[s * 3 + AX_PA[t]]), qf[s * 3 + AX_PB[t]], sb, 0, 0, 0);
            });
            float s[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = sa[i] + sb[i];
            // ---- online softmax over this lane's query column (its 16 keys here, the other 16 in lane ^ 32)
            if (key0 + AX_KT > T) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (key0 + (i & 3) + 8 * (i >> 2) + 4 * hi >= T) s[i] = -INFINITY;
            }
            float mt = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) mt = fmaxf(mt, s[i]);
            mt = fmaxf(mt, wave_xor<32>(mt));
            const float m_new = fmaxf(m_run, mt);
            const float alpha = __expf(m_run - m_new);
            float rs = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s[i] = __expf(s[i] - m_new);
                rs += s[i];
            }
            rs += wave_xor<32>(rs);
            l_run = l_run * alpha + rs;
            m_run = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
            }
            // ---- O^T += V^T P^T: the probabilities of k-step t are registers 8 t .. 8 t + 7, split into their planes
            bf16x8 pf[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float v8[8] = {s[8 * t], s[8 * t + 1], s[8 * t + 2], s[8 * t + 3], s[8 * t + 4], s[8 * t + 5], s[8 * t + 6], s[8 * t + 7]};
                ax_split8(v8, pf[t]);
            }
            ax_wait12(vf);
            ax_static_for<24>([&](auto X) {
                constexpr int x = decltype(X)::value;
                constexpr int pr = x / 4, t = (x / 2) % 2, b = x % 2;      // product pr of k-step t, feature block b
                o[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[b * 6 + t * 3 + AX_PA[pr]]), pf[t][AX_PB[pr]], o[b], 0, 0, 0);
            });
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // pair st + 1 readable; everybody is done with pair st
    }
    __builtin_amdgcn_s_barrier();              // the loaders' tail fetches have landed: the ring can be reused

    // ---- merge the two key streams of each query block through LDS and write the normalised rows ----------------------
    // o[b][i] is O^T[feature 32 b + (i & 3) + 8 (i >> 2) + 4 hi][query r]
    constexpr int O_LD = 68;
    float* Os = reinterpret_cast<float*>(lds);                 // [wave][32 queries][O_LD]
    float* Ms = Os + 4 * 32 * O_LD;                            // [wave][32]
    float* Ls = Ms + 4 * 32;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) Os[(wave * 32 + r) * O_LD + 32 * b + (i & 3) + 8 * (i >> 2) + 4 * hi] = o[b][i];
    if (hi == 0) {
        Ms[wave * 32 + r] = m_run;
        Ls[wave * 32 + r] = l_run;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        const int tid = threadIdx.x;       // 0 .. 255
        const int dd = tid & 63, qg = tid >> 6;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int q = qg * 16 + i;     // 0 .. 63: query block q >> 5, row q & 31
            const int qrow = q0 + q;
            const int w0 = q >> 5, w1 = w0 + 2;      // the block's two streams (wave = qb + 2 ks)
            const float m0 = Ms[w0 * 32 + (q & 31)], m1 = Ms[w1 * 32 + (q & 31)];
            const float M = fmaxf(m0, m1);
            const float e0 = m0 == -INFINITY ? 0.f : expf(m0 - M), e1 = m1 == -INFINITY ? 0.f : expf(m1 - M);
            const float L = e0 * Ls[w0 * 32 + (q & 31)] + e1 * Ls[w1 * 32 + (q & 31)];
            const float acc = e0 * Os[(w0 * 32 + (q & 31)) * O_LD + dd] + e1 * Os[(w1 * 32 + (q & 31)) * O_LD + dd];
            if (qrow < T) aout[(long)qrow * a.ldo + head * 64 + dd] = acc / L;
        }
    }
}

// fp32 qkv [T][3 d] -> the operand image enc_attention_x3_kernel reads (the layout the qkv projection's X3 epilogue
// writes): X3 rows [T][2 d] (q | k), then V^T as X3 rows [d][vt_ld] with the key order 0-3, 8-11, 4-7, 12-15 per 16-group.
// Diagnostics only (wlk_diag_encoder_attention_x3): in the product the GEMM epilogue produces this image directly.
__global__ __launch_bounds__(256) void x3_pack_qkv_kernel(const float* __restrict__ qkv, unsigned short* __restrict__ out, int T, int d,
                                                          long vt_off, long vt_ld) {
    const long n_qk = (long)T * (2 * d / 8), n_vt = (long)d * (vt_ld / 8);
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    float v[8];
    if (idx < n_qk) {
        const int row = (int)(idx / (2 * d / 8)), c = (int)(idx - (long)row * (2 * d / 8));
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = qkv[(long)row * 3 * d + 8 * c + e];
        x3_store_chunk(out + (long)row * 3 * (2 * d) + (long)c * 24, v);
    } else if (idx < n_qk + n_vt) {
        const long j = idx - n_qk;
        const int dcol = (int)(j / (vt_ld / 8)), u = (int)(j - (long)dcol * (vt_ld / 8));
        const int r0 = 16 * (u >> 1) + 4 * (u & 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int t = r0 + (e & 3) + 8 * (e >> 2);
            v[e] = t < T ? qkv[(long)t * 3 * d + 2 * d + dcol] : 0.f;
        }
        x3_store_chunk(out + vt_off + (long)dcol * 3 * vt_ld + (long)u * 24, v);
    }
}

void launch_x3_pack_qkv(const LaunchCtx& ctx, const float* qkv, unsigned short* out, int T, int d, long vt_off, long vt_ld) {
    const long n = (long)T * (2 * d / 8) + (long)d * (vt_ld / 8);
    KernelScope ks(ctx, "x3_pack_qkv");
    hipLaunchKernelGGL(x3_pack_qkv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx.stream, qkv, out, T, d, vt_off, vt_ld);
    WLK_HIP(hipGetLastError());
}

bool enc_attention_x3_enabled() {
    static const bool on = [] {
        const char* e = getenv("WLK_X3_ATTN");
        return !(e && e[0] == '0');
    }();
    return on;
}

void launch_encoder_attention_x3(const LaunchCtx& ctx, const unsigned short* qk3, long ldqk, long vt_off, long vt_ld, float* out,
                                 long ldo, int T, int d, int n_head, const PtrTable* z, int batch) {
    if (d != n_head * 64 || ldqk % 8 != 0 || vt_ld % 16 != 0 || vt_ld < ((T + 31) / 32) * 32 || T < 64)
        throw std::invalid_argument("x3 attention: unsupported shape");
    static std::atomic<uint64_t> configured{0};
    int dev = 0;
    WLK_HIP(hipGetDevice(&dev));
    if (!(configured.load(std::memory_order_acquire) >> (dev & 63) & 1)) {
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_attention_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)AX_LDS_BYTES));
        configured.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    AttnX3Args a{};
    a.qk3 = qk3; a.ldqk = ldqk; a.vt_off = vt_off; a.vt_ld = vt_ld; a.out = out; a.ldo = ldo; a.T = T; a.d = d; a.n_head = n_head;
    a.batch = batch;
    if (batch > 0) a.z = *z;
    const int q_tiles = (T + AX_QT - 1) / AX_QT;
    KernelScope ks(ctx, "enc_attention_x3", (batch > 0 ? batch : 1) * 4.0 * (double)T * T * 64 * n_head, 0.0);
    hipLaunchKernelGGL(enc_attention_x3_kernel, dim3(q_tiles * n_head, batch > 0 ? batch : 1), dim3(512), AX_LDS_BYTES, ctx.stream, a);
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

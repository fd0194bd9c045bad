// Encoder self-attention on the bf16 matrix cores at fp32 accuracy (x3.h): softmax(Q K^T) V per 64-wide head with Q, K and
// V as three bf16 planes each, six bf16 MFMAs per fp32 product, fp32 accumulation, fp32 softmax.  gfx950 only.
// Replaces whisper/model.py:148-173 (qkv_attention) for the encoder, as enc_attention_pw_kernel (attention.hip) does
// with fp32 MFMAs: 4.6 GFLOP per launch on base.en took 72 us there (0.40 of the fp32 matrix peak, 31 % of an encode).
//
// Operands come straight from the qkv projection's epilogue (gemm_x3.hip, X3GemmArgs::x3_out):
//   Q | K : X3 rows [T][2 d] (chunks run along the feature axis) - a lane's A / B fragment is one 16-byte unit;
//   V^T   : X3 rows [d][T padded to 32] - chunks run along the KEY axis; stored chunk u of every 32-key group holds keys
//           4 u .. 4 u + 3 and 16 + 4 u .. 16 + 4 u + 3: exactly the keys whose probabilities a lane of group u holds
//           after S^T = K Q^T, so the accumulator registers of step 1 are (after the split into planes) the B operand of
//           step 2 without any data movement between lanes.
// Workgroup = (head, 64 queries), eight waves, all of them computing: wave (qs, ks) owns queries 16 qs .. 16 qs + 15 and
// the key tiles t = ks (mod 2) (flash-style online softmax per wave, the two key streams of a query block merged at the
// end); every wave also issues an eighth - six 1 KiB pieces - of the LDS-DMA that feeds a ring of three slots of two 32-key tiles (K tile
// 12 KiB + V^T tile 12 KiB per stream); one workgroup barrier per pair of tiles.  Per tile and wave: 24
// v_mfma_f32_16x16x32_bf16 for S^T, 24 for O^T += V^T P^T, an 8-value softmax per lane in between.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "wave_ops.h"
#include "x3.h"

namespace wlk {

namespace {
typedef float af32x4 __attribute__((ext_vector_type(4)));

constexpr int AX_QT = 64, AX_KT = 32;
constexpr int AX_K_ROW = 384;                              // K tile row: 8 chunks x 3 planes x 16 bytes
constexpr int AX_V_ROW = 192;                              // V^T tile row: 4 stored chunks x 3 planes x 16 bytes
constexpr int AX_K_BYTES = AX_KT * AX_K_ROW;               // 12 288
constexpr int AX_V_BYTES = 64 * AX_V_ROW;                  // 12 288
constexpr int AX_TILE_BYTES = AX_K_BYTES + AX_V_BYTES;     // one stream's tile
constexpr int AX_SLOT_BYTES = 2 * AX_TILE_BYTES;           // both streams: 49 152
constexpr int AX_NB = 3, AX_DT = AX_NB - 1;
constexpr int AX_NPW = AX_SLOT_BYTES / 1024 / 8;           // 6 DMA pieces per wave and slot
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
// key row (relative to the start of its 32-key group) of element e of stored chunk u of V^T: a lane of group u = lane >> 4
// holds, after S^T = K Q^T with 16 x 16 MFMAs, the probabilities of keys 4 u .. 4 u + 3 and 16 + 4 u .. 16 + 4 u + 3
__host__ __device__ inline int ax_vt_key(int u, int e) { return 4 * u + (e & 3) + 16 * (e >> 2); }
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
    int x3_out;                  // round 6: != 0 -> `out` (z.out[i]) is an X3 row image [T][ldo] (three bf16 planes, x3.h) instead of
                                 // fp32 rows: the operand of an X3 out projection (d >= 1024 models), no fp32 copy, no pack launch
    PtrTable z;                  // batched encodes: qk3 = z.in[i], out = z.out[i]
};

// Round 4, second version.  The first one had four compute waves (32 queries each, 32 x 32 MFMAs) beside four loader
// waves: per key tile a compute wave ran its S MFMAs, ~220 VALU instructions of softmax / plane split and its P V MFMAs
// strictly one after the other, so the matrix pipe of its SIMD idled through every softmax (47 us per launch against
// 18 us of MFMA time).  Here ALL EIGHT waves compute: wave (qs, ks) owns 16 queries (v_mfma_f32_16x16x32_bf16: same flop
// rate, half the accumulator and softmax work per wave) of key stream ks, the two waves of a SIMD belong to different
// streams and cover each other's softmax and DMA issue (6 pieces per wave and step) with their MFMAs.
// ABL (timing probe, WLK_X3_ATTN_ABL): 1 = DMA and barriers only, 2 = no DMA, 3 = MFMAs only (no softmax, no DMA)
template <int ABL>
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
    const int qs = wave & 3, ks = wave >> 2;
    const int c = lane & 15, g = lane >> 4;

    // Q fragments (B operand of S^T = K Q^T): query q0 + 16 qs + c, feature chunk 4 kstep + g, plane p - first in the queue
    bf16x8 qf[6];
    {
        const int q = min(q0 + 16 * qs + c, T - 1);
        const char* qp = reinterpret_cast<const char*>(qk3) + (long)q * row_qk + (long)(head * 8) * 48;
#pragma unroll
        for (int kstep = 0; kstep < 2; ++kstep)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                qf[kstep * 3 + p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const x3_u32x4*>(qp + ((4 * kstep + g) * 3 + p) * 16));
    }
    // ---- DMA sources: piece j of a slot covers its bytes [1024 j, 1024 j + 1024): [stream 0: K | V^T][stream 1: K | V^T];
    // the bank swizzle lives in the source address (K: chunk ^ (row & 7), V^T: chunk ^ ((row >> 1) & 3), planes in place) --
    const char* src[AX_NPW];
    long adv[AX_NPW];
    int key_row[AX_NPW];      // K pieces: the key this lane fetches relative to the tile's first key (-1: V^T piece)
    int stream_of[AX_NPW];
#pragma unroll
    for (int i = 0; i < AX_NPW; ++i) {
        const int j = wave + 8 * i;
        const int byte = 1024 * j + 16 * lane;
        const int stream = byte / AX_TILE_BYTES, in_tile = byte - stream * AX_TILE_BYTES;
        stream_of[i] = stream;
        if (in_tile < AX_K_BYTES) {
            const int row = in_tile / AX_K_ROW;
            const int up = (in_tile - row * AX_K_ROW) >> 4;
            const int unit = ((up / 3) ^ (row & 7)) * 3 + up % 3;
            key_row[i] = row;
            src[i] = reinterpret_cast<const char*>(qk3) + (long)(a.d / 8 + head * 8) * 48 + unit * 16;    // + key * row_qk
            adv[i] = 0;
        } else {
            const int off = in_tile - AX_K_BYTES;
            const int row = off / AX_V_ROW;
            const int up = (off - row * AX_V_ROW) >> 4;
            const int unit = ((up / 3) ^ ((row >> 1) & 3)) * 3 + up % 3;
            key_row[i] = -1;
            src[i] = reinterpret_cast<const char*>(qk3 + a.vt_off) + (long)(head * 64 + row) * row_vt + unit * 16 +
                     (long)stream * (AX_KT / 8) * 48;
            adv[i] = 2L * (AX_KT / 8) * 48;          // two tiles of 32 keys = 8 stored chunks further
        }
    }
    auto issue_step = [&](int step) {
        if constexpr (ABL >= 2) return;
        const int sc = min(step, n_steps - 1);         // the tail re-fetches the last pair into a free slot
        const int slot = step % AX_NB;
        ax_static_for<AX_NPW>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const char* p;
            if (key_row[i] >= 0) {
                const int key = min((2 * sc + stream_of[i]) * AX_KT + key_row[i], T - 1);   // rows past T: clamped, masked below
                p = src[i] + (long)key * row_qk;
            } else {
                // a second stream without a tile (odd tile count) re-reads its last real tile
                p = src[i] + (long)min(sc, (n_tiles - 1 - stream_of[i]) / 2) * adv[i];
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                             (__attribute__((address_space(3))) void*)(lds + slot * AX_SLOT_BYTES + (wave + 8 * i) * 1024),
                                             16, 0, 0);
        });
    };
    issue_step(0);
    issue_step(1);

    // ---- fragment addresses of this wave's stream --------------------------------------------------------------------
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned tile_base = lds_base + (unsigned)ks * AX_TILE_BYTES;
    unsigned k_addr[12], v_addr[12];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int kstep = 0; kstep < 2; ++kstep)
#pragma unroll
            for (int p = 0; p < 3; ++p) {       // key 16 kb + c, feature chunk 4 kstep + g
                const int row = 16 * kb + c;
                k_addr[kb * 6 + kstep * 3 + p] = tile_base + (unsigned)(row * AX_K_ROW) + (unsigned)((((4 * kstep + g) ^ (row & 7)) * 3 + p) * 16);
            }
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int p = 0; p < 3; ++p) {           // feature 16 db + c, stored key chunk g
            const int row = 16 * db + c;
            v_addr[db * 3 + p] = tile_base + AX_K_BYTES + (unsigned)(row * AX_V_ROW) + (unsigned)(((g ^ ((row >> 1) & 3)) * 3 + p) * 16);
        }
    af32x4 o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = af32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    // (Measured and dropped, profiles/r04j_x3_attention_ablation.txt, r04k_x3_attention_skewed_ablation.txt: running stream 1 half a step out of phase - it finishes
    // tile st - 1 while stream 0 computes S(st) - so that one wave of a SIMD is in its softmax while the other is in its
    // MFMAs: 60.8 us instead of 48.9.  What the two waves of a SIMD share is issue bandwidth, not just the matrix pipe.)
    af32x4 vf[12];
    float s[8];
    auto phase_s = [&](int tile, unsigned off) {          // fragments of `tile` -> registers, s = K Q^T
        af32x4 kf[12];
        ax_read12(kf, k_addr, off);
        ax_read12(vf, v_addr, off);
        ax_wait12(kf);
        // key blocks 0 / 1 are two independent accumulator chains
        af32x4 s2[2] = {af32x4{0.f, 0.f, 0.f, 0.f}, af32x4{0.f, 0.f, 0.f, 0.f}};
        ax_static_for<24>([&](auto X) {
            constexpr int x = decltype(X)::value;
            constexpr int t = x / 4, kstep = (x / 2) % 2, kb = x % 2;      // product t of feature step kstep, key block kb
            s2[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf[kb * 6 + kstep * 3 + AX_PA[t]]),
                                                             qf[kstep * 3 + AX_PB[t]], s2[kb], 0, 0, 0);
        });
        // s[4 kb + i] = score of key key0 + 16 kb + 4 g + i against this lane's query
#pragma unroll
        for (int i = 0; i < 4; ++i) { s[i] = s2[0][i]; s[4 + i] = s2[1][i]; }
        const int key0 = tile * AX_KT;
        if (__builtin_amdgcn_readfirstlane(key0 + AX_KT > T)) {      // only the last tile has keys past the end
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (key0 + 16 * (i >> 2) + 4 * g + (i & 3) >= T) s[i] = -INFINITY;
        }
        ax_wait12(vf);
    };
    auto phase_pv = [&]() {                                 // online softmax of s, O^T += V^T P^T
        bf16x8 pf[3];
        if constexpr (ABL != 3) {
            // this lane's query column: its 8 keys here, the others in lanes ^ 16, ^ 32
            float mt = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
            mt = fmaxf(mt, wave_xor<16>(mt));
            mt = fmaxf(mt, wave_xor<32>(mt));
            const float m_new = fmaxf(m_run, mt);
            const float alpha = __expf(m_run - m_new);
            float rs = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s[i] = __expf(s[i] - m_new);
                rs += s[i];
            }
            rs += wave_xor<16>(rs);
            rs += wave_xor<32>(rs);
            l_run = l_run * alpha + rs;
            m_run = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int db = 0; db < 4; ++db) o[db] *= alpha;
            }
            // the lane's 8 probabilities are, split into planes, its B fragment
            ax_split8(s, pf);
        } else {
#pragma unroll
            for (int p = 0; p < 3; ++p) pf[p] = __builtin_bit_cast(bf16x8, vf[p]);
        }
        ax_static_for<24>([&](auto X) {
            constexpr int x = decltype(X)::value;
            constexpr int t = x / 4, db = x % 4;      // product t of feature block db: four independent chains
            o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf[db * 3 + AX_PA[t]]), pf[AX_PB[t]], o[db], 0, 0, 0);
        });
    };

    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // the Q fragments and this wave's pieces of pair 0
    __builtin_amdgcn_s_barrier();
    for (int st = 0; st < n_steps; ++st) {
        issue_step(st + AX_DT);
        const unsigned off = (unsigned)(st % AX_NB) * (unsigned)AX_SLOT_BYTES;
        const int tile = 2 * st + ks;
        if (ABL != 1 && tile < n_tiles) {      // wave-uniform
            phase_s(tile, off);
            phase_pv();
        }
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");   // this wave's pieces of pair st + 1 have landed
        __builtin_amdgcn_s_barrier();          // pair st + 1 readable; everybody is done with pair st's LDS image
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // the tail fetches have landed: the ring can be reused

    // ---- merge the two key streams of each 16-query block through LDS and write the normalised rows -------------------
    // o[db][i] is O^T[feature 16 db + 4 g + i][query c]
    constexpr int O_LD = 68;
    float* Os = reinterpret_cast<float*>(lds);                 // [wave][16 queries][O_LD]
    float* Ms = Os + 8 * 16 * O_LD;                            // [wave][16]
    float* Ls = Ms + 8 * 16;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int i = 0; i < 4; ++i) Os[(wave * 16 + c) * O_LD + 16 * db + 4 * g + i] = o[db][i];
    if (g == 0) {
        Ms[wave * 16 + c] = m_run;
        Ls[wave * 16 + c] = l_run;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (a.x3_out) {
        // one (query, 8-feature chunk) per thread: 64 queries x 8 chunks = the 512 threads; same merge arithmetic per element
        const int tid = threadIdx.x;
        const int q = tid >> 3, ch = tid & 7;
        const int qrow = q0 + q;
        const int w0 = q >> 4, w1 = w0 + 4;
        const float m0 = Ms[w0 * 16 + (q & 15)], m1 = Ms[w1 * 16 + (q & 15)];
        const float M = fmaxf(m0, m1);
        const float e0 = m0 == -INFINITY ? 0.f : expf(m0 - M), e1 = m1 == -INFINITY ? 0.f : expf(m1 - M);
        const float L = e0 * Ls[w0 * 16 + (q & 15)] + e1 * Ls[w1 * 16 + (q & 15)];
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int dd = 8 * ch + e;
            v[e] = (e0 * Os[(w0 * 16 + (q & 15)) * O_LD + dd] + e1 * Os[(w1 * 16 + (q & 15)) * O_LD + dd]) / L;
        }
        if (qrow < T)
            x3_store_chunk(reinterpret_cast<unsigned short*>(aout) + (long)qrow * 3 * a.ldo + (long)(head * 8 + ch) * 24, v);
    } else {
        const int tid = threadIdx.x;       // 0 .. 511
        const int dd = tid & 63, qg = tid >> 6;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = qg * 8 + i;      // 0 .. 63: 16-query block q >> 4, row q & 15
            const int qrow = q0 + q;
            const int w0 = q >> 4, w1 = w0 + 4;      // the block's two streams (wave = qs + 4 ks)
            const float m0 = Ms[w0 * 16 + (q & 15)], m1 = Ms[w1 * 16 + (q & 15)];
            const float M = fmaxf(m0, m1);
            const float e0 = m0 == -INFINITY ? 0.f : expf(m0 - M), e1 = m1 == -INFINITY ? 0.f : expf(m1 - M);
            const float L = e0 * Ls[w0 * 16 + (q & 15)] + e1 * Ls[w1 * 16 + (q & 15)];
            const float acc = e0 * Os[(w0 * 16 + (q & 15)) * O_LD + dd] + e1 * Os[(w1 * 16 + (q & 15)) * O_LD + dd];
            if (qrow < T) aout[(long)qrow * a.ldo + head * 64 + dd] = acc / L;
        }
    }
}

// fp32 qkv [T][3 d] -> the operand image enc_attention_x3_kernel reads (the layout the qkv projection's X3 epilogue
// writes): X3 rows [T][2 d] (q | k), then V^T as X3 rows [d][vt_ld] whose stored chunk u of every 32-key group holds the
// keys 4 u .. 4 u + 3 and 16 + 4 u .. 16 + 4 u + 3 (ax_vt_key).
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
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int t = 32 * (u >> 2) + ax_vt_key(u & 3, e);
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
                                 long ldo, int T, int d, int n_head, const PtrTable* z, int batch, bool x3_out) {
    if (d != n_head * 64 || ldqk % 8 != 0 || vt_ld % 32 != 0 || vt_ld < ((T + 31) / 32) * 32 || T < 64)
        throw std::invalid_argument("x3 attention: unsupported shape");
    static std::atomic<uint64_t> configured{0};
    int dev = 0;
    WLK_HIP(hipGetDevice(&dev));
    if (!(configured.load(std::memory_order_acquire) >> (dev & 63) & 1)) {
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_attention_x3_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AX_LDS_BYTES));
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_attention_x3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AX_LDS_BYTES));
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_attention_x3_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AX_LDS_BYTES));
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_attention_x3_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AX_LDS_BYTES));
        configured.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    AttnX3Args a{};
    a.qk3 = qk3; a.ldqk = ldqk; a.vt_off = vt_off; a.vt_ld = vt_ld; a.out = out; a.ldo = ldo; a.T = T; a.d = d; a.n_head = n_head;
    a.batch = batch;
    a.x3_out = x3_out ? 1 : 0;
    if (x3_out && ldo % 8 != 0) throw std::invalid_argument("x3 attention: X3 result rows need ldo % 8 == 0");
    if (batch > 0) a.z = *z;
    const int q_tiles = (T + AX_QT - 1) / AX_QT;
    KernelScope ks(ctx, "enc_attention_x3", (batch > 0 ? batch : 1) * 4.0 * (double)T * T * 64 * n_head, 0.0);
    static const int abl = [] {
        const char* e = getenv("WLK_X3_ATTN_ABL");
        return e ? atoi(e) : 0;
    }();
    const dim3 grid(q_tiles * n_head, batch > 0 ? batch : 1);
    if (abl == 1) hipLaunchKernelGGL(enc_attention_x3_kernel<1>, grid, dim3(512), AX_LDS_BYTES, ctx.stream, a);
    else if (abl == 2) hipLaunchKernelGGL(enc_attention_x3_kernel<2>, grid, dim3(512), AX_LDS_BYTES, ctx.stream, a);
    else if (abl == 3) hipLaunchKernelGGL(enc_attention_x3_kernel<3>, grid, dim3(512), AX_LDS_BYTES, ctx.stream, a);
    else hipLaunchKernelGGL(enc_attention_x3_kernel<0>, grid, dim3(512), AX_LDS_BYTES, ctx.stream, a);
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

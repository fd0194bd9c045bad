// fp32 GEMM / GEMV for the Whisper encoder and decoder on gfx950.
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T + bias[N] )
//
// Both operands are K-contiguous ("NT"), which is how the reference stores activations
// [tokens, features] and nn.Linear weights [out, in] (whisper/model.py:44-50).  The two 1-D
// convolutions of the encoder stem (whisper/model.py:230-231,244-245) are the same GEMM with
// overlapping A rows (lda < K) over a time-major, zero-padded activation buffer, so no im2col
// buffer is ever materialised.
//
// Matrix core: v_mfma_f32_32x32x2_f32 - exact fp32 (bitwise an fmaf chain), 64 cycles per
// instruction per SIMD, 157 TFLOP/s chip peak.  One wave owns a 32x32 tile (16 accumulator
// VGPRs); a 256-thread workgroup owns 64x64 and walks K in steps of 32 through a double-
// buffered LDS stage.  LDS rows are padded to 36 floats so that the per-lane ds_read_b128 of
// 4 consecutive k values is conflict-free in every 16-lane service group (bank = 4*(9*row mod 16)).
//
// k-permutation trick: MFMA step j of k-group s takes k = 8s+j from lanes 0-31 and k = 8s+4+j
// from lanes 32-63, for A and B alike.  A contraction is order-free, so each lane fetches its
// four k values with ONE 16-byte LDS read instead of four strided 4-byte reads.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace wlk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32, LDS_LD = BK + 4;

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// One operand-tile slice held in registers between its global load and its LDS store.
template <int NA, int NW>
struct Stage {
    float4 a[NA], w[NW];
};

// BM x BN workgroup tile, one 32x32 MFMA tile per wave ((BM/32)*(BN/32) waves).  Global -> register
// -> LDS staging runs TWO k-tiles ahead of the MFMA stream: the loads of tile kt+2 are issued before
// the math on tile kt, and are only waited for one full iteration later, so a cold-L2 / Infinity-Cache
// round trip (every kernel starts with an invalidated L2 on this multi-XCD part) is covered even when
// a CU holds a single workgroup - the common case for the 1500-row encoder GEMMs (192..768 tiles on
// 256 CUs).  Guards are branch-free (clamped address + select) so the loads stay unpredicated.
template <int BM, int BN>
__global__ __launch_bounds__(64 * (BM / 32) * (BN / 32)) void gemm_nt_f32_kernel(GemmArgs g) {
    constexpr int NT = 64 * (BM / 32) * (BN / 32);
    constexpr int NA = BM * 8 / NT, NW = BN * 8 / NT;
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Ws[2][BN * LDS_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / (BN / 32), wc = wave % (BN / 32);
    // batched encodes: this workgroup works on session blockIdx.y's buffers (locals, not writes into `g`: a modified
    // argument struct would be materialised in scratch memory)
    const bool batched = g.batch > 0;
    const float* const gA = batched ? table_at(g.z.in, blockIdx.y) : g.A;
    float* const gC = batched ? table_at(g.z.out, blockIdx.y) : g.C;
    const float* const gR = batched ? table_at(g.z.res, blockIdx.y) : g.R;
    // XCD-aware tile mapping.  Workgroup b is dispatched to XCD b % 8 (observed; used for speed only) and
    // each XCD has a private 4 MiB L2 that starts cold at every launch.  With the plain (x = N tile,
    // y = M tile) order every XCD touches ALL of A, so 8 copies of A cross the fabric (TCC miss rate 50 %
    // on the 1500x512x2048 GEMM).  Cutting the tile grid into 4 row bands x 2 column bands, one per XCD,
    // brings fabric traffic from 8A + W down to 2A + 4W bytes (TCC misses 825k -> 348k on that GEMM).
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    int tile_m, tile_n;
    if (tiles_m >= 8) {
        const int band_m = (tiles_m + 3) / 4, band_n = (tiles_n + 1) / 2;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        tile_m = (xcd >> 1) * band_m + slot / band_n;
        tile_n = (xcd & 1) * band_n + slot % band_n;
        if (slot >= band_m * band_n || tile_m >= tiles_m || tile_n >= tiles_n) return;   // padding workgroups
    } else {   // skinny problems (decoder prefill): plain order, every XCD gets work
        tile_m = blockIdx.x / tiles_n;
        tile_n = blockIdx.x - tile_m * tiles_n;
        if (tile_m >= tiles_m) return;
    }
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    // Operand tiles are fetched with buffer loads: the hardware bounds check returns 0 for any
    // offset past the descriptor's size, so row / K tails need no branches and the compiler can
    // count outstanding loads exactly (counted vmcnt keeps the second prefetch stage in flight).
    constexpr unsigned kOob = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(gA), 0, (int)((((long)g.M - 1) * g.lda + g.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.W), 0, (int)((long)g.N * g.K * 4), 0x00020000);
    unsigned a_byte[NA], w_byte[NW];
    int a_lds[NA], w_lds[NW], a_c[NA], w_c[NW];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int idx = tid + NT * i, row = idx >> 3, c4 = idx & 7;
        a_byte[i] = (m0 + row) < g.M ? (unsigned)(((long)(m0 + row) * g.lda + c4 * 4) * 4) : kOob;
        a_lds[i] = row * LDS_LD + c4 * 4;
        a_c[i] = c4 * 4;
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int idx = tid + NT * i, row = idx >> 3, c4 = idx & 7;
        w_byte[i] = (n0 + row) < g.N ? (unsigned)(((long)(n0 + row) * g.K + c4 * 4) * 4) : kOob;
        w_lds[i] = row * LDS_LD + c4 * 4;
        w_c[i] = c4 * 4;
    }
    const int nk = (g.K + BK - 1) / BK;

    auto fetch = [&](Stage<NA, NW>& st, int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const unsigned off = (k0 + a_c[i]) < g.K ? a_byte[i] + (unsigned)k0 * 4u : kOob;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)off, 0, 0);
            st.a[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const unsigned off = (k0 + w_c[i]) < g.K ? w_byte[i] + (unsigned)k0 * 4u : kOob;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (int)off, 0, 0);
            st.w[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
    };
    auto stash = [&](const Stage<NA, NW>& st, int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<float4*>(&As[buf][a_lds[i]]) = st.a[i];
#pragma unroll
        for (int i = 0; i < NW; ++i) *reinterpret_cast<float4*>(&Ws[buf][w_lds[i]]) = st.w[i];
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int a_off = (wr * 32 + (lane & 31)) * LDS_LD + (lane >> 5) * 4;
    const int w_off = (wc * 32 + (lane & 31)) * LDS_LD + (lane >> 5) * 4;
    // Straight-line software pipeline, two k-tiles per trip, no conditionals inside: tiles past the end of K are
    // fetched out of bounds (-> zeros, contributing nothing), so the load count in flight is a compile-time
    // constant and hipcc emits counted vmcnt waits instead of vmcnt(0).
    //
    // The 16 MFMAs of a k-tile are one dependent chain (64 cycles each) and a wave issues in order, so every LDS
    // fragment read is placed right AFTER an MFMA has been issued and is consumed a full MFMA group later: the
    // fragments of groups 2-3 are read during group 0-1's matrix work, those of the next tile's groups 0-1 (after
    // the barrier that publishes that tile) during groups 2-3.  Same accumulation order as the plain loop.
    Stage<NA, NW> s0, s1;
    fetch(s0, 0);
    fetch(s1, 1);
    stash(s0, 0);
    __syncthreads();
    const int nk2 = (nk + 1) & ~1;
    {
#define WLK_MFMA(a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)
#define WLK_SB() __builtin_amdgcn_sched_barrier(0)
        auto rd_a = [&](int buf, int grp) { return *reinterpret_cast<const float4*>(&As[buf][a_off + grp * 8]); };
        auto rd_w = [&](int buf, int grp) { return *reinterpret_cast<const float4*>(&Ws[buf][w_off + grp * 8]); };
        float4 fa0 = rd_a(0, 0), fb0 = rd_w(0, 0), fa1 = rd_a(0, 1), fb1 = rd_w(0, 1), fa2, fb2, fa3, fb3;
        // one k-tile in LDS buffer `buf`; `nxt` is the register stage that goes to the other buffer meanwhile
        auto tile = [&](int buf, const Stage<NA, NW>& nxt) {
            WLK_MFMA(fa0.x, fb0.x); WLK_SB();
            fa2 = rd_a(buf, 2); fb2 = rd_w(buf, 2); WLK_SB();
            WLK_MFMA(fa0.y, fb0.y); WLK_MFMA(fa0.z, fb0.z); WLK_MFMA(fa0.w, fb0.w);
            WLK_MFMA(fa1.x, fb1.x); WLK_SB();
            fa3 = rd_a(buf, 3); fb3 = rd_w(buf, 3); WLK_SB();
            WLK_MFMA(fa1.y, fb1.y); WLK_MFMA(fa1.z, fb1.z); WLK_MFMA(fa1.w, fb1.w); WLK_SB();
            stash(nxt, buf ^ 1);
            __syncthreads();
            WLK_MFMA(fa2.x, fb2.x); WLK_SB();
            fa0 = rd_a(buf ^ 1, 0); fb0 = rd_w(buf ^ 1, 0); WLK_SB();
            WLK_MFMA(fa2.y, fb2.y); WLK_MFMA(fa2.z, fb2.z); WLK_MFMA(fa2.w, fb2.w);
            WLK_MFMA(fa3.x, fb3.x); WLK_SB();
            fa1 = rd_a(buf ^ 1, 1); fb1 = rd_w(buf ^ 1, 1); WLK_SB();
            WLK_MFMA(fa3.y, fb3.y); WLK_MFMA(fa3.z, fb3.z); WLK_MFMA(fa3.w, fb3.w); WLK_SB();
        };
        for (int kt = 0; kt < nk2; kt += 2) {
            fetch(s0, kt + 2);
            WLK_SB();
            tile(0, s1);       // tile kt; tile kt+1 goes to buffer 1
            fetch(s1, kt + 3);
            WLK_SB();
            tile(1, s0);       // tile kt+1; tile kt+2 goes to buffer 0
        }
#undef WLK_MFMA
#undef WLK_SB
    }

    // epilogue: acc[r] is C[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31] of the wave tile.
    // The 16 residual reads are issued together (one latency, not sixteen) before any store.
    const int col = n0 + wc * 32 + (lane & 31);
    if (col < g.N) {
        const float b = g.bias ? g.bias[col] : 0.f;
        const bool do_scale = (g.flags & kGemmScaleCols) && (g.scale_period ? col % g.scale_period : col) < g.scale_cols;
        const int row_base = m0 + wr * 32 + 4 * (lane >> 5);
        float res[16];
        if (g.flags & kGemmResidual) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + (r & 3) + 8 * (r >> 2);
                res[r] = gR[(long)min(row, g.M - 1) * g.ldr + col];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) res[r] = 0.f;
        }
        const bool gelu = (g.flags & kGemmGelu) != 0;
        const bool relu = (g.flags & kGemmRelu) != 0, swish = (g.flags & kGemmSwish) != 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_base + (r & 3) + 8 * (r >> 2);
            float v = acc[r] + b;
            if (do_scale) v *= g.scale;
            if (gelu) v = gelu_erf(v);
            if (relu) v = fmaxf(v, 0.f);
            if (swish) v = v / (1.0f + expf(-v));
            v += res[r];
            if (row < g.M) gC[(long)row * g.ldc + col] = v;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Under-filled grids (decoder prefill: M ~ 60 rows; Sortformer: M <= 401 rows).  One wave computes a
// 32x32xK tile at 153.6 GFLOP/s at best (one SIMD's share of the matrix peak), so a 32x32x2048 tile costs
// >= 27 us however idle the rest of the chip is.  Here the four waves of a workgroup own the SAME 32x32
// output tile and split K between them: the workgroup stages 32x128 slabs of A and W, wave w multiplies
// k-sub-tile w of every slab, and the four partial tiles are added in wave order through LDS (fixed order:
// run-to-run identical).  4x the waves per output tile, no cross-workgroup reduction, same k-permutation
// and buffer-load pipeline as the main kernel.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_nt_f32_kwave_kernel(GemmArgs g) {
    constexpr int KW = 4, SLAB = BK * KW, SUB = 32 * LDS_LD;
    __shared__ __attribute__((aligned(16))) float As[2][KW * SUB];
    __shared__ __attribute__((aligned(16))) float Ws[2][KW * SUB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (g.N + 31) / 32;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * 32, n0 = tile_n * 32;

    constexpr unsigned kOob = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.A), 0, (int)((((long)g.M - 1) * g.lda + g.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.W), 0, (int)((long)g.N * g.K * 4), 0x00020000);
    // a slab row is 32 float4 (512 contiguous bytes); thread t moves float4 (t & 31) of rows (t >> 5) + 8 i
    unsigned a_byte[4], w_byte[4];
    int lds_at[4];
    const int c4 = tid & 31, kcol = c4 * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 5) + 8 * i;
        a_byte[i] = (m0 + row) < g.M ? (unsigned)(((long)(m0 + row) * g.lda + kcol) * 4) : kOob;
        w_byte[i] = (n0 + row) < g.N ? (unsigned)(((long)(n0 + row) * g.K + kcol) * 4) : kOob;
        lds_at[i] = (c4 >> 3) * SUB + row * LDS_LD + (c4 & 7) * 4;
    }
    struct Slab {
        float4 a[4], w[4];
    };
    auto fetch = [&](Slab& st, int ks) {
        const int k0 = ks * SLAB;
        const bool in = (k0 + kcol) < g.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)(in ? a_byte[i] + (unsigned)k0 * 4u : kOob), 0, 0);
            st.a[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (int)(in ? w_byte[i] + (unsigned)k0 * 4u : kOob), 0, 0);
            st.w[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
    };
    auto stash = [&](const Slab& st, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&As[buf][lds_at[i]]) = st.a[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&Ws[buf][lds_at[i]]) = st.w[i];
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int frag = wave * SUB + (lane & 31) * LDS_LD + (lane >> 5) * 4;
    auto mma = [&](int buf) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 a4 = *reinterpret_cast<const float4*>(&As[buf][frag + s * 8]);
            const float4 b4 = *reinterpret_cast<const float4*>(&Ws[buf][frag + s * 8]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
        }
    };
    const int ns = (g.K + SLAB - 1) / SLAB, ns2 = (ns + 1) & ~1;
    Slab s0, s1;
    fetch(s0, 0);
    fetch(s1, 1);
    stash(s0, 0);
    __syncthreads();
    for (int ks = 0; ks < ns2; ks += 2) {
        fetch(s0, ks + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(0);
        stash(s1, 1);
        __syncthreads();
        fetch(s1, ks + 3);
        __builtin_amdgcn_sched_barrier(0);
        mma(1);
        stash(s0, 0);
        __syncthreads();
    }
    // fold the four k-partials in wave order (the loop's last barrier already separates the final LDS reads)
    float* red = &As[0][0];
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < KW - 1; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[(w * 16 + r) * 64 + lane];

    const int col = n0 + (lane & 31);
    if (col >= g.N) return;
    const float b = g.bias ? g.bias[col] : 0.f;
    const bool do_scale = (g.flags & kGemmScaleCols) && (g.scale_period ? col % g.scale_period : col) < g.scale_cols;
    const int row_base = m0 + 4 * (lane >> 5);
    // decoder prefill: the k / v thirds of the projection also go to the self-attention caches (row = beam * n_tok + t)
    float* kv_dst = nullptr;
    int kv_col = 0, kv_off = 0;
    if (g.kcache && col >= g.kv_d) {
        kv_dst = col < 2 * g.kv_d ? g.kcache : g.vcache;
        kv_col = col < 2 * g.kv_d ? col - g.kv_d : col - 2 * g.kv_d;
        kv_off = *g.kv_pos;
    }
    float res[16];
    if (g.flags & kGemmResidual) {
#pragma unroll
        for (int r = 0; r < 16; ++r) res[r] = g.R[(long)min(row_base + (r & 3) + 8 * (r >> 2), g.M - 1) * g.ldr + col];
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) res[r] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row_base + (r & 3) + 8 * (r >> 2);
        float v = acc[r] + b;
        if (do_scale) v *= g.scale;
        if (g.flags & kGemmGelu) v = gelu_erf(v);
        if (g.flags & kGemmRelu) v = fmaxf(v, 0.f);
        if (g.flags & kGemmSwish) v = v / (1.0f + expf(-v));
        v += res[r];
        if (row < g.M) {
            g.C[(long)row * g.ldc + col] = v;
            if (kv_dst) {
                const int bm = row / g.kv_ntok, t = row - bm * g.kv_ntok;
                kv_dst[((long)bm * g.kv_ctx + kv_off + t) * g.kv_d + kv_col] = v;
            }
        }
    }
}

// 32x32 output tiles at or below which the k-wave kernel is used (WLK_KWAVE_MAX_TILES overrides; 0 disables)
static long kwave_max_tiles() {
    static const long v = [] {
        const char* e = getenv("WLK_KWAVE_MAX_TILES");
        return e ? atol(e) : 416L;
    }();
    return v;
}

bool gemm_takes_kwave(int M, int N, int K) {
    const long tiles32 = (long)((N + 31) / 32) * ((M + 31) / 32);
    return K >= 256 && tiles32 <= kwave_max_tiles();
}

void launch_gemm(const LaunchCtx& ctx, const GemmArgs& g, const char* tag) {
    if (g.M <= 0 || g.N <= 0) return;
    if (g.K % 4 != 0 || g.lda % 4 != 0) throw std::invalid_argument("gemm: K and lda must be multiples of 4");
    if ((((long)g.M - 1) * g.lda + g.K) * 4 >= (1L << 31) || (long)g.N * g.K * 4 >= (1L << 31))
        throw std::invalid_argument("gemm: operand larger than 2 GiB");
    const double nb = std::max(g.batch, 1);
    KernelScope ks(ctx, tag, nb * 2.0 * g.M * g.N * g.K,
                   4.0 * (nb * (double)g.M * g.K + (double)g.N * g.K + nb * (double)g.M * g.N));
    if (g.batch > 0 && (gemm_takes_kwave(g.M, g.N, g.K) || g.force_kwave || (long)((g.N + 63) / 64) * ((g.M + 63) / 64) < 64))
        throw std::invalid_argument("gemm: batched launches are only available on the 64x64 path");
    if (g.batch > kMaxBatch) throw std::invalid_argument("gemm: batch too large");
    const long tiles64 = (long)((g.N + 63) / 64) * ((g.M + 63) / 64);
    const int tiles_n = (g.N + 63) / 64;
    const long tiles32 = (long)((g.N + 31) / 32) * ((g.M + 31) / 32);
    if (g.kcache && !(gemm_takes_kwave(g.M, g.N, g.K) || g.force_kwave))
        throw std::invalid_argument("gemm: fused KV-cache append is only available on the k-wave path");
    if (gemm_takes_kwave(g.M, g.N, g.K) || (g.force_kwave && g.K >= 256)) {
        hipLaunchKernelGGL(gemm_nt_f32_kwave_kernel, dim3((unsigned)tiles32), dim3(256), 0, ctx.stream, g);
    } else if (tiles64 >= 64) {
        const int tiles_m = (g.M + 63) / 64;
        const int blocks = tiles_m >= 8 ? 8 * ((tiles_m + 3) / 4) * ((tiles_n + 1) / 2) : tiles_m * tiles_n;
        hipLaunchKernelGGL((gemm_nt_f32_kernel<64, 64>), dim3(blocks, std::max(g.batch, 1)), dim3(256), 0, ctx.stream, g);
    } else {  // few tiles (decoder prefill): halve the tile height so that more CUs get a workgroup
        const int tiles_m = (g.M + 31) / 32;
        const int blocks = tiles_m >= 8 ? 8 * ((tiles_m + 3) / 4) * ((tiles_n + 1) / 2) : tiles_m * tiles_n;
        hipLaunchKernelGGL((gemm_nt_f32_kernel<32, 64>), dim3(blocks), dim3(128), 0, ctx.stream, g);
    }
    WLK_HIP(hipGetLastError());
}

// -------------------------------------------------------------------------------------------------
// GEMV path: M <= 8 activation rows (one decode step, beam <= 8).  Pure weight streaming: every
// wave owns 4 consecutive output features, its 64 lanes stride over K in float4 units
// (1 KiB coalesced per wave-instruction), the M x K activations sit in LDS, and the partial
// sums are folded with DPP-free xor shuffles.  HBM-bound by construction: K*4 bytes per output
// feature against 2*M*K flops.
// -------------------------------------------------------------------------------------------------
template <int MR, int RPW>
__global__ __launch_bounds__(256) void gemv_f32_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [MR][K]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int K4 = g.K >> 2;
    // the first weight chunk of this wave's first output group does not depend on the activations:
    // issue it BEFORE the x staging round trip so the two memory latencies overlap instead of adding up
    const int n_groups = (g.N + RPW - 1) / RPW;
    const int grp0 = blockIdx.x * 4 + wave;
    float4 wpre[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int n = min(grp0 * RPW + r, g.N - 1);
        const bool ok = grp0 < n_groups && lane < K4;
        const float4 t = *reinterpret_cast<const float4*>(g.W + (long)(ok ? n : 0) * g.K + (ok ? lane : 0) * 4);
        wpre[r] = t;
    }
    for (int i = tid; i < MR * K4; i += 256) {
        const int m = i / K4, c = i - m * K4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < g.M) v = *reinterpret_cast<const float4*>(g.A + (long)m * g.lda + c * 4);
        reinterpret_cast<float4*>(xs)[i] = v;
    }
    __syncthreads();
    if (g.ln_gamma) {
        // fused LayerNorm of the staged rows; same reduction order as layernorm_kernel (lane-strided
        // partial sums, xor-shuffle fold), so fused and unfused paths agree bit for bit
        for (int m = wave; m < g.M; m += 4) {
            float* xr = xs + m * g.K;
            float sum = 0.f;
            for (int c = lane; c < g.K; c += 64) sum += xr[c];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
            const float mean = sum / (float)g.K;
            float sq = 0.f;
            for (int c = lane; c < g.K; c += 64) {
                const float t = xr[c] - mean;
                sq += t * t;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
            const float rstd = 1.0f / sqrtf(sq / (float)g.K + 1e-5f);
            for (int c = lane; c < g.K; c += 64) xr[c] = (xr[c] - mean) * rstd * g.ln_gamma[c] + g.ln_beta[c];
        }
        __syncthreads();
    }

    for (int grp = grp0; grp < n_groups; grp += gridDim.x * 4) {
        const int n_base = grp * RPW;
        float acc[RPW][MR];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int m = 0; m < MR; ++m) acc[r][m] = 0.f;
        // K is walked in batches of 4 lane-strided float4 chunks: all loads of a batch are issued
        // before the first FMA so that several KiB per wave are in flight (weights come from HBM / MALL)
        constexpr int UB = 4;
        for (int c0 = lane; c0 < K4; c0 += 64 * UB) {
            float4 w[UB][RPW];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int c = c0 + 64 * u;
                const bool ok = c < K4;
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int n = min(n_base + r, g.N - 1);
                    if (u == 0 && grp == grp0 && c0 == lane) w[u][r] = wpre[r];
                    else w[u][r] = *reinterpret_cast<const float4*>(g.W + (long)n * g.K + (ok ? c : 0) * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int c = c0 + 64 * u;
                if (c < K4) {
#pragma unroll
                    for (int m = 0; m < MR; ++m) {
                        const float4 x = reinterpret_cast<const float4*>(xs)[m * K4 + c];
#pragma unroll
                        for (int r = 0; r < RPW; ++r) {
                            acc[r][m] = fmaf(w[u][r].x, x.x, acc[r][m]);
                            acc[r][m] = fmaf(w[u][r].y, x.y, acc[r][m]);
                            acc[r][m] = fmaf(w[u][r].z, x.z, acc[r][m]);
                            acc[r][m] = fmaf(w[u][r].w, x.w, acc[r][m]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                float v = acc[r][m];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
                acc[r][m] = v;
            }
        if (lane < RPW * MR) {
            const int r = lane / MR, m = lane - r * MR;
            const int n = n_base + r;
            if (n < g.N && m < g.M) {
                float v = 0.f;
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                    for (int mm = 0; mm < MR; ++mm)
                        if (rr == r && mm == m) v = acc[rr][mm];
                if (g.bias) v += g.bias[n];
                if ((g.flags & kGemmScaleCols) && n < g.scale_cols) v *= g.scale;
                if (g.flags & kGemmGelu) v = gelu_erf(v);
                if (g.flags & kGemmRelu) v = fmaxf(v, 0.f);
                if (g.flags & kGemmSwish) v = v / (1.0f + expf(-v));
                if (g.flags & kGemmResidual) v += g.R[(long)m * g.ldr + n];
                g.C[(long)m * g.ldc + n] = v;
                if (g.kv_rows && n >= g.kv_d) {          // batched steps: every row has its own cache
                    const StepRow sr = g.kv_rows[m];
                    const long at = g.kv_layer_off + (long)sr.offset * g.kv_d;
                    if (n < 2 * g.kv_d) sr.kcache[at + n - g.kv_d] = v;
                    else sr.vcache[at + n - 2 * g.kv_d] = v;
                } else if (g.kcache && n >= g.kv_d) {
                    const long at = ((long)m * g.kv_ctx + *g.kv_pos) * g.kv_d;
                    if (n < 2 * g.kv_d) g.kcache[at + n - g.kv_d] = v;
                    else g.vcache[at + n - 2 * g.kv_d] = v;
                }
            }
        }
    }
}

// Single-row variant (beam 1 decode steps, the common case): no LDS and no barrier at all.  Every wave loads the
// activation row itself, in the same lane-strided float4 mapping it uses for the weights, so the x fetch, the
// gamma/beta fetch and the weight fetch are ONE memory round trip instead of "stage x -> barrier -> read LDS".
// The fused LayerNorm statistics use the scalar lane-strided order of layernorm_kernel (so fused == unfused bit
// for bit) and the dot products the same fmaf order as gemv_f32_kernel: results are identical to that kernel.
template <int UB, int RPW>
__global__ __launch_bounds__(256) void gemv1_f32_kernel(GemmArgs g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K4 = g.K >> 2;
    const int n_groups = (g.N + RPW - 1) / RPW;
    if (g.mg_pm && (int)blockIdx.x >= (int)gridDim.x - g.mg_side_blocks) {
        // side job (cross_merge_kernel's tail): softmax row of the k-th alignment head of this layer -> alignment window
        int k = (int)blockIdx.x - ((int)gridDim.x - g.mg_side_blocks), head = -1;
        for (int h = 0; h < g.mg_heads; ++h)
            if (g.mg_head_rank[h] >= 0 && k-- == 0) head = h;
        if (head < 0) return;
        const long base = (long)head * kCrossSplitWays;
        float M = g.mg_pm[base];
#pragma unroll
        for (int s = 1; s < kCrossSplitWays; ++s) M = fmaxf(M, g.mg_pm[base + s]);
        float L = 0.f;
#pragma unroll
        for (int s = 0; s < kCrossSplitWays; ++s) L += g.mg_pl[base + s] * expf(g.mg_pm[base + s] - M);
        float* dst = g.mg_ring + (((long)g.mg_head_rank[head] * g.mg_n_beam + g.mg_beam_of_row[0]) * g.mg_ring_rows +
                                  g.mg_ring_row[0]) * (long)g.mg_T;
        const float* srow = g.mg_scores + (long)head * g.mg_T;
        for (int j = threadIdx.x; j < g.mg_T; j += 256) dst[j] = expf(srow[j] - M) / L;
        return;
    }
    const int grp = blockIdx.x * 4 + wave;
    if (grp >= n_groups) return;
    const int n_base = grp * RPW;
    float4 w[UB][RPW], x[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        const int c = lane + 64 * u;
        const bool ok = c < K4;
#pragma unroll
        for (int r = 0; r < RPW; ++r)
            w[u][r] = *reinterpret_cast<const float4*>(g.W + (long)min(n_base + r, g.N - 1) * g.K + (ok ? c : 0) * 4);
        if (!g.mg_pm) {
            x[u] = *reinterpret_cast<const float4*>(g.A + (ok ? c : 0) * 4);
        } else {   // cross_merge_kernel's arithmetic for the head that owns dims 4c .. 4c+3
            const int cc = ok ? c : 0;
            const int head = (4 * cc) >> 6, dd = (4 * cc) & 63;
            const long base = (long)head * kCrossSplitWays;
            float M = g.mg_pm[base];
#pragma unroll
            for (int s = 1; s < kCrossSplitWays; ++s) M = fmaxf(M, g.mg_pm[base + s]);
            float L = 0.f;
            float f[kCrossSplitWays];
#pragma unroll
            for (int s = 0; s < kCrossSplitWays; ++s) {
                f[s] = expf(g.mg_pm[base + s] - M);
                L += g.mg_pl[base + s] * f[s];
            }
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int s = 0; s < kCrossSplitWays; ++s) {
                const float4 o = *reinterpret_cast<const float4*>(g.mg_po + (base + s) * 64 + dd);
                acc.x += o.x * f[s]; acc.y += o.y * f[s]; acc.z += o.z * f[s]; acc.w += o.w * f[s];
            }
            x[u] = make_float4(acc.x / L, acc.y / L, acc.z / L, acc.w / L);
        }
    }
    if (g.ln_gamma) {
        float v[UB * 4];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < UB * 4; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < g.K ? g.A[c] : 0.f;
            sum += v[i];
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
        const float mean = sum / (float)g.K;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < UB * 4; ++i) {
            const float t = (lane + 64 * i) < g.K ? v[i] - mean : 0.f;
            sq += t * t;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
        const float rstd = 1.0f / sqrtf(sq / (float)g.K + 1e-5f);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int c = lane + 64 * u;
            const int cc = (c < K4 ? c : 0) * 4;
            const float4 ga = *reinterpret_cast<const float4*>(g.ln_gamma + cc);
            const float4 be = *reinterpret_cast<const float4*>(g.ln_beta + cc);
            x[u].x = (x[u].x - mean) * rstd * ga.x + be.x;
            x[u].y = (x[u].y - mean) * rstd * ga.y + be.y;
            x[u].z = (x[u].z - mean) * rstd * ga.z + be.z;
            x[u].w = (x[u].w - mean) * rstd * ga.w + be.w;
        }
    }
    float acc[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        if (lane + 64 * u < K4) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                acc[r] = fmaf(w[u][r].x, x[u].x, acc[r]);
                acc[r] = fmaf(w[u][r].y, x[u].y, acc[r]);
                acc[r] = fmaf(w[u][r].z, x[u].z, acc[r]);
                acc[r] = fmaf(w[u][r].w, x[u].w, acc[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc[r] += __shfl_xor(acc[r], off, 64);
    if (lane < RPW) {
        const int n = n_base + lane;
        if (n < g.N) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < RPW; ++r)
                if (r == lane) v = acc[r];
            if (g.bias) v += g.bias[n];
            if ((g.flags & kGemmScaleCols) && n < g.scale_cols) v *= g.scale;
            if (g.flags & kGemmGelu) v = gelu_erf(v);
            if (g.flags & kGemmRelu) v = fmaxf(v, 0.f);
            if (g.flags & kGemmSwish) v = v / (1.0f + expf(-v));
            if (g.flags & kGemmResidual) v += g.R[n];
            g.C[n] = v;
            if (g.kcache && n >= g.kv_d) {
                const long at = (long)(*g.kv_pos) * g.kv_d;
                if (n < 2 * g.kv_d) g.kcache[at + n - g.kv_d] = v;
                else g.vcache[at + n - 2 * g.kv_d] = v;
            }
        }
    }
}

static bool gemv1_enabled() {
    static const bool on = getenv("WLK_NO_GEMV1") == nullptr;
    return on;
}

bool gemv1_folds_merge(int K) { return gemv1_enabled() && K <= 2048 && K % 64 == 0 && getenv("WLK_NO_MERGE_FOLD") == nullptr; }

void launch_gemv(const LaunchCtx& ctx, const GemmArgs& g, const char* tag) {
    if (g.M <= 0 || g.N <= 0) return;
    if (!gemv_applicable(g.M, g.K) || g.lda % 4 != 0) throw std::invalid_argument("gemv: unsupported shape");
    const int mr = gemv_row_bucket(g.M);
    // output features per wave per pass: few for narrow layers (more workgroups in flight), more for
    // the 51864-wide vocabulary projection (amortises the shuffle folds)
    const int rpw = g.N >= 16384 ? 4 : (g.N >= 2048 ? 2 : 1);
    const int n_groups = (g.N + rpw - 1) / rpw;
    int blocks = (n_groups + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    const size_t lds = (size_t)mr * g.K * sizeof(float);
    KernelScope ks(ctx, tag, 2.0 * g.M * g.N * g.K, 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N));
    if (g.mg_pm && !(g.M == 1 && g.K <= 2048 && rpw <= 2 && gemv1_enabled() && !g.ln_gamma))
        throw std::invalid_argument("gemv: the merged cross-attention operand needs the single-row kernel");
    if (g.M == 1 && g.K <= 2048 && rpw <= 2 && gemv1_enabled() && !g.kv_rows) {
        const int ub = (g.K / 4 + 63) / 64;
        blocks += g.mg_pm ? g.mg_side_blocks : 0;
#define WLK_GEMV1(UBv)                                                                                             \
    do {                                                                                                           \
        if (rpw == 2) hipLaunchKernelGGL((gemv1_f32_kernel<UBv, 2>), dim3(blocks), dim3(256), 0, ctx.stream, g);   \
        else hipLaunchKernelGGL((gemv1_f32_kernel<UBv, 1>), dim3(blocks), dim3(256), 0, ctx.stream, g);            \
    } while (0)
        if (ub <= 2) WLK_GEMV1(2);
        else if (ub <= 4) WLK_GEMV1(4);
        else WLK_GEMV1(8);
#undef WLK_GEMV1
        WLK_HIP(hipGetLastError());
        return;
    }
#define WLK_GEMV(MRv, RPWv) \
    hipLaunchKernelGGL((gemv_f32_kernel<MRv, RPWv>), dim3(blocks), dim3(256), lds, ctx.stream, g)
#define WLK_GEMV_R(MRv)                      \
    do {                                     \
        if (rpw == 4) WLK_GEMV(MRv, 4);      \
        else if (rpw == 2) WLK_GEMV(MRv, 2); \
        else WLK_GEMV(MRv, 1);               \
    } while (0)
    switch (mr) {
        case 1: WLK_GEMV_R(1); break;
        case 2: WLK_GEMV_R(2); break;
        case 4: WLK_GEMV_R(4); break;
        default: WLK_GEMV_R(8); break;
    }
#undef WLK_GEMV_R
#undef WLK_GEMV
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

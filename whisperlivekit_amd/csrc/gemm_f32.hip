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
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "wave_ops.h"
#include "align_body.h"

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
    WLK_PIN_GEMM_ARGS(g);
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
//
// KP (round 6): the k-PIPE kernel's partition of K instead - wave w multiplies k = 32 t + 8 w .. 8 w + 7 of every 32-deep
// slab t (chunk pair w of each of the slab's four sub-tiles) in slab order, same in-wave order of the eight values, same
// wave-order fold, same epilogue expression.  Every output element is then bit for bit what gemm_nt_f32_kpipe_kernel
// computes for it with ANY tile: a row keeps its arithmetic when other rows are stacked under it and the launch moves
// from 32 x 32 tiles to one-tile-per-CU tiles (launch_gemm_kp; the Sortformer's stacked sessions).  K % 128 == 0.
// NBUF = 1 (round 6 probe): ONE slab buffer in LDS instead of two - 36.9 KB per workgroup, four workgroups per compute unit
// instead of two (a 291 x 2048 projection's 640 tiles are then resident at once instead of in two rounds), at the price of a
// second barrier per slab.  Same operands into the same MFMAs in the same order.
template <bool KP, int NBUF = 2>
__global__ __launch_bounds__(256) void gemm_nt_f32_kwave_kernel(GemmArgs g) {
    WLK_PIN_GEMM_ARGS(g);
    constexpr int KW = 4, SLAB = BK * KW, SUB = 32 * LDS_LD;
    __shared__ __attribute__((aligned(16))) float As[NBUF][KW * SUB];
    __shared__ __attribute__((aligned(16))) float Ws[NBUF][KW * SUB];
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
    const int frag = (KP ? wave * 8 : wave * SUB) + (lane & 31) * LDS_LD + (lane >> 5) * 4;
    auto mma = [&](int buf) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 a4 = *reinterpret_cast<const float4*>(&As[buf][frag + s * (KP ? SUB : 8)]);
            const float4 b4 = *reinterpret_cast<const float4*>(&Ws[buf][frag + s * (KP ? SUB : 8)]);
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
        if constexpr (NBUF == 1) __syncthreads();
        stash(s1, NBUF - 1);
        __syncthreads();
        fetch(s1, ks + 3);
        __builtin_amdgcn_sched_barrier(0);
        mma(NBUF - 1);
        if constexpr (NBUF == 1) __syncthreads();
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

// -------------------------------------------------------------------------------------------------
// The decoder prefill (M ~ 60 prompt rows) on 16 x 16 tiles: the k-wave kernel above, with v_mfma_f32_16x16x4_f32.
// A 32 x 32 tile per workgroup leaves a 60 x 512 projection on 32 of 256 compute units, each streaming 32 weight rows
// and running 16 dependent 64-cycle MFMAs per wave and slab (11 us per launch, 32 launches per prefill); 16 x 16 tiles
// are 128 workgroups that each stream 16 weight rows and run 8 MFMAs of 32 cycles per wave and slab.
// Every wave feeds the k values of its quarter in the order the 32 x 32 x 2 sequence accumulates them (k-group s: 8s,
// 8s+4, 8s+1, 8s+5 | 8s+2, 8s+6, 8s+3, 8s+7), and the four partials are folded in wave order as above.
// -------------------------------------------------------------------------------------------------
// NPL > 0: the LayerNorm in front of the projection (ln1 -> qkv, lnx -> xq, ln2 -> fc1) is part of the kernel.  K = 64 NPL is
// the model width; each wave derives mean and rstd of four of the tile's sixteen rows exactly as layernorm_kernel does
// (lane-strided elements summed in index order, butterfly fold, two passes) while the first two slabs are in flight, and
// the A values are normalised - (x - mean) * rstd * gamma + beta, the same expression - on their way into LDS.  Bit for
// bit the LayerNorm launch + this kernel (tests/test_gpu_parity.py::test_prefill_gemm_fuses_the_layernorm); 18 launches
// less per prefill.
// KP (round 6): the k-pipe kernel's partition of K, as in gemm_nt_f32_kwave_kernel<true> - wave w takes the eight k values
// 8 w .. 8 w + 7 of each of the slab's four 32-deep sub-tiles (in the same in-group order), so the 16 x 16 tiles join the "kp"
// family: bit for bit the 32 x 32 k-wave tiles' and every k-pipe tile's result.
template <int NPL, bool KP = false>
__global__ __launch_bounds__(256) void gemm_nt_f32_kwave16_kernel(GemmArgs g) {
    WLK_PIN_GEMM_ARGS(g);
    constexpr bool LN = NPL > 0;
    __shared__ float ln_stat[LN ? 32 : 1];
    constexpr int KW = 4, SLAB = BK * KW, SUB = 16 * LDS_LD;
    __shared__ __attribute__((aligned(16))) float As[2][KW * SUB];
    __shared__ __attribute__((aligned(16))) float Ws[2][KW * SUB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (g.N + 15) / 16;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * 16, n0 = tile_n * 16;

    constexpr unsigned kOob = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.A), 0, (int)((((long)g.M - 1) * g.lda + g.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.W), 0, (int)((long)g.N * g.K * 4), 0x00020000);
    // a slab row is 32 float4 (512 contiguous bytes); thread t moves float4 (t & 31) of rows (t >> 5) + 8 i, i = 0, 1
    unsigned a_byte[2], w_byte[2];
    int lds_at[2];
    const int c4 = tid & 31, kcol = c4 * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 5) + 8 * i;
        a_byte[i] = (m0 + row) < g.M ? (unsigned)(((long)(m0 + row) * g.lda + kcol) * 4) : kOob;
        w_byte[i] = (n0 + row) < g.N ? (unsigned)(((long)(n0 + row) * g.K + kcol) * 4) : kOob;
        lds_at[i] = (c4 >> 3) * SUB + row * LDS_LD + (c4 & 7) * 4;
    }
    struct Slab {
        float4 a[2], w[2];
        float4 ga, be;      // LN: gamma / beta of this thread's four k columns of the slab
    };
    auto fetch = [&](Slab& st, int ks) {
        const int k0 = ks * SLAB;
        const bool in = (k0 + kcol) < g.K;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)(in ? a_byte[i] + (unsigned)k0 * 4u : kOob), 0, 0);
            st.a[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (int)(in ? w_byte[i] + (unsigned)k0 * 4u : kOob), 0, 0);
            st.w[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
        if constexpr (LN) {
            const int kk = min(k0 + kcol, g.K - 4);
            st.ga = *reinterpret_cast<const float4*>(g.ln_gamma + kk);
            st.be = *reinterpret_cast<const float4*>(g.ln_beta + kk);
        }
    };
    [[maybe_unused]] float ln_mean[2] = {0.f, 0.f}, ln_rstd[2] = {1.f, 1.f};
    // LN: the workgroups of column tile 0 also keep the normalised rows (GemmArgs::ln_out) - `ks` is the slab being stashed
    [[maybe_unused]] const bool keep_ln = LN && g.ln_out != nullptr && tile_n == 0;
    auto stash = [&](const Slab& st, int buf, [[maybe_unused]] int ks = 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float4 a4 = st.a[i];
            if constexpr (LN) {
                a4.x = (a4.x - ln_mean[i]) * ln_rstd[i] * st.ga.x + st.be.x;
                a4.y = (a4.y - ln_mean[i]) * ln_rstd[i] * st.ga.y + st.be.y;
                a4.z = (a4.z - ln_mean[i]) * ln_rstd[i] * st.ga.z + st.be.z;
                a4.w = (a4.w - ln_mean[i]) * ln_rstd[i] * st.ga.w + st.be.w;
                const int row = m0 + (tid >> 5) + 8 * i, kk = ks * SLAB + kcol;
                if (keep_ln && row < g.M && kk < g.K) *reinterpret_cast<float4*>(g.ln_out + (long)row * g.ld_ln_out + kk) = a4;
            }
            *reinterpret_cast<float4*>(&As[buf][lds_at[i]]) = a4;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(&Ws[buf][lds_at[i]]) = st.w[i];
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // lane (row / column = lane & 15, k-slot g4 = lane >> 4): of k-group s it supplies k = 8s + 4 (g4 & 1) + (g4 >> 1) to the
    // first MFMA and that + 2 to the second - elements (g4 >> 1) and (g4 >> 1) + 2 of the float4 at 8s + 4 (g4 & 1)
    const int g4 = lane >> 4;
    const int frag = (KP ? wave * 8 : wave * SUB) + (lane & 15) * LDS_LD + (g4 & 1) * 4;
    const bool odd = (g4 >> 1) != 0;
    auto mma = [&](int buf) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 a4 = *reinterpret_cast<const float4*>(&As[buf][frag + s * (KP ? SUB : 8)]);
            const float4 b4 = *reinterpret_cast<const float4*>(&Ws[buf][frag + s * (KP ? SUB : 8)]);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(odd ? a4.y : a4.x, odd ? b4.y : b4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(odd ? a4.w : a4.z, odd ? b4.w : b4.z, acc, 0, 0, 0);
        }
    };
    const int ns = (g.K + SLAB - 1) / SLAB, ns2 = (ns + 1) & ~1;
    Slab s0, s1;
    fetch(s0, 0);
    fetch(s1, 1);
    // the epilogue's operands (bias, residual, cache position) travel with the first two slabs instead of behind the
    // last fold: acc[r] is C[row = 4 (lane >> 4) + r][col = lane & 15] of the tile; absent operands read the weights
    const int col = n0 + (lane & 15);
    const int colc = min(col, g.N - 1);
    const int row_base = m0 + 4 * g4;
    const bool has_res = (g.flags & kGemmResidual) != 0;
    float b = (g.bias ? g.bias + colc : g.W)[0];
    float res[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) res[r] = (has_res ? g.R + (long)min(row_base + r, g.M - 1) * g.ldr + colc : g.W)[0];
    int kv_off = (g.kcache ? g.kv_pos : reinterpret_cast<const int*>(g.W))[0];
    // LN: rows m0 + 4 wave + q - layernorm_kernel's statistics, all four rows' elements requested together (and before the
    // pins below: a pin waits for what it names)
    [[maybe_unused]] float v[4][LN ? NPL : 1];
    if constexpr (LN) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* xr = g.A + (long)min(m0 + 4 * wave + q, g.M - 1) * g.lda;
#pragma unroll
            for (int i = 0; i < NPL; ++i) v[q][i] = xr[lane + 64 * i];
        }
    }
    pin_loaded(b);
#pragma unroll
    for (int r = 0; r < 4; ++r) pin_loaded(res[r]);
    asm volatile("" : "+v"(kv_off));
    if constexpr (LN) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < NPL; ++i) sum += v[q][i];
            sum = wave_sum(sum);
            const float mean = sum / (float)g.K;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                const float t = v[q][i] - mean;
                sq += t * t;
            }
            sq = wave_sum(sq);
            const float rstd = 1.0f / sqrtf(sq / (float)g.K + 1e-5f);
            if (lane == 0) {
                ln_stat[2 * (4 * wave + q)] = mean;
                ln_stat[2 * (4 * wave + q) + 1] = rstd;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (tid >> 5) + 8 * i;
            ln_mean[i] = ln_stat[2 * row];
            ln_rstd[i] = ln_stat[2 * row + 1];
        }
    }
    stash(s0, 0, 0);
    __syncthreads();
    for (int ks = 0; ks < ns2; ks += 2) {
        fetch(s0, ks + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(0);
        stash(s1, 1, ks + 1);
        __syncthreads();
        fetch(s1, ks + 3);
        __builtin_amdgcn_sched_barrier(0);
        mma(1);
        stash(s0, 0, ks + 2);
        __syncthreads();
    }
    // fold the four k-partials in wave order
    float* red = &As[0][0];
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave - 1) * 4 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < KW - 1; ++w)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += red[(w * 4 + r) * 64 + lane];

    if (col >= g.N) return;
    if (!g.bias) b = 0.f;
    const bool do_scale = (g.flags & kGemmScaleCols) && (g.scale_period ? col % g.scale_period : col) < g.scale_cols;
    float* kv_dst = nullptr;
    int kv_col = 0;
    if (g.kcache && col >= g.kv_d) {
        kv_dst = col < 2 * g.kv_d ? g.kcache : g.vcache;
        kv_col = col < 2 * g.kv_d ? col - g.kv_d : col - 2 * g.kv_d;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (!has_res) res[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row_base + r;
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

// -------------------------------------------------------------------------------------------------
// Encoder-sized problems (M ~ 1500 rows, N = d .. 12d): the "one tile per compute unit" kernel.
//
// What the 64x64 kernel above loses on these shapes is not arithmetic but quantisation and fixed cost: 1500 x 512
// is 192 workgroups of 64x64 on 256 compute units, 1500 x 1536 is 2.25 rounds, every launch pays its prologue and
// epilogue at one wave per SIMD.  Here the tile is chosen so that the whole problem is ONE round of 256 workgroups
// (1500 rows = 16 tiles of 96; N = 512 / 1536 / 2048 -> 32 / 96 / 128 columns = 16 tiles; 6144 -> three full rounds),
// and the four waves of a workgroup do not split the tile but K: every wave multiplies the WHOLE (32 TM) x (32 TN) tile
// over its own quarter of each 64-deep K slab (TM x TN independent accumulator tiles: 192 accumulator registers for
// 96 x 128), and the four partial tiles are added in wave order through LDS at the end (fixed order: run-to-run
// identical; the order does not depend on the batch, so a session stacked with others gets its solo arithmetic).
// Per MFMA this needs (TM + TN) / (TM TN) LDS fragment reads instead of 2, and per flop the fewest bytes from L2.
//
// Slabs arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass), one 1 KiB piece per wave
// instruction = 8 rows x one 128-byte line, issued for slab t+1 before the math on slab t and waited for at the one
// barrier per slab.  The DMA writes lane-linearly, so the bank swizzle lives in the SOURCE address: lane l of a piece
// fetches 16-byte chunk (l & 7) ^ ((row >> 1) & 7) of its row's line, and the fragment reads apply the same XOR - each
// 16-lane service group of a ds_read_b128 then touches 16 distinct bank quads (rows of equal parity in a group have
// distinct (row >> 1) & 7).  The k-permutation of the kernels above applies unchanged: one 16-byte read = 4 MFMA steps.
// -------------------------------------------------------------------------------------------------
// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>) - indices into register arrays
// stay constants whatever the optimiser thinks of the body's size (a runtime-indexed accumulator array goes to scratch)
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// Workgroup barrier that orders LDS traffic only: every LDS access of this wave has completed (lgkmcnt(0)), nothing is
// said about global memory.  __syncthreads() also waits for outstanding global stores (vmcnt(0)) - inside the fold that
// was ~2 us of store-acknowledge latency per round (measured with s_memtime stamps: 23k of a workgroup's 83k cycles).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Fold of the four k-partials + epilogue, shared by the k-split and the k-pipe kernel.
//
// Every wave finishes a quarter of EVERY tile - accumulator registers 4w .. 4w+3 = rows 8w .. 8w+3 (+4 for lanes 32-63):
// balanced for any tile count, and no accumulator is ever selected by a runtime index.  What shaped the rest (all
// measured with s_memtime stamps, 96 x 128 tile, 83k cycles per workgroup):
//  * the tile loop is a RUNTIME loop - unrolled over 12 tiles the erf-heavy epilogue was ~200 KB of straight-line code
//    that every workgroup pulled through the instruction cache once, at < 2 bytes per cycle: 57 us per workgroup;
//  * there is NO global load inside the loop: vmcnt retires in order, so a wait for a bias or residual value issued
//    behind the previous tile's stores is a wait for those stores to be acknowledged (~2000 cycles per tile).  Residual
//    operands and per-column constants are fetched at kernel start into registers (EpilogueOperands, in flight
//    during the main loop), parked in LDS next to the partials, and read back by runtime tile index;
//  * barriers order LDS only (lds_barrier): __syncthreads() would also wait for the stores.
// LDS use: NT x 4 KiB (residuals) + 2 KiB (column constants) + P x 16 KiB (partials of the P tiles of a pass).
template <int TM, int TN>
struct EpilogueOperands {
    float res[TM * TN][4];     // residual values of this wave's quarter of every tile (0 without kGemmResidual)
    float bias[TN], mul[TN];   // per tile column
};

template <int TM, int TN>
__device__ __forceinline__ void ksplit_fetch_epilogue(const GemmArgs& g, EpilogueOperands<TM, TN>& eo, int wave, int lane, int m0,
                                                      int n0, const float* gR) {
    const bool has_res = (g.flags & kGemmResidual) != 0, scaled = (g.flags & kGemmScaleCols) != 0;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + 32 * j + (lane & 31);
        eo.bias[j] = g.bias && col < g.N ? g.bias[col] : 0.f;
        eo.mul[j] = scaled && (g.scale_period ? col % g.scale_period : col) < g.scale_cols ? g.scale : 1.0f;
    }
#pragma unroll
    for (int q = 0; q < TM * TN; ++q) {
        const int i = q / TN, j = q % TN;
        const int col = min(n0 + 32 * j + (lane & 31), g.N - 1);
        const int row0 = m0 + 32 * i + 8 * wave + 4 * (lane >> 5);
#pragma unroll
        for (int e = 0; e < 4; ++e) eo.res[q][e] = has_res ? gR[(long)min(row0 + e, g.M - 1) * g.ldr + col] : 0.f;
    }
}

template <int TM, int TN, int LDS_FLOATS>
__device__ __forceinline__ void ksplit_fold_store(const GemmArgs& g, f32x16 (&acc)[TM][TN], const EpilogueOperands<TM, TN>& eo,
                                                  float* lds, int wave, int lane, int m0, int n0, float* gC) {
    constexpr int NT = TM * TN;
    constexpr int RES_FLOATS = NT * 1024, COL_FLOATS = 512;
    constexpr int P_FIT = (LDS_FLOATS - RES_FLOATS - COL_FLOATS) / 4096;
    static_assert(P_FIT >= 1, "no room for the fold");
    constexpr int P = P_FIT < NT ? P_FIT : NT;                     // tiles per pass (4 waves x 4 KiB each)
    constexpr int PASSES = (NT + P - 1) / P;
    float* resbuf = lds;                          // [tile][wave][lane][4]
    float* colbuf = lds + RES_FLOATS;             // [bias | mul][TN][32]   (written by wave 0's lanes 0-31)
    float* red = colbuf + COL_FLOATS;             // [tile of the pass][wave][4 x (64 lanes x 16 bytes)]
    const bool gelu = (g.flags & kGemmGelu) != 0, relu = (g.flags & kGemmRelu) != 0, swish = (g.flags & kGemmSwish) != 0;
    static_for<NT>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        *reinterpret_cast<float4*>(resbuf + (q * 4 + wave) * 256 + lane * 4) =
            make_float4(eo.res[q][0], eo.res[q][1], eo.res[q][2], eo.res[q][3]);
    });
    if (wave == 0 && lane < 32) {
        static_for<TN>([&](auto J) {
            constexpr int j = decltype(J)::value;
            colbuf[j * 32 + lane] = eo.bias[j];
            colbuf[128 + j * 32 + lane] = eo.mul[j];
        });
    }
    static_for<PASSES>([&](auto PP) {
        constexpr int pass = decltype(PP)::value, q_lo = pass * P, q_hi = (q_lo + P < NT ? q_lo + P : NT);
        if constexpr (pass > 0) lds_barrier();       // the previous pass's reads are done
        static_for<q_hi - q_lo>([&](auto T) {
            constexpr int t = decltype(T)::value, q = q_lo + t, i = q / TN, j = q % TN;
            float* dst = red + (t * 4 + wave) * 1024 + lane * 4;
            static_for<4>([&](auto V) {
                constexpr int v = decltype(V)::value;
                *reinterpret_cast<float4*>(dst + v * 256) =
                    make_float4(acc[i][j][4 * v], acc[i][j][4 * v + 1], acc[i][j][4 * v + 2], acc[i][j][4 * v + 3]);
            });
        });
        lds_barrier();
#pragma unroll 1
        for (int q = q_lo; q < q_hi; ++q) {
            const float* p = red + ((q - q_lo) * 4) * 1024 + wave * 256 + lane * 4;
            float4 s4 = *reinterpret_cast<const float4*>(p);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 t4 = *reinterpret_cast<const float4*>(p + w * 1024);
                s4.x += t4.x; s4.y += t4.y; s4.z += t4.z; s4.w += t4.w;
            }
            const float4 r4 = *reinterpret_cast<const float4*>(resbuf + (q * 4 + wave) * 256 + lane * 4);
            const int i = q / TN, j = q - i * TN;
            const float b = colbuf[j * 32 + (lane & 31)], mul = colbuf[128 + j * 32 + (lane & 31)];
            const int col = n0 + 32 * j + (lane & 31);
            const int row0 = m0 + 32 * i + 8 * wave + 4 * (lane >> 5);
            if (col < g.N) {
                const float res[4] = {r4.x, r4.y, r4.z, r4.w};
                // the four outputs side by side, one uniform branch per activation (not per element): four independent
                // dependency chains for a wave that has nobody to share its SIMD with
                float v[4] = {(s4.x + b) * mul, (s4.y + b) * mul, (s4.z + b) * mul, (s4.w + b) * mul};   // mul == 1.0f where no
                                                                                                       // scale applies: bit-neutral
                if (gelu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (swish) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + expf(-v[e]));
                }
                float* out = gC + (long)row0 * g.ldc + col;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (row0 + e < g.M) out[(long)e * g.ldc] = v[e] + res[e];
            }
        }
    });
}

// KS = slab depth (32 or 64 floats of K per trip; each of the four waves contracts KS / 4 of them)
template <int TM, int TN, int KS>
struct KSplitCfg {
    static constexpr int BM = 32 * TM, BN = 32 * TN;
    static constexpr int HALVES = KS / 32;                                        // 128-byte lines per slab row
    static constexpr int A_PIECES = BM / 8 * HALVES, W_PIECES = BN / 8 * HALVES;  // 1 KiB pieces per slab
    static constexpr int PIECES_PER_WAVE = (A_PIECES + W_PIECES) / 4;
    static constexpr int SLAB_FLOATS = (BM + BN) * KS;
    static constexpr int RED_FLOATS = TM * TN * 1024 + 512 + 2 * 4096;            // the fold: residuals + constants + >= 2 tiles per pass
    static constexpr int LDS_FLOATS = (2 * SLAB_FLOATS > RED_FLOATS ? 2 * SLAB_FLOATS : RED_FLOATS);
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);
};

// ABL (timing ablations of the probe only, results are wrong): 1 = no DMA inside the loop, 2 = no MFMA, 3 = no fragment reads
template <int TM, int TN, int KS, int ABL = 0>
__global__ __launch_bounds__(256) void gemm_nt_f32_ksplit_kernel(GemmArgs g) {
    WLK_PIN_GEMM_ARGS(g);
    using Cfg = KSplitCfg<TM, TN, KS>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NP = Cfg::PIECES_PER_WAVE, HALVES = Cfg::HALVES, STEPS = KS / 32;
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool batched = g.batch > 0;
    const float* const gA = batched ? table_at(g.z.in, blockIdx.y) : g.A;
    float* const gC = batched ? table_at(g.z.out, blockIdx.y) : g.C;
    const float* const gR = batched ? table_at(g.z.res, blockIdx.y) : g.R;
    // XCD-aware tile mapping (see gemm_nt_f32_kernel): 4 row bands x 2 column bands, one per XCD
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    int tile_m, tile_n;
    if (tiles_m >= 8) {
        const int band_m = (tiles_m + 3) / 4, band_n = (tiles_n + 1) / 2;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        tile_m = (xcd >> 1) * band_m + slot / band_n;
        tile_n = (xcd & 1) * band_n + slot % band_n;
        if (slot >= band_m * band_n || tile_m >= tiles_m || tile_n >= tiles_n) return;   // padding workgroups
    } else {
        tile_m = blockIdx.x / tiles_n;
        tile_n = blockIdx.x - tile_m * tiles_n;
        if (tile_m >= tiles_m) return;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // DMA sources of this wave's pieces (piece j = wave + 4 i: A pieces first, then W; piece = (row group of 8, line)).
    // Rows past the end of the operand are clamped: they only feed output rows / columns that are never stored.
    const float* src[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int j = wave + 4 * i;
        const bool is_a = j < Cfg::A_PIECES;
        const int jj = is_a ? j : j - Cfg::A_PIECES;
        const int row_local = (jj / HALVES) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row_local >> 1) & 7);
        const int col = (jj % HALVES) * 32 + chunk * 4;
        if (is_a) src[i] = gA + (long)min(m0 + row_local, g.M - 1) * g.lda + col;
        else src[i] = g.W + (long)min(n0 + row_local, g.N - 1) * g.K + col;
    }
    // one DMA piece of this wave: slab kt -> buffer buf
    auto issue_piece = [&](auto I, int kt, int buf) {
        constexpr int i = decltype(I)::value;
        const int j = wave + 4 * i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (long)kt * KS),
                                         (__attribute__((address_space(3))) void*)(lds + buf * Cfg::SLAB_FLOATS + j * 256), 16, 0, 0);
    };

    // fragment addresses (floats): row r of an operand sits in piece (r >> 3, line), 8 x 16-byte chunks per line; wave w
    // owns chunks [w KS / 16, (w + 1) KS / 16) of a slab row; read step s takes chunk 2 s (lanes 0-31) / 2 s + 1 (32-63)
    const int cg0 = (KS / 16) * wave + (lane >> 5);     // chunk of step 0 within the slab row; step 1 is + 2 (same line)
    const int line = cg0 >> 3, c0 = cg0 & 7;
    int a_off[TM], w_off[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int r = 32 * t + (lane & 31);
        a_off[t] = (((r >> 3) * HALVES + line) * 64 + (r & 7) * 8 + (c0 ^ ((r >> 1) & 7))) * 4;
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int r = 32 * t + (lane & 31);
        w_off[t] = Cfg::A_PIECES * 256 + (((r >> 3) * HALVES + line) * 64 + (r & 7) * 8 + (c0 ^ ((r >> 1) & 7))) * 4;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // One slab per trip, one basic block: fragment reads of slab kt (all STEPS (TM + TN) of them) FIRST - a ds_read placed
    // behind an outstanding LDS-DMA would make hipcc drain the DMA queue (it cannot tell the two buffers apart) - then
    // the MFMAs with the DMA pieces of slab kt + 1 (other buffer) dealt between the first of them, then the one barrier
    // (hipcc waits for the DMA in front of it; the sched_barrier keeps the matrix work from sinking below it, which
    // would expose the whole DMA latency).  The last trip re-fetches its own slab into the idle buffer instead of
    // branching; the barrier drains it before the fold reuses LDS.  The other buffer's last reads sit before the
    // previous trip's barrier.
    const int nslab = g.K / KS;
    EpilogueOperands<TM, TN> eo;          // residual / bias values: in flight during the main loop (see ksplit_fold_store)
    ksplit_fetch_epilogue<TM, TN>(g, eo, wave, lane, m0, n0, gR);
    static_for<NP>([&](auto I) { issue_piece(I, 0, 0); });
    __syncthreads();
    constexpr int NMF = 4 * STEPS * TM * TN;            // MFMAs per trip
    constexpr int GAP = NMF / NP >= 2 ? 2 : 1;          // one DMA piece every GAP MFMAs (a piece's ~60 issue cycles fit inside one
                                                        // MFMA's 64); front-loaded, so the rest of the trip covers the DMA latency
    for (int kt = 0; kt < nslab; ++kt) {
        const int cur = kt & 1;
        const int nxt = min(kt + 1, nslab - 1);
        const float* base = lds + cur * Cfg::SLAB_FLOATS;
        float4 fa[STEPS][TM], fb[STEPS][TN];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {      // step 1 = chunk + 2: one XOR on the float offset (bit 3)
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                if constexpr (ABL == 3) fa[s][t] = make_float4(1.f, 2.f, 3.f, (float)kt);
                else fa[s][t] = *reinterpret_cast<const float4*>(base + (a_off[t] ^ (s * 8)));
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                if constexpr (ABL == 3) fb[s][t] = make_float4(1.f, 2.f, 3.f, (float)kt);
                else fb[s][t] = *reinterpret_cast<const float4*>(base + (w_off[t] ^ (s * 8)));
            }
        }
        static_for<NMF>([&](auto X) {
            constexpr int x = decltype(X)::value;
            constexpr int s = x / (4 * TM * TN), c = (x / (TM * TN)) & 3, q = x % (TM * TN), i = q / TN, j = q % TN;
            const float a = c == 0 ? fa[s][i].x : c == 1 ? fa[s][i].y : c == 2 ? fa[s][i].z : fa[s][i].w;
            const float b = c == 0 ? fb[s][j].x : c == 1 ? fb[s][j].y : c == 2 ? fb[s][j].z : fb[s][j].w;
            if constexpr (ABL == 2) {
                asm volatile("" ::"v"(a), "v"(b));
            } else {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
            }
            if constexpr (ABL != 1 && x % GAP == GAP - 1 && x / GAP < NP) {
                issue_piece(std::integral_constant<int, x / GAP>{}, nxt, cur ^ 1);
            }
        });
        if constexpr (ABL == 0) {
            // issue order of the trip: all fragment reads, then GAP MFMAs / one DMA piece, NP times, then the remaining
            // MFMAs (left to itself hipcc bunches the pieces behind one MFMA: ~60 cycles of issue each, the pipe idles)
            __builtin_amdgcn_sched_group_barrier(0x100, STEPS * (TM + TN), 0);
            static_for<NP>([&](auto) {
                __builtin_amdgcn_sched_group_barrier(0x008, GAP, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            });
            if constexpr (NMF - NP * GAP > 0) __builtin_amdgcn_sched_group_barrier(0x008, NMF - NP * GAP, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }

    ksplit_fold_store<TM, TN, Cfg::LDS_FLOATS>(g, acc, eo, lds, wave, lane, m0, n0, gC);
}

// -------------------------------------------------------------------------------------------------
// The same kernel with the slab pipeline under manual control ("k-pipe"): 32-deep slabs in a ring of NB LDS buffers,
// the DMA of slab t + NB - 1 issued during trip t, COUNTED waits (s_waitcnt vmcnt((NB - 2) NP): only slab t + 1 has to
// have landed, the younger slabs stay in flight across the barrier), fragment reads of slab t + 1 issued in the
// middle of trip t so that their latency hides behind the second half of the trip's MFMAs.  hipcc cannot express
// this: it drains the whole DMA queue in front of every __syncthreads() and in front of any LDS read that follows an
// LDS-DMA (it cannot tell the ring's buffers apart), which is exactly the latency a one-workgroup-per-CU kernel has
// no other wave to hide behind.  So the fragment reads, both wait kinds and the barrier are inline asm / raw builtins
// here and the ordering is argued by hand:
//   RAW  slab t+1 is read (trip t, after the barrier) only after EVERY wave has waited for its own pieces of it:
//        pieces retire in issue order, at the wait the youngest (NB - 2) NP belong to slabs t+2 .. t+NB-1;
//   WAR  buffer (t + NB - 1) % NB = (t - 1) % NB is refilled from trip t on; slab t-1 was read in trip t-2 (waited for
//        with lgkmcnt(0) at that trip's end), and the barrier of trip t-1 lies between;
//   the fragments' lgkmcnt(0) takes the fragment registers as in/out operands, so no MFMA that reads them can be
//        scheduled above it.
// Trips past the end re-fetch the last slab into a dead buffer (uniform counting, no branches); the loop's exit
// drains everything before the fold reuses LDS.
// -------------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));
// NB = ring slots (32-deep slabs), STEPS = slabs per trip (= per barrier).  Thin tiles take STEPS = 2: a trip of 96 x 32 is
// only 12 MFMAs (768 cycles) per slab, and the ~180 cycles per trip in which the wave cannot feed the matrix pipe (the
// barrier, the fragment-read issue, DMA issue slots longer than an MFMA's shadow, loop control) were 19 % of its loop.
template <int TM, int TN, int NB, int STEPS>
struct KPipeCfg {
    static_assert(NB % STEPS == 0 && NB / STEPS >= 2, "the ring holds whole trips, at least two");
    static constexpr int BM = 32 * TM, BN = 32 * TN, KS = 32;
    static constexpr int A_PIECES = BM / 8, W_PIECES = BN / 8;                   // 1 KiB pieces per slab (8 rows x 128 B)
    static constexpr int PIECES_PER_WAVE = (A_PIECES + W_PIECES) / 4;
    static constexpr int SLAB_FLOATS = (BM + BN) * KS;
    static constexpr int RED_FLOATS = TM * TN * 1024 + 512 + 2 * 4096;
    static constexpr int LDS_FLOATS = (NB * SLAB_FLOATS > RED_FLOATS ? NB * SLAB_FLOATS : RED_FLOATS);
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define WLK_VM(n) else if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory");
    WLK_VM(1) WLK_VM(2) WLK_VM(3) WLK_VM(4) WLK_VM(5) WLK_VM(6) WLK_VM(7) WLK_VM(8) WLK_VM(9) WLK_VM(10) WLK_VM(11) WLK_VM(12)
    WLK_VM(13) WLK_VM(14) WLK_VM(15) WLK_VM(16) WLK_VM(17) WLK_VM(18) WLK_VM(19) WLK_VM(20) WLK_VM(21) WLK_VM(22) WLK_VM(23)
    WLK_VM(24) WLK_VM(25) WLK_VM(26) WLK_VM(27) WLK_VM(28) WLK_VM(30) WLK_VM(32) WLK_VM(33) WLK_VM(35) WLK_VM(36) WLK_VM(40)
    WLK_VM(42) WLK_VM(44) WLK_VM(48)
#undef WLK_VM
    else static_assert(N < 0, "add the vmcnt literal");
}

// s_waitcnt lgkmcnt(0) tied to the N fragment registers it makes valid (in/out operands: no MFMA that reads them can be
// scheduled above the wait)
template <int N>
__device__ __forceinline__ void wait_lgkm_tied(f32x4v* f) {
    static_assert(N >= 2 && N <= 8, "operand list");
    if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]));
    else if constexpr (N == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]));
    else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
    else if constexpr (N == 5) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]));
    else if constexpr (N == 6) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]));
    else if constexpr (N == 7)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
}

template <int TM, int TN, int NB, int STEPS>
__global__ __launch_bounds__(256) void gemm_nt_f32_kpipe_kernel(GemmArgs g) {
    WLK_PIN_GEMM_ARGS(g);
    using Cfg = KPipeCfg<TM, TN, NB, STEPS>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NP = Cfg::PIECES_PER_WAVE, KS = 32;
    constexpr int DT = NB / STEPS - 1;                  // trips of DMA in flight beyond the one being multiplied
    constexpr int NF = TM + TN, NFT = STEPS * NF;       // fragment registers (float4) per slab / per trip
    static_assert(NFT <= 8, "fragment operand list");
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const long long t_start = g.dbg_clock ? (long long)__builtin_readcyclecounter() : 0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool batched = g.batch > 0;
    const float* const gA = batched ? table_at(g.z.in, blockIdx.y) : g.A;
    float* const gC = batched ? table_at(g.z.out, blockIdx.y) : g.C;
    const float* const gR = batched ? table_at(g.z.res, blockIdx.y) : g.R;
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    int tile_m, tile_n;
    if (tiles_m >= 8) {
        const int band_m = (tiles_m + 3) / 4, band_n = (tiles_n + 1) / 2;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        tile_m = (xcd >> 1) * band_m + slot / band_n;
        tile_n = (xcd & 1) * band_n + slot % band_n;
        if (slot >= band_m * band_n || tile_m >= tiles_m || tile_n >= tiles_n) return;   // padding workgroups
    } else {
        tile_m = blockIdx.x / tiles_n;
        tile_n = blockIdx.x - tile_m * tiles_n;
        if (tile_m >= tiles_m) return;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const float* src[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int j = wave + 4 * i;
        const bool is_a = j < Cfg::A_PIECES;
        const int jj = is_a ? j : j - Cfg::A_PIECES;
        const int row_local = jj * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row_local >> 1) & 7);
        if (is_a) src[i] = gA + (long)min(m0 + row_local, g.M - 1) * g.lda + chunk * 4;
        else src[i] = g.W + (long)min(n0 + row_local, g.N - 1) * g.K + chunk * 4;
    }
    const int nslab = g.K / KS;
    auto issue_piece = [&](auto I, int slab, int slot) {     // slab (clamped) -> ring slot
        constexpr int i = decltype(I)::value;
        const int j = wave + 4 * i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (long)min(slab, nslab - 1) * KS),
                                         (__attribute__((address_space(3))) void*)(lds + slot * Cfg::SLAB_FLOATS + j * 256), 16, 0, 0);
    };

    // fragment byte addresses inside a ring slot: wave w owns chunks 2w, 2w + 1 of every row's 128-byte line
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
    const int c0 = 2 * wave + (lane >> 5);
    unsigned f_addr[NF];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int r = 32 * t + (lane & 31);
        f_addr[t] = lds_base + (unsigned)(((r >> 3) * 64 + (r & 7) * 8 + (c0 ^ ((r >> 1) & 7))) * 16);
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int r = 32 * t + (lane & 31);
        f_addr[TM + t] = lds_base + (unsigned)(Cfg::A_PIECES * 1024 + ((r >> 3) * 64 + (r & 7) * 8 + (c0 ^ ((r >> 1) & 7))) * 16);
    }
    // fragments of trip `trip` (slabs trip STEPS + s in slot (trip STEPS + s) % NB) -> f[s NF + t]
    auto read_frags = [&](f32x4v (&f)[NFT], int trip) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const unsigned off = (unsigned)((trip * STEPS + s) % NB) * (unsigned)(Cfg::SLAB_FLOATS * 4);
#pragma unroll
            for (int t = 0; t < NF; ++t) asm volatile("ds_read_b128 %0, %1" : "=v"(f[s * NF + t]) : "v"(f_addr[t] + off));
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int NMF = 4 * STEPS * TM * TN;             // MFMAs per trip
    constexpr int HEAD = NMF / 2;                        // ... in front of the barrier
    constexpr int NPT = STEPS * NP;                      // DMA pieces per wave per trip
    constexpr int GAP = HEAD / NPT >= 1 ? HEAD / NPT : 1;   // one DMA piece every GAP MFMAs of the head
    static_assert(NPT <= HEAD, "more DMA pieces than head MFMAs");

    // residual / bias values first: they are older than every DMA piece, so the counted waits below cover them, and they
    // land during the first trips (see ksplit_fold_store)
    EpilogueOperands<TM, TN> eo;
    ksplit_fetch_epilogue<TM, TN>(g, eo, wave, lane, m0, n0, gR);
    // prologue: trips 0 .. DT-1 in flight, trip 0 landed and read
    static_for<DT * STEPS>([&](auto S) {
        constexpr int sl = decltype(S)::value;
        static_for<NP>([&](auto I) { issue_piece(I, sl, sl % NB); });
    });
    wait_vmcnt<(DT - 1) * NPT>();
    __builtin_amdgcn_s_barrier();
    f32x4v f0[NFT], f1[NFT];
    read_frags(f0, 0);
    wait_lgkm_tied<NFT>(f0);

    // one trip: MFMAs of trip `tt` from f; DMA of trip tt + DT; reads of trip tt + 1 into fn
    auto trip = [&](f32x4v (&f)[NFT], f32x4v (&fn)[NFT], int tt) {
        auto mfma_at = [&](auto X) {
            constexpr int x = decltype(X)::value;
            constexpr int s = x / (4 * TM * TN), c = (x / (TM * TN)) & 3, q = x % (TM * TN), i = q / TN, j = q % TN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[s * NF + i][c], f[s * NF + TM + j][c], acc[i][j], 0, 0, 0);
        };
        static_for<HEAD>([&](auto X) {
            constexpr int x = decltype(X)::value;
            mfma_at(X);
            if constexpr (x % GAP == GAP - 1 && x / GAP < NPT) {
                constexpr int pc = x / GAP, st = pc / NP;
                const int slab = (tt + DT) * STEPS + st;
                issue_piece(std::integral_constant<int, pc % NP>{}, slab, slab % NB);
            }
        });
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt<(DT - 1) * NPT>();
        __builtin_amdgcn_s_barrier();
        read_frags(fn, tt + 1);
        __builtin_amdgcn_sched_barrier(0);
        static_for<NMF - HEAD>([&](auto X) { mfma_at(std::integral_constant<int, HEAD + decltype(X)::value>{}); });
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkm_tied<NFT>(fn);
    };
    const long long t_loop = g.dbg_clock ? (long long)__builtin_readcyclecounter() : 0;
    const int ntrip = nslab / STEPS;                     // launch_gemm only sends K % 64 == 0 here
    int tt = 0;
    for (; tt + 1 < ntrip; tt += 2) {
        trip(f0, f1, tt);
        trip(f1, f0, tt + 1);
    }
    if (tt < ntrip) trip(f0, f1, tt);
    wait_vmcnt<0>();
    __syncthreads();
    const long long t_fold = g.dbg_clock ? (long long)__builtin_readcyclecounter() : 0;

    ksplit_fold_store<TM, TN, Cfg::LDS_FLOATS>(g, acc, eo, lds, wave, lane, m0, n0, gC);
    if (g.dbg_clock && threadIdx.x == 0) {
        long long* d = g.dbg_clock + 4 * ((long)blockIdx.y * gridDim.x + blockIdx.x);
        d[0] = t_start; d[1] = t_loop; d[2] = t_fold; d[3] = (long long)__builtin_readcyclecounter();
    }
}

// (Two variants of this kernel were measured in round 3 and removed again - profiles/r03_gemm_kpipe_variants.txt: TWO slabs per
// trip / barrier (ring of 6) and DMA issued by four dedicated LOADER waves instead of the compute waves.  Neither moves
// the 96 x 32 loop off 944 cycles per slab (768 of MFMA): that tile needs 32 KB per slab per CU = 8.8 TB/s chip-wide out
// of L2 at the rate it runs, i.e. the thin tile sits on the L2 -> LDS bandwidth, not on issue slots or barriers.)
template <int TM, int TN, int NB, int STEPS>
static void launch_kpipe(const LaunchCtx& ctx, const GemmArgs& g) {
    using Cfg = KPipeCfg<TM, TN, NB, STEPS>;
    static_assert(Cfg::LDS_BYTES <= 160 * 1024, "ring does not fit the 160 KiB LDS");
    static std::atomic<uint64_t> configured{0};
    int dev = 0;
    WLK_HIP(hipGetDevice(&dev));
    if (!(configured.load(std::memory_order_acquire) >> (dev & 63) & 1)) {
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_f32_kpipe_kernel<TM, TN, NB, STEPS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES));
        configured.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const int tiles_m = (g.M + Cfg::BM - 1) / Cfg::BM, tiles_n = (g.N + Cfg::BN - 1) / Cfg::BN;
    const int blocks = tiles_m >= 8 ? 8 * ((tiles_m + 3) / 4) * ((tiles_n + 1) / 2) : tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm_nt_f32_kpipe_kernel<TM, TN, NB, STEPS>), dim3(blocks, std::max(g.batch, 1)), dim3(256), Cfg::LDS_BYTES,
                       ctx.stream, g);
}

// ks code of a k-pipe configuration: 100 + ring slots (103 / 104: a ring of 3 / 4 slabs, one slab per trip)
static bool dispatch_kpipe(const LaunchCtx& ctx, const GemmArgs& g, int tm, int tn, int ks) {
#define WLK_KP(a, b, nb, st) if (tm == a && tn == b) { launch_kpipe<a, b, nb, st>(ctx, g); return true; }
    if (ks == 104) {
        WLK_KP(3, 4, 4, 1) WLK_KP(3, 3, 4, 1) WLK_KP(3, 2, 4, 1) WLK_KP(3, 1, 4, 1) WLK_KP(2, 4, 4, 1) WLK_KP(2, 2, 4, 1) WLK_KP(2, 1, 4, 1) WLK_KP(4, 2, 4, 1)
    } else if (ks == 103) {
        WLK_KP(3, 4, 3, 1) WLK_KP(3, 3, 3, 1) WLK_KP(3, 2, 3, 1) WLK_KP(3, 1, 3, 1) WLK_KP(2, 4, 3, 1) WLK_KP(2, 2, 3, 1) WLK_KP(2, 1, 3, 1) WLK_KP(4, 2, 3, 1)
    }
#undef WLK_KP
    return false;
}

// Tile of the k-split kernel for an (M, N, K) problem, or {0, 0} when it does not apply: the shape that needs the fewest
// rounds of 256 workgroups x tile area (ties: the larger tile).  Depends on the problem only, never on the batch.
struct KSplitTile { int tm, tn, ks; };
// Which problems take the one-tile-per-CU kernels, and with which tile.  Measured on MI355X (scripts/gemm_time_probe.py,
// profiles/r03_gemm_probe_*.txt), M = 1500: they win where the output is narrow - N = 512: 12.3 / 34.6 / 28.0 us (out
// projection, fc2, conv2) against 14.8 / 45.3 / 36.5 for the 64x64 kernel; N = 768 (small): 27 / 86 / 68 against
// 32 / 110 / 85; N = 1280 (large-v3): 3-5 %; N = 1536 (base qkv, 96 x 96 tiles): 28.7 against 31.9 - and lose from
// N = 2048 on: there the 64x64 kernel's three co-resident workgroups per CU hide each other's prologue and epilogue,
// which a lone 96 x 128 workgroup cannot (its main loop runs at 88 % of the MFMA rate, but ~2.5 us of cold-start DMA
// and ~6 us of fold + erf epilogue at one wave per SIMD are exposed: 40 vs 35.5 us on fc1).  The choice depends on
// (M, N, K) only - never on the batch - so a session stacked with others keeps its solo arithmetic.
// `forced` (diagnostics, force_kernel 4): any applicable problem, so that every tile instantiation stays tested.
static KSplitTile ksplit_tile(int M, int N, int K, bool forced = false) {
    static const int mode = [] {
        const char* e = getenv("WLK_GEMM");
        return e && e[0] == 'c' ? 0 : 1;      // WLK_GEMM=classic: the 64x64 kernel everywhere (A/B switch)
    }();
    if (K % 64 != 0 || K < 256 || M < 512) return {0, 0, 0};
    // K >= 512: with fewer than 16 slabs the fixed cost of a launch dominates either kernel, and the one measured case
    // (large-v3 conv1, K = 384, M = 3000) favours the 64x64 kernel (36 vs 44 us)
    if (!forced && (!mode || N > 1536 || K < 512)) return {0, 0, 0};
    static const KSplitTile cand[] = {{3, 4, 104}, {3, 3, 104}, {3, 2, 104}, {3, 1, 104}, {2, 4, 104}, {2, 2, 104}, {2, 1, 104}, {4, 2, 104}};
    KSplitTile best{0, 0, 0};
    double best_cost = 0.0;
    for (const KSplitTile& c : cand) {
        const long tiles = (long)((M + 32 * c.tm - 1) / (32 * c.tm)) * ((N + 32 * c.tn - 1) / (32 * c.tn));
        // rounds of 256 workgroups x (MFMA steps of a tile + fixed cost of a round + a charge for thin tiles' L2 traffic)
        const double cost = ((tiles + 255) / 256) * (c.tm * c.tn * (K / 8.0) + 48.0 + 2.0 * (c.tm + c.tn) * (K / 64.0));
        if (best.tm == 0 || cost < best_cost) { best = c; best_cost = cost; }
    }
    return best;
}

template <int TM, int TN, int KS, int ABL = 0>
static void launch_ksplit(const LaunchCtx& ctx, const GemmArgs& g) {
    using Cfg = KSplitCfg<TM, TN, KS>;
    static_assert(Cfg::LDS_BYTES <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    static std::atomic<uint64_t> configured{0};     // per device: more than 64 KiB of dynamic LDS has to be asked for
    int dev = 0;
    WLK_HIP(hipGetDevice(&dev));
    if (!(configured.load(std::memory_order_acquire) >> (dev & 63) & 1)) {
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_f32_ksplit_kernel<TM, TN, KS, ABL>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES));
        configured.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const int tiles_m = (g.M + Cfg::BM - 1) / Cfg::BM, tiles_n = (g.N + Cfg::BN - 1) / Cfg::BN;
    const int blocks = tiles_m >= 8 ? 8 * ((tiles_m + 3) / 4) * ((tiles_n + 1) / 2) : tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm_nt_f32_ksplit_kernel<TM, TN, KS, ABL>), dim3(blocks, std::max(g.batch, 1)), dim3(256), Cfg::LDS_BYTES,
                       ctx.stream, g);
}

template <int KS, int ABL>
static bool dispatch_ksplit(const LaunchCtx& ctx, const GemmArgs& g, int tm, int tn) {
#define WLK_KS(a, b) if (tm == a && tn == b) { launch_ksplit<a, b, KS, ABL>(ctx, g); return true; }
    WLK_KS(3, 4) WLK_KS(3, 3) WLK_KS(3, 2) WLK_KS(3, 1) WLK_KS(2, 4) WLK_KS(2, 2) WLK_KS(2, 1) WLK_KS(4, 2)
#undef WLK_KS
    return false;
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

bool gemm_takes_ksplit(int M, int N, int K) { return ksplit_tile(M, N, K).tm != 0; }

// prompt-sized row counts (decoder prefill): 16 x 16 tiles - four times the workgroups, a quarter of the chain each
static bool kwave16_enabled() {
    static const bool on = [] {
        const char* e = getenv("WLK_KWAVE16");
        return !(e && e[0] == '0');
    }();
    return on;
}

// Can launch_gemm take the LayerNorm in front of this projection (GemmArgs::ln_gamma / ln_beta, A = the un-normalised rows)?
// Only the 16 x 16 prefill kernel does, for K = a Whisper width.
static bool gemm_can_fuse_layernorm(int M, int N, int K) {
    return kwave16_enabled() && M > 8 && M <= 128 && gemm_takes_kwave(M, N, K) && ksplit_tile(M, N, K).tm == 0 &&
           (K == 384 || K == 512 || K == 768 || K == 1024 || K == 1280);
}
// ... and do the decoder's prefill chains ask for it?  Measured neutral (profiles/r04p_ab_prefill_ln_fuse.txt: 18 launches
// less per prefill, but 128-512 workgroups each re-derive the statistics of their sixteen rows - 200.7-201.7 fused against
// 200.5-203.5 audio-s/s, 8 streams 358-363 either way), so the separate launch stays the default; WLK_PREFILL_LN_FUSE=1
// turns the fused form on (bit-identical: tests/test_gpu_parity.py::test_prefill_gemm_fuses_the_layernorm).
static std::atomic<int> g_prefill_ln_fuse{-1};          // -1: not read yet; the switch is read ONCE (this sits on the decode path)
void x3_refresh_env_switches();                          // gemm_x3.hip: WLK_X3_PERSIST
void refresh_env_switches() {
    g_prefill_ln_fuse.store(-1, std::memory_order_relaxed);
    x3_refresh_env_switches();
}   // wlk_diag_env_refresh: the parity test flips the switch inside one process
bool gemm_fuses_layernorm(int M, int N, int K) {
    int on = g_prefill_ln_fuse.load(std::memory_order_relaxed);
    if (on < 0) {
        const char* e = getenv("WLK_PREFILL_LN_FUSE");
        on = e && e[0] == '1';
        g_prefill_ln_fuse.store(on, std::memory_order_relaxed);
    }
    return on && gemm_can_fuse_layernorm(M, N, K);
}

void launch_gemm(const LaunchCtx& ctx, const GemmArgs& g, const char* tag) {
    if (g.M <= 0 || g.N <= 0) return;
    if (g.K % 4 != 0 || g.lda % 4 != 0) throw std::invalid_argument("gemm: K and lda must be multiples of 4");
    if ((((long)g.M - 1) * g.lda + g.K) * 4 >= (1L << 31) || (long)g.N * g.K * 4 >= (1L << 31))
        throw std::invalid_argument("gemm: operand larger than 2 GiB");
    const double nb = std::max(g.batch, 1);
    KernelScope ks(ctx, tag, nb * 2.0 * g.M * g.N * g.K,
                   4.0 * (nb * (double)g.M * g.K + (double)g.N * g.K + nb * (double)g.M * g.N));
    if (g.batch > kMaxBatch) throw std::invalid_argument("gemm: batch too large");
    const bool want_kwave = g.force_kwave || g.force_kernel == 2;
    // encoder-sized problems: one tile per compute unit, K split over the waves (force_kernel 3 = the 64x64 kernel)
    const KSplitTile kt = (want_kwave || g.kcache || g.force_kernel == 3) ? KSplitTile{0, 0, 0} : ksplit_tile(g.M, g.N, g.K, g.force_kernel == 4);
    if (g.force_kernel == 4 && !kt.tm) throw std::invalid_argument("gemm: the k-split kernel does not take this shape");
    if (g.ln_gamma && !(gemm_can_fuse_layernorm(g.M, g.N, g.K) && !kt.tm && g.force_kernel == 0 && g.batch == 0))
        throw std::invalid_argument("gemm: only the 16 x 16 prefill kernel takes the LayerNorm (gemm_fuses_layernorm)");
    if (kt.tm) {
        // probe override (scripts/gemm_tile_probe.py): WLK_KSPLIT_FORCE="tm,tn,ks" (ks 64: compiler-scheduled k-split; 103 / 104: k-pipe)
        static const KSplitTile forced = [] {
            KSplitTile f{0, 0, 0};
            if (const char* e = getenv("WLK_KSPLIT_FORCE")) sscanf(e, "%d,%d,%d", &f.tm, &f.tn, &f.ks);
            return f;
        }();
        const KSplitTile use = forced.tm ? forced : kt;
        bool ok;
        if (use.ks >= 100) ok = dispatch_kpipe(ctx, g, use.tm, use.tn, use.ks);
        else ok = dispatch_ksplit<64, 0>(ctx, g, use.tm, use.tn);
        if (!ok) throw std::logic_error("gemm: k-split tile without an instantiation");
        WLK_HIP(hipGetLastError());
        return;
    }
    if (g.batch > 0 && (gemm_takes_kwave(g.M, g.N, g.K) || want_kwave || (long)((g.N + 63) / 64) * ((g.M + 63) / 64) < 64))
        throw std::invalid_argument("gemm: batched launches are only available on the tiled paths");
    const long tiles64 = (long)((g.N + 63) / 64) * ((g.M + 63) / 64);
    const int tiles_n = (g.N + 63) / 64;
    const long tiles32 = (long)((g.N + 31) / 32) * ((g.M + 31) / 32);
    if (g.kcache && !(gemm_takes_kwave(g.M, g.N, g.K) || want_kwave))
        throw std::invalid_argument("gemm: fused KV-cache append is only available on the k-wave path");
    if ((gemm_takes_kwave(g.M, g.N, g.K) && g.force_kernel != 3) || (want_kwave && g.K >= 256)) {
        if (kwave16_enabled() && g.M <= 128 && g.force_kernel != 2) {
            const long tiles16 = (long)((g.N + 15) / 16) * ((g.M + 15) / 16);
            const dim3 grid((unsigned)tiles16);
            switch (g.ln_gamma ? g.K / 64 : 0) {
                case 0: hipLaunchKernelGGL(gemm_nt_f32_kwave16_kernel<0>, grid, dim3(256), 0, ctx.stream, g); break;
                case 6: hipLaunchKernelGGL(gemm_nt_f32_kwave16_kernel<6>, grid, dim3(256), 0, ctx.stream, g); break;
                case 8: hipLaunchKernelGGL(gemm_nt_f32_kwave16_kernel<8>, grid, dim3(256), 0, ctx.stream, g); break;
                case 12: hipLaunchKernelGGL(gemm_nt_f32_kwave16_kernel<12>, grid, dim3(256), 0, ctx.stream, g); break;
                case 16: hipLaunchKernelGGL(gemm_nt_f32_kwave16_kernel<16>, grid, dim3(256), 0, ctx.stream, g); break;
                case 20: hipLaunchKernelGGL(gemm_nt_f32_kwave16_kernel<20>, grid, dim3(256), 0, ctx.stream, g); break;
                default: throw std::invalid_argument("gemm: fused LayerNorm needs K = 384 / 512 / 768 / 1024 / 1280");
            }
        } else {
            hipLaunchKernelGGL(gemm_nt_f32_kwave_kernel<false>, dim3((unsigned)tiles32), dim3(256), 0, ctx.stream, g);
        }
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

// The "kp" family (round 6): kernels that share ONE per-element arithmetic - the k-pipe's - whatever their tile, so that
// the choice may depend on M without a row's result depending on it.  The streaming Sortformer runs one session's step
// (M <= 401 rows: 32 x 32 tiles, four waves split K) or several sessions' steps stacked (M up to 8 x 401: one tile per CU)
// through here, and a session's activities are bit for bit those of its step alone (tests/test_gpu_sortformer.py).
//   K % 128 == 0, K >= 256:  M >= 512 -> gemm_nt_f32_kpipe_kernel with the tile ksplit_tile picks (any N, any K of that form);
//                            else     -> gemm_nt_f32_kwave_kernel<true>;
//   K = 128 .. 255, K % 32 == 0 (the Transformer half's K = 192): k-wave tiles at EVERY row count (16 x 16 / 32 x 32 below 512
//                            rows, 32 x 32 from there on) - the second slab is half zeros, the arithmetic the family's;
//   other K:                 the plain tiled kernel - one wave per 32 x 32 tile walks K in order, the same for its 64 x 64 and
//                            32 x 64 workgroup shapes.
// the k-pipe tile of a kp-family launch (any tile gives the same bits; this is speed only)
// Measured on MI355X for the Sortformer's shapes at the row counts stacked steps produce (scripts/kp_tile_probe.py,
// profiles/r06_kp_tile_probe.txt): ksplit_tile's "rounds of 256 workgroups" model is right for M = 1500 and up to 25 % off
// elsewhere (thin tiles run two or three workgroups per CU), so the shapes that matter carry their measured best tile.
// Measured (scripts/kp_tile_probe.py, profiles/r06_kp_tile_probe.txt; us per launch, 32 x 32 / 16 x 16 tiles, M = 50 / 100 / 200 /
// 291 / 401): N 512 K 512: 7.8/4.5 7.8/4.5 7.9/5.6 8.0/6.8 8.0/7.9;  N 192 K 768: 9.3/5.3 ... 9.5/6.7;  N 512 K 2048: 17.6/9.3 17.8/9.5
// 18.1/13.3 18.5/17.5 18.7/21.3 (the 32 x 32 wave runs 256 dependent MFMAs whatever M is);  N 1536 K 512: 7.6/5.5 7.8/6.9 10.9/9.9
// 11.2/13.0 16.6/16.7;  N 2048 K 512: 8.7/6.0 8.9/8.2 12.1/12.9 18.9/17.3 22.2/21.9;  N 1024 K 512: 7.5/4.6 7.7/5.6 7.8/7.9 10.7/9.6
// 10.8/12.2.  16 x 16 tiles win while their four-fold workgroup count still fits the chip a few times over; the crossover
// moves down with the bytes each workgroup pulls (K).
static bool kp16_rule(int M, int N, int K, long tiles16) {
    if (N <= 512 && K <= 1024) return true;                 // narrow outputs: always (M < 512 here)
    if (K > 1024) return M <= 320;                          // long reductions: the 32 x 32 chain is the cost until the grid is full
    if (N == 1024) return M <= 320;
    return tiles16 <= 1300;                                 // wide outputs (q|k|v, feed-forward in)
}
static KSplitTile kp_tile(int M, int N, int K) {
    const int band = M < 900 ? 0 : M < 1800 ? 1 : M < 2800 ? 2 : 3;
    struct Row { int n, k; int t[4][2]; };
    static const Row rows[] = {
        {2048, 512, {{3, 2}, {2, 1}, {2, 2}, {2, 1}}},    // feed-forward in
        {512, 2048, {{2, 1}, {3, 1}, {3, 2}, {2, 4}}},    // feed-forward out
        {1536, 512, {{3, 2}, {2, 2}, {2, 2}, {2, 2}}},    // q | k | v
        {512, 512, {{2, 1}, {3, 1}, {3, 2}, {2, 4}}},     // attention out, pointwise conv 2
        {1024, 512, {{3, 1}, {3, 2}, {2, 1}, {2, 1}}},    // pointwise conv 1
        {192, 768, {{2, 1}, {2, 1}, {2, 1}, {3, 1}}},     // Transformer dense out
        {192, 512, {{2, 1}, {2, 1}, {2, 1}, {3, 1}}},     // encoder projection
    };
    for (const Row& r : rows)
        if (r.n == N && r.k == K) return KSplitTile{r.t[band][0], r.t[band][1], 104};
    if (N == 256 && K == 256) return M < 4000 ? KSplitTile{2, 1, 104} : M < 9000 ? KSplitTile{2, 2, 104} : KSplitTile{2, 1, 104};   // stem pointwise
    return ksplit_tile(M, N, K, true);
}
// below 512 rows: 16 x 16 tiles (four times the workgroups, a quarter of the dependent MFMA chain per wave) or 32 x 32 tiles?
// force_kernel 6 / 7 (diagnostics): 16 x 16 / 32 x 32 whatever the shape.  WLK_KP16=0 / 1 overrides the rule.
static bool kp_takes_16(int M, int N, int K, int force) {
    if (force == 6) return true;
    if (force == 7 || force == 8) return false;
    static const int env = [] { const char* e = getenv("WLK_KP16"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    if (env >= 0) return env == 1;
    const long tiles16 = (long)((N + 15) / 16) * ((M + 15) / 16);
    return kp16_rule(M, N, K, tiles16);
}
// Can launch_gemm_kp take the LayerNorm in front (GemmArgs::ln_gamma / ln_beta, A = the un-normalised rows)?  Opt-in
// (WLK_SF_LN_FUSE=1): measured a LOSS on the single-session Sortformer step (68 launches less per chunk, bit-identical, but 4.13 ms
// per chunk against 3.89: hundreds of workgroups re-deriving the statistics of their rows cost more than the 4.5 us launches they
// replace - the third time this trade was measured, after the encoder's large tiles in round 1 and the prefill in round 4).
bool gemm_kp_fuses_layernorm(int M, int N, int K) {
    static const bool on = [] { const char* e = getenv("WLK_SF_LN_FUSE"); return e && e[0] == '1'; }();
    // K = 192 (the Transformer half, round 6; opt-in WLK_SF_TF_LN_FUSE=1): sixteen rows of 192 are only 12 KB per workgroup and the
    // column-tile-0 workgroups keep the normalised rows for the block's residual (GemmArgs::ln_out) - 35 launches less per chunk,
    // bit-identical, and STILL a loss: 3.61 against 3.57 ms per chunk (profiles/r06t_tf_ln_fuse.txt).  A LayerNorm launch's 4.4 us
    // are its own chain (row load -> two dependent reductions -> store), which moves into every consumer workgroup's critical
    // path when folded; the kernel boundary it saves is the smaller part.  Needs the k-wave tiles for K = 192 (WLK_KP_SHORT_K).
    static const bool on192 = [] {
        const char* e = getenv("WLK_SF_TF_LN_FUSE");
        const char* k = getenv("WLK_KP_SHORT_K");
        return e && e[0] == '1' && !(k && k[0] == '0');
    }();
    (void)N;
    return M > 0 && M < 512 && ((on && K == 512) || (on192 && K == 192));
}
bool gemm_kp_takes_kpipe(int M, int N, int K) { return K % 128 == 0 && K >= 256 && M >= 512 && ksplit_tile(M, N, K, true).tm != 0; }
void launch_gemm_kp(const LaunchCtx& ctx, const GemmArgs& g, const char* tag) {
    if (g.M <= 0 || g.N <= 0) return;
    if (g.K % 4 != 0 || g.lda % 4 != 0) throw std::invalid_argument("gemm: K and lda must be multiples of 4");
    if (g.batch > 0 || g.kcache) throw std::invalid_argument("gemm (kp family): plain projections only");
    if (g.ln_gamma) {
        // the LayerNorm in front of the projection inside the 16 x 16 kernel (gemm_nt_f32_kwave16_kernel<NPL, true>: every workgroup
        // derives layernorm_kernel's statistics of its sixteen rows while the first slabs are in flight) - bit for bit the separate
        // launch, one kernel boundary less.  Rows < 512, K = 512 only (gemm_kp_fuses_layernorm); the stacked steps keep the launch.
        if (!gemm_kp_fuses_layernorm(g.M, g.N, g.K)) throw std::invalid_argument("gemm (kp family): this shape does not take the LayerNorm");
        KernelScope ks(ctx, tag, 2.0 * g.M * g.N * g.K, 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N));
        const long tiles16 = (long)((g.N + 15) / 16) * ((g.M + 15) / 16);
        if (g.K == 192) hipLaunchKernelGGL((gemm_nt_f32_kwave16_kernel<3, true>), dim3((unsigned)tiles16), dim3(256), 0, ctx.stream, g);
        else hipLaunchKernelGGL((gemm_nt_f32_kwave16_kernel<8, true>), dim3((unsigned)tiles16), dim3(256), 0, ctx.stream, g);
        WLK_HIP(hipGetLastError());
        return;
    }
    // K = 192 (the Sortformer's Transformer width; any multiple of 32 from 128 to 255): the k-wave tiles too, at EVERY row count -
    // the last 128-deep slab is half zeros, a row's arithmetic (wave w: k = 32 t + 8 w .. + 7, partials folded in wave order) does
    // not depend on M.  Until round 6 these shapes took the 64 x 64 family, where ONE wave walks K (7.9 us per launch for 64 MFLOP);
    // WLK_KP_SHORT_K=0 keeps that.
    static const bool short_k = [] { const char* e = getenv("WLK_KP_SHORT_K"); return !(e && e[0] == '0'); }();
    const bool short_k_kwave = short_k && g.K % 32 == 0 && g.K >= 128 && g.K < 256;
    if (!short_k_kwave && (g.K % 128 != 0 || g.K < 256)) {
        GemmArgs p = g;
        p.force_kernel = 3;
        launch_gemm(ctx, p, tag);
        return;
    }
    if ((((long)g.M - 1) * g.lda + g.K) * 4 >= (1L << 31) || (long)g.N * g.K * 4 >= (1L << 31))
        throw std::invalid_argument("gemm: operand larger than 2 GiB");
    KernelScope ks(ctx, tag, 2.0 * g.M * g.N * g.K, 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N));
    if (!short_k_kwave && gemm_kp_takes_kpipe(g.M, g.N, g.K)) {
        KSplitTile kt = kp_tile(g.M, g.N, g.K);
        if (g.force_kernel >= 500) kt = KSplitTile{(g.force_kernel - 500) / 10, (g.force_kernel - 500) % 10, 104};   // tile probe
        if (!dispatch_kpipe(ctx, g, kt.tm, kt.tn, kt.ks)) throw std::logic_error("gemm: k-pipe tile without an instantiation");
    } else if ((!short_k_kwave || g.M < 512) && kp_takes_16(g.M, g.N, g.K, g.force_kernel)) {
        const long tiles16 = (long)((g.N + 15) / 16) * ((g.M + 15) / 16);
        hipLaunchKernelGGL((gemm_nt_f32_kwave16_kernel<0, true>), dim3((unsigned)tiles16), dim3(256), 0, ctx.stream, g);
    } else {
        const long tiles32 = (long)((g.N + 31) / 32) * ((g.M + 31) / 32);
        // more than two tiles per compute unit: the one-buffer form (36.9 KB of LDS, three to four workgroups per CU) keeps them all
        // resident where the two-buffer form runs a second round - 291 x 2048 x 512: 19.0 -> 15.2 us, 401 rows 22.2 -> 19.2, q|k|v at
        // 401 rows 16.8 -> 14.0; below that the extra barrier per slab costs 0-3 % (profiles/r06q_kwave_nbuf1_probe.txt).  Same bits.
        // force_kernel 7 / 8 (diagnostics): two buffers / one buffer whatever the grid.
        const bool one_buf = g.force_kernel == 8 || (g.force_kernel != 7 && tiles32 > 512);
        if (one_buf) hipLaunchKernelGGL((gemm_nt_f32_kwave_kernel<true, 1>), dim3((unsigned)tiles32), dim3(256), 0, ctx.stream, g);
        else hipLaunchKernelGGL(gemm_nt_f32_kwave_kernel<true>, dim3((unsigned)tiles32), dim3(256), 0, ctx.stream, g);
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
    WLK_PIN_GEMM_ARGS(g);
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [MR][K]
    // side workgroups (GemmArgs::side_align): the z-score of the step's alignment window - it depends on the last
    // cross-attention, not on the logits - runs on the first workgroups of the launch, beside the weight stream
    const int side = g.side_blocks;
    if ((int)blockIdx.x < side) {
        const AlignArgs a = *g.side_align;
        const int fblock = (int)blockIdx.x % g.side_zf, rest = (int)blockIdx.x / g.side_zf;
        align_zscore_body(a, fblock, rest % a.n_align, rest / a.n_align);
        return;
    }
    const int blk = (int)blockIdx.x - side, n_blk = (int)gridDim.x - side;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int K4 = g.K >> 2;
    // the first weight chunk of this wave's first output group does not depend on the activations:
    // issue it BEFORE the x staging round trip so the two memory latencies overlap instead of adding up
    const int n_groups = (g.N + RPW - 1) / RPW;
    const int grp0 = blk * 4 + wave;
    float4 wpre[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int n = min(grp0 * RPW + r, g.N - 1);
        const bool ok = grp0 < n_groups && lane < K4;
        const float4 t = *reinterpret_cast<const float4*>(g.W + (long)(ok ? n : 0) * g.K + (ok ? lane : 0) * 4);
        wpre[r] = t;
    }
    // Round 4: the LayerNorm affine of the first 64 kLnKeep features per lane, and the epilogue operands of this wave's
    // first output group (bias, residual, per-row cache pointers), are requested here - in flight with the activations
    // and the first weights - instead of behind the reductions / behind the dot products (each was one more memory round
    // trip of a kernel that is a chain of them).  Absent operands are simply not requested (uniform branches).
    constexpr int kLnKeep = 8;
    [[maybe_unused]] float gq[kLnKeep], bq[kLnKeep];
    // (unconditional loads from pointers that are always readable - an absent operand points at the weights - because a
    // load inside a branch makes hipcc wait for ALL outstanding loads where the branch joins)
    {
        const float* gp = g.ln_gamma ? g.ln_gamma : g.W;
        const float* bp = g.ln_gamma ? g.ln_beta : g.W;
#pragma unroll
        for (int i = 0; i < kLnKeep; ++i) {
            const int c = lane + 64 * i;
            gq[i] = gp[c < g.K ? c : 0];
            bq[i] = bp[c < g.K ? c : 0];
        }
    }
    const int ep_r = lane / MR, ep_m = lane - ep_r * MR;           // epilogue role of this lane (lanes < RPW * MR)
    const int ep_n0 = min(grp0 * RPW + (ep_r < RPW ? ep_r : 0), g.N - 1);
    const int ep_mm = ep_m < g.M ? ep_m : 0;
    const float bias0 = (g.bias ? g.bias + ep_n0 : g.W)[0];
    const float res0 = ((g.flags & kGemmResidual) ? g.R + (long)ep_mm * g.ldr + ep_n0 : g.W)[0];
    static_assert(sizeof(StepRow) == 64, "the dummy StepRow read below stays inside the first weight row");
    const StepRow kv_row0 = (g.kv_rows ? g.kv_rows + ep_mm : reinterpret_cast<const StepRow*>(g.W))[0];
    const int kv_pos0 = (g.kcache ? g.kv_pos : reinterpret_cast<const int*>(g.W))[0];
    for (int i = tid; i < MR * K4; i += 256) {
        const int m = i / K4, c = i - m * K4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < g.M) v = *reinterpret_cast<const float4*>(g.A + (long)m * g.lda + c * 4);
        reinterpret_cast<float4*>(xs)[i] = v;
    }
    __syncthreads();
    if (g.ln_gamma) {
        // fused LayerNorm of the staged rows; same reduction order as layernorm_kernel (lane-strided
        // partial sums, butterfly fold), so fused and unfused paths agree bit for bit
        for (int m = wave; m < g.M; m += 4) {
            float* xr = xs + m * g.K;
            float sum = 0.f;
            for (int c = lane; c < g.K; c += 64) sum += xr[c];
            sum = wave_sum(sum);
            const float mean = sum / (float)g.K;
            float sq = 0.f;
            for (int c = lane; c < g.K; c += 64) {
                const float t = xr[c] - mean;
                sq += t * t;
            }
            sq = wave_sum(sq);
            const float rstd = 1.0f / sqrtf(sq / (float)g.K + 1e-5f);
#pragma unroll
            for (int i = 0; i < kLnKeep; ++i) {
                const int c = lane + 64 * i;
                if (c < g.K) xr[c] = (xr[c] - mean) * rstd * gq[i] + bq[i];
            }
            for (int c = lane + 64 * kLnKeep; c < g.K; c += 64) xr[c] = (xr[c] - mean) * rstd * g.ln_gamma[c] + g.ln_beta[c];
        }
        __syncthreads();
    }

    for (int grp = grp0; grp < n_groups; grp += n_blk * 4) {
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
                acc[r][m] = wave_sum(acc[r][m]);
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
                const bool first = grp == grp0;           // this group's epilogue operands were requested at the top
                if (g.bias) v += first ? bias0 : g.bias[n];
                if ((g.flags & kGemmScaleCols) && n < g.scale_cols) v *= g.scale;
                if (g.flags & kGemmGelu) v = gelu_erf(v);
                if (g.flags & kGemmRelu) v = fmaxf(v, 0.f);
                if (g.flags & kGemmSwish) v = v / (1.0f + expf(-v));
                if (g.flags & kGemmResidual) v += first ? res0 : g.R[(long)m * g.ldr + n];
                g.C[(long)m * g.ldc + n] = v;
                if (g.kv_rows && n >= g.kv_d) {          // batched steps: every row has its own cache
                    const StepRow sr = first ? kv_row0 : g.kv_rows[m];
                    const long at = g.kv_layer_off + (long)sr.offset * g.kv_d;
                    if (n < 2 * g.kv_d) sr.kcache[at + n - g.kv_d] = v;
                    else sr.vcache[at + n - 2 * g.kv_d] = v;
                } else if (g.kcache && n >= g.kv_d) {
                    const long at = ((long)m * g.kv_ctx + (first ? kv_pos0 : *g.kv_pos)) * g.kv_d;
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
//
// A decode step is ~50 of these launches back to back, each a few microseconds long, so what the kernel costs is its
// chain of dependent latencies, not its bytes.  Round 4 took the chain apart (ISA of the round-3 kernel):
//   * the by-value GemmArgs (480 bytes) was fetched by five s_load + s_waitcnt stages, one per branch that first
//     touched a field, before the first vector load was issued -> the operands the first loads need (W, A, K, N, the
//     LayerNorm affine, bias, residual) are leading scalar kernel arguments, which hipcc's kernarg preload
//     (-mllvm -amdgpu-kernarg-preload-count) delivers in SGPRs at wave start; the rest sits in one small struct that is
//     requested at the top and pinned there (WLK_PIN_S) so that the compiler cannot sink its loads into the epilogue;
//   * gamma / beta were requested AFTER the first reduction (a second memory round trip), bias / residual / the cache
//     position in the epilogue (a third) -> every load of the kernel is issued before the first wait;
//   * the LayerNorm's lane-strided statistics loads were eight predicated branches -> clamped addresses + selects;
//   * each reduction was six ds_bpermute_b32 + waits (~100 cycles each) -> VALU butterflies (wave_ops.h), same order.
// The arithmetic is untouched: bit-identical to the round-3 kernel and to the staged kernel
// (tests/test_gpu_parity.py::test_barrier_free_single_row_gemv_is_bit_identical_to_the_staged_kernel).

struct Gemv1Tail {             // needed only once the dot products are folded
    float* C;
    float* kcache;             // nullptr = no cache append
    float* vcache;
    const int* kv_pos;         // always readable (the launcher points it at the weights when there is no cache)
    int flags;
    float scale;
    int scale_cols;
    int kv_d;
};
struct Gemv1Merge {            // A row = cross-attention output still in split form (GemmArgs::mg_*)
    const float* pm;
    const float* pl;
    const float* po;
    const float* scores;
    const int* head_rank;
    const int* side_heads;     // [side_blocks] heads of this layer that are alignment heads
    float* ring;
    const int* ring_row;
    const int* beam_of_row;
    int T, ring_rows, n_beam, side_blocks;
};

template <int UB, int RPW, bool LN, bool MG, bool FULL>
__global__ __launch_bounds__(256) void gemv1_f32_kernel(const float* __restrict__ W, const float* __restrict__ A, int K, int N,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ bias, const float* R, Gemv1Tail t,
                                                        Gemv1Merge mg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K4 = K >> 2;
    if constexpr (MG) {
        if ((int)blockIdx.x >= (int)gridDim.x - mg.side_blocks) {
            // side job (cross_merge_kernel's tail): softmax row of the k-th alignment head of this layer -> alignment window
            const int head = mg.side_heads[(int)blockIdx.x - ((int)gridDim.x - mg.side_blocks)];
            const int rank = mg.head_rank[head];
            const int ring_row = mg.ring_row[0], beam = mg.beam_of_row[0];
            const long base = (long)head * kCrossSplitWays;
            float pmv[kCrossSplitWays], plv[kCrossSplitWays];
#pragma unroll
            for (int s = 0; s < kCrossSplitWays; ++s) { pmv[s] = mg.pm[base + s]; plv[s] = mg.pl[base + s]; }
            const float* srow = mg.scores + (long)head * mg.T;
            constexpr int kMaxPer = 8;                    // score values per thread held in flight (T <= 2048)
            float sv[kMaxPer];
#pragma unroll
            for (int i = 0; i < kMaxPer; ++i) {
                const int j = threadIdx.x + 256 * i;
                sv[i] = srow[j < mg.T ? j : 0];
            }
            float M = pmv[0];
#pragma unroll
            for (int s = 1; s < kCrossSplitWays; ++s) M = fmaxf(M, pmv[s]);
            float L = 0.f;
#pragma unroll
            for (int s = 0; s < kCrossSplitWays; ++s) L += plv[s] * expf(pmv[s] - M);
            float* dst = mg.ring + (((long)rank * mg.n_beam + beam) * mg.ring_rows + ring_row) * (long)mg.T;
#pragma unroll
            for (int i = 0; i < kMaxPer; ++i) {
                const int j = threadIdx.x + 256 * i;
                if (j < mg.T) dst[j] = expf(sv[i] - M) / L;
            }
            for (int j = threadIdx.x + 256 * kMaxPer; j < mg.T; j += 256) dst[j] = expf(srow[j] - M) / L;
            return;
        }
    }
    const int n_groups = (N + RPW - 1) / RPW;
    const int grp = blockIdx.x * 4 + wave;
    if (grp >= n_groups) return;
    const int n_base = grp * RPW;

    // ---- every load of the kernel, before the first wait -------------------------------------------------------------
    float4 w[UB][RPW], x[UB];
    [[maybe_unused]] float4 ga[UB], be[UB];
    [[maybe_unused]] float v[UB * 4];
    [[maybe_unused]] float pmv[UB][kCrossSplitWays], plv[UB][kCrossSplitWays];
    [[maybe_unused]] float4 pov[UB][kCrossSplitWays];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        const int c = lane + 64 * u;
        const int cc = (FULL || c < K4) ? c : 0;
#pragma unroll
        for (int r = 0; r < RPW; ++r)
            w[u][r] = *reinterpret_cast<const float4*>(W + (long)min(n_base + r, N - 1) * K + cc * 4);
        if constexpr (!MG) {
            x[u] = *reinterpret_cast<const float4*>(A + cc * 4);
        } else {   // cross_merge_kernel's operands for the head that owns dims 4c .. 4c+3
            const int head = (4 * cc) >> 6, dd = (4 * cc) & 63;
            const long base = (long)head * kCrossSplitWays;
#pragma unroll
            for (int s = 0; s < kCrossSplitWays; ++s) {
                pmv[u][s] = mg.pm[base + s];
                plv[u][s] = mg.pl[base + s];
                pov[u][s] = *reinterpret_cast<const float4*>(mg.po + (base + s) * 64 + dd);
            }
        }
        if constexpr (LN) {
            ga[u] = *reinterpret_cast<const float4*>(gamma + cc * 4);
            be[u] = *reinterpret_cast<const float4*>(beta + cc * 4);
        }
    }
    if constexpr (LN) {
#pragma unroll
        for (int i = 0; i < UB * 4; ++i) {
            const int c = lane + 64 * i;
            const float f = A[(FULL || c < K) ? c : 0];
            v[i] = (FULL || c < K) ? f : 0.f;
        }
    }
    const int n_out = min(n_base + (lane < RPW ? lane : 0), N - 1);
    const float bias_v = bias[n_out];
    const float res_v = R[n_out];
    const int kv_pos = *t.kv_pos;
    WLK_PIN_S(t.C); WLK_PIN_S(t.kcache); WLK_PIN_S(t.vcache); WLK_PIN_S(t.flags); WLK_PIN_S(t.scale);
    WLK_PIN_S(t.scale_cols); WLK_PIN_S(t.kv_d); WLK_PIN_S(kv_pos);
    // nothing below may move above this line and no load below it: hipcc otherwise rolls the long variants into a window
    // of twelve loads with counted waits (UB 20: 3.3 memory round trips per wave instead of one)
    __builtin_amdgcn_sched_barrier(0);

    // ---- arithmetic (statement for statement the round-3 kernel's) ---------------------------------------------------
    if constexpr (MG) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            float M = pmv[u][0];
#pragma unroll
            for (int s = 1; s < kCrossSplitWays; ++s) M = fmaxf(M, pmv[u][s]);
            float L = 0.f;
            float f[kCrossSplitWays];
#pragma unroll
            for (int s = 0; s < kCrossSplitWays; ++s) {
                f[s] = expf(pmv[u][s] - M);
                L += plv[u][s] * f[s];
            }
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int s = 0; s < kCrossSplitWays; ++s) {
                const float4 o = pov[u][s];
                acc.x += o.x * f[s]; acc.y += o.y * f[s]; acc.z += o.z * f[s]; acc.w += o.w * f[s];
            }
            x[u] = make_float4(acc.x / L, acc.y / L, acc.z / L, acc.w / L);
        }
    }
    if constexpr (LN) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < UB * 4; ++i) sum += v[i];
        sum = wave_sum(sum);
        const float mean = sum / (float)K;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < UB * 4; ++i) {
            const float d = (FULL || (lane + 64 * i) < K) ? v[i] - mean : 0.f;
            sq += d * d;
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
    float acc[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        if (FULL || lane + 64 * u < K4) {
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
    for (int r = 0; r < RPW; ++r) acc[r] = wave_sum(acc[r]);
    // a use outside the epilogue's branch: hipcc otherwise sinks these two loads into it (one more memory round trip)
    asm volatile("" ::"v"(bias_v), "v"(res_v));
    if (lane < RPW) {
        const int n = n_base + lane;
        if (n < N) {
            float o = 0.f;
#pragma unroll
            for (int r = 0; r < RPW; ++r)
                if (r == lane) o = acc[r];
            if (bias != W) o += bias_v;                   // the launcher passes W for an absent operand
            if ((t.flags & kGemmScaleCols) && n < t.scale_cols) o *= t.scale;
            if (t.flags & kGemmGelu) o = gelu_erf(o);
            if (t.flags & kGemmRelu) o = fmaxf(o, 0.f);
            if (t.flags & kGemmSwish) o = o / (1.0f + expf(-o));
            if (t.flags & kGemmResidual) o += res_v;
            t.C[n] = o;
            if (t.kcache && n >= t.kv_d) {
                const long at = (long)kv_pos * t.kv_d;
                if (n < 2 * t.kv_d) t.kcache[at + n - t.kv_d] = o;
                else t.vcache[at + n - 2 * t.kv_d] = o;
            }
        }
    }
}

// The merged operand at the widths where one wave per output feature cannot afford to re-derive it (d > 512: small,
// medium, large): the WORKGROUP merges the cross-attention's split partials once - thread t owns float4 chunks t and
// t + 256 of the attention output - and hands the row to its four waves through LDS, while every wave's weight row is
// already in flight.  41 KB of partials per workgroup instead of per wave, 100 VGPRs instead of 440.  Arithmetic: the
// merge is cross_merge_kernel's / gemv1_f32_kernel<MG>'s statement for statement, the dot product gemv1_f32_kernel's -
// bit-identical to the separate merge launch + plain GEMV it replaces (tests/test_gpu_parity.py::
// test_merge_folded_into_out_projection_is_bit_identical runs small / large shapes through both).
template <int UB>
__global__ __launch_bounds__(256) void gemv1_mgl_f32_kernel(const float* __restrict__ W, int K, int N,
                                                            const float* __restrict__ bias, const float* R, Gemv1Tail t,
                                                            Gemv1Merge mg) {
    __shared__ __attribute__((aligned(16))) float xs[2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K4 = K >> 2;
    if ((int)blockIdx.x >= (int)gridDim.x - mg.side_blocks) {
        // side job (cross_merge_kernel's tail): softmax row of the k-th alignment head of this layer -> alignment window
        const int head = mg.side_heads[(int)blockIdx.x - ((int)gridDim.x - mg.side_blocks)];
        const int rank = mg.head_rank[head];
        const int ring_row = mg.ring_row[0], beam = mg.beam_of_row[0];
        const long base = (long)head * kCrossSplitWays;
        float pmv[kCrossSplitWays], plv[kCrossSplitWays];
#pragma unroll
        for (int s = 0; s < kCrossSplitWays; ++s) { pmv[s] = mg.pm[base + s]; plv[s] = mg.pl[base + s]; }
        const float* srow = mg.scores + (long)head * mg.T;
        constexpr int kMaxPer = 8;
        float sv[kMaxPer];
#pragma unroll
        for (int i = 0; i < kMaxPer; ++i) {
            const int j = threadIdx.x + 256 * i;
            sv[i] = srow[j < mg.T ? j : 0];
        }
        float M = pmv[0];
#pragma unroll
        for (int s = 1; s < kCrossSplitWays; ++s) M = fmaxf(M, pmv[s]);
        float L = 0.f;
#pragma unroll
        for (int s = 0; s < kCrossSplitWays; ++s) L += plv[s] * expf(pmv[s] - M);
        float* dst = mg.ring + (((long)rank * mg.n_beam + beam) * mg.ring_rows + ring_row) * (long)mg.T;
#pragma unroll
        for (int i = 0; i < kMaxPer; ++i) {
            const int j = threadIdx.x + 256 * i;
            if (j < mg.T) dst[j] = expf(sv[i] - M) / L;
        }
        for (int j = threadIdx.x + 256 * kMaxPer; j < mg.T; j += 256) dst[j] = expf(srow[j] - M) / L;
        return;
    }
    const int n = min((int)blockIdx.x * 4 + wave, N - 1);      // (a wave past the last feature recomputes it and stores nothing)
    // ---- every load of the kernel, before the first wait -------------------------------------------------------------
    float4 w[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) w[u] = *reinterpret_cast<const float4*>(W + (long)n * K + (lane + 64 * u) * 4);   // K = 256 UB
    constexpr int NCH = (UB + 3) / 4;                          // float4 chunks of the row per thread
    float pmv[NCH][kCrossSplitWays], plv[NCH][kCrossSplitWays];
    float4 pov[NCH][kCrossSplitWays];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = (int)threadIdx.x + 256 * j;
        const int cc = c < K4 ? c : 0;
        const int head = (4 * cc) >> 6, dd = (4 * cc) & 63;
        const long base = (long)head * kCrossSplitWays;
#pragma unroll
        for (int s = 0; s < kCrossSplitWays; ++s) {
            pmv[j][s] = mg.pm[base + s];
            plv[j][s] = mg.pl[base + s];
            pov[j][s] = *reinterpret_cast<const float4*>(mg.po + (base + s) * 64 + dd);
        }
    }
    float bias_v = bias[n];
    float res_v = R[n];
    const int kv_pos = *t.kv_pos;
    WLK_PIN_S(t.C); WLK_PIN_S(t.kcache); WLK_PIN_S(t.vcache); WLK_PIN_S(t.flags); WLK_PIN_S(t.scale);
    WLK_PIN_S(t.scale_cols); WLK_PIN_S(t.kv_d); WLK_PIN_S(kv_pos);
    __builtin_amdgcn_sched_barrier(0);

    // ---- the merged row, once per workgroup ----------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        float M = pmv[j][0];
#pragma unroll
        for (int s = 1; s < kCrossSplitWays; ++s) M = fmaxf(M, pmv[j][s]);
        float L = 0.f;
        float f[kCrossSplitWays];
#pragma unroll
        for (int s = 0; s < kCrossSplitWays; ++s) {
            f[s] = expf(pmv[j][s] - M);
            L += plv[j][s] * f[s];
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s = 0; s < kCrossSplitWays; ++s) {
            const float4 o = pov[j][s];
            acc.x += o.x * f[s]; acc.y += o.y * f[s]; acc.z += o.z * f[s]; acc.w += o.w * f[s];
        }
        const int c = (int)threadIdx.x + 256 * j;
        if (c < K4) reinterpret_cast<float4*>(xs)[c] = make_float4(acc.x / L, acc.y / L, acc.z / L, acc.w / L);
    }
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        const float4 x = reinterpret_cast<const float4*>(xs)[lane + 64 * u];
        acc = fmaf(w[u].x, x.x, acc);
        acc = fmaf(w[u].y, x.y, acc);
        acc = fmaf(w[u].z, x.z, acc);
        acc = fmaf(w[u].w, x.w, acc);
    }
    acc = wave_sum(acc);
    asm volatile("" ::"v"(bias_v), "v"(res_v));
    if (lane == 0 && (int)blockIdx.x * 4 + wave < N) {
        float o = acc;
        if (bias != W) o += bias_v;
        if ((t.flags & kGemmScaleCols) && n < t.scale_cols) o *= t.scale;
        if (t.flags & kGemmGelu) o = gelu_erf(o);
        if (t.flags & kGemmRelu) o = fmaxf(o, 0.f);
        if (t.flags & kGemmSwish) o = o / (1.0f + expf(-o));
        if (t.flags & kGemmResidual) o += res_v;
        t.C[n] = o;
    }
}

static bool gemv1_enabled() {
    static const bool on = getenv("WLK_NO_GEMV1") == nullptr;
    return on;
}

// Where the merged operand pays: every WAVE re-derives its slice of the attention output from the split partials, 8 x the
// bytes of the activation row it replaces, and holds them in registers (UB 8: 442 VGPRs).  On base.en that is 8 MB per
// launch out of the L2s and a launch saved; on large-v3 (K = 1280) 52 MB and one wave per SIMD: dec_xout 20 us against
// 6.9 + 6.9 for merge kernel + plain GEMV (profiles/r04m_large_v3_folds.txt).  Default: K <= 512; WLK_MERGE_FOLD=1 folds
// wherever the kernel can, WLK_NO_MERGE_FOLD=1 never.
// From there to K = 2048 the workgroup-level form (gemv1_mgl_f32_kernel) takes over where K is a multiple of 256 (small,
// medium, large); WLK_MERGE_FOLD_LDS=0 keeps merge kernel + plain GEMV at those widths.
static bool gemv1_merge_through_lds(int K) {
    const char* e = getenv("WLK_MERGE_FOLD_LDS");
    const char* w = getenv("WLK_MERGE_FOLD");
    return K > 512 && K <= 2048 && K % 256 == 0 && !(e && e[0] == '0') && !(w && w[0] == '1');
}
bool gemv1_folds_merge(int K) {
    // read per call: test_merge_folded_into_out_projection_is_bit_identical flips the switch inside one process
    if (getenv("WLK_NO_MERGE_FOLD") != nullptr || !gemv1_enabled() || K > 2048 || K % 64 != 0) return false;
    const char* e = getenv("WLK_MERGE_FOLD");
    return (e && e[0] == '1') || K <= 512 || gemv1_merge_through_lds(K);
}

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
    // the whole K of a row in flight per wave: up to 2048 for every variant, up to 5120 (the MLP's second projection of
    // the large models, 20 float4 per lane) for the plain one-feature-per-wave variant
    const bool plain1 = rpw == 1 && !g.ln_gamma && !g.mg_pm;
    if (g.side_blocks > 0 && (!g.side_align || g.side_zf <= 0 || rpw != 4))
        throw std::invalid_argument("gemv: side workgroups ride only in the vocabulary projection's launch");
    if (g.M == 1 && (g.K <= 2048 || (plain1 && g.K <= 5120)) && rpw <= 2 && gemv1_enabled() && !g.kv_rows) {
        const int ub = (g.K / 4 + 63) / 64;
        blocks += g.mg_pm ? g.mg_side_blocks : 0;
        // absent operands point at the weights: the kernel requests every operand up front, unconditionally
        const float* bias = g.bias ? g.bias : g.W;
        const float* res = (g.flags & kGemmResidual) ? g.R : g.W;
        Gemv1Tail t{g.C, g.kcache, g.vcache, g.kcache ? g.kv_pos : reinterpret_cast<const int*>(g.W), g.flags, g.scale,
                    g.scale_cols, g.kv_d};
        Gemv1Merge mg{g.mg_pm, g.mg_pl, g.mg_po, g.mg_scores, g.mg_head_rank, g.mg_side_heads, g.mg_ring, g.mg_ring_row,
                      g.mg_beam_of_row, g.mg_T, g.mg_ring_rows, g.mg_n_beam, g.mg_side_blocks};
        if (g.mg_pm && g.mg_side_blocks > 0 && !g.mg_side_heads)
            throw std::invalid_argument("gemv: the merged operand's side blocks need the layer's alignment head list");
        if (g.mg_pm && rpw == 1 && !g.kcache && gemv1_merge_through_lds(g.K)) {
#define WLK_GEMV1_MGL(UBv)                                                                                             \
    hipLaunchKernelGGL((gemv1_mgl_f32_kernel<UBv>), dim3(blocks), dim3(256), 0, ctx.stream, g.W, g.K, g.N, bias, res, t, mg)
            switch (ub) {
                case 3: WLK_GEMV1_MGL(3); break;
                case 4: WLK_GEMV1_MGL(4); break;
                case 5: WLK_GEMV1_MGL(5); break;
                case 6: WLK_GEMV1_MGL(6); break;
                case 8: WLK_GEMV1_MGL(8); break;
                default: throw std::invalid_argument("gemv: no workgroup-merge instantiation for this width");
            }
#undef WLK_GEMV1_MGL
            WLK_HIP(hipGetLastError());
            return;
        }
#define WLK_GEMV1_F(UBv, RPWv, LNv, MGv, FULLv)                                                                         \
    hipLaunchKernelGGL((gemv1_f32_kernel<UBv, RPWv, LNv, MGv, FULLv>), dim3(blocks), dim3(256), 0, ctx.stream, g.W, g.A, \
                       g.K, g.N, g.ln_gamma, g.ln_beta, bias, res, t, mg)
    // FULL: K fills every lane of every chunk (all Whisper widths but tiny's 384) - no predicates, one basic block, every
    // load above the barrier
#define WLK_GEMV1_I(UBv, RPWv, LNv, MGv)                             \
    do {                                                             \
        if (g.K == 256 * UBv) WLK_GEMV1_F(UBv, RPWv, LNv, MGv, true); \
        else WLK_GEMV1_F(UBv, RPWv, LNv, MGv, false);                \
    } while (0)
#define WLK_GEMV1_R(UBv, LNv, MGv)                 \
    do {                                           \
        if (rpw == 2) WLK_GEMV1_I(UBv, 2, LNv, MGv); \
        else WLK_GEMV1_I(UBv, 1, LNv, MGv);        \
    } while (0)
#define WLK_GEMV1_NOMG(UBv)                                    \
    do {                                                       \
        if (g.ln_gamma) WLK_GEMV1_R(UBv, true, false);         \
        else WLK_GEMV1_R(UBv, false, false);                   \
    } while (0)
#define WLK_GEMV1(UBv)                                         \
    do {                                                       \
        if (g.mg_pm) WLK_GEMV1_R(UBv, false, true);            \
        else WLK_GEMV1_NOMG(UBv);                              \
    } while (0)
        // UB = float4 chunks per lane = registers held and loads in flight: exact for every Whisper width (K = d or 4 d,
        // d in 384 .. 1280), so that a wave of the large models does not carry dead registers into its occupancy
        // (K = 1280 as UB 8: 188 VGPRs, 512 of fc1's 640 workgroups resident -> two rounds)
        switch (ub) {
            case 1: case 2: WLK_GEMV1(2); break;
            case 3: if (g.mg_pm) WLK_GEMV1(4); else WLK_GEMV1_NOMG(3); break;
            case 4: WLK_GEMV1(4); break;
            case 5: if (g.mg_pm) WLK_GEMV1(8); else WLK_GEMV1_NOMG(5); break;
            case 6: if (g.mg_pm) WLK_GEMV1(8); else WLK_GEMV1_NOMG(6); break;
            case 7: case 8: WLK_GEMV1(8); break;
            default:
                if (ub <= 12) WLK_GEMV1_I(12, 1, false, false);
                else if (ub <= 16) WLK_GEMV1_I(16, 1, false, false);
                else WLK_GEMV1_I(20, 1, false, false);
                break;
        }
#undef WLK_GEMV1
#undef WLK_GEMV1_NOMG
#undef WLK_GEMV1_R
#undef WLK_GEMV1_I
#undef WLK_GEMV1_F
        WLK_HIP(hipGetLastError());
        return;
    }
    blocks += g.side_blocks;     // the AlignAtt z-score of a graph step rides in front of the weight stream
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

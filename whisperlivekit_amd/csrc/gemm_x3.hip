// fp32-accurate GEMM on the bf16 matrix cores:  C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias[N])  with both operands in
// the X3 format (x3.h: three bf16 planes per fp32 value, six bf16 MFMAs per fp32 product).  gfx950 only.
//
// Wide kernel (N >= 1024: the encoder's qkv / fc1 projections and the cross-attention K|V projection of all decoder
// layers): one 96 x 128 tile per workgroup - 16 x 16 = 256 workgroups on the 1500 x 2048 fc1 problem, one per CU.
// The four waves split the tile's COLUMNS (wave w owns 96 x 32: three 32 x 32 accumulators), so no cross-wave fold is
// needed; K is walked in 32-deep slabs (one slab row = 4 chunks x 3 planes x 16 bytes = 192 contiguous bytes of the X3
// row) that arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB pieces, issued by four LOADER waves) into a ring of three
// slabs, two slabs in flight beyond the one being multiplied, one workgroup barrier per slab.  The DMA
// writes lane-linearly, so the bank swizzle lives in the source address: unit u of LDS row r sits at unit u ^ ((r >> 2) & 3)
// - each 16-lane service group of a fragment ds_read_b128 then touches 16 distinct 16-byte bank groups.
// Per slab and compute wave: 36 MFMAs (2 k-steps x 3 row blocks x 6 plane products; 1152 cycles of matrix pipe), 24
// fragment reads; per loader wave 11 DMA pieces.  Algorithmic work = 2 M N K flop at fp32 accuracy; the matrix pipe executes 6x that in bf16.
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "x3.h"

namespace wlk {

typedef float xf32x16 __attribute__((ext_vector_type(16)));
typedef float xf32x4 __attribute__((ext_vector_type(4)));

// exact-erf GELU as a CALL: inlined 48 times per lane (3 row blocks x 16 accumulator registers) the erf polynomial was
// ~50 KB of straight-line code that every workgroup pulled through the instruction cache once - 6 us of the 28 us the
// 1500 x 2048 x 512 fc1 launch took (profiles/r04e_x3_ablation.txt: fixed cost 13.5 us with GELU, 7.6 us without)
__device__ __attribute__((noinline)) float x3_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int N, int I = 0, typename F>
__device__ __forceinline__ void x3_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        x3_static_for<N, I + 1>(f);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32 [rows][ld] -> X3 (weights at wlk_model_finalize; operands of the diagnostics)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void x3_pack_kernel(const float* __restrict__ src, long ld_src, unsigned short* __restrict__ dst,
                                                      long ld_dst, int rows, int cols) {
    const int chunks = cols >> 3;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)rows * chunks) return;
    const int row = (int)(idx / chunks), q = (int)(idx - (long)row * chunks);
    const float4 a = *reinterpret_cast<const float4*>(src + (long)row * ld_src + q * 8);
    const float4 b = *reinterpret_cast<const float4*>(src + (long)row * ld_src + q * 8 + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    x3_store_chunk(dst + (long)row * 3 * ld_dst + (long)q * 24, v);
}

void launch_x3_pack(const LaunchCtx& ctx, const float* src, long ld_src, unsigned short* dst, long ld_dst, int rows, int cols) {
    if (cols % 8 != 0 || ld_src % 4 != 0 || ld_dst % 8 != 0) throw std::invalid_argument("x3 pack: columns must be whole chunks of 8");
    const long n = (long)rows * (cols / 8);
    if (n <= 0) return;
    KernelScope ks(ctx, "x3_pack");
    hipLaunchKernelGGL(x3_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx.stream, src, ld_src, dst, ld_dst, rows, cols);
    WLK_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void x3_unpack_kernel(const unsigned short* __restrict__ src, long ld_src, float* __restrict__ dst,
                                                        long ld_dst, int rows, int cols) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)rows * cols) return;
    const int row = (int)(idx / cols), k = (int)(idx - (long)row * cols);
    const unsigned short* u = src + (long)row * 3 * ld_src + (long)(k >> 3) * 24 + (k & 7);
    const float h = __uint_as_float((unsigned)u[0] << 16), m = __uint_as_float((unsigned)u[8] << 16), l = __uint_as_float((unsigned)u[16] << 16);
    dst[(long)row * ld_dst + k] = (h + m) + l;
}

void launch_x3_unpack(const LaunchCtx& ctx, const unsigned short* src, long ld_src, float* dst, long ld_dst, int rows, int cols) {
    const long n = (long)rows * cols;
    if (n <= 0) return;
    KernelScope ks(ctx, "x3_unpack");
    hipLaunchKernelGGL(x3_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx.stream, src, ld_src, dst, ld_dst, rows, cols);
    WLK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------------
// the wide kernel
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int XW_BM = 96, XW_BN = 128;
constexpr int XW_ROW_BYTES = 192;                         // one slab row: 4 chunks x 3 planes x 16 bytes
constexpr int XW_ROWS = XW_BM + XW_BN;                    // A rows, then W rows
constexpr int XW_SLAB_BYTES = XW_ROWS * XW_ROW_BYTES;     // 43 008
constexpr int XW_PIECES = XW_SLAB_BYTES / 1024;           // 42 DMA pieces per slab
constexpr int XW_NPW = 11;                                // per loader wave (loaders 2 and 3 issue piece 41 once more: same bytes)
constexpr int XW_NB = 3;                                  // ring slots
constexpr int XW_NF = 12;                                 // fragments per k-step and wave: 3 x 3 A + 3 B
constexpr int XW_STAGE_SPLIT = XW_SLAB_BYTES / ((XW_BN + 4) * 4);   // rows of an X3 result's LDS transpose that fit one ring slot (81)
constexpr size_t XW_LDS_BYTES = (size_t)XW_NB * XW_SLAB_BYTES + (size_t)(XW_BM - XW_STAGE_SPLIT) * (XW_BN + 4) * 4;
// plane products of one fp32 product, small terms first: (lo, hi) (hi, lo) (mid, mid) (mid, hi) (hi, mid) (hi, hi)
constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
static_assert(XW_PIECES * 1024 == XW_SLAB_BYTES, "a slab is a whole number of 1 KiB pieces");

template <int N>
__device__ __forceinline__ void xw_wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else static_assert(N < 0, "add the vmcnt literal");
}
// s_waitcnt lgkmcnt(0) tied to the fragment registers it makes valid: no MFMA that reads them is scheduled above it
__device__ __forceinline__ void xw_wait_frags(xf32x4 (&f)[XW_NF]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]));
    asm volatile("" : "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]));
}
__device__ __forceinline__ void xw_read_frags(xf32x4 (&f)[XW_NF], const unsigned (&addr)[XW_NF], unsigned off) {
#pragma unroll
    for (int t = 0; t < XW_NF; ++t) asm volatile("ds_read_b128 %0, %1" : "=v"(f[t]) : "v"(addr[t] + off));
}
}  // namespace

// Eight waves: waves 0-3 multiply (wave w owns the tile's columns 32 w .. 32 w + 31: three 32 x 32 accumulators, no
// cross-wave fold), waves 4-7 only feed the ring.  Why loaders: a 1 KiB LDS-DMA piece costs its issuing wave 60-180
// cycles of issue time, a bf16 MFMA 32 - with the compute waves issuing their own pieces (the fp32 k-pipe kernel's
// schedule, where a piece hides behind a 64-cycle MFMA) the first version of this kernel spent two thirds of every slab
// issuing DMA instead of MFMAs (32 us on the 1500 x 2048 x 512 problem against 34.6 for the fp32 kernel).  With a loader
// wave beside each compute wave on every SIMD the two instruction streams overlap.
// One workgroup barrier per slab: the loaders arrive when their pieces of slab t + 1 have landed, the compute waves when
// they have read the last fragments of slab t; behind it slab t + 1 is readable and the slot of slab t - 1 is free.
//
// Round 5 - PERSISTENT workgroups, one slab stream across tiles.  The launch is at most one workgroup per CU (the ring
// is 126 KB, so a CU holds one anyway); a workgroup walks its XCD's tile list with stride = workgroups per XCD, and the
// slabs of all its tiles form ONE stream through the ring: when the compute waves multiply the last slab of tile i the
// loaders have slabs 0 and 1 of tile i + 1 landed / in flight and issue slab 2 behind the next barrier - the cold start
// of a tile (two slabs from L2 / the fabric) and the epilogue stores of the previous one overlap instead of adding up
// (round 4, one workgroup per tile: 81 us for the 768-tile cross-K|V projection whose loaders alone need 58 and whose
// MFMAs alone 52).  Barrier count per tile = slabs (+ 2 around the LDS transpose of an X3 result), identical on both
// sides; a workgroup whose list is empty returns before the first barrier.  A launch of <= one tile per CU is the round-4
// schedule, instruction for instruction in the loop.
// ABL (timing probe only, WLK_X3_ABL): 1 = the loaders run, the compute waves skip their MFMAs and fragment reads;
// 2 = the compute waves run, the loaders issue nothing; 3 = MFMAs only (no fragment reads, no DMA)
template <int ABL>
__global__ __launch_bounds__(512) void gemm_x3_wide_kernel(X3GemmArgs g) {
    asm volatile("" ::"s"(g.A3), "s"(g.lda), "s"(g.W3), "s"(g.bias), "s"(g.C), "s"(g.ldc), "s"(g.R), "s"(g.ldr), "s"(g.M),
                 "s"(g.N), "s"(g.K), "s"(g.flags), "s"(g.scale), "s"(g.scale_cols), "s"(g.scale_period), "s"(g.batch));
    __builtin_amdgcn_sched_barrier(0);
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool batched = g.batch > 0;
    // ---- the tile walk: slot -> (session, tile row, tile column).  Banded (tiles_m >= 8): XCD-aware as in
    // gemm_nt_f32_kernel - 4 row bands x 2 column bands, one per XCD (workgroup id & 7), slots of a band row-major, sessions
    // outermost; this workgroup takes slots first, first + stride, ... of its XCD.  Plain: slots = tiles, row-major.
    const int tiles_m = (g.M + XW_BM - 1) / XW_BM, tiles_n = (g.N + XW_BN - 1) / XW_BN;
    const bool banded = g.walk_banded != 0;
    const int band_m = banded ? (tiles_m + 3) / 4 : tiles_m, band_n = banded ? (tiles_n + 1) / 2 : tiles_n;
    const int per_session = band_m * band_n;
    const int xcd = banded ? (int)(blockIdx.x & 7) : 0;
    const int first = banded ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int stride = banded ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int limit = g.walk_slots;
    const int band_m0 = banded ? (xcd >> 1) * band_m : 0, band_n0 = banded ? (xcd & 1) * band_n : 0;
    auto decode = [&](int slot, int& tm, int& tn, int& b) -> bool {
        b = slot / per_session;
        const int r = slot - b * per_session;
        const int rm = r / band_n;
        tm = band_m0 + rm;
        tn = band_n0 + (r - rm * band_n);
        return tm < tiles_m && tn < tiles_n;
    };
    auto advance = [&](int slot) {          // first slot >= `slot` (in this workgroup's sequence) that is a tile
        int tm, tn, b;
        while (slot < limit && !decode(slot, tm, tn, b)) slot += stride;
        return slot;
    };
    const int slot0 = advance(first);
    if (slot0 >= limit) return;             // padding workgroup (both roles leave before any barrier)
    const int nslab = g.K / 32;
    const bool x3_out = g.x3_out;

    if (wave >= 4) {
        // ---- loader: piece j covers LDS bytes [1024 j, 1024 j + 1024) of a slab; lane l lands at byte 1024 j + 16 l = a
        // (row, swizzled unit) of the slab image, and fetches that row's logical unit from the X3 operand -----------------
        const int lw = wave - 4;
        const char* src[XW_NPW];
        int piece_of[XW_NPW], row_of[XW_NPW], unit_of[XW_NPW];
#pragma unroll
        for (int i = 0; i < XW_NPW; ++i) {
            const int j = min(lw + 4 * i, XW_PIECES - 1);
            piece_of[i] = j;
            const int byte = 1024 * j + 16 * lane;
            row_of[i] = byte / XW_ROW_BYTES;
            unit_of[i] = (((byte - row_of[i] * XW_ROW_BYTES) >> 4) ^ ((row_of[i] >> 2) & 3)) * 16;
        }
        auto set_src = [&](int slot) {
            int tm, tn, b;
            decode(slot, tm, tn, b);
            const int m0 = tm * XW_BM, n0 = tn * XW_BN;
            const unsigned short* const gA = batched ? reinterpret_cast<const unsigned short*>(table_at(g.z.in, (unsigned)b)) : g.A3;
#pragma unroll
            for (int i = 0; i < XW_NPW; ++i) {
                const int row = row_of[i];
                const bool is_a = row < XW_BM;
                const int grow = is_a ? min(m0 + row, g.M - 1) : min(n0 + row - XW_BM, g.N - 1);
                const char* const base = is_a ? reinterpret_cast<const char*>(gA) : reinterpret_cast<const char*>(g.W3);
                const long row_bytes = (is_a ? g.lda : (long)g.K) * 6;
                if (g.pair_probe)      // timing probe (WLK_X3_PAIRPROBE): the addresses a row-pair-interleaved operand would have
                    src[i] = base + (long)(grow >> 1) * 2 * row_bytes + (grow & 1) * XW_ROW_BYTES + unit_of[i];
                else
                    src[i] = base + (long)grow * row_bytes + unit_of[i];
            }
        };
        int issue_slot = slot0, s_next = 0, ring = 0;
        bool more = true;
        set_src(issue_slot);
        auto issue_one = [&]() {     // the next slab of the stream into the next ring slot (past the end: the last slab again, into a free slot)
            if constexpr (ABL < 2) {
                const long adv = (long)s_next * (g.pair_probe ? 2 * XW_ROW_BYTES : XW_ROW_BYTES);
                unsigned char* const dst = lds + ring * XW_SLAB_BYTES;
                x3_static_for<XW_NPW>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + adv),
                                                     (__attribute__((address_space(3))) void*)(dst + piece_of[i] * 1024), 16, 0, 0);
                });
            }
            ring = ring == XW_NB - 1 ? 0 : ring + 1;
            if (more && ++s_next == nslab) {
                const int ns = advance(issue_slot + stride);
                if (ns < limit) {
                    issue_slot = ns;
                    set_src(ns);
                    s_next = 0;
                } else {
                    more = false;
                    s_next = nslab - 1;
                }
            }
        };
        issue_one();
        issue_one();
        xw_wait_vmcnt<XW_NPW>();
        __builtin_amdgcn_s_barrier();
        for (int slot = slot0; slot < limit; slot = advance(slot + stride)) {
            for (int tt = 0; tt < nslab; ++tt) {
                issue_one();
                xw_wait_vmcnt<XW_NPW>();           // this loader's pieces of the slab behind the one being multiplied have landed
                __builtin_amdgcn_s_barrier();
            }
            if (x3_out) {                          // the compute waves transpose the tile through the ring slot just drained
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_barrier();
            }
        }
        xw_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();          // nothing of this loader is in flight any more
        return;
    }

    // ---- compute waves ------------------------------------------------------------------------------------------------
    // fragment byte addresses inside a ring slot: lane (r, hi) reads row r (+ 32 i), chunk 2 s + hi, plane p
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int r = lane & 31, hi = lane >> 5, swz = (r >> 2) & 3;
    unsigned f_addr[2][XW_NF];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned unit = (unsigned)(((2 * s + hi) * 3 + p) ^ swz) * 16u;
#pragma unroll
            for (int i = 0; i < 3; ++i) f_addr[s][i * 3 + p] = lds_base + (unsigned)((32 * i + r) * XW_ROW_BYTES) + unit;
            f_addr[s][9 + p] = lds_base + (unsigned)((XW_BM + 32 * wave + r) * XW_ROW_BYTES) + unit;
        }
    }
    xf32x16 acc[3];
    auto mfma_step = [&](const xf32x4 (&f)[XW_NF]) {     // 18 MFMAs: product t of row block i - three independent chains
        x3_static_for<18>([&](auto X) {
            constexpr int x = decltype(X)::value;
            constexpr int t = x / 3, i = x % 3;
            // the WEIGHT rows are the instruction's row operand: accumulator register q of lane (r, hi) is
            // C[m = 32 i + r][n = (q & 3) + 8 (q >> 2) + 4 hi] - four consecutive columns of ONE row per register quad, so the
            // epilogue stores 16 bytes per instruction and lane (round 5; with the activations as the row operand a lane held
            // one column of 16 rows: 48 four-byte stores per lane, and the store tail was issue-bound)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[9 + PB[t]]),
                                                             __builtin_bit_cast(bf16x8, f[i * 3 + PA[t]]), acc[i], 0, 0, 0);
        });
    };
    // this lane's sixteen columns of a tile: n0 + 32 wave + 8 j + 4 hi + (0 .. 3), j = 0 .. 3
    const int col_in_tile = 32 * wave + 4 * hi;
    auto load_bias = [&](int slot, xf32x4 (&b)[4]) {
        int tm, tn, bz;
        decode(slot, tm, tn, bz);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = min(tn * XW_BN + col_in_tile + 8 * j, g.N - 4);
            b[j] = g.bias ? *reinterpret_cast<const xf32x4*>(g.bias + c) : xf32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    xf32x4 bias_next[4];
    load_bias(slot0, bias_next);

    __builtin_amdgcn_s_barrier();              // slab 0 has landed
    unsigned ring_off = 0;                     // byte offset of the ring slot of the slab being multiplied
    xf32x4 g0[XW_NF], g1[XW_NF];
    xw_read_frags(g0, f_addr[0], ring_off);
    xw_wait_frags(g0);
    for (int slot = slot0; slot < limit;) {
        int tile_m, tile_n, bz;
        decode(slot, tile_m, tile_n, bz);
        const int next_slot = advance(slot + stride);
        const int m0 = tile_m * XW_BM, n0 = tile_n * XW_BN;
        const int col0 = n0 + col_in_tile;           // + 8 j + e
        xf32x4 bias[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bias[j] = bias_next[j];
        if (next_slot < limit) load_bias(next_slot, bias_next);       // requested a whole tile ahead of its use
        auto scaled = [&](int col) { return (g.flags & kGemmScaleCols) && (g.scale_period ? col % g.scale_period : col) < g.scale_cols; };
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
        for (int tt = 0; tt < nslab; ++tt) {
            if constexpr (ABL != 1 && ABL != 3) xw_read_frags(g1, f_addr[1], ring_off);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ABL != 1) mfma_step(g0);
            __builtin_amdgcn_sched_barrier(0);
            xw_wait_frags(g1);
            __builtin_amdgcn_s_barrier();          // the next slab of the stream is readable; everybody is done with this slab's fragments
            ring_off = ring_off == (unsigned)((XW_NB - 1) * XW_SLAB_BYTES) ? 0u : ring_off + (unsigned)XW_SLAB_BYTES;
            if constexpr (ABL != 1 && ABL != 3) xw_read_frags(g0, f_addr[0], ring_off);     // (last slab of a tile: slab 0 of the next)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ABL != 1) mfma_step(g1);
            __builtin_amdgcn_sched_barrier(0);
            xw_wait_frags(g0);
        }
        float* const gC = batched ? table_at(g.z.out, (unsigned)bz) : g.C;
        const float* const gR = batched ? table_at(g.z.res, (unsigned)bz) : g.R;
        if (x3_out) {
            // ---- result in the X3 format: the tile goes through LDS so that 8 consecutive columns - or, for the
            // transposed part, 8 rows in the attention kernel's key order - meet in one thread, which splits them into the
            // three planes and writes the chunk's 48 contiguous bytes.  Staging = the ring slot of the slab just multiplied
            // (the loaders issue nothing into it before the second barrier below; the other two slots hold the next tile's
            // first slabs) for rows 0 .. 80, plus a small tail region behind the ring for rows 81 .. 95
            constexpr int PITCH = XW_BN + 4;
            const unsigned prev_off = ring_off == 0u ? (unsigned)((XW_NB - 1) * XW_SLAB_BYTES) : ring_off - (unsigned)XW_SLAB_BYTES;
            float* const stage_lo = reinterpret_cast<float*>(lds + prev_off);
            float* const stage_hi = reinterpret_cast<float*>(lds + XW_NB * XW_SLAB_BYTES) - XW_STAGE_SPLIT * PITCH;
            auto stage_row = [&](int row) -> float* { return (row < XW_STAGE_SPLIT ? stage_lo : stage_hi) + row * PITCH; };
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float* const srow = stage_row(32 * i + r) + col_in_tile;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xf32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][4 * j + e] + bias[j][e];
                        if (scaled(col0 + 8 * j + e)) v[e] *= g.scale;
                    }
                    *reinterpret_cast<xf32x4*>(srow + 8 * j) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            unsigned short* const c3 = batched ? reinterpret_cast<unsigned short*>(gC) : g.C3;
            const int tid = threadIdx.x;           // 0 .. 255: the compute waves
            if (n0 < g.vt_col0) {                  // (tiles do not straddle vt_col0: it is a multiple of the tile width)
                for (int item = tid; item < XW_BM * (XW_BN / 8); item += 256) {
                    const int row = item >> 4, c = item & 15;
                    if (m0 + row < g.M && n0 + 8 * c < g.N) {
                        const float4 a = *reinterpret_cast<const float4*>(stage_row(row) + 8 * c);
                        const float4 b = *reinterpret_cast<const float4*>(stage_row(row) + 8 * c + 4);
                        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                        x3_store_chunk(c3 + (long)(m0 + row) * 3 * g.ldc3 + (long)((n0 >> 3) + c) * 24, v);
                    }
                }
            } else {
                unsigned short* const vt = c3 + g.vt_off;
                for (int item = tid; item < XW_BN * (XW_BM / 8); item += 256) {
                    const int dcol = item / (XW_BM / 8), u = item - dcol * (XW_BM / 8);
                    // stored chunk u of the tile's 96 rows = 32-row group u >> 2, chunk u & 3: rows 4 (u & 3) .. + 3 and 16 more
                    const int r0 = 32 * (u >> 2) + 4 * (u & 3);
                    if (n0 + dcol < g.N && m0 + 32 * (u >> 2) < g.vt_ld) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int row = r0 + (e & 3) + 16 * (e >> 2);
                            v[e] = m0 + row < g.M ? stage_row(row)[dcol] : 0.f;
                        }
                        x3_store_chunk(vt + (long)(n0 + dcol - g.vt_col0) * 3 * g.vt_ld + (long)((m0 >> 3) + u) * 24, v);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // the staging slot is the loaders' again
        } else {
            // epilogue: acc[i][4 j + e] is C[m0 + 32 i + r][col0 + 8 j + e]: 16-byte stores (and residual loads)
            const bool gelu = (g.flags & kGemmGelu) != 0, has_res = (g.flags & kGemmResidual) != 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int row = m0 + 32 * i + r;
                const bool row_ok = row < g.M;
                xf32x4 res[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    res[j] = has_res ? *reinterpret_cast<const xf32x4*>(gR + (long)min(row, g.M - 1) * g.ldr + min(col0 + 8 * j, g.N - 4))
                                     : xf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xf32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][4 * j + e] + bias[j][e];
                        if (scaled(col0 + 8 * j + e)) v[e] *= g.scale;
                        if (gelu) v[e] = x3_gelu_erf(v[e]);
                        v[e] += res[j][e];
                    }
                    if (row_ok && col0 + 8 * j < g.N) *reinterpret_cast<xf32x4*>(gC + (long)row * g.ldc + col0 + 8 * j) = v;
                }
            }
        }
        slot = next_slot;
    }
    __builtin_amdgcn_s_barrier();              // the loaders' tail fetches have landed: the workgroup may leave
}

bool gemm_x3_wide_applicable(int M, int N, int K, long lda) {
    static const bool on = [] {
        const char* e = getenv("WLK_X3");
        return !(e && e[0] == '0');
    }();
    return on && M >= 256 && N >= 1024 && N % 4 == 0 && K >= 64 && K % 32 == 0 && lda % 8 == 0;
}

static std::atomic<int> g_x3_persist{-1};                // -1: WLK_X3_PERSIST not read yet
void x3_refresh_env_switches() { g_x3_persist.store(-1, std::memory_order_relaxed); }

void launch_gemm_x3(const LaunchCtx& ctx, const X3GemmArgs& g, const char* tag) {
    if (g.M <= 0 || g.N <= 0) return;
    if (g.K % 32 != 0 || g.lda % 8 != 0 || g.K < 64) throw std::invalid_argument("x3 gemm: K must be a multiple of 32 (>= 64), lda of 8");
    if (g.N % 4 != 0 || (!g.x3_out && g.ldc % 4 != 0) || ((g.flags & kGemmResidual) && g.ldr % 4 != 0))
        throw std::invalid_argument("x3 gemm: N, ldc and ldr must be multiples of 4 (16-byte epilogue accesses)");
    if (g.batch <= 0 && (((uintptr_t)g.bias | (g.x3_out ? 0 : (uintptr_t)g.C) | ((g.flags & kGemmResidual) ? (uintptr_t)g.R : 0)) & 15))
        throw std::invalid_argument("x3 gemm: bias, C and R must be 16-byte aligned");
    if (g.batch > 0 && ((uintptr_t)g.bias & 15)) throw std::invalid_argument("x3 gemm: bias must be 16-byte aligned");
    if (g.flags & ~(kGemmGelu | kGemmResidual | kGemmScaleCols)) throw std::invalid_argument("x3 gemm: unsupported epilogue flag");
    if (g.x3_out && ((g.flags & (kGemmGelu | kGemmResidual)) || g.vt_col0 % XW_BN != 0 || g.ldc3 % 8 != 0 || g.vt_ld % 32 != 0 ||
                     g.vt_ld < g.M))
        throw std::invalid_argument("x3 gemm: unsupported X3 result layout");
    static std::atomic<uint64_t> configured{0};
    int dev = 0;
    WLK_HIP(hipGetDevice(&dev));
    if (!(configured.load(std::memory_order_acquire) >> (dev & 63) & 1)) {
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_wide_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XW_LDS_BYTES));
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_wide_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XW_LDS_BYTES));
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_wide_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XW_LDS_BYTES));
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_wide_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XW_LDS_BYTES));
        configured.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const int tiles_m = (g.M + XW_BM - 1) / XW_BM, tiles_n = (g.N + XW_BN - 1) / XW_BN;
    const int batch = g.batch > 0 ? g.batch : 1;
    // WLK_X3_MAP=1 (probe): plain row-major tile order instead of the XCD bands; WLK_X3_PERSIST=0: one workgroup per
    // tile (the round-4 launch: same kernel, every list has one entry)
    static const int map_mode = [] {
        const char* e = getenv("WLK_X3_MAP");
        return e ? atoi(e) : 0;
    }();
    int persist = g_x3_persist.load(std::memory_order_relaxed);
    if (persist < 0) {
        const char* e = getenv("WLK_X3_PERSIST");
        persist = !(e && e[0] == '0');
        g_x3_persist.store(persist, std::memory_order_relaxed);
    }
    static std::atomic<int> cu_count[64];
    int cus = cu_count[dev & 63].load(std::memory_order_relaxed);
    if (cus == 0) {
        WLK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (cus < 8) cus = 8;
        cu_count[dev & 63].store(cus, std::memory_order_relaxed);
    }
    static const int pair_probe = [] {
        const char* e = getenv("WLK_X3_PAIRPROBE");
        return e ? atoi(e) : 0;
    }();
    X3GemmArgs gg = g;
    gg.pair_probe = pair_probe && g.M % 2 == 0 && g.N % 2 == 0;
    gg.walk_banded = tiles_m >= 8 && map_mode == 0;
    int blocks;
    if (gg.walk_banded) {           // per XCD: slots of its band, all sessions; at most one resident workgroup per CU of the XCD
        gg.walk_slots = ((tiles_m + 3) / 4) * ((tiles_n + 1) / 2) * batch;
        blocks = 8 * (persist ? std::min(gg.walk_slots, cus / 8) : gg.walk_slots);
    } else {
        gg.walk_slots = tiles_m * tiles_n * batch;
        blocks = persist ? std::min(gg.walk_slots, cus) : gg.walk_slots;
    }
    // algorithmic work (what the roofline fraction is computed from): 2 M N K flop, operands and result once
    KernelScope ks(ctx, tag, 2.0 * batch * (double)g.M * g.N * g.K,
                   batch * (6.0 * ((double)g.M * g.K) + 4.0 * (double)g.M * g.N) + 6.0 * (double)g.N * g.K);
    static const int abl = [] {
        const char* e = getenv("WLK_X3_ABL");
        return e ? atoi(e) : 0;
    }();
    const dim3 grid(blocks);
    if (abl == 1) hipLaunchKernelGGL(gemm_x3_wide_kernel<1>, grid, dim3(512), XW_LDS_BYTES, ctx.stream, gg);
    else if (abl == 2) hipLaunchKernelGGL(gemm_x3_wide_kernel<2>, grid, dim3(512), XW_LDS_BYTES, ctx.stream, gg);
    else if (abl == 3) hipLaunchKernelGGL(gemm_x3_wide_kernel<3>, grid, dim3(512), XW_LDS_BYTES, ctx.stream, gg);
    else hipLaunchKernelGGL(gemm_x3_wide_kernel<0>, grid, dim3(512), XW_LDS_BYTES, ctx.stream, gg);
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

// fp32-accurate GEMM on the bf16 matrix cores:  C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias[N])  with both operands as three
// bf16 planes per fp32 value (x3.h; six bf16 MFMAs per fp32 product).  gfx950 only.
//
// Wide kernel (N >= 1024: the encoder's qkv / fc1 projections and the cross-attention K|V projection of all decoder
// layers): 96 x 128 tiles - 16 x 16 = 256 on the 1500 x 2048 fc1 problem, one per CU - walked by persistent workgroups.
// Four compute waves split a tile's COLUMNS (wave w owns 96 x 32: three 32 x 32 accumulators, no cross-wave fold); K is
// walked in 32-deep slabs.  ACTIVATIONS: X3 rows (one slab row = 4 chunks x 3 planes x 16 bytes = 192 contiguous bytes)
// that arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB pieces, issued by three LOADER waves) into a ring of five slots,
// four slabs in flight beyond the one being multiplied, one workgroup barrier per slab; the DMA writes lane-linearly, so
// the bank swizzle lives in the source address (unit u of LDS row r sits at unit u ^ ((r >> 2) & 3): each 16-lane
// service group of a fragment ds_read_b128 touches 16 distinct 16-byte bank groups).  WEIGHTS: fragment-major (W3F
// below), read by the wave that multiplies them straight into the MFMA's registers.  Per slab and compute wave: 36
// MFMAs (2 k-steps x 3 row blocks x 6 plane products; 1152 cycles of matrix pipe), 18 fragment reads, 6 weight loads;
// per loader wave 6 DMA pieces.  Algorithmic work = 2 M N K flop at fp32 accuracy; the matrix pipe executes 6x that in
// bf16.  Round 5's measurements behind this shape: DESIGN.md 13, profiles/r05[c-j]_x3_*.
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include <cstdio>

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
// the weight operand of the wide kernel: FRAGMENT-MAJOR ("W3F").  A compute wave multiplies ITS 32 weight rows only, so the
// weight fragments never need to meet other waves' data in LDS - they go from global memory straight into the registers the
// MFMA reads.  For that the packed weight is laid out in the order the wave's loads want it: for every block of 32 rows,
// K-slab t (32 elements), k-step s (2) and plane p (3): 1 KiB = lane l's 16 bytes at 16 l, lane (r = l & 31, hi = l >> 5)
// holding row r's elements 32 t + 16 s + 8 hi .. + 7 of plane p - one fully coalesced global_load_dwordx4 per fragment.
// Rows are padded to a multiple of 32 with zeros; same 6 bytes per element as the row format.
// ---------------------------------------------------------------------------------------------------------------------
size_t x3_w_elems(int n, int k) { return (size_t)((n + 31) / 32 * 32) * 3 * (size_t)k; }

__global__ __launch_bounds__(256) void x3_pack_w_kernel(const float* __restrict__ src, long ld_src, unsigned short* __restrict__ dst,
                                                        int n, int k) {
    const int nslab = k >> 5;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;           // one (row block, slab, k-step, lane): the three planes
    const long total = (long)((n + 31) / 32) * nslab * 2 * 64;
    if (idx >= total) return;
    const int l = (int)(idx & 63), s2 = (int)((idx >> 6) & 1);
    const long bt = idx >> 7;
    const int t = (int)(bt % nslab), nb = (int)(bt / nslab);
    const int row = 32 * nb + (l & 31), k0 = 32 * t + 16 * s2 + 8 * (l >> 5);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = row < n ? src[(long)row * ld_src + k0 + e] : 0.f;
    unsigned h[4], m[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const X3Triple x = x3_split(v[2 * i]), y = x3_split(v[2 * i + 1]);
        h[i] = x3_bits(x.h) | (x3_bits(y.h) << 16);
        m[i] = x3_bits(x.m) | (x3_bits(y.m) << 16);
        lo[i] = x3_bits(x.l) | (x3_bits(y.l) << 16);
    }
    x3_u32x4* d = reinterpret_cast<x3_u32x4*>(reinterpret_cast<char*>(dst) + (((long)nb * nslab + t) * 2 + s2) * 3072 + 16 * l);
    d[0] = x3_u32x4{h[0], h[1], h[2], h[3]};
    d[64] = x3_u32x4{m[0], m[1], m[2], m[3]};
    d[128] = x3_u32x4{lo[0], lo[1], lo[2], lo[3]};
}

void launch_x3_pack_w(const LaunchCtx& ctx, const float* src, long ld_src, unsigned short* dst, int n, int k) {
    if (k % 32 != 0 || n <= 0) throw std::invalid_argument("x3 weight pack: K must be a multiple of 32");
    const long total = (long)((n + 31) / 32) * (k / 32) * 128;
    KernelScope ks(ctx, "x3_pack_w");
    hipLaunchKernelGGL(x3_pack_w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx.stream, src, ld_src, dst, n, k);
    WLK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------------
// the wide kernel
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int XW_BM = 96, XW_BN = 128;
constexpr int XW_ROW_BYTES = 192;                         // one slab row: 4 chunks x 3 planes x 16 bytes
constexpr int XW_SLAB_BYTES = XW_BM * XW_ROW_BYTES;       // 18 432: the ACTIVATION rows of a slab (the weights bypass LDS)
constexpr int XW_PIECES = XW_SLAB_BYTES / 1024;           // 18 DMA pieces per slab
constexpr int XW_LOADERS = 3;                             // loader waves
constexpr int XW_NPW = XW_PIECES / XW_LOADERS;            // 6 pieces per loader wave and slab
constexpr int XW_NB = 5;                                  // ring slots
constexpr int XW_DT = XW_NB - 1;                          // slabs in flight beyond the one being multiplied
constexpr int XW_THREADS = (4 + XW_LOADERS) * 64;
constexpr int XW_NFA = 9;                                 // activation fragments per k-step: 3 row blocks x 3 planes
constexpr int XW_WSLAB_BYTES = 6 * 1024;                  // one slab of one 32-row weight block in W3F
constexpr int XW_STAGE_PITCH = XW_BN + 4;                 // fp32 words per row of an X3 result's LDS transpose
constexpr int XW_STAGE_OFF = XW_NB * XW_SLAB_BYTES;       // ... which has its own region behind the ring
constexpr int XW_WPITCH = 36;                             // fp32 words per row of a wave's PRIVATE 96 x 32 transpose (fp32 result)
constexpr size_t XW_LDS_BYTES = (size_t)XW_STAGE_OFF + (size_t)4 * XW_BM * XW_WPITCH * 4;       // 147 456 (>= the X3 transpose's 50 688)
static_assert((size_t)4 * XW_BM * XW_WPITCH >= (size_t)XW_BM * XW_STAGE_PITCH, "the staging region holds either transpose");
// plane products of one fp32 product, small terms first: (lo, hi) (hi, lo) (mid, mid) (mid, hi) (hi, mid) (hi, hi)
constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
static_assert(XW_PIECES * 1024 == XW_SLAB_BYTES && XW_NPW * XW_LOADERS == XW_PIECES, "a slab is a whole number of 1 KiB pieces per loader");

__device__ __forceinline__ void xw_wait_landed() {      // all but the newest XW_DT - 1 slabs of this loader have landed
    static_assert((XW_DT - 1) * XW_NPW == 18, "update the vmcnt literal");
    asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
}
// s_waitcnt lgkmcnt(0) tied to the fragment registers it makes valid: no MFMA that reads them is scheduled above it
__device__ __forceinline__ void xw_wait_frags(xf32x4 (&f)[XW_NFA]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]));
    asm volatile("" : "+v"(f[6]), "+v"(f[7]), "+v"(f[8]));
}
__device__ __forceinline__ void xw_read_frags(xf32x4 (&f)[XW_NFA], const unsigned (&addr)[XW_NFA], unsigned off) {
#pragma unroll
    for (int t = 0; t < XW_NFA; ++t) asm volatile("ds_read_b128 %0, %1" : "=v"(f[t]) : "v"(addr[t] + off));
}
}  // namespace

// Seven waves: waves 0-3 multiply (wave w owns the tile's columns 32 w .. 32 w + 31: three 32 x 32 accumulators, no
// cross-wave fold), waves 4-6 only feed the ring of ACTIVATION slabs.  Why loaders: a 1 KiB LDS-DMA piece costs its issuing
// wave 60-180 cycles of issue time, a bf16 MFMA 32 - compute waves that issue their own pieces spend more time on DMA
// than on MFMAs.  Round 5: with activations AND weights going through LDS (42 pieces per slab) the four loaders of round 4
// were themselves issue-bound next to a compute wave on their SIMD (the loaders alone ran a 16-slab tile in 0.8 us per
// slab, the MFMAs alone in 0.65, together 1.3).  The weights are now read by the wave that multiplies them, straight into
// registers (W3F above: six coalesced 1 KiB loads per slab, issued one slab ahead), which leaves 18 pieces per slab for
// three loaders, 18 KB per ring slot - so five slots, four slabs in flight - and frees the LDS of 57 % of its traffic.
// One workgroup barrier per slab: the loaders arrive when their pieces of slab t + 1 have landed, the compute waves when
// they have read the last fragments of slab t; behind it slab t + 1 is readable and the slot of slab t - 1 is free.
// The WEIGHT rows are the MFMA's row operand: accumulator register q of lane (r, hi) is C[m = 32 i + r][n = (q & 3) +
// 8 (q >> 2) + 4 hi] - four consecutive columns of ONE row per register quad, so the epilogue stores 16 bytes per
// instruction and lane (with the activations as the row operand a lane held one column of 16 rows: 48 four-byte stores
// per lane; that store tail cost 3 us of a 96 x 128 tile).
//
// PERSISTENT workgroups, one slab stream across tiles.  The launch is at most one workgroup per CU; a workgroup walks its
// XCD's tile list with stride = workgroups per XCD, and the slabs of all its tiles form ONE stream through the ring (and
// through the weight registers): the cold start of a tile overlaps the epilogue stores of the previous one.  Barrier
// count per tile = slabs (+ 2 around the LDS transpose of an X3 result), identical on both sides; a workgroup whose list
// is empty returns before the first barrier.  (Measured neutral by itself - profiles/r05c_x3_persist_probe.txt - the loop
// was the loss, not the tile boundary; kept because the deeper ring makes the boundary visible.)
// ABL (timing probe only, WLK_X3_ABL): 1 = loaders and weight loads run, no MFMAs and fragment reads; 2 = no DMA;
// 3 = MFMAs only; 4 = no weight loads; 5 = weight loads from one cached address; 6 = weight loads in front of a k-step's MFMAs
// instead of between them; 7 / 8 = s_setprio 1 on the compute / loader waves (6 - 8 compute correct results)
template <int ABL>
__global__ __launch_bounds__(XW_THREADS) void gemm_x3_wide_kernel(X3GemmArgs g) {
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
        int rm, cn;
        if (g.walk_colmajor) {          // consecutive slots = the band's rows of ONE column, then the next column: the workgroups
            cn = r / band_m;            // resident at a time share weight blocks 4-fold (and a persistent workgroup keeps its
            rm = r - cn * band_m;       // activation rows from tile to tile)
        } else {
            rm = r / band_n;
            cn = r - rm * band_n;
        }
        tm = band_m0 + rm;
        tn = band_n0 + cn;
        return tm < tiles_m && tn < tiles_n;
    };
    auto advance = [&](int slot) {          // first slot >= `slot` (in this workgroup's sequence) that is a tile
        int tm, tn, b;
        while (slot < limit && !decode(slot, tm, tn, b)) slot += stride;
        return slot;
    };
    const int slot0 = advance(first);
    if (slot0 >= limit) return;             // padding workgroup (both roles leave before any barrier)
    if constexpr (ABL == 7) { if (wave < 4) __builtin_amdgcn_s_setprio(1); }      // probes (correct results): issue priority
    if constexpr (ABL == 8) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }     // of the compute / of the loader waves
    const int nslab = g.K / 32;             // even (launch_gemm_x3)
    const bool x3_out = g.x3_out;

    // ---- the fp32 epilogue, shared by ALL seven waves.  The compute waves have turned their 96 x 32 parts of the tile around
    // in LDS (below); the tile is then 48 items of 8 rows x 32 columns (part wp = k / 12, row group k % 12), item k for wave
    // k % 7: bias, scale, exact-erf GELU, residual, one store instruction = eight whole 128-byte rows.  With the loaders
    // taking their share the erf epilogue of fc1 (48 calls per lane on four waves, ~8 us of a 28 us launch) is 28 calls
    // per lane.  The bias quads are requested in front of the barrier that publishes the transposed tile.  No second
    // barrier: a part is rewritten only behind the next tile's slab barriers, which every wave reaches after its items.
    auto shared_epilogue = [&](int m0, int n0, int bz) {
        float* const gC = batched ? table_at(g.z.out, (unsigned)bz) : g.C;
        const float* const gR = batched ? table_at(g.z.res, (unsigned)bz) : g.R;
        const float* const stage = reinterpret_cast<const float*>(lds + XW_STAGE_OFF);
        const int colq = lane & 7, rsub = lane >> 3;
        constexpr int ITEMS = 4 * (XW_BM / 8);
        auto bias_of_item = [&](int k) -> xf32x4 {       // the quad of item k's columns (clamped: an item past the end reads item 47's)
            const int c = min(n0 + 32 * (min(k, ITEMS - 1) / (XW_BM / 8)) + 4 * colq, g.N - 4);
            return g.bias ? *reinterpret_cast<const xf32x4*>(g.bias + c) : xf32x4{0.f, 0.f, 0.f, 0.f};
        };
        xf32x4 bq = bias_of_item(wave);                  // requested in front of the barrier, the next item's during the current one
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (compute waves: their part of the transpose is written)
        __builtin_amdgcn_s_barrier();
        const bool gelu = (g.flags & kGemmGelu) != 0, has_res = (g.flags & kGemmResidual) != 0;
#pragma unroll 1
        for (int k = wave; k < ITEMS; k += XW_THREADS / 64) {
            const xf32x4 bq_next = bias_of_item(k + XW_THREADS / 64);
            const int wp = k / (XW_BM / 8), it = k - wp * (XW_BM / 8);
            const int row_t = 8 * it + rsub, row = m0 + row_t;
            const int col = n0 + 32 * wp + 4 * colq;
            xf32x4 v = *reinterpret_cast<const xf32x4*>(stage + (wp * XW_BM + row_t) * XW_WPITCH + 4 * colq);
            const xf32x4 res = has_res ? *reinterpret_cast<const xf32x4*>(gR + (long)min(row, g.M - 1) * g.ldr + min(col, g.N - 4))
                                       : xf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] += bq[e];
                if ((g.flags & kGemmScaleCols) && (g.scale_period ? (col + e) % g.scale_period : col + e) < g.scale_cols) v[e] *= g.scale;
                if (gelu) v[e] = x3_gelu_erf(v[e]);
                v[e] += res[e];
            }
            if (row < g.M && col < g.N) *reinterpret_cast<xf32x4*>(gC + (long)row * g.ldc + col) = v;
            bq = bq_next;
        }
    };

    if (wave >= 4) {
        // ---- loader: piece j covers LDS bytes [1024 j, 1024 j + 1024) of a slab; lane l lands at byte 1024 j + 16 l = a
        // (row, swizzled unit) of the slab image, and fetches that row's logical unit from the X3 activations ---------------
        const int lw = wave - 4;
        const char* src[XW_NPW];
        int row_of[XW_NPW], unit_of[XW_NPW];
#pragma unroll
        for (int i = 0; i < XW_NPW; ++i) {
            const int byte = 1024 * (lw + XW_LOADERS * i) + 16 * lane;
            row_of[i] = byte / XW_ROW_BYTES;
            unit_of[i] = (((byte - row_of[i] * XW_ROW_BYTES) >> 4) ^ ((row_of[i] >> 2) & 3)) * 16;
        }
        auto set_src = [&](int slot) {
            int tm, tn, b;
            decode(slot, tm, tn, b);
            const int m0 = tm * XW_BM;
            const char* const gA = reinterpret_cast<const char*>(batched ? reinterpret_cast<const unsigned short*>(table_at(g.z.in, (unsigned)b)) : g.A3);
#pragma unroll
            for (int i = 0; i < XW_NPW; ++i) src[i] = gA + (long)min(m0 + row_of[i], g.M - 1) * g.lda * 6 + unit_of[i];
        };
        int issue_slot = slot0, s_next = 0, ring = 0;
        bool more = true;
        set_src(issue_slot);
        auto issue_one = [&]() {     // the next slab of the stream into the next ring slot (past the end: the last slab again, into a free slot)
            if constexpr (ABL != 2 && ABL != 3) {
                const long adv = (long)s_next * XW_ROW_BYTES;
                unsigned char* const dst = lds + ring * XW_SLAB_BYTES + lw * 1024;
                x3_static_for<XW_NPW>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + adv),
                                                     (__attribute__((address_space(3))) void*)(dst + i * XW_LOADERS * 1024), 16, 0, 0);
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
#pragma unroll
        for (int i = 0; i < XW_DT; ++i) issue_one();
        xw_wait_landed();
        __builtin_amdgcn_s_barrier();
        for (int slot = slot0; slot < limit; slot = advance(slot + stride)) {
            for (int tt = 0; tt < nslab; ++tt) {
                issue_one();
                xw_wait_landed();                  // this loader's pieces of the slab behind the one being multiplied have landed
                __builtin_amdgcn_s_barrier();
            }
            if (x3_out) {                          // the compute waves' two barriers around the LDS transpose of the tile
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_barrier();
            } else {
                int tm, tn, bz;
                decode(slot, tm, tn, bz);
                shared_epilogue(tm * XW_BM, tn * XW_BN, bz);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // nothing of this loader is in flight any more
        return;
    }

    // ---- compute waves ------------------------------------------------------------------------------------------------
    // activation fragment byte addresses inside a ring slot: lane (r, hi) reads row r (+ 32 i), chunk 2 s + hi, plane p
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int r = lane & 31, hi = lane >> 5, swz = (r >> 2) & 3;
    unsigned f_addr[2][XW_NFA];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned unit = (unsigned)(((2 * s + hi) * 3 + p) ^ swz) * 16u;
#pragma unroll
            for (int i = 0; i < 3; ++i) f_addr[s][i * 3 + p] = lds_base + (unsigned)((32 * i + r) * XW_ROW_BYTES) + unit;
        }
    xf32x16 acc[3];
    // MFMAs [LO, HI) of the 18 of one k-step: product t = x / 3 of row block i = x % 3 - three independent chains; w0 .. w2 =
    // this k-step's three weight planes
    auto mfma_range = [&](auto LO, auto HI, const xf32x4 (&f)[XW_NFA], const xf32x4& w0, const xf32x4& w1, const xf32x4& w2) {
        constexpr int lo = decltype(LO)::value, hi_ = decltype(HI)::value;
        x3_static_for<hi_ - lo>([&](auto X) {
            constexpr int x = lo + decltype(X)::value;
            constexpr int t = x / 3, i = x % 3;
            const xf32x4& w = PB[t] == 0 ? w0 : (PB[t] == 1 ? w1 : w2);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, f[i * 3 + PA[t]]),
                                                             acc[i], 0, 0, 0);
        });
    };
    // the weight stream of this wave: block (4 tile_n + wave) of W3F, one 6 KiB slab after the other
    const int n_blocks = (g.N + 31) / 32;
    typedef const __attribute__((address_space(1))) xf32x4* wptr_t;
    auto w_base = [&](int slot) -> const char* {
        int tm, tn, bz;
        decode(slot, tm, tn, bz);
        return reinterpret_cast<const char*>(g.W3) + (long)min(4 * tn + wave, n_blocks - 1) * nslab * XW_WSLAB_BYTES + 16 * lane;
    };
    const char* const w_fixed = reinterpret_cast<const char*>(g.W3) + 16 * lane;
    // one of a slab's six weight fragments: requested BETWEEN the MFMAs of the slab before (k_step below)
    auto load_w1 = [&](xf32x4 (&w)[6], const char* p, int x) {
        if constexpr (ABL == 5) p = w_fixed;          // probe: every weight load hits the same (cached) kilobytes
        if constexpr (ABL != 3 && ABL != 4) w[x] = *(wptr_t)(p + 1024 * x);
    };
    // this lane's sixteen columns of a tile: n0 + 32 wave + 8 j + 4 hi + (0 .. 3), j = 0 .. 3
    const int col_in_tile = 32 * wave + 4 * hi;
    // ... for an X3 result (bias added before the shared transpose; an fp32 result gets it in shared_epilogue)
    auto load_bias = [&](int slot, xf32x4 (&b)[4]) {
        int tm, tn, bz;
        decode(slot, tm, tn, bz);
        if (x3_out) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = min(tn * XW_BN + col_in_tile + 8 * j, g.N - 4);
                b[j] = g.bias ? *reinterpret_cast<const xf32x4*>(g.bias + c) : xf32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    xf32x4 wa[6], wb[6];                       // weight fragments of the slab being multiplied / of the next one
#pragma unroll
    for (int x = 0; x < 6; ++x) wa[x] = wb[x] = xf32x4{0.f, 0.f, 0.f, 0.f};
    const char* w_cur = w_base(slot0);
#pragma unroll
    for (int x = 0; x < 6; ++x) load_w1(wa, w_cur, x);
    xf32x4 bias_next[4];
    load_bias(slot0, bias_next);

    __builtin_amdgcn_s_barrier();              // slab 0 has landed
    unsigned ring_off = 0;                     // byte offset of the ring slot of the slab being multiplied
    xf32x4 g0[XW_NFA], g1[XW_NFA];
    xw_read_frags(g0, f_addr[0], ring_off);
    xw_wait_frags(g0);
    // one slab: k-step 0 from g0 (already read), k-step 1 from g1; behind the barrier the next slab's first fragments.
    // The next slab's weight fragments are requested one at a time between the MFMAs (a vector-memory instruction costs its
    // wave ~100 cycles of issue: six in a row in front of the MFMAs were 0.34 us per slab, profiles/r05e, r05f)
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 6> I6;
    typedef std::integral_constant<int, 12> I12;
    typedef std::integral_constant<int, 18> I18;
    // one k-step (`half` of the slab): 18 MFMAs from fragments f and weight planes w[3 half ..], and the loads of the NEXT
    // slab's planes of the same half, one after every six MFMAs (ABL 6, probe: all three in front)
    auto k_step = [&](const xf32x4 (&f)[XW_NFA], const xf32x4 (&w)[6], xf32x4 (&nxt)[6], const char* p_next, int half) {
        const xf32x4 &w0 = w[3 * half], &w1 = w[3 * half + 1], &w2 = w[3 * half + 2];
        if constexpr (ABL == 6) {
#pragma unroll
            for (int x = 0; x < 3; ++x) load_w1(nxt, p_next, 3 * half + x);
            __builtin_amdgcn_sched_barrier(0);
            mfma_range(I0{}, I18{}, f, w0, w1, w2);
        } else if constexpr (ABL == 1) {
#pragma unroll
            for (int x = 0; x < 3; ++x) load_w1(nxt, p_next, 3 * half + x);
        } else {
            mfma_range(I0{}, I6{}, f, w0, w1, w2);
            __builtin_amdgcn_sched_barrier(0);
            load_w1(nxt, p_next, 3 * half);
            __builtin_amdgcn_sched_barrier(0);
            mfma_range(I6{}, I12{}, f, w0, w1, w2);
            __builtin_amdgcn_sched_barrier(0);
            load_w1(nxt, p_next, 3 * half + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_range(I12{}, I18{}, f, w0, w1, w2);
            __builtin_amdgcn_sched_barrier(0);
            load_w1(nxt, p_next, 3 * half + 2);
        }
    };
    auto slab = [&](const xf32x4 (&w)[6], xf32x4 (&nxt)[6], const char* p_next) {
        if constexpr (ABL != 1 && ABL != 3) xw_read_frags(g1, f_addr[1], ring_off);
        __builtin_amdgcn_sched_barrier(0);
        k_step(g0, w, nxt, p_next, 0);
        __builtin_amdgcn_sched_barrier(0);
        xw_wait_frags(g1);
        __builtin_amdgcn_s_barrier();          // the next slab of the stream is readable; everybody is done with this slab's fragments
        ring_off = ring_off == (unsigned)((XW_NB - 1) * XW_SLAB_BYTES) ? 0u : ring_off + (unsigned)XW_SLAB_BYTES;
        if constexpr (ABL != 1 && ABL != 3) xw_read_frags(g0, f_addr[0], ring_off);     // (last slab of a tile: slab 0 of the next)
        __builtin_amdgcn_sched_barrier(0);
        k_step(g1, w, nxt, p_next, 1);
        __builtin_amdgcn_sched_barrier(0);
        xw_wait_frags(g0);
    };
    for (int slot = slot0; slot < limit;) {
        int tile_m, tile_n, bz;
        decode(slot, tile_m, tile_n, bz);
        const int next_slot = advance(slot + stride);
        const int m0 = tile_m * XW_BM, n0 = tile_n * XW_BN;
        const int col0 = n0 + col_in_tile;           // + 8 j + e
        const char* const w_next = next_slot < limit ? w_base(next_slot) : w_cur;      // (no next tile: a harmless re-read)
        xf32x4 bias[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bias[j] = bias_next[j];
        if (next_slot < limit) load_bias(next_slot, bias_next);       // requested a whole tile ahead of its use
        auto scaled = [&](int col) { return (g.flags & kGemmScaleCols) && (g.scale_period ? col % g.scale_period : col) < g.scale_cols; };
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
        for (int tt = 0; tt < nslab; tt += 2) {
            slab(wa, wb, w_cur + (long)(tt + 1) * XW_WSLAB_BYTES);           // weights one slab ahead of their MFMAs
            slab(wb, wa, tt + 2 < nslab ? w_cur + (long)(tt + 2) * XW_WSLAB_BYTES : w_next);
        }
        w_cur = w_next;
        float* const gC = batched ? table_at(g.z.out, (unsigned)bz) : g.C;
        if (x3_out) {
            // ---- result in the X3 format: the tile goes through LDS (its own region behind the ring) so that 8 consecutive
            // columns - or, for the transposed part, 8 rows in the attention kernel's key order - meet in one thread, which
            // splits them into the three planes and writes the chunk's 48 contiguous bytes
            constexpr int PITCH = XW_STAGE_PITCH;
            float* const stage = reinterpret_cast<float*>(lds + XW_STAGE_OFF);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float* const srow = stage + (32 * i + r) * PITCH + col_in_tile;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xf32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][4 * j + e] + bias[j][e];
                        if (scaled(col0 + 8 * j + e)) v[e] *= g.scale;
                    }
                    if (g.flags & kGemmGelu) {       // round 6: fc1 of the d >= 1024 models hands fc2 its X3 operand directly
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = x3_gelu_erf(v[e]);
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
                        const float4 a = *reinterpret_cast<const float4*>(stage + row * PITCH + 8 * c);
                        const float4 b = *reinterpret_cast<const float4*>(stage + row * PITCH + 8 * c + 4);
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
                            v[e] = m0 + row < g.M ? stage[row * PITCH + dcol] : 0.f;
                        }
                        x3_store_chunk(vt + (long)(n0 + dcol - g.vt_col0) * 3 * g.vt_ld + (long)((m0 >> 3) + u) * 24, v);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // the staging region may be written again
        } else {
            // ---- fp32 result.  acc[i][4 j + e] is C[m0 + 32 i + r][col0 + 8 j + e]: stored from there, an instruction would
            // touch 32 rows with 32 bytes each (a quarter of a line per request).  The wave turns its 96 x 32 part around in
            // its own LDS region instead; shared_epilogue (all seven waves) then works on items of eight whole 128-byte rows
            float* const wst = reinterpret_cast<float*>(lds + XW_STAGE_OFF) + wave * (XW_BM * XW_WPITCH);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<xf32x4*>(wst + (32 * i + r) * XW_WPITCH + 8 * j + 4 * hi) =
                        xf32x4{acc[i][4 * j], acc[i][4 * j + 1], acc[i][4 * j + 2], acc[i][4 * j + 3]};
            shared_epilogue(m0, n0, bz);
        }
        slot = next_slot;
    }
    __builtin_amdgcn_s_barrier();              // the loaders' tail fetches have landed: the workgroup may leave
}

// ---------------------------------------------------------------------------------------------------------------------
// the wide kernel with TWO compute waves per SIMD (round 6).  The kernel above keeps ONE wave per SIMD on the matrix pipe, and
// whatever that wave waits for - the ~100 cycles of issue every weight load costs it, its fragment reads, the slab barrier -
// the pipe waits for too: 1.1 us per slab against 0.6 for the 36 MFMAs alone.  Here the two k-steps of a slab
// belong to two waves: wave v multiplies columns 32 (v & 3) .. + 31 of the tile with the slab's elements 16 (v >> 2) .. + 15
// (18 MFMAs, 9 fragment reads, 3 weight loads per slab - the workgroup's LDS and weight traffic is unchanged), waves v and
// v + 4 share a SIMD, and one wave's stalls are the other one's issue slots.  Eight compute waves + three loaders = eleven
// waves, three on a SIMD, i.e. 168 registers each: the fragments of planes mid and lo are read into the registers the MFMAs
// have just consumed (lo after the first three MFMAs of a slab, mid after twelve), only plane hi - used by the last six - is
// double-buffered, which needs slab t + 1 readable while slab t is multiplied: the slab barrier sits in FRONT of a slab's
// MFMAs (the loaders' protocol is the one above, one slot less of look-ahead).  At the end of a tile the k-step-1 wave hands
// its accumulators to its partner through the staging region (lane-private addresses: the same lane of the partner wave
// owns the same elements), one more barrier per tile; the sum of the two halves then takes the epilogues above.  Not
// bit-identical to the one-wave kernel (a tile's K sum is regrouped into the even and the odd k-steps); rows still do not
// depend on what is stacked under them, and WLK_X3_KSPLIT=0 selects the kernel above.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int XK_CW = 8;
constexpr int XK_WAVES = XK_CW + XW_LOADERS;
constexpr int XK_THREADS = XK_WAVES * 64;
template <int NRB>
__device__ __forceinline__ void xk_read(xf32x4 (&f)[NRB], unsigned addr) {       // one plane's fragments of the tile's row blocks
    asm volatile("ds_read_b128 %0, %1" : "=v"(f[0]) : "v"(addr));
    asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(f[1]) : "v"(addr));
    if constexpr (NRB == 3) asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(f[2]) : "v"(addr));
}
static_assert(32 * XW_ROW_BYTES == 6144, "row-block stride of the fragment reads");
template <int NRB>
__device__ __forceinline__ void xk_wait(xf32x4 (&a)[NRB], xf32x4 (&b)[NRB], xf32x4 (&c)[NRB]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(c[0]), "+v"(c[1]));
    if constexpr (NRB == 3) asm volatile("" : "+v"(a[2]), "+v"(b[2]), "+v"(c[2]));
}
// all but the newest (ring depth - 2) slabs of this loader have landed
template <int NPW>
__device__ __forceinline__ void xk_wait_landed() {
    static_assert(NPW == 6 || NPW == 4, "update the vmcnt literals");
    if constexpr (NPW == 6) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}
static_assert(XW_DT - 1 == 3, "xk_wait_landed's literals are 3 slabs of NPW pieces");
}  // namespace

// BM = 96 (three 32-row blocks per wave: the shapes above) or 64 (two: the N = d projections of the d >= 1024 models, whose 96-row
// tiles are 160 workgroups on 256 CUs - 24 x 10 = 240 tiles of 64 rows fill the chip with two thirds of the work each).  Per-element
// arithmetic does not depend on BM.
template <int BM>
__global__ __launch_bounds__(XK_THREADS) void gemm_x3_wide2_kernel(X3GemmArgs g) {
    constexpr int NRB = BM / 32;                              // row blocks of a wave's part of the tile
    constexpr int SLAB_BYTES = BM * XW_ROW_BYTES;             // activation rows of a slab: 18 432 / 12 288
    constexpr int NPW = SLAB_BYTES / 1024 / XW_LOADERS;       // DMA pieces per loader wave and slab: 6 / 4
    constexpr int STAGE_OFF = XW_NB * SLAB_BYTES;             // the staging region behind the ring
    static_assert(NPW * XW_LOADERS * 1024 == SLAB_BYTES, "a slab is a whole number of 1 KiB pieces per loader");
    asm volatile("" ::"s"(g.A3), "s"(g.lda), "s"(g.W3), "s"(g.bias), "s"(g.C), "s"(g.ldc), "s"(g.R), "s"(g.ldr), "s"(g.M),
                 "s"(g.N), "s"(g.K), "s"(g.flags), "s"(g.scale), "s"(g.scale_cols), "s"(g.scale_period), "s"(g.batch));
    __builtin_amdgcn_sched_barrier(0);
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool batched = g.batch > 0;
    // ---- the tile walk (as in gemm_x3_wide_kernel) --------------------------------------------------------------------
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + XW_BN - 1) / XW_BN;
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
        int rm, cn;
        if (g.walk_colmajor) {
            cn = r / band_m;
            rm = r - cn * band_m;
        } else {
            rm = r / band_n;
            cn = r - rm * band_n;
        }
        tm = band_m0 + rm;
        tn = band_n0 + cn;
        return tm < tiles_m && tn < tiles_n;
    };
    auto advance = [&](int slot) {
        int tm, tn, b;
        while (slot < limit && !decode(slot, tm, tn, b)) slot += stride;
        return slot;
    };
    const int slot0 = advance(first);
    if (slot0 >= limit) return;
    const int nslab = g.K / 32;             // even (launch_gemm_x3)
    const bool x3_out = g.x3_out;

    // ---- the fp32 epilogue, shared by all eleven waves (item k of the 48 for wave k % 11) ------------------------------------
    auto shared_epilogue = [&](int m0, int n0, int bz) {
        float* const gC = batched ? table_at(g.z.out, (unsigned)bz) : g.C;
        const float* const gR = batched ? table_at(g.z.res, (unsigned)bz) : g.R;
        const float* const stage = reinterpret_cast<const float*>(lds + STAGE_OFF);
        const int colq = lane & 7, rsub = lane >> 3;
        constexpr int ITEMS = 4 * (BM / 8);
        auto bias_of_item = [&](int k) -> xf32x4 {
            const int c = min(n0 + 32 * (min(k, ITEMS - 1) / (BM / 8)) + 4 * colq, g.N - 4);
            return g.bias ? *reinterpret_cast<const xf32x4*>(g.bias + c) : xf32x4{0.f, 0.f, 0.f, 0.f};
        };
        xf32x4 bq = bias_of_item(wave);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool gelu = (g.flags & kGemmGelu) != 0, has_res = (g.flags & kGemmResidual) != 0;
#pragma unroll 1
        for (int k = wave; k < ITEMS; k += XK_WAVES) {
            const xf32x4 bq_next = bias_of_item(k + XK_WAVES);
            const int wp = k / (BM / 8), it = k - wp * (BM / 8);
            const int row_t = 8 * it + rsub, row = m0 + row_t;
            const int col = n0 + 32 * wp + 4 * colq;
            xf32x4 v = *reinterpret_cast<const xf32x4*>(stage + (wp * BM + row_t) * XW_WPITCH + 4 * colq);
            const xf32x4 res = has_res ? *reinterpret_cast<const xf32x4*>(gR + (long)min(row, g.M - 1) * g.ldr + min(col, g.N - 4))
                                       : xf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] += bq[e];
                if ((g.flags & kGemmScaleCols) && (g.scale_period ? (col + e) % g.scale_period : col + e) < g.scale_cols) v[e] *= g.scale;
                if (gelu) v[e] = x3_gelu_erf(v[e]);
                v[e] += res[e];
            }
            if (row < g.M && col < g.N) *reinterpret_cast<xf32x4*>(gC + (long)row * g.ldc + col) = v;
            bq = bq_next;
        }
    };

    if (wave >= XK_CW) {
        // ---- loader (the code of gemm_x3_wide_kernel's loaders; barriers per tile: slabs + fold + epilogue) ----------------
        const int lw = wave - XK_CW;
        const char* src[NPW];
        int row_of[NPW], unit_of[NPW];
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int byte = 1024 * (lw + XW_LOADERS * i) + 16 * lane;
            row_of[i] = byte / XW_ROW_BYTES;
            unit_of[i] = (((byte - row_of[i] * XW_ROW_BYTES) >> 4) ^ ((row_of[i] >> 2) & 3)) * 16;
        }
        auto set_src = [&](int slot) {
            int tm, tn, b;
            decode(slot, tm, tn, b);
            const int m0 = tm * BM;
            const char* const gA = reinterpret_cast<const char*>(batched ? reinterpret_cast<const unsigned short*>(table_at(g.z.in, (unsigned)b)) : g.A3);
#pragma unroll
            for (int i = 0; i < NPW; ++i) src[i] = gA + (long)min(m0 + row_of[i], g.M - 1) * g.lda * 6 + unit_of[i];
        };
        int issue_slot = slot0, s_next = 0, ring = 0;
        bool more = true;
        set_src(issue_slot);
        auto issue_one = [&]() {
            const long adv = (long)s_next * XW_ROW_BYTES;
            unsigned char* const dst = lds + ring * SLAB_BYTES + lw * 1024;
            x3_static_for<NPW>([&](auto I) {
                constexpr int i = decltype(I)::value;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + adv),
                                                 (__attribute__((address_space(3))) void*)(dst + i * XW_LOADERS * 1024), 16, 0, 0);
            });
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
#pragma unroll
        for (int i = 0; i < XW_DT; ++i) issue_one();
        xk_wait_landed<NPW>();
        __builtin_amdgcn_s_barrier();
        for (int slot = slot0; slot < limit; slot = advance(slot + stride)) {
            for (int tt = 0; tt < nslab; ++tt) {
                issue_one();
                xk_wait_landed<NPW>();
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_s_barrier();          // the fold: the k-step-1 waves' accumulators are in the staging region
            if (x3_out) {
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_barrier();
            } else {
                int tm, tn, bz;
                decode(slot, tm, tn, bz);
                shared_epilogue(tm * BM, tn * XW_BN, bz);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        return;
    }

    // ---- compute waves ------------------------------------------------------------------------------------------------
    const int cb = wave & 3, half = wave >> 2;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int r = lane & 31, hi = lane >> 5, swz = (r >> 2) & 3;
    unsigned fa[3];                        // plane p's fragment of row block 0 in ring slot 0: lane (r, hi) reads row r, chunk 2 half + hi
#pragma unroll
    for (int p = 0; p < 3; ++p) fa[p] = lds_base + (unsigned)(r * XW_ROW_BYTES) + (unsigned)((((2 * half + hi) * 3 + p) ^ swz) * 16);
    xf32x16 acc[NRB];
    auto mfma3 = [&](const xf32x4& w, const xf32x4 (&f)[NRB]) {
#pragma unroll
        for (int i = 0; i < NRB; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, f[i]), acc[i], 0, 0, 0);
    };
    const int n_blocks = (g.N + 31) / 32;
    typedef const __attribute__((address_space(1))) xf32x4* wptr_t;
    auto w_base = [&](int slot) -> const char* {       // this wave's half of slab 0 of its weight block
        int tm, tn, bz;
        decode(slot, tm, tn, bz);
        return reinterpret_cast<const char*>(g.W3) + (long)min(4 * tn + cb, n_blocks - 1) * nslab * XW_WSLAB_BYTES + half * (XW_WSLAB_BYTES / 2) + 16 * lane;
    };
    const int col_in_tile = 32 * cb + 4 * hi;
    auto load_bias = [&](int slot, xf32x4 (&b)[4]) {
        int tm, tn, bz;
        decode(slot, tm, tn, bz);
        if (x3_out && half == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = min(tn * XW_BN + col_in_tile + 8 * j, g.N - 4);
                b[j] = g.bias ? *reinterpret_cast<const xf32x4*>(g.bias + c) : xf32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    xf32x4 wa[3], wb[3];                       // weight planes of the slab being multiplied / of the next one
#pragma unroll
    for (int x = 0; x < 3; ++x) wb[x] = xf32x4{0.f, 0.f, 0.f, 0.f};
    const char* w_cur = w_base(slot0);
#pragma unroll
    for (int x = 0; x < 3; ++x) wa[x] = *(wptr_t)(w_cur + 1024 * x);
    xf32x4 bias_next[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bias_next[j] = xf32x4{0.f, 0.f, 0.f, 0.f};
    load_bias(slot0, bias_next);

    __builtin_amdgcn_s_barrier();              // slab 0 has landed
    xf32x4 f0a[NRB], f0b[NRB], f1[NRB], f2[NRB];       // plane hi (two buffers), mid, lo
    xk_read<NRB>(f0a, fa[0]);
    xk_read<NRB>(f1, fa[1]);
    xk_read<NRB>(f2, fa[2]);
    xk_wait<NRB>(f0a, f1, f2);
    unsigned ring_off = SLAB_BYTES;         // byte offset of the ring slot of the slab being READ (one ahead of the MFMAs)
    // one slab: the barrier behind which the next slab is readable, then the six plane products in the one-wave kernel's order
    // (small terms first: (lo, hi) (hi, lo) (mid, mid) (mid, hi) (hi, mid) (hi, hi), activation plane first) with the next
    // slab's fragment reads and weight loads between them.  The weights are requested one slab ahead, in the order of their
    // use and all in the first half of the slab: a whole slab of MFMAs lies between a load and its first use.  (Two slabs
    // ahead, with the products regrouped by weight plane so that a plane's registers are free early, measured SLOWER:
    // profiles/r06k_x3_ksplit_regrouped_probe.txt - weight latency is not what the loop waits for.)
    auto slab = [&](const xf32x4 (&F0c)[NRB], xf32x4 (&F0n)[NRB], const xf32x4 (&Wc)[3], xf32x4 (&Wn)[3], const char* pn) {
        __builtin_amdgcn_s_barrier();
        xk_read<NRB>(F0n, fa[0] + ring_off);
        __builtin_amdgcn_sched_barrier(0);
        mfma3(Wc[0], f2);
        __builtin_amdgcn_sched_barrier(0);
        xk_read<NRB>(f2, fa[2] + ring_off);
        Wn[0] = *(wptr_t)(pn);
        __builtin_amdgcn_sched_barrier(0);
        mfma3(Wc[2], F0c);
        __builtin_amdgcn_sched_barrier(0);
        Wn[2] = *(wptr_t)(pn + 2048);
        __builtin_amdgcn_sched_barrier(0);
        mfma3(Wc[1], f1);
        __builtin_amdgcn_sched_barrier(0);
        Wn[1] = *(wptr_t)(pn + 1024);
        __builtin_amdgcn_sched_barrier(0);
        mfma3(Wc[0], f1);
        __builtin_amdgcn_sched_barrier(0);
        xk_read<NRB>(f1, fa[1] + ring_off);
        __builtin_amdgcn_sched_barrier(0);
        mfma3(Wc[1], F0c);
        mfma3(Wc[0], F0c);
        __builtin_amdgcn_sched_barrier(0);
        xk_wait<NRB>(F0n, f1, f2);
        ring_off = ring_off == (unsigned)((XW_NB - 1) * SLAB_BYTES) ? 0u : ring_off + (unsigned)SLAB_BYTES;
    };
    for (int slot = slot0; slot < limit;) {
        int tile_m, tile_n, bz;
        decode(slot, tile_m, tile_n, bz);
        const int next_slot = advance(slot + stride);
        const int m0 = tile_m * BM, n0 = tile_n * XW_BN;
        const int col0 = n0 + col_in_tile;
        const char* const w_next = next_slot < limit ? w_base(next_slot) : w_cur;
        xf32x4 bias[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bias[j] = bias_next[j];
        if (next_slot < limit) load_bias(next_slot, bias_next);
        auto scaled = [&](int col) { return (g.flags & kGemmScaleCols) && (g.scale_period ? col % g.scale_period : col) < g.scale_cols; };
#pragma unroll
        for (int i = 0; i < NRB; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
        for (int tt = 0; tt < nslab; tt += 2) {
            slab(f0a, f0b, wa, wb, w_cur + (long)(tt + 1) * XW_WSLAB_BYTES);
            slab(f0b, f0a, wb, wa, tt + 2 < nslab ? w_cur + (long)(tt + 2) * XW_WSLAB_BYTES : w_next);
        }
        w_cur = w_next;
        float* const gC = batched ? table_at(g.z.out, (unsigned)bz) : g.C;
        if (x3_out) {
            constexpr int PITCH = XW_STAGE_PITCH;
            float* const stage = reinterpret_cast<float*>(lds + STAGE_OFF);
            if (half == 1) {
#pragma unroll
                for (int i = 0; i < NRB; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<xf32x4*>(stage + (32 * i + r) * PITCH + col_in_tile + 8 * j) =
                            xf32x4{acc[i][4 * j], acc[i][4 * j + 1], acc[i][4 * j + 2], acc[i][4 * j + 3]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // fold
            if (half == 0) {
#pragma unroll
                for (int i = 0; i < NRB; ++i) {
                    float* const srow = stage + (32 * i + r) * PITCH + col_in_tile;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        xf32x4 v = *reinterpret_cast<const xf32x4*>(srow + 8 * j);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = (acc[i][4 * j + e] + v[e]) + bias[j][e];
                            if (scaled(col0 + 8 * j + e)) v[e] *= g.scale;
                        }
                        if (g.flags & kGemmGelu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = x3_gelu_erf(v[e]);
                        }
                        *reinterpret_cast<xf32x4*>(srow + 8 * j) = v;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            unsigned short* const c3 = batched ? reinterpret_cast<unsigned short*>(gC) : g.C3;
            const int tid = threadIdx.x;           // 0 .. 511: the compute waves
            if (n0 < g.vt_col0) {
                for (int item = tid; item < BM * (XW_BN / 8); item += XK_CW * 64) {
                    const int row = item >> 4, c = item & 15;
                    if (m0 + row < g.M && n0 + 8 * c < g.N) {
                        const float4 a = *reinterpret_cast<const float4*>(stage + row * PITCH + 8 * c);
                        const float4 b = *reinterpret_cast<const float4*>(stage + row * PITCH + 8 * c + 4);
                        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                        x3_store_chunk(c3 + (long)(m0 + row) * 3 * g.ldc3 + (long)((n0 >> 3) + c) * 24, v);
                    }
                }
            } else {
                unsigned short* const vt = c3 + g.vt_off;
                for (int item = tid; item < XW_BN * (BM / 8); item += XK_CW * 64) {
                    const int dcol = item / (BM / 8), u = item - dcol * (BM / 8);
                    const int r0 = 32 * (u >> 2) + 4 * (u & 3);
                    if (n0 + dcol < g.N && m0 + 32 * (u >> 2) < g.vt_ld) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int row = r0 + (e & 3) + 16 * (e >> 2);
                            v[e] = m0 + row < g.M ? stage[row * PITCH + dcol] : 0.f;
                        }
                        x3_store_chunk(vt + (long)(n0 + dcol - g.vt_col0) * 3 * g.vt_ld + (long)((m0 >> 3) + u) * 24, v);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // the staging region may be written again
        } else {
            float* const wst = reinterpret_cast<float*>(lds + STAGE_OFF) + cb * (BM * XW_WPITCH);
            if (half == 1) {
#pragma unroll
                for (int i = 0; i < NRB; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<xf32x4*>(wst + (32 * i + r) * XW_WPITCH + 8 * j + 4 * hi) =
                            xf32x4{acc[i][4 * j], acc[i][4 * j + 1], acc[i][4 * j + 2], acc[i][4 * j + 3]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // fold
            if (half == 0) {
#pragma unroll
                for (int i = 0; i < NRB; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float* const p = wst + (32 * i + r) * XW_WPITCH + 8 * j + 4 * hi;
                        const xf32x4 o = *reinterpret_cast<const xf32x4*>(p);
                        *reinterpret_cast<xf32x4*>(p) = xf32x4{acc[i][4 * j] + o[0], acc[i][4 * j + 1] + o[1], acc[i][4 * j + 2] + o[2], acc[i][4 * j + 3] + o[3]};
                    }
            }
            shared_epilogue(m0, n0, bz);
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
    return on && M >= 256 && N >= 1024 && N % 4 == 0 && K >= 64 && K % 64 == 0 && lda % 8 == 0;
}

static std::atomic<int> g_x3_persist{-1};                // -1: WLK_X3_PERSIST not read yet
static std::atomic<int> g_x3_ksplit{-1};                 // -1: WLK_X3_KSPLIT not read yet (1: gemm_x3_wide2_kernel, the default)
static std::atomic<int> g_x3_bm{-1};                     // -1: WLK_X3_BM not read yet (0: by the rounds x rows rule)
void x3_refresh_env_switches() {
    g_x3_persist.store(-1, std::memory_order_relaxed);
    g_x3_ksplit.store(-1, std::memory_order_relaxed);
    g_x3_bm.store(-1, std::memory_order_relaxed);
}

void launch_gemm_x3(const LaunchCtx& ctx, const X3GemmArgs& g, const char* tag) {
    if (g.M <= 0 || g.N <= 0) return;
    if (g.K % 64 != 0 || g.lda % 8 != 0 || g.K < 64) throw std::invalid_argument("x3 gemm: K must be a multiple of 64, lda of 8");
    if (g.N % 4 != 0 || (!g.x3_out && g.ldc % 4 != 0) || ((g.flags & kGemmResidual) && g.ldr % 4 != 0))
        throw std::invalid_argument("x3 gemm: N, ldc and ldr must be multiples of 4 (16-byte epilogue accesses)");
    if (g.batch <= 0 && (((uintptr_t)g.bias | (g.x3_out ? 0 : (uintptr_t)g.C) | ((g.flags & kGemmResidual) ? (uintptr_t)g.R : 0)) & 15))
        throw std::invalid_argument("x3 gemm: bias, C and R must be 16-byte aligned");
    if (g.batch > 0 && ((uintptr_t)g.bias & 15)) throw std::invalid_argument("x3 gemm: bias must be 16-byte aligned");
    if (g.flags & ~(kGemmGelu | kGemmResidual | kGemmScaleCols)) throw std::invalid_argument("x3 gemm: unsupported epilogue flag");
    if (g.x3_out && ((g.flags & kGemmResidual) || g.vt_col0 % XW_BN != 0 || g.ldc3 % 8 != 0 || g.vt_ld % 32 != 0 ||
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
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_wide_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XW_LDS_BYTES));
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_wide_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XW_LDS_BYTES));
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_wide_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XW_LDS_BYTES));
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_wide_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XW_LDS_BYTES));
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_wide_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XW_LDS_BYTES));
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_wide2_kernel<96>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XW_LDS_BYTES));
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_wide2_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XW_LDS_BYTES));
        configured.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const int batch = g.batch > 0 ? g.batch : 1;
    // Timing ablations of scripts/x3_probe.py.  Variants 1-5 skip MFMAs, DMA pieces or weight loads and produce WRONG results, so a
    // stray WLK_X3_ABL on a serving box must not reach them: they need WLK_PROBES=1 beside it and announce themselves once.
    static const int abl = [] {
        const char* e = getenv("WLK_X3_ABL");
        int v = e ? atoi(e) : 0;
        if (v >= 1 && v <= 5) {
            const char* ok = getenv("WLK_PROBES");
            if (!(ok && ok[0] == '1')) {
                fprintf(stderr, "libwlk_hip: WLK_X3_ABL=%d ignored (result-corrupting timing ablation; set WLK_PROBES=1 to run it)\n", v);
                v = 0;
            } else {
                fprintf(stderr, "libwlk_hip: WLK_X3_ABL=%d ACTIVE - the X3 GEMM results of this process are WRONG (timing probe)\n", v);
            }
        }
        return v;
    }();
    int ksplit = g_x3_ksplit.load(std::memory_order_relaxed);
    if (ksplit < 0) {
        const char* e = getenv("WLK_X3_KSPLIT");
        ksplit = !(e && e[0] == '0');
        g_x3_ksplit.store(ksplit, std::memory_order_relaxed);
    }
    const bool wide2 = ksplit && abl == 0;
    static std::atomic<int> cu_count[64];
    int cus = cu_count[dev & 63].load(std::memory_order_relaxed);
    if (cus == 0) {
        WLK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (cus < 8) cus = 8;
        cu_count[dev & 63].store(cus, std::memory_order_relaxed);
    }
    // tile height (two-wave kernel only): 64-row tiles where they cut the rounds x rows a CU works through by a tenth or more - the
    // N = d projections of the d >= 1024 models (N = 1280: 160 tiles of 96 rows on 256 CUs, 240 of 64: 40.1 -> 31.7 us at K = 1280,
    // 140.9 -> 120.0 at K = 5120) and large-v3's fc1 (640 tiles = 3 rounds of 96 rows, 960 = 4 rounds of 64: 137.4 -> 131.4 us); at
    // equal rounds x rows the 96-row tile's loop wins (18 MFMAs per three weight loads against 12: base fc1 25.7 vs 31.1 us).
    // profiles/r06l_x3_bm64_probe.txt.  WLK_X3_BM=64 / 96 forces either; an element's arithmetic does not depend on it.
    int bm_env = g_x3_bm.load(std::memory_order_relaxed);
    if (bm_env < 0) {
        const char* e = getenv("WLK_X3_BM");
        bm_env = e ? atoi(e) : 0;
        g_x3_bm.store(bm_env, std::memory_order_relaxed);
    }
    const int tiles_n = (g.N + XW_BN - 1) / XW_BN;
    auto rounds_rows = [&](int bm) { return (long)(((long)((g.M + bm - 1) / bm) * tiles_n * batch + cus - 1) / cus) * bm; };
    int bm = XW_BM;
    if (wide2 && (bm_env == 64 || (bm_env == 0 && 10 * rounds_rows(64) <= 9 * rounds_rows(XW_BM)))) bm = 64;
    const int tiles_m = (g.M + bm - 1) / bm;
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
    // column-major walk inside a band: the workgroups resident at a time share weight blocks 4-fold and a persistent
    // workgroup keeps its activation rows from tile to tile (cross-K|V 71 -> 67 us, large-v3 fc1 154 -> 147); with several
    // sessions in one launch the row-major order measured better (212 vs 222 us at 8 sessions), so it is taken for one
    // session only.  WLK_X3_COLMAJOR=0 / 1 forces either
    static const int colmajor = [] {
        const char* e = getenv("WLK_X3_COLMAJOR");
        return e ? atoi(e) : -1;
    }();
    X3GemmArgs gg = g;
    gg.walk_colmajor = colmajor >= 0 ? colmajor : (batch <= 1);
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
    const dim3 grid(blocks);
    if (wide2) {
        if (bm == 64) hipLaunchKernelGGL(gemm_x3_wide2_kernel<64>, grid, dim3(XK_THREADS), XW_LDS_BYTES, ctx.stream, gg);
        else hipLaunchKernelGGL(gemm_x3_wide2_kernel<96>, grid, dim3(XK_THREADS), XW_LDS_BYTES, ctx.stream, gg);
        WLK_HIP(hipGetLastError());
        return;
    }
    if (abl == 1) hipLaunchKernelGGL(gemm_x3_wide_kernel<1>, grid, dim3(XW_THREADS), XW_LDS_BYTES, ctx.stream, gg);
    else if (abl == 2) hipLaunchKernelGGL(gemm_x3_wide_kernel<2>, grid, dim3(XW_THREADS), XW_LDS_BYTES, ctx.stream, gg);
    else if (abl == 3) hipLaunchKernelGGL(gemm_x3_wide_kernel<3>, grid, dim3(XW_THREADS), XW_LDS_BYTES, ctx.stream, gg);
    else if (abl == 4) hipLaunchKernelGGL(gemm_x3_wide_kernel<4>, grid, dim3(XW_THREADS), XW_LDS_BYTES, ctx.stream, gg);
    else if (abl == 5) hipLaunchKernelGGL(gemm_x3_wide_kernel<5>, grid, dim3(XW_THREADS), XW_LDS_BYTES, ctx.stream, gg);
    else if (abl == 6) hipLaunchKernelGGL(gemm_x3_wide_kernel<6>, grid, dim3(XW_THREADS), XW_LDS_BYTES, ctx.stream, gg);
    else if (abl == 7) hipLaunchKernelGGL(gemm_x3_wide_kernel<7>, grid, dim3(XW_THREADS), XW_LDS_BYTES, ctx.stream, gg);
    else if (abl == 8) hipLaunchKernelGGL(gemm_x3_wide_kernel<8>, grid, dim3(XW_THREADS), XW_LDS_BYTES, ctx.stream, gg);
    else hipLaunchKernelGGL(gemm_x3_wide_kernel<0>, grid, dim3(XW_THREADS), XW_LDS_BYTES, ctx.stream, gg);
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

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
constexpr int XW_DT = XW_NB - 1;                          // slabs in flight beyond the one being multiplied
constexpr int XW_NF = 12;                                 // fragments per k-step and wave: 3 x 3 A + 3 B
constexpr size_t XW_LDS_BYTES = (size_t)XW_NB * XW_SLAB_BYTES;
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
    const unsigned short* const gA = batched ? reinterpret_cast<const unsigned short*>(table_at(g.z.in, blockIdx.y)) : g.A3;
    float* const gC = batched ? table_at(g.z.out, blockIdx.y) : g.C;
    const float* const gR = batched ? table_at(g.z.res, blockIdx.y) : g.R;
    // XCD-aware tile mapping, as in gemm_nt_f32_kernel: 4 row bands x 2 column bands, one per XCD
    const int tiles_m = (g.M + XW_BM - 1) / XW_BM, tiles_n = (g.N + XW_BN - 1) / XW_BN;
    int tile_m, tile_n;
    if (tiles_m >= 8 && g.map_mode == 2) {          // probe: 2 row bands x 4 column bands
        const int band_m = (tiles_m + 1) / 2, band_n = (tiles_n + 3) / 4;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        tile_m = (xcd >> 2) * band_m + slot / band_n;
        tile_n = (xcd & 3) * band_n + slot % band_n;
        if (slot >= band_m * band_n || tile_m >= tiles_m || tile_n >= tiles_n) return;
    } else if (tiles_m >= 8 && g.map_mode == 0) {
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
    const int m0 = tile_m * XW_BM, n0 = tile_n * XW_BN;
    const int nslab = g.K / 32;

    if (wave >= 4) {
        // ---- loader: piece j covers LDS bytes [1024 j, 1024 j + 1024) of a slab; lane l lands at byte 1024 j + 16 l = a
        // (row, swizzled unit) of the slab image, and fetches that row's logical unit from the X3 operand -----------------
        const int lw = wave - 4;
        const char* src[XW_NPW];
        int piece_of[XW_NPW];
#pragma unroll
        for (int i = 0; i < XW_NPW; ++i) {
            const int j = min(lw + 4 * i, XW_PIECES - 1);
            piece_of[i] = j;
            const int byte = 1024 * j + 16 * lane;
            const int row = byte / XW_ROW_BYTES;
            const int unit = ((byte - row * XW_ROW_BYTES) >> 4) ^ ((row >> 2) & 3);
            if (row < XW_BM)
                src[i] = reinterpret_cast<const char*>(gA) + ((long)min(m0 + row, g.M - 1) * 3 * g.lda) * 2 + unit * 16;
            else
                src[i] = reinterpret_cast<const char*>(g.W3) + ((long)min(n0 + row - XW_BM, g.N - 1) * 3 * g.K) * 2 + unit * 16;
        }
        auto issue_slab = [&](int slab) {     // slab (clamped: the tail re-fetches the last one into a free slot)
            if constexpr (ABL >= 2) return;
            const int slot = slab % XW_NB;
            const long adv = (long)min(slab, nslab - 1) * XW_ROW_BYTES;
            x3_static_for<XW_NPW>([&](auto I) {
                constexpr int i = decltype(I)::value;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + adv),
                                                 (__attribute__((address_space(3))) void*)(lds + slot * XW_SLAB_BYTES + piece_of[i] * 1024),
                                                 16, 0, 0);
            });
        };
        issue_slab(0);
        issue_slab(1);
        xw_wait_vmcnt<XW_NPW>();
        __builtin_amdgcn_s_barrier();
        for (int tt = 0; tt < nslab; ++tt) {
            issue_slab(tt + XW_DT);
            xw_wait_vmcnt<XW_NPW>();           // this loader's pieces of slab tt + 1 have landed
            __builtin_amdgcn_s_barrier();
        }
        xw_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();          // nothing of this loader is in flight any more: the ring may be reused (epilogue staging)
        return;
    }

    // ---- compute waves ------------------------------------------------------------------------------------------------
    const int col = n0 + 32 * wave + (lane & 31);
    const float bias = g.bias ? g.bias[min(col, g.N - 1)] : 0.f;
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
    auto read_frags = [&](xf32x4 (&f)[XW_NF], int slab, const unsigned (&addr)[XW_NF]) {
        xw_read_frags(f, addr, (unsigned)(slab % XW_NB) * (unsigned)XW_SLAB_BYTES);
    };
    xf32x16 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    auto mfma_step = [&](const xf32x4 (&f)[XW_NF]) {     // 18 MFMAs: product t of row block i - three independent chains
        x3_static_for<18>([&](auto X) {
            constexpr int x = decltype(X)::value;
            constexpr int t = x / 3, i = x % 3;
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[i * 3 + PA[t]]),
                                                             __builtin_bit_cast(bf16x8, f[9 + PB[t]]), acc[i], 0, 0, 0);
        });
    };

    __builtin_amdgcn_s_barrier();              // slab 0 has landed
    xf32x4 g0[XW_NF], g1[XW_NF];
    read_frags(g0, 0, f_addr[0]);
    xw_wait_frags(g0);
    for (int tt = 0; tt < nslab; ++tt) {
        if constexpr (ABL != 1 && ABL != 3) read_frags(g1, tt, f_addr[1]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ABL != 1) mfma_step(g0);
        __builtin_amdgcn_sched_barrier(0);
        xw_wait_frags(g1);
        __builtin_amdgcn_s_barrier();          // slab tt + 1 readable; everybody is done with slab tt's fragments
        if constexpr (ABL != 1 && ABL != 3) read_frags(g0, tt + 1, f_addr[0]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ABL != 1) mfma_step(g1);
        __builtin_amdgcn_sched_barrier(0);
        xw_wait_frags(g0);
    }

    __builtin_amdgcn_s_barrier();              // the loaders' tail fetches have landed
    if (g.x3_out) {
        // ---- result in the X3 format: the tile goes through LDS (the ring is free) so that 8 consecutive columns - or, for
        // the transposed part, 8 rows in the attention kernel's key order - meet in one thread, which splits them into the
        // three planes and writes the chunk's 48 contiguous bytes
        constexpr int PITCH = XW_BN + 4;
        float* stage = reinterpret_cast<float*>(lds);
        const bool do_scale = (g.flags & kGemmScaleCols) && (g.scale_period ? col % g.scale_period : col) < g.scale_cols;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float v = acc[i][q] + bias;
                if (do_scale) v *= g.scale;
                stage[(32 * i + (q & 3) + 8 * (q >> 2) + 4 * hi) * PITCH + 32 * wave + (lane & 31)] = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        unsigned short* const c3 = batched ? reinterpret_cast<unsigned short*>(table_at(g.z.out, blockIdx.y)) : g.C3;
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
        return;
    }
    // epilogue: acc[i][q] is C[row = 32 i + (q & 3) + 8 (q >> 2) + 4 (lane >> 5)][col] of the tile
    if (col < g.N) {
        const bool do_scale = (g.flags & kGemmScaleCols) && (g.scale_period ? col % g.scale_period : col) < g.scale_cols;
        const bool gelu = (g.flags & kGemmGelu) != 0, has_res = (g.flags & kGemmResidual) != 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int row_base = m0 + 32 * i + 4 * hi;
            float res[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = row_base + (q & 3) + 8 * (q >> 2);
                res[q] = has_res ? gR[(long)min(row, g.M - 1) * g.ldr + col] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = row_base + (q & 3) + 8 * (q >> 2);
                float v = acc[i][q] + bias;
                if (do_scale) v *= g.scale;
                if (gelu) v = x3_gelu_erf(v);
                v += res[q];
                if (row < g.M) gC[(long)row * g.ldc + col] = v;
            }
        }
    }
}

bool gemm_x3_wide_applicable(int M, int N, int K, long lda) {
    static const bool on = [] {
        const char* e = getenv("WLK_X3");
        return !(e && e[0] == '0');
    }();
    return on && M >= 256 && N >= 1024 && K >= 64 && K % 32 == 0 && lda % 8 == 0;
}

void launch_gemm_x3(const LaunchCtx& ctx, const X3GemmArgs& g, const char* tag) {
    if (g.M <= 0 || g.N <= 0) return;
    if (g.K % 32 != 0 || g.lda % 8 != 0 || g.K < 64) throw std::invalid_argument("x3 gemm: K must be a multiple of 32 (>= 64), lda of 8");
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
    static const int map_mode = [] {
        const char* e = getenv("WLK_X3_MAP");
        return e ? atoi(e) : 0;
    }();
    X3GemmArgs gg = g;
    gg.map_mode = map_mode;
    int blocks = tiles_m * tiles_n;
    if (tiles_m >= 8 && map_mode == 0) blocks = 8 * ((tiles_m + 3) / 4) * ((tiles_n + 1) / 2);
    if (tiles_m >= 8 && map_mode == 2) blocks = 8 * ((tiles_m + 1) / 2) * ((tiles_n + 3) / 4);
    const int batch = g.batch > 0 ? g.batch : 1;
    // algorithmic work (what the roofline fraction is computed from): 2 M N K flop, operands and result once
    KernelScope ks(ctx, tag, 2.0 * batch * (double)g.M * g.N * g.K,
                   batch * (6.0 * ((double)g.M * g.K) + 4.0 * (double)g.M * g.N) + 6.0 * (double)g.N * g.K);
    static const int abl = [] {
        const char* e = getenv("WLK_X3_ABL");
        return e ? atoi(e) : 0;
    }();
    const dim3 grid(blocks, batch);
    if (abl == 1) hipLaunchKernelGGL(gemm_x3_wide_kernel<1>, grid, dim3(512), XW_LDS_BYTES, ctx.stream, gg);
    else if (abl == 2) hipLaunchKernelGGL(gemm_x3_wide_kernel<2>, grid, dim3(512), XW_LDS_BYTES, ctx.stream, gg);
    else if (abl == 3) hipLaunchKernelGGL(gemm_x3_wide_kernel<3>, grid, dim3(512), XW_LDS_BYTES, ctx.stream, gg);
    else hipLaunchKernelGGL(gemm_x3_wide_kernel<0>, grid, dim3(512), XW_LDS_BYTES, ctx.stream, gg);
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

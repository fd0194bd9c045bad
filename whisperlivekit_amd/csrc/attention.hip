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
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "wave_ops.h"

namespace wlk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

long long* g_attn_dbg_clock = nullptr;

constexpr int QT = 32, KT = 32, NWAVE = 4, K_LD = 68, O_LD = 65;
constexpr int kAttnLdsFloats = NWAVE * KT * K_LD + NWAVE * KT * 64;  // K tiles + V tiles
constexpr int kAttnLdsTotal = kAttnLdsFloats + QT * K_LD;            // + Q tile

// One kernel serves the encoder (q, k, v interleaved in one [T][3d] buffer) and the decoder's
// PREFILL cross-attention (q rows = all fed tokens of all beams, k/v = the 1500 projected encoder
// positions).  For the latter, heads that are AlignAtt alignment heads also dump their raw scores
// into the session's alignment window (softmaxed in place afterwards by ring_softmax_kernel), so
// the [H, q, 1500] QK tensor of the reference is never formed for the other heads.
__global__ __launch_bounds__(256) void flash_attention_kernel(FlashArgs a) {
    WLK_PIN_FLASH_ARGS(a);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // batched encodes: session blockIdx.y (locals: writing into `a` would put the argument struct into scratch memory)
    const bool batched = a.batch > 0;
    const float* const aq = batched ? table_at(a.z.in, blockIdx.y) : a.q;
    const float* const ak = batched ? aq + a.z_k_off : a.k;
    const float* const av = batched ? aq + a.z_v_off : a.v;
    float* const aout = batched ? table_at(a.z.out, blockIdx.y) : a.out;
    const int T = a.Tk;
    const int n_head = a.n_head;
    float* Ks = lds;                            // [NWAVE][KT][K_LD]
    float* Vs = lds + NWAVE * KT * K_LD;        // [NWAVE][KT][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int head = blockIdx.x % n_head;       // head == XCD for 8 heads: K/V of a head stay in one L2
    const int q_tiles = (a.Tq + QT - 1) / QT;
    const int qt_idx = (blockIdx.x / n_head) % q_tiles;
    const int ks = blockIdx.x / (n_head * q_tiles);   // key split index (0 when a.k_splits == 1)
    const int q0 = qt_idx * QT;
    const long ld = a.ldkv;
    const int half = lane >> 5;
    const int lq = lane & 31;


    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // staging map: per iteration 128 keys x 64 floats for K and for V = 2048 float4 each, 8 per thread
    // stacked prefills: this query tile's session supplies the keys / values and the alignment window
    // (round 4: global address space kept explicit - a pointer loaded from the tile table made these flat loads; no
    // select on the loaded VALUES - hipcc turned `ok ? k4 : 0` into one predicated branch per load with waits in between:
    // rows past T are fetched from row 0 instead, their scores are set to -inf below and their probabilities are 0)
    const gcf_ptr tk = to_global(a.tile_rows ? a.tile_rows[qt_idx].cross_kv + a.tile_kv_off : ak);
    const gcf_ptr kbase = tk + head * 64;
    const gcf_ptr vbase = (a.tile_rows ? tk + a.tile_v_off : to_global(av)) + head * 64;
    float4 rk[8], rv[8];
    auto fetch = [&](int it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i;
            const int key = it * (NWAVE * KT) + (idx >> 4);
            const int c4 = idx & 15;
            const long off = (long)(key < T ? key : 0) * ld + c4 * 4;
            rk[i] = ldg4(kbase + off);
            rv[i] = ldg4(vbase + off);
        }
    };
    // key range of this workgroup: iterations [it_lo, it_hi) of 128 keys each
    const int n_iter_all = (T + NWAVE * KT - 1) / (NWAVE * KT);
    const int it_per = (n_iter_all + a.k_splits - 1) / a.k_splits;
    const int it_lo = ks * it_per;
    const int n_iter = min(n_iter_all, it_lo + it_per);
    fetch(it_lo);     // in flight while the Q tile is staged and the alignment-head bookkeeping below resolves
    // alignment-window bookkeeping of this lane's query row: requested together with the tiles (each used to be its own
    // memory round trip in front of the first key tile)
    // (unconditional loads from pointers that are always readable - an absent table points at the queries - and pinned
    // behind the Q staging: a load inside a branch, or one hipcc can sink into one, costs its own wait)
    const int bk_row = min(q0 + lq, a.Tq - 1);
    const int* const any_ints = reinterpret_cast<const int*>(aq);
    int rank_raw = (a.head_rank ? a.head_rank + head : any_ints)[0];
    int ring_row_v = (a.ring_row ? a.ring_row + bk_row : any_ints)[0];
    int beam_v = (a.beam_of_row ? a.beam_of_row + bk_row : any_ints)[0];
    // Q tile -> LDS once ([32][K_LD], same padded layout as the K tiles); the B fragments are
    // re-read per key tile (one ds_read_b128 per 4 MFMAs - the 64-cycle fp32 MFMA leaves the LDS idle)
    float* Qs = lds + kAttnLdsFloats;
    static_assert(QT * 16 == 2 * 256, "two float4 of the Q tile per thread");
    {
        float4 qv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 256 * j, qr = i >> 4, c4 = i & 15;
            qv[j] = *reinterpret_cast<const float4*>(aq + (long)min(q0 + qr, a.Tq - 1) * a.ldq + head * 64 + c4 * 4);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 256 * j, qr = i >> 4, c4 = i & 15;
            *reinterpret_cast<float4*>(&Qs[qr * K_LD + c4 * 4]) = q0 + qr < a.Tq ? qv[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float* Qw = Qs + lq * K_LD + half * 4;
    asm volatile("" : "+v"(rank_raw), "+v"(ring_row_v), "+v"(beam_v));
    const int rank = a.head_rank ? rank_raw : -1;
    // alignment-head score dump (decoder prefill only): pointer resolved from the values requested above
    float* dump = nullptr;
    if (rank >= 0 && q0 + lq < a.Tq) {
        if (a.tile_rows) dump = a.tile_rows[qt_idx].ring + ((long)rank * a.ring_rows + ring_row_v) * (long)T;
        else dump = a.ring + (((long)rank * a.n_beam + beam_v) * a.ring_rows + ring_row_v) * (long)T;
    }
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

    const float* Kw = Ks + wave * (KT * K_LD) + lq * K_LD + half * 4;
    const float* Vw = Vs + wave * (KT * 64) + lq;

    for (int it = it_lo; it < n_iter; ++it) {
        __syncthreads();  // previous iteration's LDS reads are done
        stash();
        __syncthreads();
        if (it + 1 < n_iter) fetch(it + 1);

        const int key0 = it * (NWAVE * KT) + wave * KT;
        if (key0 < T) {  // wave-uniform
            f32x16 s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
            // the 32 MFMAs are one dependent chain and the wave issues in order: the LDS fragments of group g+1 are
            // requested right after the first MFMA of group g has been issued (same idea as gemm_f32.hip)
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
            // rows of s: key = key0 + (r&3) + 8*(r>>2) + 4*half ; column: query lq
            if (dump) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int key = key0 + 8 * g + 4 * half;
                    if (key < T)   // T % 4 == 0 is checked by the launcher when dumping
                        *reinterpret_cast<float4*>(dump + key) = make_float4(s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]);
                }
            }
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
                s[r] = __expf(s[r] - m_new);   // v_exp_f32: ~1 ulp, keeps the VALU share small next to the MFMAs
                rs += s[r];
            }
            rs += __shfl_xor(rs, 32, 64);
            l_run = l_run * alpha + rs;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
            // V values of key r+1 are requested before the two MFMAs of key r are issued (one LDS latency ahead)
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
            if (a.k_splits == 1) {
                if (qrow < a.Tq) aout[(long)qrow * a.ldo + head * 64 + dd] = acc / L;
            } else if (qrow < a.Tq) {
                // partial state of this key range: unnormalised O (relative to M), M and L
                const long slot = ((long)qrow * n_head + head) * a.k_splits + ks;
                a.part_o[slot * 64 + dd] = acc;
                if (dd == 0) { a.part_m[slot] = M; a.part_l[slot] = L; }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Encoder self-attention with wave-private staging ("pw"): the arithmetic of flash_attention_kernel - same (head,
// 32-query tile) workgroups, wave w takes key tiles w, w+4, ..., same MFMA sequences, same online-softmax updates,
// same four-way merge, so the output is bit-identical to it - but no workgroup barrier inside the key loop.
// flash_attention_kernel stages 128 keys per trip with all 256 threads (global -> registers -> LDS) between two
// __syncthreads(), although each wave only ever reads ITS OWN 32-key tile back: a third of the kernel's time was that
// staging (scripts/probes/attn_ablation.hip).  Here every wave fetches its own tiles by LDS-DMA into a private 16 KiB
// region - K(t) while the previous tile's P.V runs, V(t) while Q.K^T(t) runs - so the only waits are the wave's own
// s_waitcnt vmcnt(0) at the two phase changes, and the four waves of a workgroup drift freely.  The DMA writes
// lane-linearly; rows are 256 bytes, so the bank swizzle (16-byte chunk c of key row r stored at chunk c ^ (r & 15)) is
// applied to the source address and to the reads.  Fragment reads are inline asm: hipcc would drain the DMA queue in
// front of every LDS read that follows an LDS-DMA.  Q fragments live in registers.  64 KiB of LDS: two workgroups per
// CU, as before.
// ---------------------------------------------------------------------------------------------------------------------
typedef float f32x4a __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void enc_attention_pw_kernel(FlashArgs a) {
    WLK_PIN_FLASH_ARGS(a);
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const bool dbg = a.dbg_clock != nullptr;
    const long long t_start = dbg ? (long long)__builtin_readcyclecounter() : 0;
    long long w_k = 0, w_v = 0, t_soft = 0;
    const bool batched = a.batch > 0;
    const float* const aq = batched ? table_at(a.z.in, blockIdx.y) : a.q;
    const float* const ak = batched ? aq + a.z_k_off : a.k;
    const float* const av = batched ? aq + a.z_v_off : a.v;
    float* const aout = batched ? table_at(a.z.out, blockIdx.y) : a.out;
    const int T = a.Tk, n_head = a.n_head;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.x % n_head;       // head == XCD for 8 heads: K/V of a head stay in one L2
    const int q_tiles = (a.Tq + QT - 1) / QT;
    const int qt_idx = (blockIdx.x / n_head) % q_tiles;
    const int ks = blockIdx.x / (n_head * q_tiles);   // key-range split (0 when a.k_splits == 1)
    const int q0 = qt_idx * QT;
    const long ld = a.ldkv;
    const int half = lane >> 5, lq = lane & 31;

    float4 qf[8];                               // Q[q0 + lq][8 g + 4 half .. +4]
    {
        const bool ok = q0 + lq < a.Tq;
        const float* qp = aq + (long)(ok ? q0 + lq : 0) * a.ldq + head * 64 + half * 4;
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

    // private regions of this wave: K tile [32 keys][16 chunks of 16 bytes], V tile the same
    float* Kb = lds + wave * 4096;
    float* Vb = Kb + 2048;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
    const unsigned kb_addr = lds_base + (unsigned)wave * 16384u, vb_addr = kb_addr + 8192u;
    // DMA piece p (0..7) of a tile = key rows 4p .. 4p+3; lane l fills slot (row 4p + (l >> 4), chunk l & 15) with source
    // chunk (l & 15) ^ (row & 15)
    const int prow = lane >> 4;
    const float* kbase = ak + head * 64;
    const float* vbase = av + head * 64;
    auto dma_piece = [&](const float* base, float* dst, int key0, int p) {
        const int row = 4 * p + prow;
        const int chunk = (lane & 15) ^ (row & 15);
        const float* src = base + (long)min(key0 + row, T - 1) * ld + chunk * 4;   // rows past T: clamped, masked below
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + p * 256), 16, 0, 0);
    };
    // K fragment of group g: row lq, chunk 2g + half  ->  stored chunk (2g) ^ (half ^ (lq & 15))
    const unsigned k_row = kb_addr + (unsigned)lq * 256u;
    const unsigned k_h16 = (unsigned)(half ^ (lq & 15)) << 4;
    // V values of key kn = rs + 4 half (rs = (r & 3) + 8 (r >> 2)): d = lq and d = lq + 32
    const unsigned v_base = vb_addr + (unsigned)half * 1024u + (unsigned)(lq & 3) * 4u;
    const unsigned v_c0 = (unsigned)((lq >> 2) ^ (4 * half));
    auto v_addr = [&](int rs, int hi) { return v_base + (unsigned)rs * 256u + (((v_c0 ^ (unsigned)(hi * 8)) ^ (unsigned)(rs & 15)) << 4); };

    // key tiles [t_lo, t_hi) of this workgroup: whole groups of four tiles per split, so that a split's four streams
    // are tiles t_lo + w, t_lo + w + 4, ...
    const int n_tiles_all = (T + KT - 1) / KT;
    const int per_split = ((n_tiles_all + a.k_splits - 1) / a.k_splits + NWAVE - 1) / NWAVE * NWAVE;
    const int t_lo = ks * per_split;
    const int n_tiles = min(n_tiles_all, t_lo + per_split);
    const bool live = q0 < a.Tq;
    if (live && t_lo + wave < n_tiles) {
#pragma unroll
        for (int p = 0; p < 8; ++p) dma_piece(kbase, Kb, (t_lo + wave) * KT, p);
    }
    const long long t_loop = dbg ? (long long)__builtin_readcyclecounter() : 0;
    for (int t = t_lo + wave; live && t < n_tiles; t += NWAVE) {
        const int key0 = t * KT;
        long long tw = dbg ? (long long)__builtin_readcyclecounter() : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // K(t) has landed (the only DMA this wave has in flight)
        if (dbg) w_k += (long long)__builtin_readcyclecounter() - tw;
        // ---- S^T = K Q^T, with the eight pieces of V(t) dealt between the MFMAs
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
        f32x4a kc, kn;
        asm volatile("ds_read_b128 %0, %1" : "=v"(kc) : "v"(k_row + k_h16));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kc));
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + 1 < 8) asm volatile("ds_read_b128 %0, %1" : "=v"(kn) : "v"(k_row + (k_h16 ^ (unsigned)((g + 1) * 32))));
            __builtin_amdgcn_sched_barrier(0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[0], qf[g].x, s, 0, 0, 0);
            dma_piece(vbase, Vb, key0, g);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[1], qf[g].y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[2], qf[g].z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[3], qf[g].w, s, 0, 0, 0);
            if (g + 1 < 8) {
                __builtin_amdgcn_sched_barrier(0);      // the wait stays BEHIND the group's MFMAs: they cover the read's latency
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kn));
                kc = kn;
            }
        }
        // ---- online softmax (flash_attention_kernel's, verbatim)
        tw = dbg ? (long long)__builtin_readcyclecounter() : 0;
        float mt = -INFINITY;
        if (key0 + KT > T) {        // only the last tile has keys past the end (wave-uniform)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (key >= T) s[r] = -INFINITY;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[r]);
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
        // once the running maxima have settled alpha is exactly 1.0f in every lane and the rescale (32 multiplies through
        // the accumulator file) is the identity: skipped then, bit for bit the same result
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
        }
        // ---- O^T += V P^T, with the eight pieces of K(t + 4) dealt between the MFMAs
        if (dbg) { const long long n = (long long)__builtin_readcyclecounter(); t_soft += n - tw; tw = n; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // V(t) has landed
        if (dbg) w_v += (long long)__builtin_readcyclecounter() - tw;
        const bool more = t + NWAVE < n_tiles;
        float v0, v1, v0n, v1n;
        asm volatile("ds_read_b32 %0, %1" : "=v"(v0) : "v"(v_addr(0, 0)));
        asm volatile("ds_read_b32 %0, %1" : "=v"(v1) : "v"(v_addr(0, 1)));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r + 1 < 16) {
                const int rsn = ((r + 1) & 3) + 8 * ((r + 1) >> 2);
                asm volatile("ds_read_b32 %0, %1" : "=v"(v0n) : "v"(v_addr(rsn, 0)));
                asm volatile("ds_read_b32 %0, %1" : "=v"(v1n) : "v"(v_addr(rsn, 1)));
            }
            __builtin_amdgcn_sched_barrier(0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[r], o0, 0, 0, 0);
            if (more && (r & 1) == 0) dma_piece(kbase, Kb, key0 + NWAVE * KT, r >> 1);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[r], o1, 0, 0, 0);
            if (r + 1 < 16) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0n), "+v"(v1n));
                v0 = v0n;
                v1 = v1n;
            }
        }
    }

    // merge the four key-stream states through LDS: flash_attention_kernel's (every wave's DMA is drained: a wave leaves
    // the loop with nothing in flight, and the barrier comes with hipcc's vmcnt(0))
    const long long t_merge = dbg ? (long long)__builtin_readcyclecounter() : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
                const float mw = Ms[w * QT + q];
                const float e = mw == -INFINITY ? 0.f : expf(mw - M);     // a stream without keys (short inputs / empty ranges)
                L += e * Ls[w * QT + q];
                acc += e * Os[(w * QT + q) * O_LD + dd];
            }
            if (a.k_splits == 1) {
                if (qrow < a.Tq) aout[(long)qrow * a.ldo + head * 64 + dd] = acc / L;
            } else if (qrow < a.Tq) {
                // partial state of this key range (flash_attention_kernel's layout; folded by flash_merge_kernel)
                float* const part_o = batched ? const_cast<float*>(table_at(a.z.res, blockIdx.y)) : a.part_o;
                float* const part_m = batched ? part_o + (size_t)a.Tq * n_head * a.k_splits * 64 : a.part_m;
                float* const part_l = batched ? part_m + (size_t)a.Tq * n_head * a.k_splits : a.part_l;
                const long slot = ((long)qrow * n_head + head) * a.k_splits + ks;
                part_o[slot * 64 + dd] = acc;
                if (dd == 0) { part_m[slot] = M; part_l[slot] = L; }
            }
        }
    }
    if (dbg && tid == 0) {
        long long* d = a.dbg_clock + 8 * ((long)blockIdx.y * gridDim.x + blockIdx.x);
        d[0] = t_start; d[1] = t_loop; d[2] = t_merge; d[3] = (long long)__builtin_readcyclecounter();
        d[4] = w_k; d[5] = w_v; d[6] = t_soft; d[7] = 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Encoder self-attention, 64 queries per workgroup WITH the 32-query kernel's arithmetic ("q64x"): eight waves =
// two query sub-tiles x the same four key streams (stream w = key tiles w, w+4, w+8, ...), the four K/V tiles of an
// iteration staged once in LDS and read by both sub-tiles (half the global -> LDS traffic per MFMA, which
// scripts/probes/attn_ablation.hip shows to be a third of the 32-query kernel's time), Q fragments in registers.
// Every (query tile, stream) runs exactly the MFMA sequence, online-softmax updates and four-way merge of
// flash_attention_kernel, so the output is bit-identical to it - unlike the key-split q64 kernel below, which regroups
// the keys.  192 workgroups of 512 threads for base.en (one per CU on 192 CUs, two waves per SIMD).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void enc_attention_q64x_kernel(FlashArgs a) {
    WLK_PIN_FLASH_ARGS(a);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const bool batched = a.batch > 0;
    const float* const aq = batched ? table_at(a.z.in, blockIdx.y) : a.q;
    const float* const ak = batched ? aq + a.z_k_off : a.k;
    const float* const av = batched ? aq + a.z_v_off : a.v;
    float* const aout = batched ? table_at(a.z.out, blockIdx.y) : a.out;
    const int T = a.Tk, n_head = a.n_head;
    float* Ks = lds;                            // [NWAVE][KT][K_LD]
    float* Vs = lds + NWAVE * KT * K_LD;        // [NWAVE][KT][64]
    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
    const int qs = wave8 >> 2, wave = wave8 & 3;      // query sub-tile, key stream
    const int head = blockIdx.x % n_head;
    const int qt_idx = blockIdx.x / n_head;
    const int q0 = qt_idx * (2 * QT) + qs * QT;
    const long ld = a.ldkv;
    const int half = lane >> 5, lq = lane & 31;

    float4 qf[8];                               // Q[q0 + lq][8 g + 4 half .. +4] (what the 32-query kernel reads from LDS)
    {
        const bool ok = q0 + lq < a.Tq;
        const float* qp = aq + (long)(ok ? q0 + lq : 0) * a.ldq + head * 64 + half * 4;
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
    const float* kbase = ak + head * 64;
    const float* vbase = av + head * 64;
    float4 rk[4], rv[4];                        // 128 keys x 16 float4 for K and for V over 512 threads
    auto fetch = [&](int it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 512 * i;
            const int key = it * (NWAVE * KT) + (idx >> 4);
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
            const int idx = tid + 512 * i;
            const int kl = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<float4*>(&Ks[(kl >> 5) * (KT * K_LD) + (kl & 31) * K_LD + c4 * 4]) = rk[i];
            *reinterpret_cast<float4*>(&Vs[kl * 64 + c4 * 4]) = rv[i];
        }
    };
    const int n_iter = (T + NWAVE * KT - 1) / (NWAVE * KT);
    fetch(0);
    const float* Kw = Ks + wave * (KT * K_LD) + lq * K_LD + half * 4;
    const float* Vw = Vs + wave * (KT * 64) + lq;
    const bool live = q0 < a.Tq;                // the second sub-tile of the last query tile may be empty
    for (int it = 0; it < n_iter; ++it) {
        __syncthreads();
        stash();
        __syncthreads();
        if (it + 1 < n_iter) fetch(it + 1);
        const int key0 = it * (NWAVE * KT) + wave * KT;
        if (key0 < T && live) {                 // wave-uniform
            f32x16 s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
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
        }
    }
    // merge of the four streams of each sub-tile: flash_attention_kernel's, once per sub-tile
    __syncthreads();
    float* Os = lds;                               // [2 * NWAVE][QT][O_LD]
    float* Ms = lds + 2 * NWAVE * QT * O_LD;       // [2 * NWAVE][QT]
    float* Ls = Ms + 2 * NWAVE * QT;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dd = (r & 3) + 8 * (r >> 2) + 4 * half;
        Os[(wave8 * QT + lq) * O_LD + dd] = o0[r];
        Os[(wave8 * QT + lq) * O_LD + 32 + dd] = o1[r];
    }
    if (half == 0) {
        Ms[wave8 * QT + lq] = m_run;
        Ls[wave8 * QT + lq] = l_run;
    }
    __syncthreads();
    {
        const int dd = tid & 63;
        const int qg = tid >> 6;                   // 0..7: eight queries each
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q64 = qg * 8 + i;            // 0..63
            const int sub = q64 >> 5, q = q64 & 31;
            const int qrow = qt_idx * (2 * QT) + q64;
            const int wb = sub * NWAVE;
            float M = Ms[wb * QT + q];
#pragma unroll
            for (int w = 1; w < NWAVE; ++w) M = fmaxf(M, Ms[(wb + w) * QT + q]);
            float L = 0.f, acc = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) {
                const float e = expf(Ms[(wb + w) * QT + q] - M);
                L += e * Ls[(wb + w) * QT + q];
                acc += e * Os[((wb + w) * QT + q) * O_LD + dd];
            }
            if (qrow < a.Tq) aout[(long)qrow * a.ldo + head * 64 + dd] = acc / L;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Encoder self-attention, second form: 64 queries per workgroup, K/V tiles shared by two waves, balanced key splits.
//
// Where the 32-query kernel above loses its time on the 1500 x 1500 x 8-head problem (scripts/probes/attn_ablation.hip,
// MI355X): of 97 us, 32 us are the global -> LDS traffic of K and V (every one of the 376 workgroups re-reads the
// 768 KB of its head: 289 MB per launch, 16 flops per byte), 11 us are grid quantisation (376 workgroups of 76 KB LDS
// = 2 per CU on 120 CUs and 1 on the rest: the loaded CUs take twice as long), 9 us the softmax arithmetic that a
// second wave per SIMD would cover, 11 us prologue / epilogue.  Here
//   * a workgroup owns 64 queries: wave (qs, kt) multiplies query sub-tile qs with key tile kt of the staged PAIR of
//     key tiles, so every K/V byte that reaches LDS feeds two waves (half the traffic), Q fragments stay in registers;
//   * LDS per workgroup is 34 KB (two K tiles + two V tiles; the merge area overlays them), ~150 VGPRs: 3 workgroups
//     per CU, 3 waves per SIMD - softmax and staging of one wave hide behind the MFMAs of the other two;
//   * the key-tile pairs of a (query tile, head) are dealt round-robin over k_splits workgroups so that the grid is a
//     whole number of "3 per CU" rounds (base: 24 x 8 x 4 = 768 = 3 x 256); the k_splits partial softmax states are
//     folded by flash_merge_kernel in a fixed order (deterministic).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int QT2 = 64;
constexpr int kAttn2Stage = 2 * KT * K_LD + 2 * KT * 64;                 // floats: K pair + V pair
constexpr int kAttn2Merge = NWAVE * QT * O_LD + 2 * NWAVE * QT;          // floats: O, m, l of the four waves
constexpr int kAttn2LdsFloats = kAttn2Stage > kAttn2Merge ? kAttn2Stage : kAttn2Merge;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void enc_attention_q64_kernel(FlashArgs a) {
    WLK_PIN_FLASH_ARGS(a);
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

    const float* kbase = ak + head * 64;
    const float* vbase = av + head * 64;
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
    if (ks < n_pairs) fetch(ks);
    for (int pair = ks; pair < n_pairs; pair += a.k_splits) {
        __syncthreads();                      // the previous pair's LDS reads are done
        stash();
        __syncthreads();
        if (pair + a.k_splits < n_pairs) fetch(pair + a.k_splits);
        const int key0 = pair * (2 * KT) + kt * KT;
        if (key0 < T) {                       // wave-uniform
            f32x16 s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
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

// folds the k_splits partial softmax states of every (query row, head): out = sum_s e^{m_s-M} O_s / sum_s e^{m_s-M} l_s
__global__ __launch_bounds__(64) void flash_merge_kernel(FlashArgs a) {
    WLK_PIN_FLASH_ARGS(a);
    const int row = blockIdx.x, head = blockIdx.y, dd = threadIdx.x;
    const bool batched = a.batch > 0;
    float* const aout = batched ? table_at(a.z.out, blockIdx.z) : a.out;
    const float* const part_o = batched ? table_at(a.z.res, blockIdx.z) : a.part_o;
    const float* const part_m = batched ? part_o + (size_t)a.Tq * a.n_head * a.k_splits * 64 : a.part_m;
    const float* const part_l = batched ? part_m + (size_t)a.Tq * a.n_head * a.k_splits : a.part_l;
    const long base = ((long)row * a.n_head + head) * a.k_splits;
    float M = -INFINITY;
    for (int s = 0; s < a.k_splits; ++s) M = fmaxf(M, part_m[base + s]);
    float L = 0.f, acc = 0.f;
    for (int s = 0; s < a.k_splits; ++s) {
        const float ms = part_m[base + s];
        const float f = ms == -INFINITY ? 0.f : expf(ms - M);
        L += part_l[base + s] * f;
        acc += part_o[(base + s) * 64 + dd] * f;
    }
    aout[(long)row * a.ldo + head * 64 + dd] = acc / L;
}

// key splits of the 64-query kernel: the split count whose grid is closest to whole rounds of 3 workgroups per CU
int enc_attention_q64_splits(int T, int n_head, int batch) {
    static const int forced = [] {
        const char* e = getenv("WLK_ENC_KSPLIT");
        return e ? atoi(e) : 0;
    }();
    if (forced > 0) return std::min(forced, 8);
    const long units = (long)((T + QT2 - 1) / QT2) * n_head * std::max(batch, 1);
    const int n_pairs = (T + 2 * KT - 1) / (2 * KT);
    // measured on MI355X (T = 1500): 192 units (8 heads) 88 us unsplit, 70 us with 4 splits (768 workgroups = three
    // per CU) incl. the merge kernel; 480 units (20 heads) 139 us unsplit vs 151-197 us split - once the grid already
    // fills the chip about twice over, the partial-state traffic and the merge launch cost more than the balance wins
    if (units >= 400) return 1;
    int best = 1;
    double best_eff = 0.0;
    for (int ks = 1; ks <= 6 && ks <= n_pairs; ++ks) {
        const double rounds = (double)units * ks / 768.0;
        const double eff = rounds / std::ceil(rounds) - 0.03 * (ks - 1);     // each extra split costs merge traffic
        if (eff > best_eff + 1e-9) { best_eff = eff; best = ks; }
    }
    return best;
}

constexpr int kAttnXLdsFloats = (NWAVE * KT * K_LD + NWAVE * KT * 64) > (2 * NWAVE * QT * O_LD + 4 * NWAVE * QT)
                                    ? (NWAVE * KT * K_LD + NWAVE * KT * 64)
                                    : (2 * NWAVE * QT * O_LD + 4 * NWAVE * QT);

static void launch_enc_q64x(const LaunchCtx& ctx, const FlashArgs& a, const char* tag) {
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    WLK_HIP(hipGetDevice(&dev));
    const size_t lds = kAttnXLdsFloats * sizeof(float);
    if (dev < 64 && !attr_set[dev].load(std::memory_order_acquire)) {
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(enc_attention_q64x_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev].store(true, std::memory_order_release);
    }
    const int q_tiles = (a.Tq + 2 * QT - 1) / (2 * QT);
    const double nb = std::max(a.batch, 1);
    KernelScope ks(ctx, tag, nb * 4.0 * a.Tq * (double)a.Tk * 64.0 * a.n_head,
                   nb * 4.0 * 64.0 * a.n_head * (2.0 * a.Tq + 2.0 * a.Tk));
    hipLaunchKernelGGL(enc_attention_q64x_kernel, dim3(q_tiles * a.n_head, std::max(a.batch, 1)), dim3(512), lds, ctx.stream, a);
    WLK_HIP(hipGetLastError());
}

// Key-range splits of the wave-private kernel.  376 workgroups (base.en: 47 query tiles x 8 heads) on 256 CUs leave 136
// CUs idle for half of the launch (two co-resident workgroups share their SIMDs' matrix pipes, so a CU with two takes
// twice as long as a CU with one); with two key ranges per (query tile, head) there are 752 workgroups of half the
// length, 2.94 per CU.  Costs: the partial states go through memory and a merge launch follows, and the sum over keys
// is grouped as (range 0) + (range 1): not bit-identical to the unsplit kernel (the merge kernel's arithmetic, as in the
// decoder's split cross-attention).  Depends on (T, heads) only - never on the batch.
int enc_attention_pw_splits(int T, int n_head) {
    static const int forced = [] {
        const char* e = getenv("WLK_ENC_KSPLIT");
        return e ? atoi(e) : 0;
    }();
    if (forced > 0) return std::min(forced, 4);
    // Measured (profiles/r03_attention_ab.txt): base.en back to back 73.5 us unsplit, 69.6 with two ranges incl. the merge
    // launch - and NOTHING inside the encoder chain (75.7-76.1 vs 74.8-76.1 us per launch, 145.6 vs 145.4 audio-s/s): the
    // extra launch boundary and the partial states eat the balance.  So the default stays unsplit, which also keeps the
    // kernel bit-identical to flash_attention_kernel; WLK_ENC_KSPLIT=2 is the A/B switch.
    (void)T; (void)n_head;
    return 1;
}

static void launch_enc_pw(const LaunchCtx& ctx, const FlashArgs& a, const char* tag) {
    const size_t lds = 64 * 1024;            // 4 waves x (8 KiB K + 8 KiB V); the merge (34 KiB) reuses it
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    WLK_HIP(hipGetDevice(&dev));
    if (dev < 64 && !attr_set[dev].load(std::memory_order_acquire)) {
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(enc_attention_pw_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev].store(true, std::memory_order_release);
    }
    const int q_tiles = (a.Tq + QT - 1) / QT;
    const double nb = std::max(a.batch, 1);
    KernelScope ks(ctx, tag, nb * 4.0 * a.Tq * (double)a.Tk * 64.0 * a.n_head,
                   nb * 4.0 * 64.0 * a.n_head * (2.0 * a.Tq + 2.0 * a.Tk));
    FlashArgs b = a;
    b.dbg_clock = g_attn_dbg_clock;
    hipLaunchKernelGGL(enc_attention_pw_kernel, dim3(q_tiles * a.n_head * a.k_splits, std::max(a.batch, 1)), dim3(256), lds,
                       ctx.stream, b);
    WLK_HIP(hipGetLastError());
    if (a.k_splits > 1) {
        hipLaunchKernelGGL(flash_merge_kernel, dim3(a.Tq, a.n_head, std::max(a.batch, 1)), dim3(64), 0, ctx.stream, a);
        WLK_HIP(hipGetLastError());
    }
}

static void launch_enc_q64(const LaunchCtx& ctx, FlashArgs a, const char* tag) {
    const size_t lds = kAttn2LdsFloats * sizeof(float);     // 34 KB: below the 64 KB default limit, no attribute needed
    const int q_tiles = (a.Tq + QT2 - 1) / QT2;
    const double nb = std::max(a.batch, 1);
    KernelScope ks(ctx, tag, nb * 4.0 * a.Tq * (double)a.Tk * 64.0 * a.n_head,
                   nb * 4.0 * 64.0 * a.n_head * (2.0 * a.Tq + 2.0 * a.Tk));
    hipLaunchKernelGGL(enc_attention_q64_kernel, dim3(q_tiles * a.n_head * a.k_splits, std::max(a.batch, 1)), dim3(256), lds,
                       ctx.stream, a);
    WLK_HIP(hipGetLastError());
    if (a.k_splits > 1) {
        hipLaunchKernelGGL(flash_merge_kernel, dim3(a.Tq, a.n_head, std::max(a.batch, 1)), dim3(64), 0, ctx.stream, a);
        WLK_HIP(hipGetLastError());
    }
}

static void launch_flash(const LaunchCtx& ctx, const FlashArgs& a, const char* tag) {
    // sessions launch from different host threads: the per-device "attribute set" marks are atomics (the call itself
    // is idempotent, so a benign double set is all a race can cause)
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    WLK_HIP(hipGetDevice(&dev));
    const size_t lds = kAttnLdsTotal * sizeof(float);
    if (dev < 64 && !attr_set[dev].load(std::memory_order_acquire)) {
        WLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attention_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev].store(true, std::memory_order_release);
    }
    if (a.head_rank && a.Tk % 4 != 0) throw std::invalid_argument("flash attention: score dump needs Tk % 4 == 0");
    const int q_tiles = (a.Tq + QT - 1) / QT;
    // QK^T and PV: 2 * Tq*Tk*64 MACs per head each; reads q,k,v once, writes out
    const double nb = std::max(a.batch, 1);
    KernelScope ks(ctx, tag, nb * 4.0 * a.Tq * (double)a.Tk * 64.0 * a.n_head,
                   nb * 4.0 * 64.0 * a.n_head * (2.0 * a.Tq + 2.0 * a.Tk));
    if (a.batch > 0 && (a.k_splits > 1 || a.head_rank)) throw std::invalid_argument("flash attention: plain form only when batched");
    // no empty key range: a range is whole 128-key iterations, so 1500 keys are at most 12 ranges (a range without keys
    // would hand the merge a (-inf, 0) state to fold)
    FlashArgs b = a;
    if (b.k_splits > 1) {
        const int n_iter_all = (b.Tk + NWAVE * KT - 1) / (NWAVE * KT);
        const int it_per = (n_iter_all + b.k_splits - 1) / b.k_splits;
        b.k_splits = (n_iter_all + it_per - 1) / it_per;
    }
    hipLaunchKernelGGL(flash_attention_kernel, dim3(q_tiles * b.n_head * b.k_splits, std::max(b.batch, 1)), dim3(256), lds,
                       ctx.stream, b);
    WLK_HIP(hipGetLastError());
    if (b.k_splits > 1) {
        hipLaunchKernelGGL(flash_merge_kernel, dim3(b.Tq, b.n_head), dim3(64), 0, ctx.stream, b);
        WLK_HIP(hipGetLastError());
    }
}

size_t flash_split_scratch_floats(int rows, int n_head, int k_splits) {
    return (size_t)rows * n_head * k_splits * (64 + 2);
}

// WLK_ENC_ATTN selects the encoder self-attention kernel.  Default "lds" = the 32-query kernel.  Two 64-query forms are
// kept as opt-in A/B variants: "q64x" (eight waves, the 32-query kernel's arithmetic, bit-identical output: 80.2 -> 74.5
// us per launch back to back) and "q64" (balanced key splits + merge: 70 us, but it regroups the key sums - one exact-tie
// AlignAtt arg-max of the 8 pinned bench streams then falls the other way).  Inside the real launch chain neither beats
// the default (same box, alternating runs: 135.0 / 134.1 audio-s/s for one stream, 273 / 268 / 262 for 8 streams with
// lds / q64x / q64): their wins are L2-warm effects of timing one kernel in a loop.  A register-fed variant without
// LDS staging was slower everywhere and is gone.
static int enc_attention_variant() {
    static const int v = [] {
        const char* e = getenv("WLK_ENC_ATTN");
        if (e && e[0] == 'q') return (e[1] == '6' && e[2] == '4' && e[3] == 'x') ? 3 : 2;
        if (e && e[0] == 'l') return 0;      // "lds": the barrier-staged 32-query kernel (round 1 / 2 default)
        return 4;                            // "pw": wave-private LDS-DMA staging, bit-identical to "lds"
    }();
    return v;
}

void launch_encoder_attention(const LaunchCtx& ctx, const float* qkv, float* out, int T, int d, int n_head,
                              int k_splits, float* split_scratch) {
    FlashArgs a;
    a.q = qkv; a.ldq = 3L * d; a.k = qkv + d; a.v = qkv + 2 * d; a.ldkv = 3L * d; a.out = out; a.ldo = d;
    a.Tq = T; a.Tk = T; a.n_head = n_head;
    if (k_splits > 1 && split_scratch) {   // key ranges on separate workgroups + merge: evens out the 376-tile grid
        a.k_splits = k_splits;
        a.part_o = split_scratch;
        a.part_m = split_scratch + (size_t)T * n_head * k_splits * 64;
        a.part_l = a.part_m + (size_t)T * n_head * k_splits;
    }
    const int variant = enc_attention_variant();
    if (variant == 2 && split_scratch) {
        a.k_splits = enc_attention_q64_splits(T, n_head, 1);
        a.part_o = split_scratch;
        a.part_m = split_scratch + (size_t)T * n_head * a.k_splits * 64;
        a.part_l = a.part_m + (size_t)T * n_head * a.k_splits;
        launch_enc_q64(ctx, a, "enc_attention");
    } else if (variant == 3) {
        launch_enc_q64x(ctx, a, "enc_attention");
    } else if (variant == 4) {
        a.k_splits = split_scratch ? enc_attention_pw_splits(T, n_head) : 1;
        if (a.k_splits > 1) {
            a.part_o = split_scratch;
            a.part_m = split_scratch + (size_t)T * n_head * a.k_splits * 64;
            a.part_l = a.part_m + (size_t)T * n_head * a.k_splits;
        }
        launch_enc_pw(ctx, a, "enc_attention");
    } else {
        launch_flash(ctx, a, "enc_attention");
    }
}

void launch_encoder_attention_batched(const LaunchCtx& ctx, const PtrTable& z, int batch, int T, int d, int n_head) {
    if (batch < 1 || batch > kMaxBatch) throw std::invalid_argument("encoder attention: bad batch");
    FlashArgs a;
    a.ldq = 3L * d; a.ldkv = 3L * d; a.ldo = d; a.Tq = T; a.Tk = T; a.n_head = n_head;
    a.batch = batch; a.z = z; a.z_k_off = d; a.z_v_off = 2L * d;
    bool have_scratch = true;            // z.res[i] = the session's split scratch (flash_split_scratch_floats(T, H, 6))
    for (int i = 0; i < batch; ++i) have_scratch &= z.res[i] != nullptr;
    if (enc_attention_variant() == 3) {
        launch_enc_q64x(ctx, a, "enc_attention");
    } else if (enc_attention_variant() == 4) {
        a.k_splits = have_scratch ? enc_attention_pw_splits(T, n_head) : 1;   // one session's split count, whatever the batch
        launch_enc_pw(ctx, a, "enc_attention");
    } else if (enc_attention_variant() == 2 && have_scratch) {
        // the split count of ONE session, whatever the batch: a session's arithmetic (and its rounding) must not depend
        // on who else encodes at the same time; B sessions are B whole copies of a balanced grid anyway
        a.k_splits = enc_attention_q64_splits(T, n_head, 1);
        launch_enc_q64(ctx, a, "enc_attention");
    } else {
        launch_flash(ctx, a, "enc_attention");
    }
}

void launch_prefill_cross_attention(const LaunchCtx& ctx, const FlashArgs& a) { launch_flash(ctx, a, "dec_cross_attention_prefill"); }

// in-place softmax of the raw alignment-head scores the prefill dumped: one workgroup per
// (query row, alignment rank) - exactly F.softmax(qk, dim=-1) of simul_whisper.py:411
__global__ __launch_bounds__(256) void ring_softmax_kernel(float* ring, const int* ring_row, const int* beam_of_row,
                                                           const int* ranks, int ring_rows, int n_beam, int T) {
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    const int rank = ranks[blockIdx.y];
    float* x = ring + (((long)rank * n_beam + beam_of_row[row]) * ring_rows + ring_row[row]) * (long)T;
    float mx = -INFINITY;
    for (int j = tid; j < T; j += 256) mx = fmaxf(mx, x[j]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < T; j += 256) {
        const float e = expf(x[j] - mx);
        x[j] = e;
        sum += e;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    for (int j = tid; j < T; j += 256) x[j] = x[j] / sum;
}

void launch_ring_softmax(const LaunchCtx& ctx, float* ring, const int* ring_row, const int* beam_of_row,
                         const int* ranks_dev, int n_ranks, int rows, int ring_rows, int n_beam, int T) {
    if (n_ranks <= 0 || rows <= 0) return;
    KernelScope ks(ctx, "align_ring_softmax", 0.0, 4.0 * 2.0 * rows * (double)n_ranks * T);
    hipLaunchKernelGGL(ring_softmax_kernel, dim3(rows, n_ranks), dim3(256), 0, ctx.stream, ring, ring_row,
                       beam_of_row, ranks_dev, ring_rows, n_beam, T);
    WLK_HIP(hipGetLastError());
}

}  // namespace wlk

// Internal declarations shared by the kernel files and the C-ABI layer.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <stdexcept>
#include <mutex>
#include <string>

namespace wlk {

constexpr int kNFft = 400;
constexpr int kHop = 160;
constexpr int kNFreq = 201;
constexpr int kMelFrames = 3000;     // frames fed to the encoder (30 s)
constexpr int kPadSamples = 480000;  // zero padding the streaming path appends (simul_whisper.py:346)
constexpr int kHeadDim = 64;         // every Whisper size uses 64-wide heads
constexpr int kAlignWindow = 16;     // decode steps kept for AlignAtt (align_att_base.py:223)

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

inline void hip_check(hipError_t e, const char* what, const char* file, int line) {
    if (e != hipSuccess) {
        throw HipError(std::string(what) + " failed: " + hipGetErrorString(e) + " (" + file + ":" +
                       std::to_string(line) + ")");
    }
}
#define WLK_HIP(expr) ::wlk::hip_check((expr), #expr, __FILE__, __LINE__)

// message returned by wlk_last_error() on the calling thread (api.hip owns the storage)
void set_last_error(const std::string& msg);

// optional per-kernel HIP-event timing (wlk_prof_begin/end)
struct Profiler;
// Blocking copies / fills that stay off the legacy (NULL) stream: hipMemcpy / hipMemset make the legacy stream depend on
// every other stream, and HIP refuses that ("would make the legacy stream depend on a capturing blocking stream") while
// ANOTHER thread is capturing one of the step graphs - which sessions of other threads do at any time.  These run on a
// per-device utility stream and wait for it (api.hip).
void copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
void memset_sync(void* dst, int value, size_t bytes);

struct LaunchCtx {
    hipStream_t stream = nullptr;
    Profiler* prof = nullptr;
};
// flops / bytes are the ALGORITHMIC work of the launch (what the roofline fraction is computed from)
void prof_before(const LaunchCtx& ctx, const char* name, double flops, double bytes);
void prof_after(const LaunchCtx& ctx);

struct KernelScope {  // RAII: brackets one launch with events when profiling is armed
    const LaunchCtx& ctx;
    KernelScope(const LaunchCtx& c, const char* name, double flops = 0.0, double bytes = 0.0) : ctx(c) {
        if (ctx.prof) prof_before(ctx, name, flops, bytes);
    }
    ~KernelScope() {
        if (ctx.prof) prof_after(ctx);
    }
};

// ---- cross-session batched decode steps (engine.hip) ---------------------------------------------
// One row of a batched single-token decode step = one beam-1 session.  The table lives in device memory and is
// rewritten before every step; kernels that touch per-session state (KV-cache append, self-/cross-attention, the
// alignment window and its read-out) index it by row instead of taking one session's pointers.
struct StepRow {
    float* kcache;           // the session's current self-attention K cache, layer 0 ([L][ctx][d], beam 1)
    float* vcache;
    const float* cross_kv;   // the session's cross-attention K|V of all layers, [T][L][2d]
    float* ring;             // the session's alignment window [n_align][ring_rows][T]
    int token;               // token fed in this step
    int offset;              // self-attention cache length before this step (= position of the fed token)
    int ring_row;            // alignment-window row this step's cross-attention rows go to
    int prefill_rows, n_single, newest_row, content_len;   // AlignArgs of this row's read-out
    int pad;
};

// ---- single-token step of one beam-1 session as ONE graph replay (api.hip: wlk_step_select) -----------
// The host writes a StepBlock into pinned (host-coherent) memory and replays the graph; the graph's first kernel reads
// the block over the bus, spreads it into device memory for the kernels behind it, and the last kernel writes a
// StepResult back into pinned memory, flags last.  No copy nodes, nothing between the decoder and the read-out.
constexpr int kStepMaxAdj = 160;   // multilingual models suppress ~110 ids on every step
struct StepBlock {
    StepRow row;                  // this step's scalars (+ the session's pointers, used by the alignment read-out)
    int n_adj;
    unsigned seq;                 // echoed in StepResult's flags
    int pad[2];
    int adj_row[kStepMaxAdj];     // logit adjustments applied before the top-k (wlk_select's contract)
    int adj_ids[kStepMaxAdj];
    float adj_deltas[kStepMaxAdj];
};
struct StepResult {
    float top_vals[2];
    int top_ids[2];
    int frame;
    unsigned flag_topk, flag_align;   // = StepBlock::seq once the fields above are visible to the host
    int pad;
};
struct StepHostOut {              // by-value kernel argument of the read-out kernels of a graph-replayed step
    StepResult* result = nullptr; // pinned, one entry per row; nullptr = device-only results
    const int* n_adj = nullptr;   // device scalars of the step: number of logit adjustments (overrides the launch
    const unsigned* seq = nullptr;   // argument when set), sequence number echoed in the result flags
};
// the same for a batched step of the engine (rows = sessions)
constexpr int kEngineFusedAdj = 1024;
struct EngineBlock {
    int n_adj;
    unsigned seq;
    int pad[2];
    StepRow rows[8];
    int adj_row[kEngineFusedAdj];
    int adj_ids[kEngineFusedAdj];
    float adj_deltas[kEngineFusedAdj];
};
void launch_embed_rows_step(const LaunchCtx& ctx, const EngineBlock* host_block, EngineBlock* dev_block, const float* tok_emb,
                            const float* pos_emb, float* x, int n_rows, int d);
// The Infinity-Cache prefetcher beside a single-token decode step (decoder.hip: mall_prefetch_step_kernel): per decoder layer the
// weight ranges it reads, and the mark the step's chain leaves for it
struct TouchRanges {
    const float* p[6];
    unsigned n16[6];                      // 16-byte units per range
};
struct ProgressMark {
    const unsigned* seq = nullptr;        // device copy of the step's sequence number (StepBlock::seq)
    unsigned long long* word = nullptr;   // (seq << 32 | layer), stored by the layer's self-attention launch; nullptr = no mark
    int layer = 0;
};
void launch_mall_prefetch_step(const LaunchCtx& ctx, const TouchRanges* table, int n_layer, unsigned seq,
                               const unsigned long long* progress, int lead, float* sink);
void launch_embed_step(const LaunchCtx& ctx, const StepBlock* host_block, StepBlock* dev_block, int* tokens_dev,
                       int* ring_row, int* beam_of_row, int* d_offset, const float* tok_emb, const float* pos_emb, float* x,
                       int d);

// ---- batched encodes (engine.hip) ------------------------------------------------------------------
// Concurrent encodes of several sessions run as ONE launch per operator with grid.y = sessions: the weights are
// shared, every session keeps its own activation buffers, and operand z of a launch comes from a by-value pointer
// table (the sessions' buffers are separate allocations, so there is no common stride).
constexpr int kMaxBatch = 8;
struct PtrTable {
    const float* in[kMaxBatch];     // A / x / qkv of session z
    float* out[kMaxBatch];          // C / y / attention output
    const float* res[kMaxBatch];    // GEMM residual operand
};
// Entry z of a table that lives in kernel arguments.  A runtime index into a by-value array would make the compiler
// copy the whole argument struct to scratch memory (measured: +50 % on the GEMM); z is wave-uniform, so a chain of
// scalar selects costs a handful of SALU instructions instead.
template <typename T>
__device__ __forceinline__ T table_at(T const (&t)[kMaxBatch], unsigned z) {
    T v = t[0];
#pragma unroll
    for (int i = 1; i < kMaxBatch; ++i) v = (z == (unsigned)i) ? t[i] : v;
    return v;
}

// ---- gemm_f32.hip ---------------------------------------------------------------------------
// C[M,N] = epilogue(A[M,K] * W[N,K]^T + bias[N]); fp32 MFMA (v_mfma_f32_32x32x2_f32).
enum GemmFlags : int {
    kGemmGelu = 1,      // exact-erf GELU after bias
    kGemmResidual = 2,  // += R[m*ldr + n] (after GELU); R may alias C
    kGemmScaleCols = 4, // columns n < scale_cols are multiplied by `scale` (after bias)
    kGemmRelu = 8,      // max(x, 0) after bias            (Sortformer sub-sampling / transformer FFN)
    kGemmSwish = 16,    // x * sigmoid(x) after bias       (Conformer feed-forward)
};
struct AlignArgs;
struct GemmArgs {
    // ---- what a kernel needs before its first operand load: one contiguous block at the start of the kernel-argument
    // segment, fetched by ONE s_load burst at kernel entry (WLK_PIN_GEMM_ARGS) instead of a stage per first use -------
    const float* A = nullptr;
    long lda = 0;
    const float* W = nullptr;  // [N][K] row-major
    const float* bias = nullptr;
    float* C = nullptr;
    long ldc = 0;
    const float* R = nullptr;
    long ldr = 0;
    int M = 0, N = 0, K = 0;
    int flags = 0;
    float scale = 1.f;
    int scale_cols = 0;
    int scale_period = 0;  // > 0: the rule is (col % scale_period) < scale_cols (several [k | v] blocks side by side)
    // batched encodes: > 0 -> grid.y = batch, operands of z from the table `z` below (A, C, R above are ignored)
    int batch = 0;
    long long* dbg_clock = nullptr;  // probe only: 4 s_memtime stamps per workgroup (start, loop start, loop end, end)
    // ---- the rest -----------------------------------------------------------------------------------------------------
    // GEMV path only (decode steps): fused pre-LayerNorm of the A rows (eps 1e-5) ...
    const float* ln_gamma = nullptr;
    const float* ln_beta = nullptr;
    // 16 x 16 k-wave kernel with the LayerNorm inside (round 6): the column-tile-0 workgroups also store the normalised rows
    // here ([M][ld_ln_out]) - for networks that use LN(x) twice (the Sortformer's post-LN Transformer: projection input AND residual)
    float* ln_out = nullptr;
    long ld_ln_out = 0;
    // ... and fused KV-cache append: output columns [kv_d, 2kv_d) / [2kv_d, 3kv_d) of row m are ALSO
    // written to kcache/vcache[(m*kv_ctx + *kv_pos) * kv_d + col] (one fed token per row)
    float* kcache = nullptr;
    float* vcache = nullptr;
    const int* kv_pos = nullptr;
    int kv_d = 0, kv_ctx = 0;
    int kv_ntok = 1;       // k-wave GEMM path: rows are [beam][kv_ntok] (decoder prefill)
    // batched steps (GEMV path): row m appends to kv_rows[m].kcache/vcache + kv_layer_off at position kv_rows[m].offset
    const StepRow* kv_rows = nullptr;
    long kv_layer_off = 0;
    // single-row GEMV (beam-1 decode step) whose A row is the cross-attention output still in split form: the merge
    // of the kCrossSplit partial softmax states (cross_merge_kernel's arithmetic) is the operand load, and the last
    // `mg_side_blocks` workgroups of the launch write the alignment heads' softmax rows into the alignment window
    const float* mg_pm = nullptr;        // [H][kCrossSplit]
    const float* mg_pl = nullptr;
    const float* mg_po = nullptr;        // [H][kCrossSplit][64]
    const float* mg_scores = nullptr;    // [H][T] raw scores of this layer
    const int* mg_head_rank = nullptr;   // [H] alignment rank or -1
    const int* mg_side_heads = nullptr;  // [mg_side_blocks] the layer's alignment heads (wlk_model::layer_heads)
    float* mg_ring = nullptr;
    const int* mg_ring_row = nullptr;
    const int* mg_beam_of_row = nullptr;
    int mg_heads = 0, mg_T = 0, mg_ring_rows = 0, mg_n_beam = 1, mg_side_blocks = 0;
    PtrTable z;
    // weight-streaming GEMV of a graph-replayed step (the vocabulary projection): the FIRST `side_blocks` workgroups of the
    // launch run the AlignAtt z-score (align_body.h) of *side_align - a device-resident AlignArgs whose `rows` name the
    // step's device block - beside the weight stream: (T + 63) / 64 = side_zf frame blocks per alignment head and beam
    const AlignArgs* side_align = nullptr;
    int side_blocks = 0, side_zf = 0;
    bool force_kwave = false;            // diagnostics: take the k-wave kernel (under-filled grids) whatever the shape
    int force_kernel = 0;                // diagnostics / A-B: 0 = by shape, 2 = k-wave, 3 = the 64x64 kernel, 4 = the k-split kernel
    bool gemm_plain_loop = false;        // A/B switch: LDS fragment reads right before use instead of a group ahead
};
// First statement of every kernel that takes GemmArgs by value: the hot block above is requested together and held in
// SGPRs from the entry block on.  Without it hipcc sinks each field's s_load into the branch that first reads it - the
// round-3 kernels waited for five to seven dependent scalar round trips before their first operand load.
#define WLK_PIN_S(x) asm volatile("" ::"s"(x))
#define WLK_PIN_GEMM_ARGS(g)                                                                                                \
    do {                                                                                                                    \
        asm volatile("" ::"s"((g).A), "s"((g).lda), "s"((g).W), "s"((g).bias), "s"((g).C), "s"((g).ldc), "s"((g).R),        \
                     "s"((g).ldr), "s"((g).M), "s"((g).N), "s"((g).K), "s"((g).flags), "s"((g).scale), "s"((g).scale_cols), \
                     "s"((g).scale_period), "s"((g).batch), "s"((g).dbg_clock));                                            \
        __builtin_amdgcn_sched_barrier(0); /* nothing of the kernel is scheduled between the loads and their one wait */    \
    } while (0)
void launch_gemm(const LaunchCtx& ctx, const GemmArgs& g, const char* tag);
// the "kp" family: one per-element arithmetic (the k-pipe kernel's) for every tile, so stacked rows keep their solo results
void launch_gemm_kp(const LaunchCtx& ctx, const GemmArgs& g, const char* tag);
bool gemm_kp_takes_kpipe(int M, int N, int K);
bool gemm_kp_fuses_layernorm(int M, int N, int K);   // launch_gemm_kp takes ln_gamma / ln_beta for this problem (rows < 512, K = 512)
bool gemm_takes_kwave(int M, int N, int K);
bool gemm_takes_ksplit(int M, int N, int K);  // ... for the one-tile-per-CU k-split kernel (encoder-sized problems)   // launch_gemm's shape rule for the k-wave kernel (under-filled grids)
// weight-streaming path for M <= 8 rows (decode steps); same contract as launch_gemm
void launch_gemv(const LaunchCtx& ctx, const GemmArgs& g, const char* tag);
inline int gemv_row_bucket(int M) { return M <= 1 ? 1 : M <= 2 ? 2 : M <= 4 ? 4 : 8; }
inline bool gemv_applicable(int M, int K) {
    return M <= 8 && (long)gemv_row_bucket(M) * K * 4 <= 64 * 1024 && K % 4 == 0;
}
// can launch_gemv take the cross-attention output in split form (single-row kernel, K = d)?
bool gemv1_folds_merge(int K);
void refresh_env_switches();                      // cached environment switches are read again at their next use (tests)
bool gemm_fuses_layernorm(int M, int N, int K);   // launch_gemm takes ln_gamma / ln_beta for this problem (prefill rows)
inline void launch_linear(const LaunchCtx& ctx, const GemmArgs& g, const char* tag) {
    if (gemv_applicable(g.M, g.K)) launch_gemv(ctx, g, tag);
    else launch_gemm(ctx, g, tag);
}

// ---- gemm_x3.hip / x3.h: fp32-accurate GEMM on the bf16 matrix cores (operands as three bf16 planes) ---------------
struct X3GemmArgs {
    const unsigned short* A3 = nullptr;   // activations in X3: row m at A3 + m * 3 * lda (bf16 units)
    long lda = 0;                         // in fp32-element units (a row holds 3 * lda bf16)
    const unsigned short* W3 = nullptr;   // weights [N][K] in the FRAGMENT-MAJOR X3 form (launch_x3_pack_w, x3_w_elems(N, K) bf16)
    const float* bias = nullptr;
    float* C = nullptr;                   // fp32 result
    long ldc = 0;
    const float* R = nullptr;
    long ldr = 0;
    int M = 0, N = 0, K = 0;
    int flags = 0;                        // kGemmGelu | kGemmResidual | kGemmScaleCols
    float scale = 1.f;
    int scale_cols = 0;
    int scale_period = 0;
    int batch = 0;                        // > 0: `batch` sessions in one launch; A3 = z.in[i] (X3), C = z.out[i], R = z.res[i]
    int walk_banded = 0;                  // set by launch_gemm_x3: the tile walk of the persistent workgroups (gemm_x3.hip) -
    int walk_colmajor = 0;                // slots of a band column by column instead of row by row
    int walk_slots = 0;                   // XCD bands or plain row-major order; slots per XCD (banded) or in total, sessions included
    // Result in the X3 format instead of fp32 (the operands of enc_attention_x3_kernel): columns [0, vt_col0) as X3 rows
    // (row m at C3 + m * 3 * ldc3), columns [vt_col0, N) TRANSPOSED - column n is row n - vt_col0 of a [N - vt_col0][vt_ld]
    // X3 matrix at C3 + vt_off whose chunks run along m; stored chunk u of every 32-row group holds rows 4 u .. 4 u + 3 and
    // 16 + 4 u .. 16 + 4 u + 3 (the keys whose probabilities a lane of the attention kernel holds).  Batched:
    // C3 = z.out[i] reinterpreted; the fp32 C is not written in this mode.
    unsigned short* C3 = nullptr;
    long ldc3 = 0;
    int vt_col0 = 0;
    long vt_off = 0;                      // in bf16 units, from C3
    long vt_ld = 0;                       // fp32-element units (multiple of 32, >= M)
    bool x3_out = false;
    PtrTable z;
};
bool gemm_x3_wide_applicable(int M, int N, int K, long lda);
void launch_gemm_x3(const LaunchCtx& ctx, const X3GemmArgs& g, const char* tag);
// fp32 weights [n][ld_src] (k used) -> the wide kernel's weight operand (fragment-major, rows padded to 32: gemm_x3.hip)
size_t x3_w_elems(int n, int k);             // bf16 elements of that operand
void launch_x3_pack_w(const LaunchCtx& ctx, const float* src, long ld_src, unsigned short* dst, int n, int k);
// fp32 [rows][ld_src] (cols used) -> X3 [rows][3 * ld_dst]
void launch_x3_pack(const LaunchCtx& ctx, const float* src, long ld_src, unsigned short* dst, long ld_dst, int rows, int cols);
// X3 [rows][3 * ld_src] -> fp32 [rows][ld_dst]: (hi + mid) + lo, the exact fp32 value the planes were split from
void launch_x3_unpack(const LaunchCtx& ctx, const unsigned short* src, long ld_src, float* dst, long ld_dst, int rows, int cols);

// ---- attention_x3.hip: encoder self-attention on the bf16 matrix cores at fp32 accuracy ---------------------------
// operand image of T rows of a d-wide model: [T][2 d] X3 rows (q | k) followed by V^T as [d][vt_ld] X3 rows
inline long x3_attn_vt_ld(int T) { return ((long)T + 31) / 32 * 32; }
inline long x3_attn_vt_off(int T, int d) { return (long)T * 3 * 2 * d; }                       // bf16 units
inline size_t x3_attn_image_elems(int T, int d) { return (size_t)x3_attn_vt_off(T, d) + (size_t)d * 3 * x3_attn_vt_ld(T); }
bool enc_attention_x3_enabled();
void launch_encoder_attention_x3(const LaunchCtx& ctx, const unsigned short* qk3, long ldqk, long vt_off, long vt_ld, float* out,
                                 long ldo, int T, int d, int n_head, const PtrTable* z, int batch, bool x3_out = false);
void launch_x3_pack_qkv(const LaunchCtx& ctx, const float* qkv, unsigned short* out, int T, int d, long vt_off, long vt_ld);

// ---- layernorm.hip --------------------------------------------------------------------------
void launch_layernorm(const LaunchCtx& ctx, const float* x, long ldx, const float* gamma, const float* beta,
                      float* y, long ldy, int rows, int d, const char* tag);
// the same for `batch` sessions at once: x = z.in[i], y = z.out[i]
void launch_layernorm_batched(const LaunchCtx& ctx, const PtrTable& z, int batch, long ldx, const float* gamma,
                              const float* beta, long ldy, int rows, int d, const char* tag);
// the same with the result in the X3 format (three bf16 planes; the operand of launch_gemm_x3): y3 row stride 3 * ldy3
// bf16; batched form: x = z.in[i], y3 = z.out[i] reinterpreted
void launch_layernorm_x3(const LaunchCtx& ctx, const float* x, long ldx, const float* gamma, const float* beta,
                         unsigned short* y3, long ldy3, int rows, int d, const char* tag);
void launch_layernorm_x3_batched(const LaunchCtx& ctx, const PtrTable& z, int batch, long ldx, const float* gamma,
                                 const float* beta, long ldy3, int rows, int d, const char* tag);

// ---- mel.hip --------------------------------------------------------------------------------
struct MelArgs {
    const float* audio;    // device, n_samples valid
    int n_samples;
    const float* window;   // [400] periodic hann
    const double* twiddle; // [400] cos(2*pi*i/400) in fp64
    const float* filters;  // [n_mels][201]
    const int* filt_lo;    // [n_mels] first non-zero bin
    const int* filt_hi;    // [n_mels] one past the last non-zero bin
    int n_mels;
    float* logmel;         // [n_active][n_mels] scratch: log10(clamp(mel))
    float* frame_max;      // [n_active]
    float* mel_t;          // [(3000+2)][n_mels] time-major, row 0 and row 3001 stay zero (conv padding)
    int n_active;          // frames that can see a non-zero sample (computed by the caller)
    int n_total;           // frames the reference's STFT yields before trimming to 3000
    // incremental form (mel half): logmel / frame_max of frames [head, first) are still valid from
    // the previous call (their 400-sample windows saw no new sample); only frames [0, head) and [first, n_active) are
    // computed.  head = first = 0: everything.
    int head = 0;
    int first = 0;
    // samples of the zero-padded signal torch.stft reflects at (audio + trailing zeros): only matters when fewer than 200
    // zeros follow the audio (wlk_log_mel with a short padding); the streaming path always has 30 s of them
    int n_padded = 0x7fffffff;
};
void launch_mel(const LaunchCtx& ctx, const MelArgs& a);
// whole-file form (whisper/audio.py:110-157 as whisper.transcribe() calls it): frames [0, n_active) through
// mel_frame_kernel into a.logmel / a.frame_max, then out[n_mels][n_total] = (max(L, max L - 8) + 4) / 4, mel-major as the
// reference returns it.  a.mel_t is not used.  `max_scratch` is one float.
void launch_mel_full(const LaunchCtx& ctx, const MelArgs& a, float* max_scratch, float* out);
void launch_pcm16_to_float(const LaunchCtx& ctx, const short* in, float* out, int n);

// ---- melspec.hip (diarization front end) ---------------------------------------------------------
struct MelSpecArgs {
    const float* audio;     // device
    int n_samples;
    const float* window;    // [win_length]
    const double* twiddle;  // [n_fft] cos(2*pi*i/n_fft), fp64
    const float* filters;   // [n_mels][n_fft/2+1]
    const int* filt_lo;
    const int* filt_hi;
    float* out;             // [n_frames][n_mels], time-major
    int n_fft, win_length, hop, n_mels;
    float preemph, log_guard;
};
void launch_melspec(const LaunchCtx& ctx, const MelSpecArgs& a, int n_frames);
}  // namespace wlk
// the extractor handle of the C ABI (api.hip: wlk_melspec_*); sortformer_api.hip launches its kernel inside a stacked step
// with the handle's read-only tables (window, twiddles, filterbank) and its own audio / output rows
struct wlk_melspec {
    int device = 0, n_fft = 0, win_length = 0, hop = 0, n_mels = 0, cap = 0;
    float preemph = 0.f, log_guard = 0.f;
    float *window = nullptr, *filters = nullptr, *audio = nullptr, *out = nullptr;
    double* twiddle = nullptr;
    int *lo = nullptr, *hi = nullptr;
    hipStream_t stream = nullptr;
    std::mutex mu;   // one device audio/out buffer and one stream: runs from different host threads are serialised
};
namespace wlk {

// ---- attention.hip --------------------------------------------------------------------------
// flash-style fp32-MFMA attention: q rows [Tq] against k/v rows [Tk], 64-wide heads, no mask
struct FlashArgs {
    // ---- hot block: everything a kernel needs before its first operand load, contiguous at the start of the
    // kernel-argument segment and requested by one s_load burst at kernel entry (WLK_PIN_FLASH_ARGS) -----------------
    const float* q = nullptr; long ldq = 0;      // query row r, head h at q + r*ldq + 64h (pre-scaled)
    const float* k = nullptr; const float* v = nullptr; long ldkv = 0;  // key t at k + t*ldkv + 64h (pre-scaled)
    float* out = nullptr; long ldo = 0;
    int Tq = 0, Tk = 0, n_head = 0;
    // batched encodes: > 0 -> grid.y = batch; q = z.in[i], k = q + z_k_off, v = q + z_v_off, out = z.out[i]
    int batch = 0;
    long z_k_off = 0, z_v_off = 0;
    // key-range split (few query tiles, e.g. decoder prefill): k_splits workgroups per (q tile, head) leave
    // partial softmax states in part_o/m/l [rows][n_head][k_splits][64|1|1]; a merge kernel folds them
    int k_splits = 1;
    int ring_rows = 0;
    // stacked prefills of several sessions (engine.hip): query tile t (32 stacked rows, all of ONE session) takes its
    // keys / values from tile_rows[t].cross_kv + tile_kv_off (values tile_v_off floats further) and dumps alignment
    // scores into tile_rows[t].ring (one beam); ring_row[] stays indexed by the stacked row
    const StepRow* tile_rows = nullptr;
    long long* dbg_clock = nullptr;   // probe only (enc_attention_pw_kernel): 8 s_memtime figures per workgroup
    // ---- the rest ---------------------------------------------------------------------------------------------------
    long tile_kv_off = 0, tile_v_off = 0;
    // decoder prefill only: raw scores of alignment heads go to the alignment window
    const int* head_rank = nullptr;              // [n_head] rank or -1
    float* ring = nullptr;
    const int* ring_row = nullptr;               // [Tq]
    const int* beam_of_row = nullptr;            // [Tq]
    int n_beam = 1;
    float* part_o = nullptr;
    float* part_m = nullptr;
    float* part_l = nullptr;
    PtrTable z;
};
#define WLK_PIN_FLASH_ARGS(a)                                                                                              \
    do {                                                                                                                   \
        asm volatile("" ::"s"((a).q), "s"((a).ldq), "s"((a).k), "s"((a).v), "s"((a).ldkv), "s"((a).out), "s"((a).ldo),      \
                     "s"((a).Tq), "s"((a).Tk), "s"((a).n_head), "s"((a).batch), "s"((a).z_k_off), "s"((a).z_v_off),         \
                     "s"((a).k_splits), "s"((a).ring_rows), "s"((a).tile_rows), "s"((a).dbg_clock));                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    } while (0)
extern long long* g_attn_dbg_clock;   // set by the timing probe (diag.hip); nullptr otherwise
size_t flash_split_scratch_floats(int rows, int n_head, int k_splits);
// encoder self-attention over qkv[T][3d] (q and k pre-scaled), out[T][d]
void launch_encoder_attention(const LaunchCtx& ctx, const float* qkv, float* out, int T, int d, int n_head,
                              int k_splits = 1, float* split_scratch = nullptr);
// encoder self-attention of `batch` sessions in one launch: qkv = z.in[i] ([T][3d]), out = z.out[i]
void launch_encoder_attention_batched(const LaunchCtx& ctx, const PtrTable& z, int batch, int T, int d, int n_head);
void launch_prefill_cross_attention(const LaunchCtx& ctx, const FlashArgs& a);
void launch_ring_softmax(const LaunchCtx& ctx, float* ring, const int* ring_row, const int* beam_of_row,
                         const int* ranks_dev, int n_ranks, int rows, int ring_rows, int n_beam, int T);

// ---- decoder.hip ----------------------------------------------------------------------------
void launch_embed(const LaunchCtx& ctx, const int* tokens, const float* tok_emb, const float* pos_emb,
                  float* x, int n_rows, int n_tok, const int* offset_dev, int d);
void launch_embed_rows(const LaunchCtx& ctx, const StepRow* rows, const float* tok_emb, const float* pos_emb, float* x,
                       int n_rows, int d);
void launch_decoder_self_attention_rows(const LaunchCtx& ctx, const float* qkv, const StepRow* rows, long layer_off,
                                        float* out, int n_rows, int d, int n_head, int ctx_len);
void launch_kv_append(const LaunchCtx& ctx, const float* qkv, float* kc, float* vc, int n_rows, int n_tok,
                      const int* offset_dev, int d, int ctx_len);
// stacked prefills: row r appends its k / v to rows[r].kcache / vcache + layer_off at position rows[r].offset
void launch_kv_append_rows(const LaunchCtx& ctx, const float* qkv, const StepRow* rows, long layer_off, int n_rows, int d);
void launch_decoder_self_attention(const LaunchCtx& ctx, const float* qkv, const float* kc, const float* vc,
                                   float* out, int n_rows, int n_tok, const int* offset_dev, int d, int n_head,
                                   int ctx_len, const ProgressMark& mark = ProgressMark{});
struct CrossAttnArgs {
    const float* q;        // [rows][d], pre-scaled
    const float* k;        // [T][ldkv] pre-scaled keys of this layer
    const float* v;        // [T][ldkv]
    long ldkv;
    float* out;            // [rows][d]
    int rows, d, n_head, T;
    const int* head_rank;  // [n_head]: alignment rank of (layer, head) or -1
    float* ring;           // alignment window base [n_align][n_rows_beam][ring_rows][T]
    const int* ring_row;   // [rows]: destination ring row of each query row
    const int* beam_of_row;// [rows]
    int ring_rows, n_beam;
    float* qk_debug;       // [rows][n_head][T] or nullptr
    // batched steps: keys/values of row r are step_rows[r].cross_kv + kv_off (+ d for v), its alignment rows go to
    // step_rows[r].ring at step_rows[r].ring_row (one beam per row)
    const StepRow* step_rows = nullptr;
    long kv_off = 0;
    // decode steps (split kernel only): the query projection folded into the kernel's prologue - q is ignored and every
    // workgroup derives its head's 64 query values itself, q_h = scale * (Wq[64 h .. 64 h + 64, :] . LN(x_row) + b),
    // with the single-row GEMV's arithmetic (gemv1_f32_kernel: lane-strided fmaf chains, xor-shuffle folds), so the
    // values are bit-identical to the separate dec_lnx_xq launch this replaces
    const float* xq_x = nullptr;       // [rows][d] residual stream
    const float* xq_w = nullptr;       // [d][d]
    const float* xq_b = nullptr;
    const float* xq_gamma = nullptr;
    const float* xq_beta = nullptr;
    float xq_scale = 1.f;
};
void launch_decoder_cross_attention(const LaunchCtx& ctx, const CrossAttnArgs& a);
bool cross_split_folds_query(int d);   // can the split kernel take the query projection (d % 256 == 0, d <= 2048)?
// decode steps: keys split over several workgroups per (row, head) + merge; scratch layout is
// [scores rows*H*T | pm rows*H*S | pl rows*H*S | po rows*H*S*64] (cross_split_scratch_floats)
void launch_decoder_cross_attention_split(const LaunchCtx& ctx, const CrossAttnArgs& a, float* scores, float* pm,
                                          float* pl, float* po, bool merge = true);
constexpr int kCrossSplitWays = 8;   // == kCrossSplit of decoder.hip (checked there)
size_t cross_split_scratch_floats(int rows, int n_head, int T);
void launch_kv_gather(const LaunchCtx& ctx, const float* src, float* dst, const int* source_rows, int n_rows,
                      int len, int d, int ctx_len, int n_layer);

// ---- sortformer.hip (a12: streaming Sortformer diarizer network) -----------------------------------
// sub-sampling stem, channels-last: feats [T][F] -> conv0 (1->C, 3x3 s2 p1, ReLU) -> [T1][F1][C]
inline int sf_sub_len(int n) { return (n - 1) / 2 + 1; }   // floor((n + 2 - 3) / 2) + 1
void launch_sf_scale_copy(const LaunchCtx& ctx, const float* src, float* dst, long n, float scale);
constexpr int kSfMaxSegments = 8;     // sessions per stacked Sortformer step
struct SfSegments {                    // by-value table of a stacked step: session s owns rows [start[s], start[s] + len[s])
    int n = 0;
    int start[kSfMaxSegments] = {0}, len[kSfMaxSegments] = {0};
};
// one stride-2 stage of the sub-sampling stem over stacked sessions: session s reads frames [in_start, in_start + in_len) and
// writes its sf_sub_len(in_len) output frames from out_start on
struct SfConvSegs {
    int n = 0, in_total = 0, out_total = 0;
    int in_start[kSfMaxSegments] = {0}, in_len[kSfMaxSegments] = {0}, out_start[kSfMaxSegments] = {0};
};
SfConvSegs sf_conv_segs_next(const SfConvSegs& prev);
void launch_sf_conv0(const LaunchCtx& ctx, const float* in, const float* w, const float* b, float* out, const SfConvSegs& sg, int F,
                     int C);
void launch_sf_dwconv2d(const LaunchCtx& ctx, const float* in, const float* w, const float* b, float* out, const SfConvSegs& sg,
                        int Fi, int C);
void launch_sf_assemble(const LaunchCtx& ctx, const float* ctx_rows, const float* chunk_rows, float* x, const SfSegments& rows,
                        const SfSegments& chunks, int d, float scale);
struct SfAttnArgs {
    const float* q = nullptr; const float* k = nullptr; const float* v = nullptr;
    long ldq = 0, ldk = 0, ldv = 0;
    float* out = nullptr; long ldo = 0;
    int T = 0, n_head = 0, dh = 0;
    float scale = 1.f;                 // applied to (ac + bd)
    // relative-position term (Conformer RelPositionMultiHeadAttention): score += (q + v_bias) . pos[pos_row0 - i + j]
    const float* pos = nullptr; long ldp = 0; int pos_row0 = 0;
    const float* bias_u = nullptr; const float* bias_v = nullptr;
    // stacked sessions (sortformer_api.hip): n_seg > 0 -> rows [seg_start[s], seg_start[s] + seg_T[s]) of q / k / v / out are
    // session s's sequence (grid.z = n_seg, T above = the longest); attention never crosses a segment
    int n_seg = 0;
    int seg_start[kSfMaxSegments] = {0}, seg_T[kSfMaxSegments] = {0};
};
constexpr int kSfMaxFrames = 512;      // attention rows the score buffer in LDS is sized for
void launch_sf_attention(const LaunchCtx& ctx, const SfAttnArgs& a);
// Conformer convolution module core: GLU over [T][2d] -> depthwise conv1d (k taps, same padding) -> BatchNorm (eval)
// -> Swish -> [T][d]; w tap-major [k][d]
void launch_sf_glu_dwconv(const LaunchCtx& ctx, const float* in, const float* w, const float* b, const float* bn_mean,
                          const float* bn_invstd, const float* bn_w, const float* bn_b, float* out, const SfSegments& rows, int d,
                          int taps);
// relu -> Linear(d,d)+relu -> Linear(d,n_spk) -> sigmoid; w1t = the first Linear's weight TRANSPOSED ([in][out])
void launch_sf_head(const LaunchCtx& ctx, const float* x, const float* w1t, const float* b1, const float* w2,
                    const float* b2, float* out, int T, int d, int n_spk);
void launch_sf_transpose(const LaunchCtx& ctx, const float* src, float* dst, int rows, int cols);

// ---- select.hip -----------------------------------------------------------------------------
// adjustments (may be n_adj = 0) are applied to the logits in place before the reduction
void launch_logsoftmax_topk(const LaunchCtx& ctx, float* logits, int n_vocab, int n_rows, int k,
                            float* top_vals, int* top_ids, void* scratch, const int* adj_row, const int* adj_ids,
                            const float* adj_deltas, int n_adj);
size_t topk_scratch_bytes(int n_rows);
void launch_token_prob(const LaunchCtx& ctx, const float* logits, int n_vocab, int n_rows, int token, float* probs);
// whisper's batch-decoder logit rules + the greedy pick of one sequence (select.hip: rules_pick_kernel); the fields mirror
// wlk_pick_params of include/wlk_hip.h
struct PickRules {
    int first_step, without_timestamps, timestamp_begin, eot, no_timestamps, ts_mode, ts_bound, max_initial;
};
void launch_rules_pick(const LaunchCtx& ctx, const float* logits, int n_vocab, const unsigned char* mask, const PickRules& p,
                       int* out_token, float* out_logprob);
struct AlignArgs {
    const float* ring;   // [n_align][n_beam][ring_rows][T]
    int n_align, n_beam, ring_rows, T;
    int prefill_rows;    // rows of the first (prefill) step still inside the window, else 0
    int n_single;        // valid single-step slots (<= 16)
    int newest_row;      // ring row of the newest query row
    int single_base;     // first ring row of the single-step slots
    int content_len;
    float* z;            // scratch [n_beam][n_align][T]
    float* attn_last;    // [n_beam][T] head-mean of the median-filtered newest row
    int* frames;         // [n_beam]
    const StepRow* rows = nullptr;   // launch_alignatt_rows
    float* part = nullptr;           // early z-score form: [n_beam][64] (value, frame) pairs, one per 256-frame block
};
void launch_alignatt(const LaunchCtx& ctx, const AlignArgs& a);
// top-k (with adjustments) and AlignAtt read-out of the same rows in two launches instead of four; returns false when
// the configuration needs the separate kernels (no alignment heads, window too large for LDS, WLK_SELECT_FUSED=0)
bool launch_select_fused(const LaunchCtx& ctx, float* logits, int n_vocab, int n_rows, int k, float* top_vals, int* top_ids,
                         void* scratch, const int* adj_row, const int* adj_ids, const float* adj_deltas, int n_adj,
                         const AlignArgs& a, const StepHostOut& host = StepHostOut{}, const float* ns_logits = nullptr,
                         int ns_token = 0, float* ns_probs = nullptr, bool early_z = false);
// early_z: the z rows were written by the side workgroups of the vocabulary projection's launch (GemmArgs::side_align, the
// same AlignArgs resident on the device) - the slice pass carries the arg-max, the last launch is the 64-thread fold
bool select_early_z_enabled();
bool select_fused_applicable(int n_rows, int k, const AlignArgs& a);
// batched steps: one read-out per row with the row's own window / counters (a.ring, prefill_rows, n_single, newest_row
// and content_len are taken from rows[r]; a.n_beam is the number of rows, each row is its own beam 0)
void launch_alignatt_rows(const LaunchCtx& ctx, const AlignArgs& a, const StepRow* rows);

}  // namespace wlk

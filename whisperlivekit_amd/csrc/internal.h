// Internal object layouts shared by the C-ABI files (api.hip, engine.hip): the model, the per-stream session and the
// small helpers around them.  Not part of the public interface (include/wlk_hip.h keeps both types opaque).
#pragma once
#include <map>
#include <memory>
#include <chrono>
#include <cstdlib>
#include <thread>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/wlk_hip.h"
#include "common.h"

namespace wlk {

const std::string& get_last_error();

template <typename F>
inline int guarded(F&& f) {
    try {
        return f();
    } catch (const HipError& e) {
        set_last_error(e.what());
        return WLK_ERR_HIP;
    } catch (const std::length_error& e) {
        set_last_error(e.what());
        return WLK_ERR_CAPACITY;
    } catch (const std::invalid_argument& e) {
        set_last_error(e.what());
        return WLK_ERR_ARG;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return WLK_ERR_STATE;
    }
}
inline int fail(int code, const std::string& msg) {
    set_last_error(msg);
    return code;
}


// profiler: HIP events around every launch on the session stream
struct Profiler {
    struct Rec {
        const char* name;
        hipEvent_t a, b;
        double flops, bytes;
    };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t pending = nullptr;
    const char* pending_name = nullptr;
    double pending_flops = 0.0, pending_bytes = 0.0;
    hipEvent_t get() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e;
        WLK_HIP(hipEventCreate(&e));
        return e;
    }
};

struct TensorSlot {
    std::string name;
    uint64_t offset, numel;
};


struct LayerW {
    const float *ln1w, *ln1b, *qkvw, *qkvb, *outw, *outb, *ln2w, *ln2b, *fc1w, *fc1b, *fc2w, *fc2b;
    const float *lnxw = nullptr, *lnxb = nullptr, *xqw = nullptr, *xqb = nullptr, *xkvw = nullptr, *xkvb = nullptr,
                *xoutw = nullptr, *xoutb = nullptr;
    // encoder layers: the wide projections' weights in the X3 format (x3.h), packed by wlk_model_finalize; nullptr = the
    // shape stays on the fp32-MFMA kernels
    const unsigned short *qkvw3 = nullptr, *fc1w3 = nullptr;
    const unsigned short *outw3 = nullptr, *fc2w3 = nullptr;      // round 6: the N = d projections where d >= 1024 (medium, large)
};

template <typename T>
inline T* dev_alloc(size_t n) {
    T* p = nullptr;
    WLK_HIP(hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)));
    return p;
}
template <typename T>
inline T* dev_alloc_zero(size_t n, hipStream_t s) {
    T* p = dev_alloc<T>(n);
    WLK_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
    return p;
}

}  // namespace wlk

struct wlk_engine;
struct wlk_session;
struct wlk_model;

// ---- stacked prefills (api.hip: wlk_prefill_group; engine.hip: the prefill lane) -----------------------------------
// Workspace of one stacked prefill chain: activations of all stacked rows, the row / tile tables, read-out staging.
struct wlk_prefill_ws {
    int cap_rows = 0;             // stacked rows (each session padded to whole 32-row query tiles) the buffers hold
    int cap_session_rows = 0;     // longest prompt of one session taken into a stack
    float *dx = nullptr, *dh = nullptr, *dqkv = nullptr, *datt = nullptr, *dq = nullptr, *dmlp = nullptr, *hsel = nullptr,
          *logits = nullptr, *part = nullptr;
    wlk::StepRow *rows_dev = nullptr, *tiles_dev = nullptr;
    int *ring_row_dev = nullptr, *zeros_dev = nullptr;
    char* pinned = nullptr;       // host staging of the tables
    size_t pinned_bytes = 0;
};
struct wlk_prefill_item {
    wlk_session* s = nullptr;
    const int64_t* tokens = nullptr;
    int n_tok = 0, sot_index = 0;
    int rc = 0;
    std::string err;
};

// ------------------------------------------------------------------------------------------------
// model
// ------------------------------------------------------------------------------------------------
struct wlk_model {
    wlk_dims D{};
    int device = 0;
    float* arena = nullptr;
    bool owns_arena = false;
    uint64_t arena_floats = 0;
    std::vector<wlk::TensorSlot> slots;
    std::map<std::string, const wlk::TensorSlot*> by_name;
    double* twiddle = nullptr;        // [400] fp64 cos table
    // cross-attention k|v projections of ALL decoder layers side by side ([L*2d, d] and [L*2d]; copies of the
    // arena's per-layer tensors made by wlk_model_finalize): one GEMM per encode instead of L
    float* xkv_all_w = nullptr;
    float* xkv_all_b = nullptr;
    unsigned short* xkv_all_w3 = nullptr;      // the same weights in the X3 format (nullptr: fp32 kernels)
    std::vector<unsigned short*> x3_owned;     // every X3 weight buffer of this model (freed with it)
    int* filt_lo = nullptr;           // [n_mels]
    int* filt_hi = nullptr;
    int* head_rank = nullptr;         // [L][H] alignment rank or -1
    int* layer_ranks = nullptr;       // [L][H] ranks of each layer's alignment heads, compacted
    int* layer_heads = nullptr;       // [L][H] the alignment heads of each layer in ascending order, compacted
    int* all_ranks = nullptr;         // [L*H] 0, 1, 2, ...: every alignment rank in one launch
    std::vector<int> layer_rank_count;
    std::vector<int> align_pairs;     // (layer, head)*
    int n_align = 0;
    bool finalized = false;
    std::vector<wlk::LayerW> enc_layers, dec_layers;   // resolved once at finalize
    const float *w_tok_emb = nullptr, *w_dec_pos = nullptr, *w_ln_w = nullptr, *w_ln_b = nullptr;
    wlk_engine* engine = nullptr;     // cross-session batched decode steps (engine.hip), created on first attach
    std::mutex engine_mu;

    const float* w(const std::string& name) const {
        auto it = by_name.find(name);
        if (it == by_name.end()) throw std::invalid_argument("unknown tensor " + name);
        return arena + it->second->offset;
    }
};


// ------------------------------------------------------------------------------------------------
// session
// ------------------------------------------------------------------------------------------------
struct wlk_session {
    wlk_model* m = nullptr;
    int beam = 1;
    int audio_cap = 0;
    hipStream_t stream = nullptr;
    wlk::Profiler prof;
    bool prof_on = false;
    bool debug = false;
    bool use_graph = true;
    hipGraphExec_t step_exec[2] = {nullptr, nullptr};   // single-token decode step, per KV buffer
    int enc_ksplit = 1;                                 // key split of the encoder attention (WLK_ENC_KSPLIT)
    float* esplit = nullptr;
    short* pcm16_dev = nullptr;                         // staging of wlk_audio_append_pcm16 (lazily allocated)

    // audio (two buffers: eviction copies the tail into the other one)
    float* audio[2] = {nullptr, nullptr};
    int audio_cur = 0;
    int audio_len = 0;

    // mel + encoder workspaces
    float *logmel = nullptr, *frame_max = nullptr, *mel_t = nullptr;
    // per-frame mel cache (incremental mel): logmel/frame_max reflect the first mel_cached_samples samples of the
    // current audio buffer for frames [mel_dirty_head, mel_cached_frames); the alternates take the shifted copy when
    // whole frames are evicted
    float *logmel_alt = nullptr, *frame_max_alt = nullptr;
    int mel_cached_samples = 0, mel_cached_frames = 0, mel_dirty_head = 0;
    bool mel_incremental = true;
    int frame_cap = 0;
    float *x1p = nullptr, *ex = nullptr, *eh = nullptr, *eqkv = nullptr, *eatt = nullptr, *emlp = nullptr,
          *enc_out = nullptr, *cross_kv = nullptr;
    // X3 images (three bf16 planes) of the LayerNorm outputs that feed the bf16-MFMA projections (gemm_x3.hip); enc_out3
    // replaces enc_out when the cross-attention K|V projection takes that path (wlk_export("enc") unpacks it)
    unsigned short *eh3 = nullptr, *enc_out3 = nullptr;
    unsigned short* emlp3 = nullptr;      // X3 image of the MLP's hidden activations [T][4 d] (operand of an X3 fc2; round 6)
    bool enc_out_is_x3 = false;
    // operand image of the X3 encoder attention (attention_x3.hip): q | k as X3 rows, V^T behind them; written by the qkv
    // projection's epilogue.  Zero-initialised once: the padding keys of V^T are never written and must stay finite.
    unsigned short* eqkv3 = nullptr;
    bool encoded = false;
    int content_len = 0;

    // decoder workspaces
    int max_rows = 0;
    int* tokens_dev = nullptr;
    float *dx = nullptr, *dh = nullptr, *dqkv = nullptr, *datt = nullptr, *dq = nullptr, *dmlp = nullptr;
    float* kcache[2] = {nullptr, nullptr};  // [L][beam][ctx][d]; second pair allocated on first reorder
    float* vcache[2] = {nullptr, nullptr};
    int kv_cur = 0;
    int self_len = 0;
    int n_steps = 0;
    int prefill_rows = 0;
    int last_rows = 0, last_ntok = 0;
    float *hsel = nullptr, *logits_last = nullptr, *logits_sot = nullptr;
    bool have_sot = false;
    int* step_in = nullptr;      // [tokens max_rows | ring_row max_rows | beam_of_row max_rows | offset]: ONE H2D per decode
    int *ring_row = nullptr, *beam_of_row = nullptr, *d_offset = nullptr;   // views into step_in
    float* ring = nullptr;
    int ring_rows = 0;
    float *z = nullptr, *attn_last = nullptr;
    float* qk_debug = nullptr;  // [L][max_rows][H][T]
    float* xsplit = nullptr;    // scratch of the split cross-attention (decode steps)
    float* fsplit = nullptr;    // scratch of the key-split flash attention (decoder prefill)
    static constexpr int kFlashSplitsMax = 16;   // scratch is sized for this many key ranges
    static int flash_splits();                   // key ranges of the prefill cross-attention (WLK_FLASH_SPLITS, default 6)

    // select scratch (device) + pinned host staging
    int *adj_row = nullptr, *src_rows = nullptr;   // adj_row: packed [rows | ids | deltas] of the current call
    float* top_vals = nullptr;
    void* topk_scratch = nullptr;
    int *top_ids = nullptr, *frames = nullptr;
    float* probs = nullptr;
    static constexpr int kAdjCap = 4096;
    void* pinned = nullptr;  // 1 MiB
    static constexpr size_t kPinnedBytes = 1 << 20;
    // wlk_decode's staging block has its own pinned allocation and an event behind its last use, so that a prefill can be
    // enqueued behind a still-running encoder instead of waiting for the stream to drain
    void* dec_stage = nullptr;
    hipEvent_t dec_stage_ev = nullptr;
    bool dec_stage_used = false;
    // the same for the audio chunks on their way to the device ring: the caller's buffer is free on return, the copy is
    // still in flight, the next append waits for it before reusing the block
    void* audio_stage = nullptr;
    hipEvent_t audio_stage_ev = nullptr;
    // whisper's batch-decoder rules on the device (wlk_rules_set / wlk_pick_greedy): bit 0 = always suppressed, bit 1 = blank
    unsigned char* rules_mask = nullptr;
    int* pick_out = nullptr;                            // [token | log-probability]
    // Infinity-Cache prefetcher beside a graph-replayed single-token step (wlk_step_select; decoder.hip): its stream, the
    // per-layer range table and the progress word the step's chain marks
    hipStream_t pf_stream = nullptr;
    wlk::TouchRanges* pf_table = nullptr;
    unsigned long long* pf_progress = nullptr;
    bool pf_marks = false;                              // set around the capture of a step graph: the chain stores its marks
    bool audio_stage_used = false;
    // single-token steps as one graph replay (wlk_step_select): host-coherent blocks the step's first / last kernel
    // read / write directly, their device-side addresses, the device copy of the input block, one graph per KV set
    wlk::StepBlock *step_host = nullptr, *step_host_dev = nullptr, *step_dev = nullptr;
    wlk::StepResult *result_host = nullptr, *result_host_dev = nullptr;
    hipGraphExec_t fstep_exec[2] = {nullptr, nullptr};
    wlk::AlignArgs* step_align_dev = nullptr;            // the step's read-out arguments, resident: the vocabulary GEMV's side workgroups read them (early z-score)
    unsigned step_seq = 0;
    uint64_t step_ns = 0, step_launch_ns = 0, step_count = 0;   // wlk_session_step_stats (WLK_STEP_TIMING=1: also printed when the session is destroyed)

    wlk::LaunchCtx ctx() { return wlk::LaunchCtx{stream, prof_on ? &prof : nullptr}; }
    // word-timestamp alignment (word_align.hip): the prefill leaves the alignment heads' raw scores in the window, and
    // its workspace (grown on demand)
    bool align_raw_scores = false;
    float* wa_buf = nullptr;
    size_t wa_cap = 0;
    wlk_engine* engine = nullptr;   // set by wlk_engine_attach: single-token steps run batched with the other attached sessions
};


// Wait for the result flags of a graph-replayed step (written by its last kernel into host-coherent memory).  A step is
// ~250 us of GPU time, so the wait spins first (lowest latency: the flags arrive over PCIe, there is no interrupt to
// wait for) - but only for WLK_STEP_SPIN_US (default 400 us, i.e. a step that runs alone and on time never leaves the
// spin); after that the thread yields between looks and, once the wait is 2 ms old, sleeps 50 us at a time: a step that
// queues behind other sessions' encoder kernels - the loaded-server case - costs the host a few wake-ups instead of a
// pegged core per session.  The stream is queried every 1024 looks so that a failed launch cannot hang the caller;
// gives up after 30 s.
inline void wlk_wait_step_flags(hipStream_t stream, const wlk::StepResult* res, int n, unsigned seq) {
    static const long spin_us = [] {
        const char* e = getenv("WLK_STEP_SPIN_US");
        return e ? atol(e) : 400L;
    }();
    const auto t_start = std::chrono::steady_clock::now();
    auto all_set = [&]() {
        for (int i = 0; i < n; ++i)
            if (__atomic_load_n(&res[i].flag_topk, __ATOMIC_ACQUIRE) != seq || __atomic_load_n(&res[i].flag_align, __ATOMIC_ACQUIRE) != seq)
                return false;
        return true;
    };
    bool spinning = true;
    for (unsigned looks = 0;; ++looks) {
        if (all_set()) return;
        if ((looks & 0x3ff) == 0x3ff) {
            const hipError_t q = hipStreamQuery(stream);
            if (q == hipSuccess) {
                if (all_set()) return;
                throw std::runtime_error("graph step: the replay finished without delivering its result");
            }
            if (q != hipErrorNotReady) WLK_HIP(q);
        }
        if (spinning) {
            __builtin_ia32_pause();
            if ((looks & 0x3f) == 0x3f &&
                std::chrono::steady_clock::now() - t_start > std::chrono::microseconds(spin_us))
                spinning = false;
            continue;
        }
        const auto waited = std::chrono::steady_clock::now() - t_start;
        if (waited > std::chrono::seconds(30)) throw std::runtime_error("graph step: no result after 30 s");
        if (waited > std::chrono::milliseconds(2)) std::this_thread::sleep_for(std::chrono::microseconds(50));
        else std::this_thread::yield();
    }
}

// api.hip
// One single-token step of a beam-1 session + its read-out (wlk_decode(first=0) followed by wlk_select(k=2)) as one
// graph replay without copy nodes.  Returns 1 (and does nothing) when the session / step does not qualify.
int wlk_step_select(wlk_session* s, int64_t token, const int32_t* adj_ids, const float* adj_deltas, int n_adj,
                    int content_mel_len, float* top_logprobs2, int32_t* top_ids2, int32_t* frame);
// wlk_no_speech_prob + wlk_select of the first step of an infer behind one synchronisation (the read-out runs whatever the
// probability turns out to be; a caller that stops on no-speech simply ignores it)
int wlk_select_first(wlk_session* s, int no_speech_token, const int32_t* adj_row, const int32_t* adj_ids,
                     const float* adj_deltas, int n_adj, int k, int content_mel_len, float* no_speech_host,
                     float* top_logprobs_host, int32_t* top_ids_host, int32_t* frames_host);
// mel_given: mel_t of every session already holds the log-mel to encode (wlk_encode_mel); the mel kernels are skipped
void wlk_encode_group(const std::vector<wlk_session*>& group, const wlk::LaunchCtx& c, std::vector<int>& content_out,
                      bool mel_given = false);
// host-side preconditions of one session's encode (model, frame capacity); empty string = fine.  The encode lane asks
// this per request before it stacks requests into one chain.
std::string wlk_encode_precheck(const wlk_session* s, const wlk_model* m);

// The prefills (first decoder pass of an infer: all prompt tokens at once) of several beam-1 sessions of one model as
// ONE launch chain on c.stream: rows of all sessions stacked for the LayerNorms and GEMMs (shared weights), per-row
// session pointers for the cache append / self-attention, per-query-tile session pointers for the cross-attention.
// Every row goes through the kernels and the arithmetic of a prefill of its session alone (wlk_decode, first = 1), so the
// results are bit-identical to it.  wlk_prefill_precheck: empty = this request can ride in a stack (otherwise the
// caller runs wlk_decode).  Leaves each session as wlk_decode(first = 1) leaves it; throws on launch failures.
std::string wlk_prefill_precheck(const wlk_prefill_item& it, const wlk_prefill_ws& ws);
void wlk_prefill_group(const std::vector<wlk_prefill_item*>& items, const wlk::LaunchCtx& c, wlk_prefill_ws& ws);
void wlk_prefill_ws_alloc(const wlk_model* m, wlk_prefill_ws& ws, int max_sessions, int session_rows);
void wlk_prefill_ws_free(wlk_prefill_ws& ws);

// engine.hip
// the prefill of an attached session through the engine's prefill lane (stacked with the prefills other sessions have
// waiting); 1 = not taken (the caller runs wlk_decode itself); blocks until this session's prefill is done
int wlk_engine_prefill(wlk_session* s, const int64_t* tokens, int n_tok, int sot_index);
bool wlk_engine_batches_encodes(const wlk_session* s);
int wlk_engine_encode(wlk_session* s, int* content_mel_len);      // blocks until this session's encode is done
namespace wlk { struct DecodeJob; }
void wlk_engine_destroy_for_model(wlk_model* m);
bool wlk_engine_wants(const wlk_session* s);
void wlk_engine_loop_enter(wlk_session* s);
void wlk_engine_loop_exit(wlk_session* s);
int wlk_engine_run_job(wlk_session* s, wlk::DecodeJob* job);   // blocks until the job's loop has stopped

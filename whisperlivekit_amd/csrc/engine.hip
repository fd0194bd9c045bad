// Cross-session batched decode steps on one GPU ("continuous batching" of the AlignAtt decode loops).
//
// The reference serves N sessions from ONE model on ONE device (whisperlivekit/core.py:246-271 builds the shared
// model, audio_processor.py:543-551 runs every session's process_iter on its own worker thread): each session then
// issues its own chain of ~55 tiny launches per generated token and re-streams the same 200 MB of decoder weights.
// Here every beam-1 session attached to the model's engine hands the single-token steps of its decode loop
// (wlk_decode_until_stop, loop.hip) to ONE worker thread per GPU, which advances all loops that are currently in
// their decode phase together: one launch chain per iteration with rows = sessions.  The weight-streaming GEMV
// kernels already take up to 8 activation rows per pass over the weights (each row keeps the fmaf order of the
// single-row kernel, so results are bit-identical to a session running alone); the kernels that touch per-session
// state (KV-cache append, self-/cross-attention, alignment window + read-out) index a StepRow table by row.
// Sessions join when their prefill is done and leave when their loop stops; nobody waits for anybody.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

#include "common.h"
#include "internal.h"
#include "loop.h"

using namespace wlk;

namespace {

struct EngineJob {
    wlk_session* s = nullptr;
    DecodeJob* job = nullptr;
    bool done = false;
    int rc = WLK_OK;
    std::string err;
};

struct EncodeReq {
    wlk_session* s = nullptr;
    int content = 0;
    bool done = false;
    int rc = WLK_OK;
    std::string err;
};

struct PrefillReq {
    wlk_prefill_item item;
    bool done = false;
};

constexpr int kEngineAdjCap = 8192;

}  // namespace

struct wlk_engine {
    wlk_model* m = nullptr;
    int max_rows = 1;
    hipStream_t stream = nullptr;
    // device workspace of one batched step
    StepRow* rows_dev = nullptr;   // = blk_dev->rows
    float *x = nullptr, *qkv = nullptr, *att = nullptr, *q = nullptr, *mlp = nullptr, *logits = nullptr, *xsplit = nullptr,
          *z = nullptr, *attn_last = nullptr;
    float* res_dev = nullptr;      // [top log-probs R*2 | top ids R*2 | frames R]
    void* topk_scratch = nullptr;
    int* adj_dev = nullptr;        // [rows n | ids n | deltas n]
    char* pinned = nullptr;
    static constexpr size_t kPinnedBytes = 256 * 1024;
    // a batched step as ONE graph replay without copy nodes (as wlk_step_select does for one session): host-coherent
    // block + results, their device-side addresses, the device copy of the block, one graph per row count
    EngineBlock *blk_host = nullptr, *blk_host_dev = nullptr, *blk_dev = nullptr;
    StepResult *res_host = nullptr, *res_host_dev = nullptr;
    hipGraphExec_t fstep_exec[9] = {};
    unsigned step_seq = 0;
    bool fused_steps = true;
    hipGraphExec_t step_exec[9] = {};   // captured launch chain per row count (adjustment count is a device scalar)
    int* n_adj_dev = nullptr;
    bool use_graph = true;

    std::thread worker;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<EngineJob*> submitted;
    bool quit = false;
    int attached = 0;
    std::atomic<int> in_loop{0};   // sessions currently inside wlk_decode_until_stop or wlk_encode
    // encode lane: concurrent encodes of the attached sessions are stacked into one launch chain (grid.y = sessions)
    // (WLK_ENGINE_ENCODE_LANES > 1: that many chains side by side, each on its own stream)
    std::vector<hipStream_t> enc_streams;
    std::vector<std::thread> enc_workers;
    int enc_batch_cap = kMaxBatch;
    std::condition_variable cv_enc_work, cv_enc_done;
    std::deque<EncodeReq*> enc_submitted;
    bool batch_encodes = true;
    int gather_us = 0;
    std::atomic<uint64_t> n_enc_batches{0}, n_enc_sessions{0};   // written by the lane workers, read by the stats calls
    void run_encodes(int lane);
    // prefill lane: the prefills (first decoder pass of an infer) of the attached sessions that are waiting at the same
    // time run as one stacked launch chain (wlk_prefill_group): every session used to run its own ~80-launch chain,
    // re-streaming the decoder weights, and those chains are what 8 streams on one GPU spent a quarter of their time in
    hipStream_t pre_stream = nullptr;
    std::thread pre_worker;
    std::condition_variable cv_pre_work, cv_pre_done;
    std::deque<PrefillReq*> pre_submitted;
    bool batch_prefills = true;
    int pre_gather_us = 0;
    int pre_min_sessions = 9;
    wlk_prefill_ws pre_ws;
    std::atomic<uint64_t> n_pre_batches{0}, n_pre_sessions{0};
    void run_prefills();
    std::atomic<uint64_t> n_iterations{0}, n_rows{0}, n_batched{0}, n_batched_rows{0};

    void run();
    void step_single(EngineJob* j, std::vector<EngineJob*>& finished);
    void step_batched(std::vector<EngineJob*>& group, std::vector<EngineJob*>& finished);
    void enqueue_decoder(int R, bool from_block = false, const AlignArgs* side_align = nullptr, int side_blocks = 0,
                         int side_zf = 0);   // embed .. logits of R rows (captured into a hipGraph per row count)
    AlignArgs* align_dev = nullptr;   // [9] the fused replay's read-out arguments per row count, resident (early z-score: select.hip)
    bool step_batched_one_replay(std::vector<EngineJob*>& group, std::vector<EngineJob*>& finished);
    void enqueue_select(int R, int n_adj);  // logit adjustments + log-softmax top-2 + AlignAtt read-out per row
};

// ---- one batched step ----------------------------------------------------------------------------------------
void wlk_engine::enqueue_decoder(int R, bool from_block, const AlignArgs* side_align, int side_blocks, int side_zf) {
    const wlk_dims& D = m->D;
    const int d = D.n_text_state, T = D.n_audio_ctx, H = D.n_text_head, V = D.n_vocab, ctx_len = D.n_text_ctx;
    const LaunchCtx c{stream, nullptr};
    const float scale = std::pow((float)kHeadDim, -0.25f);
    if (from_block) launch_embed_rows_step(c, blk_host_dev, blk_dev, m->w_tok_emb, m->w_dec_pos, x, R, d);
    else launch_embed_rows(c, rows_dev, m->w_tok_emb, m->w_dec_pos, x, R, d);
    float* sc = xsplit;
    float* pm = sc + (size_t)8 * H * T;
    float* pl = pm + (size_t)8 * H * 8;
    float* po = pl + (size_t)8 * H * 8;
    for (int i = 0; i < D.n_text_layer; ++i) {
        const LayerW& L = m->dec_layers[i];
        const long layer_off = (long)i * ctx_len * d;       // beam 1: [L][ctx][d]
        GemmArgs g;
        g.A = x; g.lda = d; g.W = L.qkvw; g.bias = L.qkvb; g.C = qkv; g.ldc = 3 * d; g.M = R; g.N = 3 * d; g.K = d;
        g.flags = kGemmScaleCols; g.scale = scale; g.scale_cols = 2 * d;
        g.ln_gamma = L.ln1w; g.ln_beta = L.ln1b;
        g.kv_rows = rows_dev; g.kv_layer_off = layer_off; g.kv_d = d; g.kv_ctx = ctx_len;
        launch_gemv(c, g, "dec_ln1_qkv_kv");
        launch_decoder_self_attention_rows(c, qkv, rows_dev, layer_off, att, R, d, H, ctx_len);
        GemmArgs o;
        o.A = att; o.lda = d; o.W = L.outw; o.bias = L.outb; o.C = x; o.ldc = d; o.M = R; o.N = d; o.K = d;
        o.flags = kGemmResidual; o.R = x; o.ldr = d;
        launch_gemv(c, o, "dec_out");
        GemmArgs qq;
        qq.A = x; qq.lda = d; qq.W = L.xqw; qq.bias = L.xqb; qq.C = q; qq.ldc = d; qq.M = R; qq.N = d; qq.K = d;
        qq.flags = kGemmScaleCols; qq.scale = scale; qq.scale_cols = d; qq.ln_gamma = L.lnxw; qq.ln_beta = L.lnxb;
        // batched steps keep the query projection as ONE multi-row GEMV (weights streamed once for all rows): folded into the
        // split cross-attention every row's 64 workgroups would stream Wq again (19 us per launch at 8 streams,
        // profiles/r04_trace8_busy.txt) - the fold only pays for a single row, where it saves a launch
        const bool fold_xq = R == 1 && cross_split_folds_query(d);
        if (!fold_xq) launch_gemv(c, qq, "dec_lnx_xq");
        CrossAttnArgs ca{};
        ca.q = q; ca.k = nullptr; ca.v = nullptr; ca.ldkv = (long)D.n_text_layer * 2 * d; ca.out = att;
        ca.rows = R; ca.d = d; ca.n_head = H; ca.T = T;
        ca.head_rank = m->n_align > 0 ? m->head_rank + (size_t)i * H : nullptr;
        ca.ring = nullptr; ca.ring_row = nullptr; ca.beam_of_row = nullptr;
        ca.ring_rows = ctx_len + kAlignWindow; ca.n_beam = 1; ca.qk_debug = nullptr;
        ca.step_rows = rows_dev; ca.kv_off = (long)i * 2 * d;
        if (fold_xq) {
            ca.xq_x = x; ca.xq_w = L.xqw; ca.xq_b = L.xqb; ca.xq_gamma = L.lnxw; ca.xq_beta = L.lnxb; ca.xq_scale = scale;
        }
        launch_decoder_cross_attention_split(c, ca, sc, pm, pl, po, true);
        GemmArgs xo;
        xo.A = att; xo.lda = d; xo.W = L.xoutw; xo.bias = L.xoutb; xo.C = x; xo.ldc = d; xo.M = R; xo.N = d; xo.K = d;
        xo.flags = kGemmResidual; xo.R = x; xo.ldr = d;
        launch_gemv(c, xo, "dec_xout");
        GemmArgs f1;
        f1.A = x; f1.lda = d; f1.W = L.fc1w; f1.bias = L.fc1b; f1.C = mlp; f1.ldc = 4 * d; f1.M = R; f1.N = 4 * d; f1.K = d;
        f1.flags = kGemmGelu; f1.ln_gamma = L.ln2w; f1.ln_beta = L.ln2b;
        launch_gemv(c, f1, "dec_ln2_fc1");
        GemmArgs f2;
        f2.A = mlp; f2.lda = 4 * d; f2.W = L.fc2w; f2.bias = L.fc2b; f2.C = x; f2.ldc = d; f2.M = R; f2.N = d; f2.K = 4 * d;
        f2.flags = kGemmResidual; f2.R = x; f2.ldr = d;
        launch_gemv(c, f2, "dec_fc2");
    }
    GemmArgs lg;
    lg.A = x; lg.lda = d; lg.W = m->w_tok_emb; lg.C = logits; lg.ldc = V; lg.M = R; lg.N = V; lg.K = d;
    lg.ln_gamma = m->w_ln_w; lg.ln_beta = m->w_ln_b;
    lg.side_align = side_align; lg.side_blocks = side_blocks; lg.side_zf = side_zf;   // fused replay: the rows' z-scores ride here
    launch_gemv(c, lg, "dec_lnf_logits");
}

void wlk_engine::enqueue_select(int R, int n_adj) {
    const wlk_dims& D = m->D;
    const int T = D.n_audio_ctx, V = D.n_vocab, ctx_len = D.n_text_ctx;
    const LaunchCtx c{stream, nullptr};
    float* top_vals = res_dev;
    int* top_ids = reinterpret_cast<int*>(res_dev) + 2 * max_rows;
    int* frames = top_ids + 2 * max_rows;
    AlignArgs a{};
    a.ring = nullptr; a.n_align = m->n_align; a.n_beam = R; a.ring_rows = ctx_len + kAlignWindow; a.T = T;
    a.prefill_rows = 0; a.n_single = 0; a.newest_row = 0; a.single_base = ctx_len; a.content_len = 0;
    a.z = z; a.attn_last = attn_last; a.frames = frames;
    a.rows = rows_dev;
    int* adj_ids = adj_dev + n_adj;
    float* adj_deltas = reinterpret_cast<float*>(adj_dev + 2 * n_adj);
    if (launch_select_fused(c, logits, V, R, 2, top_vals, top_ids, topk_scratch, adj_dev, adj_ids, adj_deltas, n_adj, a)) return;
    launch_logsoftmax_topk(c, logits, V, R, 2, top_vals, top_ids, topk_scratch, adj_dev, adj_ids, adj_deltas, n_adj);
    if (m->n_align > 0) launch_alignatt(c, a);
    else WLK_HIP(hipMemsetAsync(frames, 0, sizeof(int) * R, stream));
}

// The batched step as one graph replay: the first kernel pulls the row table and the logit adjustments out of the
// host-coherent block, the last one writes every row's result and flags into host-coherent memory.  Returns false when
// the step does not qualify (too many adjustments, no alignment heads, WLK_FUSED_STEP=0): the caller takes the chain
// with copy nodes and an eager read-out.
bool wlk_engine::step_batched_one_replay(std::vector<EngineJob*>& group, std::vector<EngineJob*>& finished) {
    const wlk_dims& D = m->D;
    const int R = (int)group.size();
    const int T = D.n_audio_ctx, V = D.n_vocab, ctx_len = D.n_text_ctx;
    if (!fused_steps || !use_graph || !blk_host) return false;
    AlignArgs a{};
    a.ring = nullptr; a.n_align = m->n_align; a.n_beam = R; a.ring_rows = ctx_len + kAlignWindow; a.T = T;
    a.single_base = ctx_len;
    a.z = z; a.attn_last = attn_last; a.frames = reinterpret_cast<int*>(res_dev) + 4 * max_rows;
    a.rows = blk_dev->rows;
    a.part = z + (size_t)max_rows * std::max(m->n_align, 1) * T;
    if (!select_fused_applicable(R, 2, a)) return false;
    EngineBlock& b = *blk_host;
    std::vector<int32_t> ids;
    std::vector<float> deltas;
    int n_adj = 0;
    for (int r = 0; r < R; ++r) {   // adjustments first: an overflow must leave the sessions untouched
        group[r]->job->adjustments(ids, deltas);
        if (n_adj + (int)ids.size() > kEngineFusedAdj) return false;
        for (size_t i = 0; i < ids.size(); ++i, ++n_adj) {
            b.adj_row[n_adj] = r;
            b.adj_ids[n_adj] = ids[i];
            b.adj_deltas[n_adj] = deltas[i];
        }
    }
    for (int r = 0; r < R; ++r) {
        wlk_session* s = group[r]->s;
        DecodeJob& job = *group[r]->job;
        if (s->self_len + 1 > ctx_len) throw std::runtime_error("text context exceeded");
        if (s->n_steps < 1) throw std::runtime_error("engine step before the prefill");
        const int after = s->n_steps + 1;
        StepRow& sr = b.rows[r];
        sr.kcache = s->kcache[s->kv_cur];
        sr.vcache = s->vcache[s->kv_cur];
        sr.cross_kv = s->cross_kv;
        sr.ring = s->ring;
        sr.token = (int)job.seq.back();
        if (sr.token < 0 || sr.token >= D.n_vocab) throw std::invalid_argument("token id out of range");
        sr.offset = s->self_len;
        sr.ring_row = ctx_len + ((s->n_steps - 1) % kAlignWindow);
        sr.prefill_rows = after <= kAlignWindow ? s->prefill_rows : 0;
        sr.n_single = std::min(after - 1, kAlignWindow);
        sr.newest_row = ctx_len + ((after - 2) % kAlignWindow);
        sr.content_len = std::min(job.P.content_mel_len, T);
        sr.pad = 0;
    }
    b.n_adj = n_adj;
    const unsigned seq = ++step_seq ? step_seq : ++step_seq;
    b.seq = seq;
    std::atomic_thread_fence(std::memory_order_release);
    hipGraphExec_t& exec = fstep_exec[R];
    if (!exec) {
        const LaunchCtx c{stream, nullptr};
        hipGraph_t graph = nullptr;
        // the rows' z-scores beside the vocabulary projection, medians beside the top-k slices, one folding wave per row at the
        // end (select.hip, early_z) - as a single session's graph step does.  Opt-in here (WLK_EARLY_Z_ENGINE=1): with eight
        // streams on the GPU the step's latency chain is not what bounds the throughput and up to 960 side workgroups compete
        // with the encode lane - 347.2 against 349.0 audio-s/s in three alternating pairs (profiles/r06m_early_z.txt)
        const char* ez = getenv("WLK_EARLY_Z_ENGINE");
        const bool early_z = ez && ez[0] == '1' && select_early_z_enabled() && V >= 16384;
        const int zf_blocks = (T + 63) / 64;
        if (early_z) WLK_HIP(hipMemcpyAsync(align_dev + R, &a, sizeof(AlignArgs), hipMemcpyHostToDevice, stream));
        WLK_HIP(hipStreamSynchronize(stream));
        WLK_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        try {
            if (early_z) enqueue_decoder(R, true, align_dev + R, zf_blocks * a.n_align * R, zf_blocks);
            else enqueue_decoder(R, true);
            StepHostOut ho;
            ho.result = res_host_dev;
            ho.n_adj = &blk_dev->n_adj;
            ho.seq = &blk_dev->seq;
            float* top_vals = res_dev;
            int* top_ids = reinterpret_cast<int*>(res_dev) + 2 * max_rows;
            if (!launch_select_fused(c, logits, V, R, 2, top_vals, top_ids, topk_scratch, blk_dev->adj_row, blk_dev->adj_ids,
                                     blk_dev->adj_deltas, 0, a, ho, nullptr, 0, nullptr, early_z))
                throw std::runtime_error("batched step: read-out not available");
        } catch (...) {
            (void)hipStreamEndCapture(stream, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            throw;
        }
        WLK_HIP(hipStreamEndCapture(stream, &graph));
        WLK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
    }
    WLK_HIP(hipGraphLaunch(exec, stream));
    wlk_wait_step_flags(stream, res_host, R, seq);
    n_batched += 1;
    n_batched_rows += R;
    for (int r = 0; r < R; ++r) {
        wlk_session* s = group[r]->s;
        s->self_len += 1;
        s->n_steps += 1;
        s->have_sot = false;
        s->last_rows = 1;
        s->last_ntok = 1;
        const volatile StepResult& res = res_host[r];
        const float lp[2] = {res.top_vals[0], res.top_vals[1]};
        const int32_t top[2] = {res.top_ids[0], res.top_ids[1]};
        if (!group[r]->job->consume(lp, top, res.frame)) finished.push_back(group[r]);
    }
    return true;
}

void wlk_engine::step_batched(std::vector<EngineJob*>& group, std::vector<EngineJob*>& finished) {
    if (step_batched_one_replay(group, finished)) return;
    const wlk_dims& D = m->D;
    const int R = (int)group.size();
    const int T = D.n_audio_ctx, ctx_len = D.n_text_ctx;
    StepRow* rows = reinterpret_cast<StepRow*>(pinned);
    int* adj = reinterpret_cast<int*>(pinned + 4096);
    std::vector<int32_t> ids, all_rows, all_ids;
    std::vector<float> deltas, all_deltas;
    for (int r = 0; r < R; ++r) {
        wlk_session* s = group[r]->s;
        DecodeJob& job = *group[r]->job;
        if (s->self_len + 1 > ctx_len) throw std::runtime_error("text context exceeded");
        if (s->n_steps < 1) throw std::runtime_error("engine step before the prefill");
        const int after = s->n_steps + 1;
        StepRow& sr = rows[r];
        sr.kcache = s->kcache[s->kv_cur];
        sr.vcache = s->vcache[s->kv_cur];
        sr.cross_kv = s->cross_kv;
        sr.ring = s->ring;
        sr.token = (int)job.seq.back();
        if (sr.token < 0 || sr.token >= D.n_vocab) throw std::invalid_argument("token id out of range");
        sr.offset = s->self_len;
        sr.ring_row = ctx_len + ((s->n_steps - 1) % kAlignWindow);
        sr.prefill_rows = after <= kAlignWindow ? s->prefill_rows : 0;
        sr.n_single = std::min(after - 1, kAlignWindow);
        sr.newest_row = ctx_len + ((after - 2) % kAlignWindow);
        sr.content_len = std::min(job.P.content_mel_len, T);
        sr.pad = 0;
        job.adjustments(ids, deltas);
        for (size_t i = 0; i < ids.size(); ++i) {
            all_rows.push_back(r);
            all_ids.push_back(ids[i]);
            all_deltas.push_back(deltas[i]);
        }
    }
    const int n_adj = (int)all_ids.size();
    if (n_adj > kEngineAdjCap) throw std::runtime_error("too many logit adjustments in one batched step");
    std::memcpy(adj, all_rows.data(), n_adj * sizeof(int));
    std::memcpy(adj + n_adj, all_ids.data(), n_adj * sizeof(int));
    std::memcpy(adj + 2 * n_adj, all_deltas.data(), n_adj * sizeof(float));
    WLK_HIP(hipMemcpyAsync(rows_dev, rows, sizeof(StepRow) * R, hipMemcpyHostToDevice, stream));
    if (n_adj > 0) WLK_HIP(hipMemcpyAsync(adj_dev, adj, (size_t)n_adj * 12, hipMemcpyHostToDevice, stream));
    if (use_graph) {
        hipGraphExec_t& exec = step_exec[R];
        if (!exec) {
            hipGraph_t graph = nullptr;
            WLK_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            try {
                enqueue_decoder(R);
            } catch (...) {
                (void)hipStreamEndCapture(stream, &graph);
                if (graph) (void)hipGraphDestroy(graph);
                throw;
            }
            WLK_HIP(hipStreamEndCapture(stream, &graph));
            WLK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(graph);
        }
        WLK_HIP(hipGraphLaunch(exec, stream));
    } else {
        enqueue_decoder(R);
    }
    enqueue_select(R, n_adj);
    float* out = reinterpret_cast<float*>(pinned + 4096 + (size_t)kEngineAdjCap * 12);
    WLK_HIP(hipMemcpyAsync(out, res_dev, (size_t)max_rows * 5 * 4, hipMemcpyDeviceToHost, stream));
    WLK_HIP(hipStreamSynchronize(stream));
    const int* out_ids = reinterpret_cast<const int*>(out) + 2 * max_rows;
    const int* out_frames = out_ids + 2 * max_rows;
    n_batched += 1;
    n_batched_rows += R;
    for (int r = 0; r < R; ++r) {
        wlk_session* s = group[r]->s;
        s->self_len += 1;
        s->n_steps += 1;
        s->have_sot = false;
        s->last_rows = 1;
        s->last_ntok = 1;
        if (!group[r]->job->consume(out + 2 * r, out_ids + 2 * r, out_frames[r])) finished.push_back(group[r]);
    }
}

void wlk_engine::step_single(EngineJob* j, std::vector<EngineJob*>& finished) {
    // a lone loop takes the session's own (graph-captured, single-row) launch chain
    DecodeJob& job = *j->job;
    const int64_t tok = job.seq.back();
    std::vector<int32_t> ids, rows;
    std::vector<float> deltas;
    float lp[2] = {0.f, 0.f};
    int32_t top[2] = {0, 0}, frame = 0;
    job.adjustments(ids, deltas);
    int rc = wlk_step_select(j->s, tok, ids.data(), deltas.data(), (int)ids.size(), job.P.content_mel_len, lp, top, &frame);
    if (rc == 1) {   // the session does not qualify for the one-replay step
        rc = wlk_decode(j->s, &tok, 1, 1, 0, job.P.sot_index);
        if (rc == WLK_OK) {
            rows.assign(ids.size(), -1);
            rc = wlk_select(j->s, rows.data(), ids.data(), deltas.data(), (int)ids.size(), 2, job.P.content_mel_len, lp, top,
                            &frame);
        }
    }
    if (rc != WLK_OK) {
        j->rc = rc;
        j->err = wlk_last_error();
        finished.push_back(j);
        return;
    }
    if (!job.consume(lp, top, frame)) finished.push_back(j);
}

void wlk_engine::run() {
    (void)hipSetDevice(m->device);
    std::vector<EngineJob*> active;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_work.wait(lk, [&] { return quit || !submitted.empty() || !active.empty(); });
            if (quit && active.empty() && submitted.empty()) return;
            while (!submitted.empty()) {
                active.push_back(submitted.front());
                submitted.pop_front();
            }
        }
        std::vector<EngineJob*> finished;
        for (size_t lo = 0; lo < active.size(); lo += (size_t)max_rows) {
            std::vector<EngineJob*> group;
            for (size_t i = lo; i < std::min(active.size(), lo + (size_t)max_rows); ++i) {
                EngineJob* j = active[i];
                if (!j->job->begin_step()) {
                    finished.push_back(j);        // text context full / token budget spent
                    continue;
                }
                // host-side preconditions of a step are checked per session BEFORE anything is launched, so a session
                // in a bad state fails alone instead of taking the rows batched with it down
                const char* why = nullptr;
                if (j->s->self_len + 1 > m->D.n_text_ctx) why = "text context exceeded";
                else if (j->s->n_steps < 1) why = "engine step before the prefill";
                else if (j->job->seq.empty() || j->job->seq.back() < 0 || j->job->seq.back() >= m->D.n_vocab) why = "token id out of range";
                if (why) {
                    j->rc = WLK_ERR_STATE;
                    j->err = why;
                    finished.push_back(j);
                    continue;
                }
                group.push_back(j);
            }
            try {
                if (group.size() == 1) step_single(group[0], finished);
                else if (group.size() > 1) step_batched(group, finished);
            } catch (const std::exception& e) {
                for (EngineJob* j : group)
                    if (std::find(finished.begin(), finished.end(), j) == finished.end()) {
                        j->rc = dynamic_cast<const HipError*>(&e) ? WLK_ERR_HIP : WLK_ERR_STATE;
                        j->err = e.what();
                        finished.push_back(j);
                    }
            }
            n_iterations += 1;
            n_rows += group.size();
        }
        if (!finished.empty()) {
            for (EngineJob* j : finished) active.erase(std::remove(active.begin(), active.end(), j), active.end());
            {
                std::lock_guard<std::mutex> lk(mu);
                for (EngineJob* j : finished) j->done = true;
            }
            cv_done.notify_all();
        }
    }
}

void wlk_engine::run_encodes(int lane) {
    (void)hipSetDevice(m->device);
    hipStream_t enc_stream = enc_streams[lane];
    for (;;) {
        std::vector<EncodeReq*> batch;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_enc_work.wait(lk, [&] { return quit || !enc_submitted.empty(); });
            if (quit && enc_submitted.empty()) return;
            if (gather_us > 0) {
                // optional gather window (WLK_ENCODE_GATHER_US, default 0 = never wait): while other sessions of this GPU
                // are inside a decode loop, i.e. about to come back with their next chunk, hold the launch briefly so
                // their encodes ride in the same chain.  A throughput-for-latency trade for saturated servers only.
                const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(gather_us);
                // target: half of the sessions that are busy on this GPU (two alternating groups keep the encode lane
                // and the decode steps of the other group running side by side)
                while (!quit && (int)enc_submitted.size() < std::min(kMaxBatch, (in_loop.load(std::memory_order_relaxed) + 1) / 2) &&
                       cv_enc_work.wait_until(lk, deadline) != std::cv_status::timeout) {
                }
            }
            while (!enc_submitted.empty() && (int)batch.size() < enc_batch_cap) {
                batch.push_back(enc_submitted.front());
                enc_submitted.pop_front();
            }
        }
        // requests that cannot be encoded (audio past the session's frame capacity, a session of another model) are
        // answered on their own before anything is launched: one misbehaving stream must not fail the others of its batch
        std::vector<EncodeReq*> good;
        for (EncodeReq* r : batch) {
            const std::string why = wlk_encode_precheck(r->s, m);
            if (why.empty()) {
                good.push_back(r);
            } else {
                r->rc = WLK_ERR_CAPACITY;
                r->err = why;
            }
        }
        auto encode = [&](const std::vector<EncodeReq*>& reqs) {     // -> status shared by `reqs`
            int rc = WLK_OK;
            std::string err;
            try {
                std::vector<wlk_session*> group;
                for (EncodeReq* r : reqs) group.push_back(r->s);
                std::vector<int> content;
                wlk_encode_group(group, LaunchCtx{enc_stream, nullptr}, content);
                WLK_HIP(hipStreamSynchronize(enc_stream));
                for (size_t i = 0; i < reqs.size(); ++i) reqs[i]->content = content[i];
            } catch (const std::length_error& e) {
                rc = WLK_ERR_CAPACITY;
                err = e.what();
            } catch (const std::exception& e) {
                rc = dynamic_cast<const HipError*>(&e) ? WLK_ERR_HIP : WLK_ERR_STATE;
                err = e.what();
            }
            for (EncodeReq* r : reqs) {
                r->rc = rc;
                r->err = err;
            }
            return rc;
        };
        if (!good.empty() && encode(good) != WLK_OK && good.size() > 1) {
            // the stacked chain failed as a whole: once more, one session at a time, so that only the session(s) that
            // actually fail report an error (a sticky HIP error fails them all again - nothing is lost by trying)
            for (EncodeReq* r : good) encode({r});
        }
        n_enc_batches.fetch_add(1, std::memory_order_relaxed);
        n_enc_sessions.fetch_add(good.size(), std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu);
            for (EncodeReq* r : batch) r->done = true;
        }
        cv_enc_done.notify_all();
    }
}

void wlk_engine::run_prefills() {
    (void)hipSetDevice(m->device);
    const LaunchCtx c{pre_stream, nullptr};
    for (;;) {
        std::vector<PrefillReq*> batch;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_pre_work.wait(lk, [&] { return quit || !pre_submitted.empty(); });
            if (quit && pre_submitted.empty()) return;
            if (pre_gather_us > 0) {
                // optional gather window (WLK_PREFILL_GATHER_US): hold the launch briefly while other sessions of this GPU
                // are still in their encode, i.e. about to arrive with their prefill
                const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(pre_gather_us);
                while (!quit && (int)pre_submitted.size() < std::min(kMaxBatch, in_loop.load(std::memory_order_relaxed)) &&
                       cv_pre_work.wait_until(lk, deadline) != std::cv_status::timeout) {
                }
            }
            int rows = 0;
            while (!pre_submitted.empty() && (int)batch.size() < kMaxBatch) {
                const int padded = (pre_submitted.front()->item.n_tok + 31) / 32 * 32;
                if (rows + padded > pre_ws.cap_rows) break;
                rows += padded;
                batch.push_back(pre_submitted.front());
                pre_submitted.pop_front();
            }
        }
        auto run = [&](const std::vector<PrefillReq*>& reqs) {
            int rc = WLK_OK;
            std::string err;
            try {
                std::vector<wlk_prefill_item*> items;
                for (PrefillReq* r : reqs) items.push_back(&r->item);
                wlk_prefill_group(items, c, pre_ws);
                WLK_HIP(hipStreamSynchronize(pre_stream));
            } catch (const std::exception& e) {
                rc = dynamic_cast<const HipError*>(&e) ? WLK_ERR_HIP : WLK_ERR_STATE;
                err = e.what();
            }
            for (PrefillReq* r : reqs) {
                r->item.rc = rc;
                r->item.err = err;
            }
            return rc;
        };
        if (run(batch) != WLK_OK && batch.size() > 1)
            for (PrefillReq* r : batch) run({r});        // once more one by one: only the session(s) that fail report it
        n_pre_batches.fetch_add(1, std::memory_order_relaxed);
        n_pre_sessions.fetch_add(batch.size(), std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu);
            for (PrefillReq* r : batch) r->done = true;
        }
        cv_pre_done.notify_all();
    }
}

// ---- lifetime -----------------------------------------------------------------------------------------------
static wlk_engine* engine_create(wlk_model* m) {
    const wlk_dims& D = m->D;
    auto e = std::make_unique<wlk_engine>();
    e->m = m;
    int rows = 8;     // the GEMV kernels stage `bucket x K` activations in <= 64 KiB of LDS (K up to 4d)
    while (rows > 1 && !gemv_applicable(rows, 4 * D.n_text_state)) rows /= 2;
    e->max_rows = rows;
    if (const char* env = std::getenv("WLK_ENGINE_MAX_ROWS")) e->max_rows = std::max(1, std::min(rows, std::atoi(env)));
    WLK_HIP(hipSetDevice(m->device));
    // the batched decode steps are chains of tiny latency-bound kernels, the encodes are GPU-filling GEMMs: the step
    // stream gets the highest priority so its workgroups are dispatched ahead of queued encoder workgroups
    // (WLK_ENGINE_PRIORITY=0 switches that off)
    int prio_lo = 0, prio_hi = 0;
    WLK_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    const char* pe = std::getenv("WLK_ENGINE_PRIORITY");
    const bool prio = !(pe && pe[0] == '0');
    WLK_HIP(hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, prio ? prio_hi : prio_lo));
    if (const char* g = std::getenv("WLK_NO_GRAPH")) e->use_graph = !(g[0] == '1');
    int lanes = 1;
    if (const char* g = std::getenv("WLK_ENGINE_ENCODE_LANES")) lanes = std::max(1, std::min(4, std::atoi(g)));
    if (const char* g = std::getenv("WLK_ENGINE_ENCODE_BATCH")) e->enc_batch_cap = std::max(1, std::min(kMaxBatch, std::atoi(g)));
    e->enc_streams.resize(lanes, nullptr);
    for (auto& st : e->enc_streams) WLK_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio_lo));
    if (const char* g = std::getenv("WLK_BATCH_ENCODE")) e->batch_encodes = !(g[0] == '0');
    if (const char* g = std::getenv("WLK_ENCODE_GATHER_US")) e->gather_us = std::max(0, std::atoi(g));
    if (const char* g = std::getenv("WLK_BATCH_PREFILL")) e->batch_prefills = !(g[0] == '0');
    if (const char* g = std::getenv("WLK_PREFILL_GATHER_US")) e->pre_gather_us = std::max(0, std::atoi(g));
    if (const char* g = std::getenv("WLK_PREFILL_MIN_SESSIONS")) e->pre_min_sessions = std::max(2, std::atoi(g));
    WLK_HIP(hipStreamCreateWithPriority(&e->pre_stream, hipStreamNonBlocking, prio ? prio_hi : prio_lo));
    if (e->batch_prefills) wlk_prefill_ws_alloc(m, e->pre_ws, kMaxBatch, std::min(256, (int)D.n_text_ctx));
    const size_t R = 8, d = D.n_text_state, T = D.n_audio_ctx, V = D.n_vocab;
    e->x = dev_alloc<float>(R * d);
    e->qkv = dev_alloc<float>(R * 3 * d);
    e->att = dev_alloc<float>(R * d);
    e->q = dev_alloc<float>(R * d);
    e->mlp = dev_alloc<float>(R * 4 * d);
    e->logits = dev_alloc<float>(R * V);
    e->xsplit = dev_alloc<float>(cross_split_scratch_floats(8, D.n_text_head, (int)T));
    e->z = dev_alloc<float>(R * std::max(m->n_align, 1) * T + R * 128);   // + the read-out's per-block (value, frame) pairs
    e->align_dev = reinterpret_cast<AlignArgs*>(dev_alloc<char>(9 * sizeof(AlignArgs)));
    e->attn_last = dev_alloc<float>(R * T);
    e->res_dev = dev_alloc<float>(R * 5);
    WLK_HIP(hipMalloc(&e->topk_scratch, topk_scratch_bytes((int)R)));
    e->adj_dev = dev_alloc<int>(3 * (size_t)kEngineAdjCap);
    WLK_HIP(hipHostMalloc(reinterpret_cast<void**>(&e->pinned), wlk_engine::kPinnedBytes, hipHostMallocDefault));
    {
        static_assert(sizeof(EngineBlock) <= 16384, "results start 16 KiB into the host-coherent block");
        void* hp = nullptr;
        WLK_HIP(hipHostMalloc(&hp, 32768, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(hp, 0, 32768);
        void* dp = nullptr;
        WLK_HIP(hipHostGetDevicePointer(&dp, hp, 0));
        e->blk_host = static_cast<EngineBlock*>(hp);
        e->blk_host_dev = static_cast<EngineBlock*>(dp);
        e->res_host = reinterpret_cast<StepResult*>(static_cast<char*>(hp) + 16384);
        e->res_host_dev = reinterpret_cast<StepResult*>(static_cast<char*>(dp) + 16384);
        e->blk_dev = reinterpret_cast<EngineBlock*>(dev_alloc<int>(sizeof(EngineBlock) / 4));
        e->rows_dev = e->blk_dev->rows;   // the row table of either step form lives inside the block's device copy
        if (const char* g = std::getenv("WLK_FUSED_STEP")) e->fused_steps = !(g[0] == '0');
    }
    WLK_HIP(hipStreamSynchronize(e->stream));
    wlk_engine* raw = e.release();
    raw->worker = std::thread([raw] { raw->run(); });
    for (int lane = 0; lane < (int)raw->enc_streams.size(); ++lane)
        raw->enc_workers.emplace_back([raw, lane] { raw->run_encodes(lane); });
    raw->pre_worker = std::thread([raw] { raw->run_prefills(); });
    return raw;
}

void wlk_engine_destroy_for_model(wlk_model* m) {
    wlk_engine* e = m->engine;
    if (!e) return;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        e->quit = true;
    }
    e->cv_work.notify_all();
    e->cv_enc_work.notify_all();
    e->cv_pre_work.notify_all();
    if (e->pre_worker.joinable()) e->pre_worker.join();
    if (e->worker.joinable()) e->worker.join();
    for (auto& w : e->enc_workers)
        if (w.joinable()) w.join();
    for (auto& st : e->enc_streams)
        if (st) (void)hipStreamDestroy(st);
    (void)hipSetDevice(m->device);
    if (e->pre_stream) (void)hipStreamDestroy(e->pre_stream);
    wlk_prefill_ws_free(e->pre_ws);
    float* fl[] = {e->x, e->qkv, e->att, e->q, e->mlp, e->logits, e->xsplit, e->z, e->attn_last, e->res_dev};
    for (float* p : fl)
        if (p) (void)hipFree(p);
    if (e->topk_scratch) (void)hipFree(e->topk_scratch);
    if (e->align_dev) (void)hipFree(e->align_dev);
    if (e->adj_dev) (void)hipFree(e->adj_dev);
    if (e->pinned) (void)hipHostFree(e->pinned);
    if (e->blk_host) (void)hipHostFree(e->blk_host);
    if (e->blk_dev) (void)hipFree(e->blk_dev);
    for (auto& g : e->fstep_exec)
        if (g) (void)hipGraphExecDestroy(g);
    for (auto& g : e->step_exec)
        if (g) (void)hipGraphExecDestroy(g);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    m->engine = nullptr;
}

// the rest of a decode loop (after its prefill step) on the engine; blocks the calling thread until the loop stops
int wlk_engine_run_job(wlk_session* s, DecodeJob* job) {
    wlk_engine* e = s->engine;
    EngineJob ej;
    ej.s = s;
    ej.job = job;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        e->submitted.push_back(&ej);
    }
    e->cv_work.notify_one();
    {
        std::unique_lock<std::mutex> lk(e->mu);
        e->cv_done.wait(lk, [&] { return ej.done; });
    }
    if (ej.rc != WLK_OK) set_last_error(ej.err);
    return ej.rc;
}

bool wlk_engine_wants(const wlk_session* s) {
    // a loop that is alone on this GPU stays on the calling thread (nothing to batch with, no hand-off latency); it is
    // asked again after every step, so it moves over as soon as a second loop shows up
    return s->engine && s->engine->in_loop.load(std::memory_order_relaxed) > 1 && !s->debug && !s->prof_on && s->beam == 1;
}

bool wlk_engine_batches_encodes(const wlk_session* s) {
    // only when another session of this GPU is busy (inside an encode or a decode loop) right now; a lone stream
    // encodes on its own stream without the hand-off
    return s->engine && s->engine->batch_encodes && !s->debug && !s->prof_on &&
           s->engine->in_loop.load(std::memory_order_relaxed) > 1;     // the caller itself is counted (wlk_encode)
}

int wlk_engine_encode(wlk_session* s, int* content_mel_len) {
    wlk_engine* e = s->engine;
    // the encode lane runs on the engine's stream: what this session still has in flight on its own stream (the audio
    // chunk on its way to the ring, an eviction shift) must be in place first
    if (hipStreamSynchronize(s->stream) != hipSuccess) return fail(WLK_ERR_HIP, "encode: the session's stream failed");
    EncodeReq req;
    req.s = s;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        e->enc_submitted.push_back(&req);
    }
    e->cv_enc_work.notify_one();
    {
        std::unique_lock<std::mutex> lk(e->mu);
        e->cv_enc_done.wait(lk, [&] { return req.done; });
    }
    if (req.rc != WLK_OK) set_last_error(req.err);
    if (content_mel_len) *content_mel_len = req.content;
    return req.rc;
}

int wlk_engine_prefill(wlk_session* s, const int64_t* tokens, int n_tok, int sot_index) {
    wlk_engine* e = s->engine;
    // the stacked chain pays once enough sessions are busy on this GPU for several prefills to be waiting together
    // (measured: neutral at 8 streams - 1.4 sessions per stack -, +3 % at 16, +8 % at 32; WLK_PREFILL_MIN_SESSIONS);
    // below that the session runs its own chain without the hand-off
    if (!e || !e->batch_prefills || e->in_loop.load(std::memory_order_relaxed) < e->pre_min_sessions) return 1;
    PrefillReq req;
    req.item.s = s;
    req.item.tokens = tokens;
    req.item.n_tok = n_tok;
    req.item.sot_index = sot_index;
    if (!wlk_prefill_precheck(req.item, e->pre_ws).empty()) return 1;      // wlk_decode reports what is wrong, if anything
    // the lane runs on its own stream: what this session still has in flight on its stream must be in place first
    if (hipStreamSynchronize(s->stream) != hipSuccess) return fail(WLK_ERR_HIP, "prefill: the session's stream failed");
    {
        std::lock_guard<std::mutex> lk(e->mu);
        e->pre_submitted.push_back(&req);
    }
    e->cv_pre_work.notify_one();
    {
        std::unique_lock<std::mutex> lk(e->mu);
        e->cv_pre_done.wait(lk, [&] { return req.done; });
    }
    if (req.item.rc != WLK_OK) set_last_error(req.item.err);
    return req.item.rc;
}

void wlk_engine_loop_enter(wlk_session* s) {
    if (s->engine) s->engine->in_loop.fetch_add(1, std::memory_order_relaxed);
}
void wlk_engine_loop_exit(wlk_session* s) {
    if (s->engine) s->engine->in_loop.fetch_sub(1, std::memory_order_relaxed);
}

extern "C" {

int wlk_engine_attach(wlk_session* s) {
    if (!s) return fail(WLK_ERR_ARG, "session is NULL");
    if (s->beam != 1) return fail(WLK_ERR_ARG, "only beam-1 sessions can share batched decode steps");
    if (s->engine) return WLK_OK;
    return guarded([&]() {
        wlk_model* m = s->m;
        std::lock_guard<std::mutex> lk(m->engine_mu);
        if (!m->engine) m->engine = engine_create(m);
        {
            std::lock_guard<std::mutex> lk2(m->engine->mu);
            m->engine->attached += 1;
        }
        s->engine = m->engine;
        return WLK_OK;
    });
}

int wlk_engine_detach(wlk_session* s) {
    if (!s) return fail(WLK_ERR_ARG, "session is NULL");
    if (!s->engine) return WLK_OK;
    {
        std::lock_guard<std::mutex> lk(s->engine->mu);
        s->engine->attached -= 1;
    }
    s->engine = nullptr;
    return WLK_OK;
}

int wlk_engine_encode_stats(wlk_model* m, uint64_t* batches, uint64_t* sessions) {
    if (!m) return fail(WLK_ERR_ARG, "model is NULL");
    std::lock_guard<std::mutex> lk(m->engine_mu);
    if (batches) *batches = m->engine ? m->engine->n_enc_batches.load(std::memory_order_relaxed) : 0;
    if (sessions) *sessions = m->engine ? m->engine->n_enc_sessions.load(std::memory_order_relaxed) : 0;
    return WLK_OK;
}

int wlk_engine_prefill_stats(wlk_model* m, uint64_t* batches, uint64_t* sessions) {
    if (!m) return fail(WLK_ERR_ARG, "model is NULL");
    std::lock_guard<std::mutex> lk(m->engine_mu);
    if (batches) *batches = m->engine ? m->engine->n_pre_batches.load(std::memory_order_relaxed) : 0;
    if (sessions) *sessions = m->engine ? m->engine->n_pre_sessions.load(std::memory_order_relaxed) : 0;
    return WLK_OK;
}

int wlk_engine_stats(wlk_model* m, uint64_t* iterations, uint64_t* rows, uint64_t* batched_steps, uint64_t* batched_rows) {
    if (!m) return fail(WLK_ERR_ARG, "model is NULL");
    std::lock_guard<std::mutex> lk(m->engine_mu);
    wlk_engine* e = m->engine;
    if (iterations) *iterations = e ? e->n_iterations.load(std::memory_order_relaxed) : 0;
    if (rows) *rows = e ? e->n_rows.load(std::memory_order_relaxed) : 0;
    if (batched_steps) *batched_steps = e ? e->n_batched.load(std::memory_order_relaxed) : 0;
    if (batched_rows) *batched_rows = e ? e->n_batched_rows.load(std::memory_order_relaxed) : 0;
    return WLK_OK;
}

}  // extern "C"

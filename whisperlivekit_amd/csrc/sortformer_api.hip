// C ABI of the streaming Sortformer diarizer network (include/wlk_hip.h, "a12 network"): packed weight arena,
// workspace, and the launch sequence of one streaming step.  Kernels: sortformer.hip, gemm_f32.hip, layernorm.hip.
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/wlk_hip.h"
#include "common.h"

namespace wlk {

static int sf_fail(int code, const std::string& msg) {
    set_last_error(msg);
    return code;
}
template <typename F>
static int sf_guarded(F&& f) {
    try {
        return f();
    } catch (const HipError& e) {
        return sf_fail(WLK_ERR_HIP, e.what());
    } catch (const std::invalid_argument& e) {
        return sf_fail(WLK_ERR_ARG, e.what());
    } catch (const std::exception& e) {
        return sf_fail(WLK_ERR_STATE, e.what());
    }
}

struct SfSlot {
    std::string name;
    uint64_t offset, numel;
};

static int sf_freq_out(const wlk_sf_dims& D) { return sf_sub_len(sf_sub_len(sf_sub_len(D.n_mels))); }

static std::vector<SfSlot> sf_layout(const wlk_sf_dims& D, uint64_t* total) {
    std::vector<SfSlot> v;
    uint64_t off = 0;
    auto add = [&](const std::string& n, uint64_t numel) {
        v.push_back({n, off, numel});
        off += (numel + 63) / 64 * 64;
    };
    const uint64_t C = D.sub_channels, d = D.fc_d_model, ff = D.fc_ff, dt = D.tf_d_model, in = D.tf_inner;
    add("pre.conv0.w", C * 9); add("pre.conv0.b", C);
    for (int i = 1; i <= 2; ++i) {
        const std::string s = std::to_string(i);
        add("pre.dw" + s + ".w", 9 * C); add("pre.dw" + s + ".b", C);
        add("pre.pw" + s + ".w", C * C); add("pre.pw" + s + ".b", C);
    }
    add("pre.out.w", d * C * sf_freq_out(D)); add("pre.out.b", d);
    add("pos.table", (uint64_t)(2 * D.max_frames - 1) * d);
    for (int i = 0; i < D.fc_layers; ++i) {
        const std::string p = "fc." + std::to_string(i) + ".";
        add(p + "ln_ff1.w", d); add(p + "ln_ff1.b", d);
        add(p + "ff1a.w", ff * d); add(p + "ff1a.b", ff);
        add(p + "ff1b.w", d * ff); add(p + "ff1b.b", d);
        add(p + "ln_att.w", d); add(p + "ln_att.b", d);
        add(p + "qkv.w", 3 * d * d); add(p + "qkv.b", 3 * d);
        add(p + "pos.w", d * d);
        add(p + "bias_u", d); add(p + "bias_v", d);
        add(p + "out.w", d * d); add(p + "out.b", d);
        add(p + "ln_conv.w", d); add(p + "ln_conv.b", d);
        add(p + "pw1.w", 2 * d * d); add(p + "pw1.b", 2 * d);
        add(p + "dw.w", (uint64_t)D.conv_kernel * d); add(p + "dw.b", d);
        add(p + "bn.mean", d); add(p + "bn.invstd", d); add(p + "bn.w", d); add(p + "bn.b", d);
        add(p + "pw2.w", d * d); add(p + "pw2.b", d);
        add(p + "ln_ff2.w", d); add(p + "ln_ff2.b", d);
        add(p + "ff2a.w", ff * d); add(p + "ff2a.b", ff);
        add(p + "ff2b.w", d * ff); add(p + "ff2b.b", d);
        add(p + "ln_out.w", d); add(p + "ln_out.b", d);
    }
    add("proj.w", dt * d); add("proj.b", dt);
    for (int i = 0; i < D.tf_layers; ++i) {
        const std::string p = "tf." + std::to_string(i) + ".";
        add(p + "qkv.w", 3 * dt * dt); add(p + "qkv.b", 3 * dt);
        add(p + "out.w", dt * dt); add(p + "out.b", dt);
        add(p + "ln1.w", dt); add(p + "ln1.b", dt);
        add(p + "in.w", in * dt); add(p + "in.b", in);
        add(p + "outd.w", dt * in); add(p + "outd.b", dt);
        add(p + "ln2.w", dt); add(p + "ln2.b", dt);
    }
    add("head.h.w", dt * dt); add("head.h.b", dt);
    add("head.s.w", (uint64_t)D.n_spk * dt); add("head.s.b", D.n_spk);
    if (total) *total = off;
    return v;
}

static int sf_check_dims(const wlk_sf_dims* d) {
    if (!d) return sf_fail(WLK_ERR_ARG, "dims is NULL");
    if (d->n_mels < 8 || d->sub_channels < 4 || d->sub_channels % 4 || d->fc_layers < 0 || d->tf_layers < 0 ||
        d->fc_heads < 1 || d->tf_heads < 1 || d->n_spk < 1 || d->n_spk > 64)
        return sf_fail(WLK_ERR_ARG, "bad Sortformer dimensions");
    if (d->fc_d_model % d->fc_heads || d->tf_d_model % d->tf_heads) return sf_fail(WLK_ERR_ARG, "width not divisible by heads");
    const int dh = d->fc_d_model / d->fc_heads, dht = d->tf_d_model / d->tf_heads;
    if (dh % 4 || dh > 64 || dht % 4 || dht > 64) return sf_fail(WLK_ERR_ARG, "head width must be a multiple of 4, <= 64");
    if (d->fc_d_model % 4 || d->fc_ff % 4 || d->tf_d_model % 4 || d->tf_inner % 4 || d->fc_d_model > 1536 || d->tf_d_model > 1536)
        return sf_fail(WLK_ERR_ARG, "layer widths must be multiples of 4 (<= 1536 for the model widths)");
    if (d->conv_kernel < 1 || d->conv_kernel % 2 == 0) return sf_fail(WLK_ERR_ARG, "conv_kernel must be odd");
    if (d->max_frames < 8 || d->max_frames > kSfMaxFrames) return sf_fail(WLK_ERR_ARG, "max_frames out of range");
    if (d->max_feat_frames < 8 || d->max_feat_frames > 8 * kSfMaxFrames) return sf_fail(WLK_ERR_ARG, "max_feat_frames out of range");
    return WLK_OK;
}

struct SfFcLayer {
    const float *ln_ff1_w, *ln_ff1_b, *ff1a_w, *ff1a_b, *ff1b_w, *ff1b_b, *ln_att_w, *ln_att_b, *qkv_w, *qkv_b, *pos_w,
        *bias_u, *bias_v, *out_w, *out_b, *ln_conv_w, *ln_conv_b, *pw1_w, *pw1_b, *dw_w, *dw_b, *bn_mean, *bn_invstd,
        *bn_w, *bn_b, *pw2_w, *pw2_b, *ln_ff2_w, *ln_ff2_b, *ff2a_w, *ff2a_b, *ff2b_w, *ff2b_b, *ln_out_w, *ln_out_b;
};
struct SfTfLayer {
    const float *qkv_w, *qkv_b, *out_w, *out_b, *ln1_w, *ln1_b, *in_w, *in_b, *outd_w, *outd_b, *ln2_w, *ln2_b;
};

}  // namespace wlk

using namespace wlk;

// One caller's step on its way through the model: inputs and outputs in PINNED host memory (the caller copies its numpy
// rows in and out itself, in parallel with other callers; the lane's copies are then truly asynchronous).
struct SfRequest {
    float* feats = nullptr;     // [max_feat_frames][n_mels]: the features the stem sees; with a PCM chunk: [n_prev rows the caller
                                //   kept from its previous chunk | the chunk's own log-mel rows, written back by the step]
    float* pcm = nullptr;       // [max_pcm] the new audio chunk (wlk_sf_step_pcm), or unused
    const wlk_melspec* mel = nullptr;
    int n_pcm = 0, n_prev = 0, zero_from = 0;
    float* ctx = nullptr;       // [max_frames][d]
    float* chunk = nullptr;     // [max chunk rows][d]      (out)
    float* preds = nullptr;     // [max_frames][n_spk]      (out)
    int n_feat = 0, n_ctx = 0, Tc = 0, T = 0;
    bool in_use = false, queued = false, done = false;
    int rc = WLK_OK;
    std::string err;
    uint64_t ticket = 0;
};

struct wlk_sortformer {
    wlk_sf_dims D{};
    int device = 0;
    float* arena = nullptr;
    uint64_t arena_floats = 0;
    std::vector<SfSlot> layout;
    std::map<std::string, const SfSlot*> index;
    bool finalized = false;
    std::mutex mu;
    std::vector<SfFcLayer> fc;
    std::vector<SfTfLayer> tf;
    // Lanes (round 6; round 5 called them workspaces): one lane runs one STACKED step at a time on its own stream - the
    // steps of up to `max_batch` sessions that were waiting when the lane became free, rows one after the other
    // ([speaker cache | FIFO | chunk] of session 0, then of session 1, ...), ONE launch chain for all of them.  Shared
    // weights are read once per chain instead of once per session, the GEMMs see M = sum of the sessions' frames (the
    // one-tile-per-CU kernels' home turf instead of 32 x 32 tiles), attention / depthwise convolution stay inside their
    // session by a by-value segment table.  Every kernel's per-row arithmetic is independent of what is stacked around
    // the row (launch_gemm_kp), so a session's result is bit for bit its step alone.  There is no gather window: a free
    // lane takes whatever is queued (work-conserving, like the ASR engine's encode lane).
    // WLK_SF_WORKSPACES = lanes (default 1), WLK_SF_BATCH = sessions per stacked step (default 8, 1 = round-5 behaviour).
    struct Lane {
        hipStream_t stream = nullptr;
        float *audio = nullptr;          // [max_batch][max_pcm]: the sessions' new audio chunks (wlk_sf_step_pcm)
        float *feats = nullptr, *ca = nullptr, *cb = nullptr, *ctxbuf = nullptr, *chunk_tmp = nullptr, *x = nullptr, *xn = nullptr,
              *wide = nullptr, *qkv = nullptr, *att = nullptr, *tx = nullptr, *ty = nullptr, *tqkv = nullptr, *tatt = nullptr,
              *th = nullptr, *preds = nullptr;
        int last_T = 0;                  // rows of the FIRST session of the lane's most recent step (exports)
        bool busy = false;
    };
    std::vector<Lane> ws;
    std::vector<SfRequest> slots;
    std::deque<int> queue;                // slot indices waiting for a lane, FIFO
    std::condition_variable cv;
    uint64_t next_ticket = 0;
    int max_batch = 8;
    int max_pcm = 64000;                  // samples of a PCM chunk a request slot holds (4 s)
    int last_ws = 0;                      // the lane of the most recent step (exports)
    uint64_t n_steps = 0, n_sessions = 0, n_rows = 0;     // stacked steps run / session steps in them / rows in them
    float* pos_full = nullptr;            // read-only after finalize
    float* head_w1t = nullptr;            // head.h.w transposed ([in][out]) at finalize: coalesced rows for sf_head_kernel
    std::vector<float*> owned;
    std::vector<float*> pinned;
    const float* P(const std::string& n) const {
        auto it = index.find(n);
        if (it == index.end()) throw std::invalid_argument("unknown packed tensor " + n);
        return arena + it->second->offset;
    }
    ~wlk_sortformer() {                  // also the clean-up of a wlk_sf_create that threw half-way
        (void)hipSetDevice(device);
        for (float* p : owned)
            if (p) (void)hipFree(p);
        for (float* p : pinned)
            if (p) (void)hipHostFree(p);
        for (auto& w : ws)
            if (w.stream) (void)hipStreamDestroy(w.stream);
    }
};

namespace wlk {
static float* sf_alloc(wlk_sortformer* m, size_t n) {
    float* p = nullptr;
    WLK_HIP(hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(float)));
    m->owned.push_back(p);
    return p;
}

// every projection of the network: the "kp" kernel family - one per-element arithmetic whatever M is (gemm_f32.hip)
static void sf_linear(const LaunchCtx& c, const float* A, long lda, const float* W, const float* b, float* C, long ldc,
                      int M, int N, int K, int flags, const float* R, long ldr, const char* tag, float scale = 1.f,
                      int scale_cols = 0) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.bias = b; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.flags = flags; g.R = R; g.ldr = ldr; g.scale = scale; g.scale_cols = scale_cols;
    launch_gemm_kp(c, g, tag);
}

// LayerNorm + projection: ONE launch while the rows are few (a single session's step: the 16 x 16 kernel normalises its A rows
// itself, bit for bit layernorm_kernel's values), LayerNorm launch + projection from 512 rows on (stacked steps: the k-pipe tiles
// take their operands by LDS-DMA and cannot normalise them on the way).  Same results either way.
static void sf_ln_linear(const LaunchCtx& c, const float* x, const float* lnw, const float* lnb, float* xn, const float* W,
                         const float* b, float* C, long ldc, int M, int N, int K, int flags, const char* tag) {
    if (gemm_kp_fuses_layernorm(M, N, K)) {
        GemmArgs g;
        g.A = x; g.lda = K; g.W = W; g.bias = b; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.flags = flags;
        g.ln_gamma = lnw; g.ln_beta = lnb;
        launch_gemm_kp(c, g, tag);
    } else {
        launch_layernorm(c, x, K, lnw, lnb, xn, K, M, K, "sf_ln");
        sf_linear(c, xn, K, W, b, C, ldc, M, N, K, flags, nullptr, 0, tag);
    }
}

static void sf_set_segments(SfAttnArgs& a, const SfSegments& rows) {
    a.n_seg = rows.n;
    a.T = 0;
    for (int s = 0; s < rows.n; ++s) {
        a.seg_start[s] = rows.start[s];
        a.seg_T[s] = rows.len[s];
        a.T = std::max(a.T, rows.len[s]);
    }
}

// the network over w_->x (already scaled) -> w_->preds, rows = the stacked sessions' [start, len) ranges
static void sf_network(wlk_sortformer* m, wlk_sortformer::Lane* w_, const LaunchCtx& c, const SfSegments& rows) {
    const wlk_sf_dims& D = m->D;
    const int d = D.fc_d_model, ff = D.fc_ff, dh = d / D.fc_heads, L = D.max_frames;
    const int T = rows.start[rows.n - 1] + rows.len[rows.n - 1];          // stacked rows
    for (int l = 0; l < D.fc_layers; ++l) {
        const SfFcLayer& w = m->fc[l];
        // x += 0.5 * FF1(LN(x))   (the 0.5 is folded into ff1b at pack time: exact, a power of two)
        sf_ln_linear(c, w_->x, w.ln_ff1_w, w.ln_ff1_b, w_->xn, w.ff1a_w, w.ff1a_b, w_->wide, ff, T, ff, d, kGemmSwish, "sf_ff_a");
        sf_linear(c, w_->wide, ff, w.ff1b_w, w.ff1b_b, w_->x, d, T, d, ff, kGemmResidual, w_->x, d, "sf_ff_b");
        // x += RelPosMHA(LN(x))
        sf_ln_linear(c, w_->x, w.ln_att_w, w.ln_att_b, w_->xn, w.qkv_w, w.qkv_b, w_->qkv, 3 * d, T, 3 * d, d, 0, "sf_qkv");
        SfAttnArgs a;
        a.q = w_->qkv; a.k = w_->qkv + d; a.v = w_->qkv + 2 * d; a.ldq = a.ldk = a.ldv = 3 * d;
        a.out = w_->att; a.ldo = d; a.n_head = D.fc_heads; a.dh = dh; a.scale = 1.0f / sqrtf((float)dh);
        a.pos = m->pos_full + (size_t)l * (2 * L - 1) * d; a.ldp = d; a.pos_row0 = L - 1;
        a.bias_u = w.bias_u; a.bias_v = w.bias_v;
        sf_set_segments(a, rows);
        launch_sf_attention(c, a);
        sf_linear(c, w_->att, d, w.out_w, w.out_b, w_->x, d, T, d, d, kGemmResidual, w_->x, d, "sf_att_out");
        // x += Conv(LN(x))
        sf_ln_linear(c, w_->x, w.ln_conv_w, w.ln_conv_b, w_->xn, w.pw1_w, w.pw1_b, w_->wide, 2 * d, T, 2 * d, d, 0, "sf_conv_pw1");
        launch_sf_glu_dwconv(c, w_->wide, w.dw_w, w.dw_b, w.bn_mean, w.bn_invstd, w.bn_w, w.bn_b, w_->att, rows, d, D.conv_kernel);
        sf_linear(c, w_->att, d, w.pw2_w, w.pw2_b, w_->x, d, T, d, d, kGemmResidual, w_->x, d, "sf_conv_pw2");
        // x += 0.5 * FF2(LN(x)); x = LN_out(x)
        sf_ln_linear(c, w_->x, w.ln_ff2_w, w.ln_ff2_b, w_->xn, w.ff2a_w, w.ff2a_b, w_->wide, ff, T, ff, d, kGemmSwish, "sf_ff_a");
        sf_linear(c, w_->wide, ff, w.ff2b_w, w.ff2b_b, w_->x, d, T, d, ff, kGemmResidual, w_->x, d, "sf_ff_b");
        launch_layernorm(c, w_->x, d, w.ln_out_w, w.ln_out_b, w_->x, d, T, d, "sf_ln");
    }
    const int dt = D.tf_d_model, dht = dt / D.tf_heads, inner = D.tf_inner;
    sf_linear(c, w_->x, d, m->P("proj.w"), m->P("proj.b"), w_->tx, dt, T, dt, d, 0, nullptr, 0, "sf_proj");
    const float qk_scale = 1.0f / sqrtf(sqrtf((float)dht));
    // Round 6, opt-in (WLK_SF_TF_LN_FUSE=1, gemm_kp_fuses_layernorm): a single session's step folds the two LayerNorms of a
    // (post-LN) Transformer block into the projection that reads them, whose column-tile-0 workgroups keep the normalised rows
    // for the block's residual (GemmArgs::ln_out).  Bit for bit the LayerNorm launch + projection, 35 launches less per chunk -
    // and measured 1 % slower, so the default stays LayerNorm launch + projection.
    auto tf_ln_linear = [&](const float* y, const float* lnw, const float* lnb, float* xn, const float* W, const float* b, float* C,
                            int N, int flags, const char* tag, float scale, int scale_cols) {
        if (gemm_kp_fuses_layernorm(T, N, dt)) {
            GemmArgs g;
            g.A = y; g.lda = dt; g.W = W; g.bias = b; g.C = C; g.ldc = N; g.M = T; g.N = N; g.K = dt; g.flags = flags;
            g.scale = scale; g.scale_cols = scale_cols;
            g.ln_gamma = lnw; g.ln_beta = lnb; g.ln_out = xn; g.ld_ln_out = dt;
            launch_gemm_kp(c, g, tag);
        } else {
            launch_layernorm(c, y, dt, lnw, lnb, xn, dt, T, dt, "sf_ln");
            sf_linear(c, xn, dt, W, b, C, N, T, N, dt, flags, nullptr, 0, tag, scale, scale_cols);
        }
    };
    for (int l = 0; l < D.tf_layers; ++l) {
        const SfTfLayer& w = m->tf[l];
        if (l == 0)
            sf_linear(c, w_->tx, dt, w.qkv_w, w.qkv_b, w_->tqkv, 3 * dt, T, 3 * dt, dt, kGemmScaleCols, nullptr, 0, "sf_tf_qkv",
                      qk_scale, 2 * dt);
        else   // tx = LN2 of the previous block (ty), inside this projection
            tf_ln_linear(w_->ty, m->tf[l - 1].ln2_w, m->tf[l - 1].ln2_b, w_->tx, w.qkv_w, w.qkv_b, w_->tqkv, 3 * dt, kGemmScaleCols,
                         "sf_tf_qkv", qk_scale, 2 * dt);
        SfAttnArgs a;
        a.q = w_->tqkv; a.k = w_->tqkv + dt; a.v = w_->tqkv + 2 * dt; a.ldq = a.ldk = a.ldv = 3 * dt;
        a.out = w_->tatt; a.ldo = dt; a.n_head = D.tf_heads; a.dh = dht; a.scale = 1.f;
        sf_set_segments(a, rows);
        launch_sf_attention(c, a);
        sf_linear(c, w_->tatt, dt, w.out_w, w.out_b, w_->ty, dt, T, dt, dt, kGemmResidual, w_->tx, dt, "sf_tf_out");
        tf_ln_linear(w_->ty, w.ln1_w, w.ln1_b, w_->tx, w.in_w, w.in_b, w_->th, inner, kGemmRelu, "sf_tf_in", 1.f, 0);
        sf_linear(c, w_->th, inner, w.outd_w, w.outd_b, w_->ty, dt, T, dt, inner, kGemmResidual, w_->tx, dt, "sf_tf_outd");
        if (l == D.tf_layers - 1) launch_layernorm(c, w_->ty, dt, w.ln2_w, w.ln2_b, w_->tx, dt, T, dt, "sf_ln");
    }
    launch_sf_head(c, w_->tx, m->head_w1t, m->P("head.h.b"), m->P("head.s.w"), m->P("head.s.b"), w_->preds, T, dt,
                   D.n_spk);
}

// One stacked step on lane `w_`: the requests in `batch` (slot indices, <= max_batch).  Everything is enqueued on the
// lane's stream - uploads from the requests' pinned blocks, the stem over all feature chunks, the network over all rows,
// the read-backs into the pinned blocks - and the stream is synchronised once.
static void sf_run_batch(wlk_sortformer* m, wlk_sortformer::Lane* w_, const std::vector<int>& batch) {
    const wlk_sf_dims& D = m->D;
    const int d = D.fc_d_model, C = D.sub_channels, nb = (int)batch.size();
    LaunchCtx c{w_->stream, nullptr};
    SfSegments rows, chunks;
    SfConvSegs s0;
    rows.n = chunks.n = nb;
    int r0 = 0, c0 = 0, f0 = 0, t0 = 0;
    for (int b = 0; b < nb; ++b) {
        SfRequest& q = m->slots[batch[b]];
        rows.start[b] = r0; rows.len[b] = q.T;
        chunks.start[b] = c0; chunks.len[b] = q.Tc;
        if (q.n_ctx > 0)
            WLK_HIP(hipMemcpyAsync(w_->ctxbuf + (size_t)r0 * d, q.ctx, (size_t)q.n_ctx * d * sizeof(float), hipMemcpyHostToDevice, w_->stream));
        if (q.n_feat > 0) {
            float* frows = w_->feats + (size_t)f0 * D.n_mels;
            if (q.n_pcm > 0) {
                // the front end inside the chain (round 6): previous rows up, the chunk's log-mel rows computed behind them on
                // this stream with the extractor's read-only tables, rows past the valid length zeroed (FilterbankFeatures'
                // seq_len rule, diarization.HipMelSpectrogram), the new rows back to the caller.  No separate extractor call:
                // under load that call's three stream operations queued behind whole launch chains of other streams in the
                // shared hardware queue (7 - 23 ms per chunk with GPU_MAX_HW_QUEUES = 2)
                const wlk_melspec* me = q.mel;
                const int n_new = q.n_feat - q.n_prev;
                float* au = w_->audio + (size_t)b * m->max_pcm;
                if (q.n_prev > 0)
                    WLK_HIP(hipMemcpyAsync(frows, q.feats, (size_t)q.n_prev * D.n_mels * sizeof(float), hipMemcpyHostToDevice, w_->stream));
                WLK_HIP(hipMemcpyAsync(au, q.pcm, (size_t)q.n_pcm * sizeof(float), hipMemcpyHostToDevice, w_->stream));
                MelSpecArgs a;
                a.audio = au; a.n_samples = q.n_pcm; a.window = me->window; a.twiddle = me->twiddle; a.filters = me->filters;
                a.filt_lo = me->lo; a.filt_hi = me->hi; a.out = frows + (size_t)q.n_prev * D.n_mels; a.n_fft = me->n_fft;
                a.win_length = me->win_length; a.hop = me->hop; a.n_mels = me->n_mels; a.preemph = me->preemph; a.log_guard = me->log_guard;
                launch_melspec(c, a, n_new);
                if (q.zero_from < n_new)
                    WLK_HIP(hipMemsetAsync(frows + (size_t)(q.n_prev + q.zero_from) * D.n_mels, 0,
                                           (size_t)(n_new - q.zero_from) * D.n_mels * sizeof(float), w_->stream));
                WLK_HIP(hipMemcpyAsync(q.feats + (size_t)q.n_prev * D.n_mels, frows + (size_t)q.n_prev * D.n_mels,
                                       (size_t)n_new * D.n_mels * sizeof(float), hipMemcpyDeviceToHost, w_->stream));
            } else {
                WLK_HIP(hipMemcpyAsync(frows, q.feats, (size_t)q.n_feat * D.n_mels * sizeof(float), hipMemcpyHostToDevice, w_->stream));
            }
            s0.in_start[s0.n] = f0; s0.in_len[s0.n] = q.n_feat; s0.out_start[s0.n] = t0;
            ++s0.n;
            f0 += q.n_feat;
            t0 += sf_sub_len(q.n_feat);
        }
        r0 += q.T;
        c0 += q.Tc;
    }
    s0.in_total = f0; s0.out_total = t0;
    if (s0.n > 0) {
        // ConvSubsampling.forward (dw_striding): conv0+ReLU, 2 x (depthwise s2, pointwise, ReLU), Linear - the feature
        // chunks of all sessions one after the other along the time axis (a session without features has no rows here)
        const int F1 = sf_sub_len(D.n_mels), F2 = sf_sub_len(F1), F3 = sf_sub_len(F2);
        const SfConvSegs s1 = sf_conv_segs_next(s0), s2 = sf_conv_segs_next(s1);
        launch_sf_conv0(c, w_->feats, m->P("pre.conv0.w"), m->P("pre.conv0.b"), w_->ca, s0, D.n_mels, C);
        launch_sf_dwconv2d(c, w_->ca, m->P("pre.dw1.w"), m->P("pre.dw1.b"), w_->cb, s1, F1, C);
        sf_linear(c, w_->cb, C, m->P("pre.pw1.w"), m->P("pre.pw1.b"), w_->ca, C, s1.out_total * F2, C, C, kGemmRelu, nullptr, 0, "sf_pre_pw");
        launch_sf_dwconv2d(c, w_->ca, m->P("pre.dw2.w"), m->P("pre.dw2.b"), w_->cb, s2, F2, C);
        sf_linear(c, w_->cb, C, m->P("pre.pw2.w"), m->P("pre.pw2.b"), w_->ca, C, s2.out_total * F3, C, C, kGemmRelu, nullptr, 0, "sf_pre_pw");
        if (s2.out_total != c0) throw std::logic_error("sortformer: chunk row count out of step with the stem");
        sf_linear(c, w_->ca, (long)F3 * C, m->P("pre.out.w"), m->P("pre.out.b"), w_->chunk_tmp, d, c0, d, F3 * C, 0, nullptr, 0, "sf_pre_out");
        for (int b = 0; b < nb; ++b) {
            SfRequest& q = m->slots[batch[b]];
            if (q.Tc > 0)
                WLK_HIP(hipMemcpyAsync(q.chunk, w_->chunk_tmp + (size_t)chunks.start[b] * d, (size_t)q.Tc * d * sizeof(float),
                                       hipMemcpyDeviceToHost, w_->stream));
        }
    }
    launch_sf_assemble(c, w_->ctxbuf, w_->chunk_tmp, w_->x, rows, chunks, d, D.xscale);
    sf_network(m, w_, c, rows);
    w_->last_T = rows.len[0];
    for (int b = 0; b < nb; ++b) {
        SfRequest& q = m->slots[batch[b]];
        WLK_HIP(hipMemcpyAsync(q.preds, w_->preds + (size_t)rows.start[b] * D.n_spk, (size_t)q.T * D.n_spk * sizeof(float),
                               hipMemcpyDeviceToHost, w_->stream));
    }
    WLK_HIP(hipStreamSynchronize(w_->stream));
}
}  // namespace wlk

extern "C" {

int wlk_sf_arena_floats(const wlk_sf_dims* dims, uint64_t* n_floats) {
    if (int rc = sf_check_dims(dims)) return rc;
    if (!n_floats) return sf_fail(WLK_ERR_ARG, "NULL argument");
    sf_layout(*dims, n_floats);
    return WLK_OK;
}

int wlk_sf_tensor_lookup(const wlk_sf_dims* dims, const char* packed_name, uint64_t* offset_floats, uint64_t* numel) {
    if (int rc = sf_check_dims(dims)) return rc;
    if (!packed_name) return sf_fail(WLK_ERR_ARG, "NULL argument");
    for (const auto& s : sf_layout(*dims, nullptr))
        if (s.name == packed_name) {
            if (offset_floats) *offset_floats = s.offset;
            if (numel) *numel = s.numel;
            return WLK_OK;
        }
    return sf_fail(WLK_ERR_ARG, std::string("unknown packed tensor ") + packed_name);
}

int wlk_sf_tensor_name(const wlk_sf_dims* dims, int index, const char** name) {
    if (int rc = sf_check_dims(dims)) return rc;
    if (!name) return sf_fail(WLK_ERR_ARG, "NULL argument");
    static thread_local std::string hold;
    const auto v = sf_layout(*dims, nullptr);
    if (index < 0 || index >= (int)v.size()) return sf_fail(WLK_ERR_ARG, "tensor index out of range");
    hold = v[index].name;
    *name = hold.c_str();
    return WLK_OK;
}

int wlk_sf_create(const wlk_sf_dims* dims, int device, wlk_sortformer** out) {
    if (int rc = sf_check_dims(dims)) return rc;
    if (!out) return sf_fail(WLK_ERR_ARG, "NULL argument");
    return sf_guarded([&]() {
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
            return sf_fail(WLK_ERR_HIP, "no HIP device: the Sortformer backend has no CPU fallback");
        WLK_HIP(hipSetDevice(device));
        auto m = std::make_unique<wlk_sortformer>();
        m->D = *dims;
        m->device = device;
        m->layout = sf_layout(*dims, &m->arena_floats);
        for (const auto& s : m->layout) m->index[s.name] = &s;
        wlk_sortformer* p = m.get();
        m->arena = sf_alloc(p, m->arena_floats);
        memset_sync(m->arena, 0, m->arena_floats * sizeof(float));
        const wlk_sf_dims& D = m->D;
        const size_t L = D.max_frames, d = D.fc_d_model, dt = D.tf_d_model, C = D.sub_channels;
        const size_t T1 = sf_sub_len(D.max_feat_frames), F1 = sf_sub_len(D.n_mels);
        m->pos_full = sf_alloc(p, (size_t)std::max(D.fc_layers, 1) * (2 * L - 1) * d);
        m->head_w1t = sf_alloc(p, dt * dt);
        // ONE lane by default (measured, scripts/diar_probe8.py, profiles/r06_diar_lanes.txt): with stacked steps a second lane
        // only splits the waiting sessions over two half-size chains that then share the GPU - 8 diarizer sessions alone
        // 642 audio-s/s on one lane against 516 on two, and beside 8 ASR streams the ASR side 250 against 216
        int n_ws = 1;
        if (const char* e = std::getenv("WLK_SF_WORKSPACES")) n_ws = std::max(1, std::min(8, std::atoi(e)));
        if (const char* e = std::getenv("WLK_SF_BATCH")) m->max_batch = std::max(1, std::min(kSfMaxSegments, std::atoi(e)));
        const size_t B = m->max_batch, R = B * L;          // sessions / rows of a stacked step
        // a diarizer step is a chain of small kernels that shares the GPU with ASR sessions' chip-filling encoder kernels
        // (config 4): WLK_SF_PRIORITY=hi puts its streams ahead of them in the dispatcher.  Measured a loss (ASR -20 %,
        // diarizer p95 worse: profiles/r05l_ab_sortformer_stream_priority.txt); default: normal priority
        int prio_lo = 0, prio_hi = 0;
        WLK_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        const char* pe = std::getenv("WLK_SF_PRIORITY");
        m->ws.resize(n_ws);
        for (auto& w : m->ws) {
            if (pe && pe[0] == 'h') WLK_HIP(hipStreamCreateWithPriority(&w.stream, hipStreamNonBlocking, prio_hi));
            else if (pe && pe[0] == 'l') WLK_HIP(hipStreamCreateWithPriority(&w.stream, hipStreamNonBlocking, prio_lo));
            else WLK_HIP(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
            w.audio = sf_alloc(p, B * (size_t)m->max_pcm);
            w.feats = sf_alloc(p, B * (size_t)D.max_feat_frames * D.n_mels);
            w.ca = sf_alloc(p, B * T1 * F1 * C);
            w.cb = sf_alloc(p, B * T1 * F1 * C);
            w.ctxbuf = sf_alloc(p, R * d);
            w.chunk_tmp = sf_alloc(p, B * (T1 / 4 + 2) * d);
            w.x = sf_alloc(p, R * d);
            w.xn = sf_alloc(p, R * d);
            w.wide = sf_alloc(p, R * std::max<size_t>(D.fc_ff, 2 * d));
            w.qkv = sf_alloc(p, R * 3 * d);
            w.att = sf_alloc(p, R * d);
            w.tx = sf_alloc(p, R * dt);
            w.ty = sf_alloc(p, R * dt);
            w.tqkv = sf_alloc(p, R * 3 * dt);
            w.tatt = sf_alloc(p, R * dt);
            w.th = sf_alloc(p, R * D.tf_inner);
            w.preds = sf_alloc(p, R * D.n_spk);
        }
        // request slots: pinned blocks the callers fill and drain themselves (enough for every lane's batch plus the next one's)
        auto pin = [&](size_t n) {
            float* h = nullptr;
            WLK_HIP(hipHostMalloc(reinterpret_cast<void**>(&h), n * sizeof(float), hipHostMallocDefault));
            p->pinned.push_back(h);
            return h;
        };
        m->slots.resize((size_t)(n_ws + 1) * B);
        for (auto& q : m->slots) {
            q.feats = pin((size_t)D.max_feat_frames * D.n_mels);
            q.pcm = pin((size_t)m->max_pcm);
            q.ctx = pin(L * d);
            q.chunk = pin((T1 / 4 + 2) * d);
            q.preds = pin(L * D.n_spk);
        }
        *out = m.release();
        return WLK_OK;
    });
}

int wlk_sf_upload(wlk_sortformer* m, const char* packed_name, const float* host, uint64_t numel) {
    if (!m || !packed_name || !host) return sf_fail(WLK_ERR_ARG, "NULL argument");
    return sf_guarded([&]() {
        auto it = m->index.find(packed_name);
        if (it == m->index.end()) return sf_fail(WLK_ERR_ARG, std::string("unknown packed tensor ") + packed_name);
        if (it->second->numel != numel)
            return sf_fail(WLK_ERR_ARG, std::string("size mismatch for ") + packed_name + ": expected " +
                                            std::to_string(it->second->numel) + ", got " + std::to_string(numel));
        WLK_HIP(hipSetDevice(m->device));
        copy_sync(m->arena + it->second->offset, host, numel * sizeof(float), hipMemcpyHostToDevice);
        m->finalized = false;
        return WLK_OK;
    });
}

int wlk_sf_finalize(wlk_sortformer* m) {
    if (!m) return sf_fail(WLK_ERR_ARG, "model is NULL");
    return sf_guarded([&]() {
        std::lock_guard<std::mutex> lock(m->mu);
        WLK_HIP(hipSetDevice(m->device));
        const wlk_sf_dims& D = m->D;
        m->fc.resize(D.fc_layers);
        for (int i = 0; i < D.fc_layers; ++i) {
            const std::string p = "fc." + std::to_string(i) + ".";
            SfFcLayer& w = m->fc[i];
            w.ln_ff1_w = m->P(p + "ln_ff1.w"); w.ln_ff1_b = m->P(p + "ln_ff1.b");
            w.ff1a_w = m->P(p + "ff1a.w"); w.ff1a_b = m->P(p + "ff1a.b");
            w.ff1b_w = m->P(p + "ff1b.w"); w.ff1b_b = m->P(p + "ff1b.b");
            w.ln_att_w = m->P(p + "ln_att.w"); w.ln_att_b = m->P(p + "ln_att.b");
            w.qkv_w = m->P(p + "qkv.w"); w.qkv_b = m->P(p + "qkv.b");
            w.pos_w = m->P(p + "pos.w"); w.bias_u = m->P(p + "bias_u"); w.bias_v = m->P(p + "bias_v");
            w.out_w = m->P(p + "out.w"); w.out_b = m->P(p + "out.b");
            w.ln_conv_w = m->P(p + "ln_conv.w"); w.ln_conv_b = m->P(p + "ln_conv.b");
            w.pw1_w = m->P(p + "pw1.w"); w.pw1_b = m->P(p + "pw1.b");
            w.dw_w = m->P(p + "dw.w"); w.dw_b = m->P(p + "dw.b");
            w.bn_mean = m->P(p + "bn.mean"); w.bn_invstd = m->P(p + "bn.invstd");
            w.bn_w = m->P(p + "bn.w"); w.bn_b = m->P(p + "bn.b");
            w.pw2_w = m->P(p + "pw2.w"); w.pw2_b = m->P(p + "pw2.b");
            w.ln_ff2_w = m->P(p + "ln_ff2.w"); w.ln_ff2_b = m->P(p + "ln_ff2.b");
            w.ff2a_w = m->P(p + "ff2a.w"); w.ff2a_b = m->P(p + "ff2a.b");
            w.ff2b_w = m->P(p + "ff2b.w"); w.ff2b_b = m->P(p + "ff2b.b");
            w.ln_out_w = m->P(p + "ln_out.w"); w.ln_out_b = m->P(p + "ln_out.b");
        }
        m->tf.resize(D.tf_layers);
        for (int i = 0; i < D.tf_layers; ++i) {
            const std::string p = "tf." + std::to_string(i) + ".";
            SfTfLayer& w = m->tf[i];
            w.qkv_w = m->P(p + "qkv.w"); w.qkv_b = m->P(p + "qkv.b");
            w.out_w = m->P(p + "out.w"); w.out_b = m->P(p + "out.b");
            w.ln1_w = m->P(p + "ln1.w"); w.ln1_b = m->P(p + "ln1.b");
            w.in_w = m->P(p + "in.w"); w.in_b = m->P(p + "in.b");
            w.outd_w = m->P(p + "outd.w"); w.outd_b = m->P(p + "outd.b");
            w.ln2_w = m->P(p + "ln2.w"); w.ln2_b = m->P(p + "ln2.b");
        }
        // linear_pos(pos_emb) of every block for the longest sequence: a shorter sequence of T frames uses rows
        // [max_frames - T, max_frames + T - 1) of it (RelPositionalEncoding centres its table the same way)
        LaunchCtx c{m->ws[0].stream, nullptr};
        const int d = D.fc_d_model, rows = 2 * D.max_frames - 1;
        for (int l = 0; l < D.fc_layers; ++l) {
            GemmArgs g;
            g.A = m->P("pos.table"); g.lda = d; g.W = m->fc[l].pos_w; g.C = m->pos_full + (size_t)l * rows * d; g.ldc = d;
            g.M = rows; g.N = d; g.K = d;
            launch_linear(c, g, "sf_pos");
        }
        launch_sf_transpose(c, m->P("head.h.w"), m->head_w1t, D.tf_d_model, D.tf_d_model);
        WLK_HIP(hipStreamSynchronize(m->ws[0].stream));
        m->finalized = true;
        return WLK_OK;
    });
}

namespace wlk {
struct SfPcmIn {                 // wlk_sf_step_pcm's front-end arguments (mel == nullptr: the caller brought the features)
    const wlk_melspec* mel = nullptr;
    const float* pcm = nullptr;
    int n_pcm = 0, n_prev = 0, zero_from = 0;
    float* feats_out = nullptr;
};
}  // namespace wlk
static int sf_step_impl(wlk_sortformer* m, const float* feats_host, int n_feat, const wlk::SfPcmIn& pin, const float* ctx_embs_host,
                        int n_ctx, float* chunk_embs_host, int chunk_capacity_rows, int* n_chunk, float* preds_host,
                        int preds_capacity_rows) {
    if (!m || !preds_host) return sf_fail(WLK_ERR_ARG, "NULL argument");
    if (!m->finalized) return sf_fail(WLK_ERR_STATE, "wlk_sf_finalize has not been called");
    if (n_feat < 0 || n_ctx < 0 || (n_feat > 0 && (!feats_host || !chunk_embs_host)) || (n_ctx > 0 && !ctx_embs_host))
        return sf_fail(WLK_ERR_ARG, "bad step arguments");
    if (n_feat > m->D.max_feat_frames) return sf_fail(WLK_ERR_CAPACITY, "feature chunk longer than max_feat_frames");
    return sf_guarded([&]() {
        const wlk_sf_dims& D = m->D;
        const int d = D.fc_d_model;
        int Tc = 0;
        if (n_feat > 0) Tc = sf_sub_len(sf_sub_len(sf_sub_len(n_feat)));
        const int T = n_ctx + Tc;
        if (n_chunk) *n_chunk = Tc;
        if (T < 1) return sf_fail(WLK_ERR_ARG, "empty step");
        if (T > D.max_frames) return sf_fail(WLK_ERR_CAPACITY, "sequence longer than max_frames");
        if (Tc > chunk_capacity_rows || T > preds_capacity_rows) return sf_fail(WLK_ERR_CAPACITY, "output buffer too small");
        // 1. a request slot; the inputs go into its pinned blocks (this thread's own memcpy, outside the lock)
        int me = -1;
        {
            std::unique_lock<std::mutex> lock(m->mu);
            m->cv.wait(lock, [&] { for (auto& q : m->slots) if (!q.in_use) return true; return false; });
            for (size_t i = 0; i < m->slots.size(); ++i)
                if (!m->slots[i].in_use) { me = (int)i; break; }
            m->slots[me].in_use = true;
        }
        SfRequest& mine = m->slots[me];
        mine.n_feat = n_feat; mine.n_ctx = n_ctx; mine.Tc = Tc; mine.T = T;
        mine.mel = pin.mel; mine.n_pcm = pin.mel ? pin.n_pcm : 0; mine.n_prev = pin.n_prev; mine.zero_from = pin.zero_from;
        mine.done = false; mine.rc = WLK_OK; mine.err.clear();
        if (pin.mel) {
            if (pin.n_prev > 0) memcpy(mine.feats, feats_host, (size_t)pin.n_prev * D.n_mels * sizeof(float));
            memcpy(mine.pcm, pin.pcm, (size_t)pin.n_pcm * sizeof(float));
        } else if (n_feat > 0) {
            memcpy(mine.feats, feats_host, (size_t)n_feat * D.n_mels * sizeof(float));
        }
        if (n_ctx > 0) memcpy(mine.ctx, ctx_embs_host, (size_t)n_ctx * d * sizeof(float));
        // 2. queue; whoever finds a free lane while requests wait runs the oldest ones as ONE stacked step (its own may or
        //    may not be among them) - no dispatcher thread, no gather window
        std::unique_lock<std::mutex> lock(m->mu);
        mine.ticket = m->next_ticket++;
        mine.queued = true;
        m->queue.push_back(me);
        while (!mine.done) {
            wlk_sortformer::Lane* lane = nullptr;
            if (!m->queue.empty())
                for (auto& w : m->ws)
                    if (!w.busy) { lane = &w; break; }
            if (!lane) {
                m->cv.wait(lock);
                continue;
            }
            std::vector<int> batch;
            while (!m->queue.empty() && (int)batch.size() < m->max_batch) {
                batch.push_back(m->queue.front());
                m->queue.pop_front();
            }
            for (int i : batch) m->slots[i].queued = false;
            lane->busy = true;
            lock.unlock();
            int rc = WLK_OK;
            std::string err;
            try {
                WLK_HIP(hipSetDevice(m->device));
                sf_run_batch(m, lane, batch);
            } catch (const HipError& e) {
                rc = WLK_ERR_HIP; err = e.what();
                (void)hipStreamSynchronize(lane->stream);        // nothing of the failed chain may still write a slot
            } catch (const std::invalid_argument& e) {
                rc = WLK_ERR_ARG; err = e.what();
                (void)hipStreamSynchronize(lane->stream);
            } catch (const std::exception& e) {
                rc = WLK_ERR_STATE; err = e.what();
                (void)hipStreamSynchronize(lane->stream);
            }
            lock.lock();
            lane->busy = false;
            m->last_ws = (int)(lane - m->ws.data());
            m->n_steps += 1;
            m->n_sessions += batch.size();
            for (int i : batch) {
                m->n_rows += m->slots[i].T;
                m->slots[i].rc = rc;
                m->slots[i].err = err;
                m->slots[i].done = true;
            }
            m->cv.notify_all();
        }
        lock.unlock();
        // 3. results out of the pinned blocks, slot back
        const int rc = mine.rc;
        const std::string err = mine.err;
        if (rc == WLK_OK) {
            if (Tc > 0) memcpy(chunk_embs_host, mine.chunk, (size_t)Tc * d * sizeof(float));
            memcpy(preds_host, mine.preds, (size_t)T * D.n_spk * sizeof(float));
            if (pin.mel) memcpy(pin.feats_out, mine.feats + (size_t)pin.n_prev * D.n_mels, (size_t)(n_feat - pin.n_prev) * D.n_mels * sizeof(float));
        }
        {
            std::lock_guard<std::mutex> g(m->mu);
            mine.in_use = false;
        }
        m->cv.notify_all();
        return rc == WLK_OK ? WLK_OK : sf_fail(rc, err);
    });
}

int wlk_sf_step(wlk_sortformer* m, const float* feats_host, int n_feat, const float* ctx_embs_host, int n_ctx,
                float* chunk_embs_host, int chunk_capacity_rows, int* n_chunk, float* preds_host,
                int preds_capacity_rows) {
    return sf_step_impl(m, feats_host, n_feat, wlk::SfPcmIn{}, ctx_embs_host, n_ctx, chunk_embs_host, chunk_capacity_rows, n_chunk,
                        preds_host, preds_capacity_rows);
}

int wlk_sf_step_pcm(wlk_sortformer* m, wlk_melspec* mel, const float* pcm_host, int n_pcm, int valid_frames,
                    const float* prev_feats_host, int n_prev, float* feats_out_host, int feats_capacity_rows, int* n_feats_out,
                    const float* ctx_embs_host, int n_ctx, float* chunk_embs_host, int chunk_capacity_rows, int* n_chunk,
                    float* preds_host, int preds_capacity_rows) {
    if (!m || !mel || !pcm_host || !feats_out_host || !n_feats_out) return sf_fail(WLK_ERR_ARG, "NULL argument");
    if (n_pcm < 1 || n_pcm > m->max_pcm || n_pcm > mel->cap) return sf_fail(WLK_ERR_CAPACITY, "audio chunk does not fit the step's buffer");
    if (mel->n_mels != m->D.n_mels || mel->device != m->device) return sf_fail(WLK_ERR_ARG, "the extractor does not match the model (mel bins / device)");
    if (n_prev < 0 || (n_prev > 0 && !prev_feats_host)) return sf_fail(WLK_ERR_ARG, "bad previous-feature rows");
    const int n_new = n_pcm / mel->hop + 1;                 // the centred STFT's frame count (wlk_melspec_run)
    *n_feats_out = n_new;
    if (n_new > feats_capacity_rows) return sf_fail(WLK_ERR_CAPACITY, "feature output buffer too small");
    wlk::SfPcmIn pin;
    pin.mel = mel; pin.pcm = pcm_host; pin.n_pcm = n_pcm; pin.n_prev = n_prev; pin.feats_out = feats_out_host;
    pin.zero_from = valid_frames < 0 ? n_new : std::min(valid_frames, n_new);
    // feats_host of the shared path = the previous rows (the new ones are produced by the step itself)
    return sf_step_impl(m, n_prev > 0 ? prev_feats_host : pcm_host, n_prev + n_new, pin, ctx_embs_host, n_ctx, chunk_embs_host,
                        chunk_capacity_rows, n_chunk, preds_host, preds_capacity_rows);
}

int wlk_sf_stats(wlk_sortformer* m, uint64_t* stacked_steps, uint64_t* session_steps, uint64_t* rows) {
    if (!m) return sf_fail(WLK_ERR_ARG, "model is NULL");
    std::lock_guard<std::mutex> lock(m->mu);
    if (stacked_steps) *stacked_steps = m->n_steps;
    if (session_steps) *session_steps = m->n_sessions;
    if (rows) *rows = m->n_rows;
    return WLK_OK;
}

int wlk_sf_export(wlk_sortformer* m, const char* what, float* host, uint64_t capacity, uint64_t* n_written) {
    if (!m || !what || !host) return sf_fail(WLK_ERR_ARG, "NULL argument");
    return sf_guarded([&]() {
        // the FIRST session's rows of the most recent step, read while no step runs on that lane (the lane is held for
        // the duration of the copy: a concurrent caller's step waits or takes another lane)
        std::unique_lock<std::mutex> lock(m->mu);
        wlk_sortformer::Lane& w = m->ws[m->last_ws];
        m->cv.wait(lock, [&] { return !w.busy; });
        w.busy = true;
        lock.unlock();
        struct Release {
            wlk_sortformer* m; wlk_sortformer::Lane* w;
            ~Release() { { std::lock_guard<std::mutex> g(m->mu); w->busy = false; } m->cv.notify_all(); }
        } release{m, &w};
        WLK_HIP(hipSetDevice(m->device));
        const float* src = nullptr;
        uint64_t n = 0;
        if (!strcmp(what, "fc_out")) { src = w.x; n = (uint64_t)w.last_T * m->D.fc_d_model; }
        else if (!strcmp(what, "tf_out")) { src = w.tx; n = (uint64_t)w.last_T * m->D.tf_d_model; }
        else return sf_fail(WLK_ERR_ARG, std::string("unknown export ") + what);
        if (n > capacity) return sf_fail(WLK_ERR_CAPACITY, "export buffer too small");
        copy_sync(host, src, n * sizeof(float), hipMemcpyDeviceToHost);
        if (n_written) *n_written = n;
        return WLK_OK;
    });
}

int wlk_sf_destroy(wlk_sortformer* m) {
    delete m;          // ~wlk_sortformer frees the arena, the lanes' buffers and streams, the pinned request blocks
    return WLK_OK;
}

}  // extern "C"

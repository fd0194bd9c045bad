// NLLB-200 (M2M-100 architecture) translation model of config 5: the network the reference loads
// through the third-party `nllw` package (whisperlivekit/core.py:320-329) - not in the reference tree.  The arithmetic
// restated here is the published M2M-100 of `transformers` (models/m2m_100/modeling_m2m_100.py, 5.15.0 in this image),
// which is what pins it (tests/golden/nllb_kat.npz):
//
//   embed(ids)   = shared[ids] * sqrt(d_model) + sinusoid[position], position = index + padding_idx + 1 (no padding
//                  inside a sequence: one sentence per call)
//   encoder      = N x { x += Wo . MHA(LN(x)) ; x += W2 . relu(W1 . LN(x)) } ; LN          (pre-LN, biases everywhere)
//   decoder      = N x { x += self-MHA(LN(x)) with KV cache ; x += cross-MHA(LN(x), encoder) ; x += FFN(LN(x)) } ; LN
//   logits       = LN(x) . shared^T                                                         (tied, no bias)
//   attention    = softmax((q * dh^-0.5) . k^T) . v, heads of 64
//
// Everything is a composition of the launchers the Whisper / Sortformer paths already use (MFMA GEMM with bias / ReLU /
// residual / column-scale epilogues, weight-streaming GEMV for the single-token steps, LayerNorm, the LDS-score
// attention for the short non-causal encoder, the KV-cache self-attention and the key-parallel cross-attention of the
// Whisper decoder, log-softmax + top-k): one small kernel is new (token + position embedding).  fp32 throughout.
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/wlk_hip.h"
#include "common.h"

namespace wlk {

static int nl_fail(int code, const std::string& msg) {
    set_last_error(msg);
    return code;
}
template <typename F>
static int nl_guarded(F&& f) {
    try {
        return f();
    } catch (const HipError& e) {
        return nl_fail(WLK_ERR_HIP, e.what());
    } catch (const std::invalid_argument& e) {
        return nl_fail(WLK_ERR_ARG, e.what());
    } catch (const std::exception& e) {
        return nl_fail(WLK_ERR_STATE, e.what());
    }
}

struct NlSlot {
    std::string name;
    uint64_t offset, numel;
};

static std::vector<NlSlot> nl_layout(const wlk_nllb_dims& D, uint64_t* total) {
    std::vector<NlSlot> v;
    uint64_t off = 0;
    auto add = [&](const std::string& n, uint64_t numel) {
        v.push_back({n, off, numel});
        off += (numel + 63) / 64 * 64;
    };
    const uint64_t d = D.d_model, f = D.ffn;
    add("shared.emb", (uint64_t)D.vocab * d);
    add("pos.table", (uint64_t)D.n_positions * d);
    auto block = [&](const std::string& p, bool cross) {
        add(p + "ln1.w", d); add(p + "ln1.b", d);
        add(p + "qkv.w", 3 * d * d); add(p + "qkv.b", 3 * d);           // rows: q | k | v
        add(p + "out.w", d * d); add(p + "out.b", d);
        if (cross) {
            add(p + "lnx.w", d); add(p + "lnx.b", d);
            add(p + "xq.w", d * d); add(p + "xq.b", d);
            add(p + "xkv.w", 2 * d * d); add(p + "xkv.b", 2 * d);       // rows: k | v
            add(p + "xout.w", d * d); add(p + "xout.b", d);
        }
        add(p + "ln2.w", d); add(p + "ln2.b", d);
        add(p + "fc1.w", f * d); add(p + "fc1.b", f);
        add(p + "fc2.w", d * f); add(p + "fc2.b", d);
    };
    for (int i = 0; i < D.enc_layers; ++i) block("enc." + std::to_string(i) + ".", false);
    add("enc.ln.w", d); add("enc.ln.b", d);
    for (int i = 0; i < D.dec_layers; ++i) block("dec." + std::to_string(i) + ".", true);
    add("dec.ln.w", d); add("dec.ln.b", d);
    if (total) *total = off;
    return v;
}

static int nl_check_dims(const wlk_nllb_dims* d) {
    if (!d) return nl_fail(WLK_ERR_ARG, "dims is NULL");
    if (d->vocab < 8 || d->d_model < 64 || d->heads < 1 || d->d_model != 64 * d->heads)
        return nl_fail(WLK_ERR_ARG, "NLLB: d_model must be 64 x heads (heads of 64, as in every released NLLB / M2M-100)");
    if (d->d_model % 4 || d->ffn % 4 || d->ffn < 4 || d->enc_layers < 1 || d->dec_layers < 1)
        return nl_fail(WLK_ERR_ARG, "NLLB: bad layer sizes");
    if (d->max_src < 1 || d->max_src > kSfMaxFrames) return nl_fail(WLK_ERR_ARG, "NLLB: max_src must be 1..512 tokens");
    if (d->max_tgt < 2 || d->max_tgt > 512) return nl_fail(WLK_ERR_ARG, "NLLB: max_tgt must be 2..512 tokens");
    if (d->pad_id < 0 || d->pad_id >= d->vocab) return nl_fail(WLK_ERR_ARG, "NLLB: pad_id out of range");
    if (d->n_positions < std::max(d->max_src, d->max_tgt) + d->pad_id + 1)
        return nl_fail(WLK_ERR_ARG, "NLLB: the position table must cover max(max_src, max_tgt) + pad_id + 1 rows");
    return WLK_OK;
}

struct NlLayer {
    const float *ln1w, *ln1b, *qkvw, *qkvb, *outw, *outb, *lnxw, *lnxb, *xqw, *xqb, *xkvw, *xkvb, *xoutw, *xoutb, *ln2w, *ln2b,
        *fc1w, *fc1b, *fc2w, *fc2b;
};

// x[r][:] = emb[tokens[r]][:] * scale + pos[pos0 + *offset + (r % n_tok)][:]
__global__ __launch_bounds__(256) void nllb_embed_kernel(const int* __restrict__ tokens, const float* __restrict__ emb,
                                                         const float* __restrict__ pos, float scale, int pos0,
                                                         const int* __restrict__ offset, int n_tok, int d,
                                                         float* __restrict__ x) {
    const int r = blockIdx.x;
    const float* e = emb + (long)tokens[r] * d;
    const float* p = pos + (long)(pos0 + (offset ? *offset : 0) + r % n_tok) * d;
    for (int c = threadIdx.x; c < d; c += 256) x[(long)r * d + c] = e[c] * scale + p[c];
}

// single-token steps replayed from a graph: tokens [rows] and the cache offset come from a host-coherent block (no copy
// node); workgroup 0 leaves the offset in device memory for the kernels behind it
__global__ __launch_bounds__(256) void nllb_embed_step_kernel(const int* __restrict__ host_block, int rows,
                                                              const float* __restrict__ emb, const float* __restrict__ pos,
                                                              float scale, int pos0, int d, int* __restrict__ offset_dev,
                                                              float* __restrict__ x) {
    const int r = blockIdx.x;
    const int tok = host_block[r], off = host_block[rows];
    if (r == 0 && threadIdx.x == 0) *offset_dev = off;
    const float* e = emb + (long)tok * d;
    const float* p = pos + (long)(pos0 + off) * d;
    for (int c = threadIdx.x; c < d; c += 256) x[(long)r * d + c] = e[c] * scale + p[c];
}

}  // namespace wlk

using namespace wlk;

struct wlk_nllb {
    wlk_nllb_dims D{};
    int device = 0;
    float* arena = nullptr;
    uint64_t arena_floats = 0;
    std::vector<NlSlot> layout;
    std::map<std::string, const NlSlot*> index;
    bool finalized = false;
    std::vector<NlLayer> enc, dec;
    const float *emb = nullptr, *pos = nullptr, *enc_lnw = nullptr, *enc_lnb = nullptr, *dec_lnw = nullptr, *dec_lnb = nullptr;
    const float* P(const std::string& n) const {
        auto it = index.find(n);
        if (it == index.end()) throw std::invalid_argument("unknown packed tensor " + n);
        return arena + it->second->offset;
    }
    ~wlk_nllb() {
        (void)hipSetDevice(device);
        if (arena) (void)hipFree(arena);
    }
};

struct wlk_nllb_session {
    wlk_nllb* m = nullptr;
    int rows = 1;
    hipStream_t stream = nullptr;
    std::vector<void*> owned;
    // encoder
    float *ex = nullptr, *eh = nullptr, *eqkv = nullptr, *eatt = nullptr, *ewide = nullptr, *enc_out = nullptr, *cross_kv = nullptr;
    int src_len = 0;
    bool encoded = false;
    // decoder
    float *kcache[2] = {nullptr, nullptr}, *vcache[2] = {nullptr, nullptr};
    int kv_cur = 0;
    float *dx = nullptr, *dh = nullptr, *dqkv = nullptr, *datt = nullptr, *dq = nullptr, *dwide = nullptr, *hsel = nullptr, *logits = nullptr;
    int *tokens_dev = nullptr, *offset_dev = nullptr, *src_rows_dev = nullptr;
    float* top_vals = nullptr;
    int* top_ids = nullptr;
    void* topk_scratch = nullptr;
    int self_len = 0;
    bool have_logits = false;
    // graph-replayed single-token steps (wlk_nllb_step)
    int* step_host = nullptr;          // pinned + mapped: [rows] tokens | offset
    int* step_host_dev = nullptr;      // the same block as the device sees it
    float* step_vals_host = nullptr;   // pinned + mapped: top-k results written by the top-k kernel itself
    int* step_ids_host = nullptr;
    float* step_vals_dev = nullptr;
    int* step_ids_dev = nullptr;
    int step_k = 1;
    hipGraphExec_t step_exec[2] = {nullptr, nullptr};
    int step_exec_k[2] = {0, 0};
    int step_exec_src[2] = {0, 0};     // the source length a captured step was recorded with: the cross-attention launch carries it
                                       // by value, so a sentence of another length needs a new recording (round 5: a session
                                       // that translated a 5-token and then a 7-token source replayed the 5-token graph)
    template <typename T>
    T* alloc(size_t n) {
        void* p = nullptr;
        WLK_HIP(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
        owned.push_back(p);
        return static_cast<T*>(p);
    }
    LaunchCtx ctx() const { return LaunchCtx{stream, nullptr}; }
    ~wlk_nllb_session() {        // also the clean-up of a half-built session (an allocation that threw in session_create)
        if (m) (void)hipSetDevice(m->device);
        if (stream) (void)hipStreamSynchronize(stream);
        for (void* p : owned) (void)hipFree(p);
        for (auto& e : step_exec) if (e) (void)hipGraphExecDestroy(e);
        if (step_host) (void)hipHostFree(step_host);
        if (step_vals_host) (void)hipHostFree(step_vals_host);
        if (step_ids_host) (void)hipHostFree(step_ids_host);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace wlk {

static void nl_linear(const LaunchCtx& c, const float* A, long lda, const float* W, const float* b, float* C, long ldc, int M,
                      int N, int K, int flags, const float* R, long ldr, const char* tag, float scale = 1.f, int scale_cols = 0) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.bias = b; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.flags = flags; g.R = R; g.ldr = ldr; g.scale = scale; g.scale_cols = scale_cols;
    launch_linear(c, g, tag);
}

static void nl_ffn(const LaunchCtx& c, const NlLayer& L, float* x, float* h, float* wide, int R, int d, int f) {
    launch_layernorm(c, x, d, L.ln2w, L.ln2b, h, d, R, d, "nllb_ln2");
    nl_linear(c, h, d, L.fc1w, L.fc1b, wide, f, R, f, d, kGemmRelu, nullptr, 0, "nllb_fc1");
    nl_linear(c, wide, f, L.fc2w, L.fc2b, x, d, R, d, f, kGemmResidual, x, d, "nllb_fc2");
}

static void nl_encode(wlk_nllb_session* s, int S) {
    wlk_nllb* m = s->m;
    const wlk_nllb_dims& D = m->D;
    const LaunchCtx c = s->ctx();
    const int d = D.d_model, H = D.heads, f = D.ffn;
    const float q_scale = 0.125f;                      // 64^-0.5, exact
    hipLaunchKernelGGL(nllb_embed_kernel, dim3(S), dim3(256), 0, s->stream, s->tokens_dev, m->emb, m->pos, D.embed_scale,
                       D.pad_id + 1, (const int*)nullptr, S, d, s->ex);
    WLK_HIP(hipGetLastError());
    for (int l = 0; l < D.enc_layers; ++l) {
        const NlLayer& L = m->enc[l];
        launch_layernorm(c, s->ex, d, L.ln1w, L.ln1b, s->eh, d, S, d, "nllb_ln1");
        nl_linear(c, s->eh, d, L.qkvw, L.qkvb, s->eqkv, 3 * d, S, 3 * d, d, kGemmScaleCols, nullptr, 0, "nllb_enc_qkv", q_scale, d);
        SfAttnArgs a;
        a.q = s->eqkv; a.k = s->eqkv + d; a.v = s->eqkv + 2 * d; a.ldq = a.ldk = a.ldv = 3 * d;
        a.out = s->eatt; a.ldo = d; a.T = S; a.n_head = H; a.dh = 64; a.scale = 1.f;
        launch_sf_attention(c, a);
        nl_linear(c, s->eatt, d, L.outw, L.outb, s->ex, d, S, d, d, kGemmResidual, s->ex, d, "nllb_enc_out");
        nl_ffn(c, L, s->ex, s->eh, s->ewide, S, d, f);
    }
    launch_layernorm(c, s->ex, d, m->enc_lnw, m->enc_lnb, s->enc_out, d, S, d, "nllb_enc_ln");
    // cross-attention keys / values of every decoder layer: [S][dec_layers][k | v]
    const long ld = (long)D.dec_layers * 2 * d;
    for (int l = 0; l < D.dec_layers; ++l) {
        const NlLayer& L = m->dec[l];
        nl_linear(c, s->enc_out, d, L.xkvw, L.xkvb, s->cross_kv + (size_t)l * 2 * d, ld, S, 2 * d, d, 0, nullptr, 0, "nllb_xkv");
    }
}

static void nl_decode(wlk_nllb_session* s, int n_tok, bool graph_step = false) {
    wlk_nllb* m = s->m;
    const wlk_nllb_dims& D = m->D;
    const LaunchCtx c = s->ctx();
    const int d = D.d_model, H = D.heads, f = D.ffn, rows = s->rows, R = rows * n_tok, ctx_len = D.max_tgt, S = s->src_len;
    const float q_scale = 0.125f;
    const bool fused = n_tok == 1 && gemv_applicable(R, d);
    if (graph_step)
        hipLaunchKernelGGL(nllb_embed_step_kernel, dim3(R), dim3(256), 0, s->stream, s->step_host_dev, rows, m->emb, m->pos,
                           D.embed_scale, D.pad_id + 1, d, s->offset_dev, s->dx);
    else
        hipLaunchKernelGGL(nllb_embed_kernel, dim3(R), dim3(256), 0, s->stream, s->tokens_dev, m->emb, m->pos, D.embed_scale,
                           D.pad_id + 1, s->offset_dev, n_tok, d, s->dx);
    WLK_HIP(hipGetLastError());
    const size_t cache_layer = (size_t)rows * ctx_len * d;
    const long ldkv = (long)D.dec_layers * 2 * d;
    for (int l = 0; l < D.dec_layers; ++l) {
        const NlLayer& L = m->dec[l];
        float* kc = s->kcache[s->kv_cur] + l * cache_layer;
        float* vc = s->vcache[s->kv_cur] + l * cache_layer;
        if (fused) {      // single-token steps: LayerNorm and the cache append ride in the weight-streaming GEMV
            GemmArgs g;
            g.A = s->dx; g.lda = d; g.W = L.qkvw; g.bias = L.qkvb; g.C = s->dqkv; g.ldc = 3 * d; g.M = R; g.N = 3 * d; g.K = d;
            g.flags = kGemmScaleCols; g.scale = q_scale; g.scale_cols = d;
            g.ln_gamma = L.ln1w; g.ln_beta = L.ln1b;
            g.kcache = kc; g.vcache = vc; g.kv_pos = s->offset_dev; g.kv_d = d; g.kv_ctx = ctx_len;
            launch_gemv(c, g, "nllb_ln1_qkv_kv");
        } else {
            launch_layernorm(c, s->dx, d, L.ln1w, L.ln1b, s->dh, d, R, d, "nllb_ln1");
            nl_linear(c, s->dh, d, L.qkvw, L.qkvb, s->dqkv, 3 * d, R, 3 * d, d, kGemmScaleCols, nullptr, 0, "nllb_dec_qkv", q_scale, d);
            launch_kv_append(c, s->dqkv, kc, vc, rows, n_tok, s->offset_dev, d, ctx_len);
        }
        launch_decoder_self_attention(c, s->dqkv, kc, vc, s->datt, rows, n_tok, s->offset_dev, d, H, ctx_len);
        nl_linear(c, s->datt, d, L.outw, L.outb, s->dx, d, R, d, d, kGemmResidual, s->dx, d, "nllb_dec_out");

        if (fused) {
            GemmArgs q;
            q.A = s->dx; q.lda = d; q.W = L.xqw; q.bias = L.xqb; q.C = s->dq; q.ldc = d; q.M = R; q.N = d; q.K = d;
            q.flags = kGemmScaleCols; q.scale = q_scale; q.scale_cols = d; q.ln_gamma = L.lnxw; q.ln_beta = L.lnxb;
            launch_gemv(c, q, "nllb_lnx_xq");
        } else {
            launch_layernorm(c, s->dx, d, L.lnxw, L.lnxb, s->dh, d, R, d, "nllb_lnx");
            nl_linear(c, s->dh, d, L.xqw, L.xqb, s->dq, d, R, d, d, kGemmScaleCols, nullptr, 0, "nllb_dec_xq", q_scale, d);
        }
        CrossAttnArgs ca{};
        ca.q = s->dq;
        ca.k = s->cross_kv + (size_t)l * 2 * d;
        ca.v = ca.k + d;
        ca.ldkv = ldkv;
        ca.out = s->datt;
        ca.rows = R; ca.d = d; ca.n_head = H; ca.T = S;
        ca.head_rank = nullptr; ca.ring = nullptr; ca.ring_row = nullptr; ca.beam_of_row = nullptr;
        ca.ring_rows = 0; ca.n_beam = 1; ca.qk_debug = nullptr;
        launch_decoder_cross_attention(c, ca);
        nl_linear(c, s->datt, d, L.xoutw, L.xoutb, s->dx, d, R, d, d, kGemmResidual, s->dx, d, "nllb_dec_xout");
        if (fused) {
            GemmArgs g;
            g.A = s->dx; g.lda = d; g.W = L.fc1w; g.bias = L.fc1b; g.C = s->dwide; g.ldc = f; g.M = R; g.N = f; g.K = d;
            g.flags = kGemmRelu; g.ln_gamma = L.ln2w; g.ln_beta = L.ln2b;
            launch_gemv(c, g, "nllb_ln2_fc1");
            nl_linear(c, s->dwide, f, L.fc2w, L.fc2b, s->dx, d, R, d, f, kGemmResidual, s->dx, d, "nllb_fc2");
        } else {
            nl_ffn(c, L, s->dx, s->dh, s->dwide, R, d, f);
        }
    }
    // final LayerNorm + tied vocabulary projection of the last fed position of every row
    if (fused) {
        GemmArgs lg;
        lg.A = s->dx; lg.lda = d; lg.W = m->emb; lg.C = s->logits; lg.ldc = D.vocab; lg.M = rows; lg.N = D.vocab; lg.K = d;
        lg.ln_gamma = m->dec_lnw; lg.ln_beta = m->dec_lnb;
        launch_gemv(c, lg, "nllb_lnf_logits");
    } else {
        launch_layernorm(c, s->dx + (size_t)(n_tok - 1) * d, (long)n_tok * d, m->dec_lnw, m->dec_lnb, s->hsel, d, rows, d, "nllb_dec_ln");
        nl_linear(c, s->hsel, d, m->emb, nullptr, s->logits, D.vocab, rows, D.vocab, d, 0, nullptr, 0, "nllb_logits");
    }
    if (graph_step)      // results straight into the host-coherent block: no copy node behind the graph either
        launch_logsoftmax_topk(c, s->logits, D.vocab, rows, s->step_k, s->step_vals_dev, s->step_ids_dev, s->topk_scratch, nullptr,
                               nullptr, nullptr, 0);
}

}  // namespace wlk

extern "C" {

int wlk_nllb_arena_floats(const wlk_nllb_dims* dims, uint64_t* n_floats) {
    if (int rc = nl_check_dims(dims)) return rc;
    if (!n_floats) return nl_fail(WLK_ERR_ARG, "NULL argument");
    nl_layout(*dims, n_floats);
    return WLK_OK;
}

int wlk_nllb_tensor_lookup(const wlk_nllb_dims* dims, const char* packed_name, uint64_t* offset_floats, uint64_t* numel) {
    if (int rc = nl_check_dims(dims)) return rc;
    if (!packed_name) return nl_fail(WLK_ERR_ARG, "NULL argument");
    for (const auto& s : nl_layout(*dims, nullptr))
        if (s.name == packed_name) {
            if (offset_floats) *offset_floats = s.offset;
            if (numel) *numel = s.numel;
            return WLK_OK;
        }
    return nl_fail(WLK_ERR_ARG, std::string("unknown packed tensor ") + packed_name);
}

int wlk_nllb_tensor_name(const wlk_nllb_dims* dims, int index, const char** name) {
    if (int rc = nl_check_dims(dims)) return rc;
    if (!name) return nl_fail(WLK_ERR_ARG, "NULL argument");
    static thread_local std::string hold;
    const auto v = nl_layout(*dims, nullptr);
    if (index < 0 || index >= (int)v.size()) return nl_fail(WLK_ERR_ARG, "tensor index out of range");
    hold = v[index].name;
    *name = hold.c_str();
    return WLK_OK;
}

int wlk_nllb_create(const wlk_nllb_dims* dims, int device, wlk_nllb** out) {
    if (int rc = nl_check_dims(dims)) return rc;
    if (!out) return nl_fail(WLK_ERR_ARG, "NULL argument");
    return nl_guarded([&]() {
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
            return nl_fail(WLK_ERR_HIP, "no HIP device: the NLLB backend has no CPU fallback");
        if (device < 0 || device >= n_dev) return nl_fail(WLK_ERR_ARG, "device index out of range");
        WLK_HIP(hipSetDevice(device));
        auto m = std::make_unique<wlk_nllb>();
        m->D = *dims;
        m->device = device;
        m->layout = nl_layout(*dims, &m->arena_floats);
        for (const auto& s : m->layout) m->index[s.name] = &s;
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&m->arena), m->arena_floats * sizeof(float)));
        memset_sync(m->arena, 0, m->arena_floats * sizeof(float));
        *out = m.release();
        return WLK_OK;
    });
}

int wlk_nllb_upload(wlk_nllb* m, const char* packed_name, const float* host, uint64_t numel) {
    if (!m || !packed_name || !host) return nl_fail(WLK_ERR_ARG, "NULL argument");
    return nl_guarded([&]() {
        auto it = m->index.find(packed_name);
        if (it == m->index.end()) return nl_fail(WLK_ERR_ARG, std::string("unknown packed tensor ") + packed_name);
        if (it->second->numel != numel)
            return nl_fail(WLK_ERR_ARG, std::string("size mismatch for ") + packed_name + ": expected " +
                                            std::to_string(it->second->numel) + ", got " + std::to_string(numel));
        WLK_HIP(hipSetDevice(m->device));
        copy_sync(m->arena + it->second->offset, host, numel * sizeof(float), hipMemcpyHostToDevice);
        m->finalized = false;
        return WLK_OK;
    });
}

int wlk_nllb_finalize(wlk_nllb* m) {
    if (!m) return nl_fail(WLK_ERR_ARG, "model is NULL");
    return nl_guarded([&]() {
        const wlk_nllb_dims& D = m->D;
        auto fill = [&](std::vector<NlLayer>& v, int n, const std::string& side, bool cross) {
            v.assign(n, NlLayer{});
            for (int i = 0; i < n; ++i) {
                const std::string p = side + "." + std::to_string(i) + ".";
                NlLayer& w = v[i];
                w.ln1w = m->P(p + "ln1.w"); w.ln1b = m->P(p + "ln1.b");
                w.qkvw = m->P(p + "qkv.w"); w.qkvb = m->P(p + "qkv.b");
                w.outw = m->P(p + "out.w"); w.outb = m->P(p + "out.b");
                if (cross) {
                    w.lnxw = m->P(p + "lnx.w"); w.lnxb = m->P(p + "lnx.b");
                    w.xqw = m->P(p + "xq.w"); w.xqb = m->P(p + "xq.b");
                    w.xkvw = m->P(p + "xkv.w"); w.xkvb = m->P(p + "xkv.b");
                    w.xoutw = m->P(p + "xout.w"); w.xoutb = m->P(p + "xout.b");
                }
                w.ln2w = m->P(p + "ln2.w"); w.ln2b = m->P(p + "ln2.b");
                w.fc1w = m->P(p + "fc1.w"); w.fc1b = m->P(p + "fc1.b");
                w.fc2w = m->P(p + "fc2.w"); w.fc2b = m->P(p + "fc2.b");
            }
        };
        fill(m->enc, D.enc_layers, "enc", false);
        fill(m->dec, D.dec_layers, "dec", true);
        m->emb = m->P("shared.emb"); m->pos = m->P("pos.table");
        m->enc_lnw = m->P("enc.ln.w"); m->enc_lnb = m->P("enc.ln.b");
        m->dec_lnw = m->P("dec.ln.w"); m->dec_lnb = m->P("dec.ln.b");
        m->finalized = true;
        return WLK_OK;
    });
}

int wlk_nllb_destroy(wlk_nllb* m) {
    delete m;        // sessions of the model must be destroyed first (they hold pointers into its arena)
    return WLK_OK;
}

int wlk_nllb_session_create(wlk_nllb* m, int rows, wlk_nllb_session** out) {
    if (!m || !out) return nl_fail(WLK_ERR_ARG, "NULL argument");
    if (!m->finalized) return nl_fail(WLK_ERR_STATE, "NLLB model not finalized");
    if (rows < 1 || rows > 8) return nl_fail(WLK_ERR_ARG, "NLLB session: 1..8 hypothesis rows");
    return nl_guarded([&]() {
        WLK_HIP(hipSetDevice(m->device));
        auto s = std::make_unique<wlk_nllb_session>();
        s->m = m;
        s->rows = rows;
        WLK_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        const wlk_nllb_dims& D = m->D;
        const size_t d = D.d_model, f = D.ffn, S = D.max_src, Tt = D.max_tgt, Rmax = (size_t)rows * Tt;
        s->ex = s->alloc<float>(S * d); s->eh = s->alloc<float>(S * d); s->eqkv = s->alloc<float>(S * 3 * d);
        s->eatt = s->alloc<float>(S * d); s->ewide = s->alloc<float>(S * f); s->enc_out = s->alloc<float>(S * d);
        s->cross_kv = s->alloc<float>(S * D.dec_layers * 2 * d);
        for (int i = 0; i < 2; ++i) {
            s->kcache[i] = s->alloc<float>((size_t)D.dec_layers * rows * Tt * d);
            s->vcache[i] = s->alloc<float>((size_t)D.dec_layers * rows * Tt * d);
        }
        s->dx = s->alloc<float>(Rmax * d); s->dh = s->alloc<float>(Rmax * d); s->dqkv = s->alloc<float>(Rmax * 3 * d);
        s->datt = s->alloc<float>(Rmax * d); s->dq = s->alloc<float>(Rmax * d); s->dwide = s->alloc<float>(Rmax * f);
        s->hsel = s->alloc<float>((size_t)rows * d);
        s->logits = s->alloc<float>((size_t)rows * D.vocab);
        s->tokens_dev = s->alloc<int>(std::max(S, Rmax));
        s->offset_dev = s->alloc<int>(1);
        s->src_rows_dev = s->alloc<int>(rows);
        s->top_vals = s->alloc<float>((size_t)rows * 16);
        s->top_ids = s->alloc<int>((size_t)rows * 16);
        s->topk_scratch = s->alloc<char>(topk_scratch_bytes(rows));
        WLK_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->step_host), (size_t)(rows + 1) * sizeof(int), hipHostMallocMapped));
        WLK_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&s->step_host_dev), s->step_host, 0));
        WLK_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->step_vals_host), (size_t)rows * 16 * sizeof(float), hipHostMallocMapped));
        WLK_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&s->step_vals_dev), s->step_vals_host, 0));
        WLK_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->step_ids_host), (size_t)rows * 16 * sizeof(int), hipHostMallocMapped));
        WLK_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&s->step_ids_dev), s->step_ids_host, 0));
        *out = s.release();
        return WLK_OK;
    });
}

int wlk_nllb_session_destroy(wlk_nllb_session* s) {
    delete s;        // ~wlk_nllb_session releases the stream, graphs and buffers
    return WLK_OK;
}

static int nl_stage_tokens(wlk_nllb_session* s, const int64_t* ids, int n, std::vector<int>& stage) {
    const wlk_nllb_dims& D = s->m->D;
    stage.resize(n);
    for (int i = 0; i < n; ++i) {
        if (ids[i] < 0 || ids[i] >= D.vocab) return nl_fail(WLK_ERR_ARG, "token id out of range");
        if (ids[i] == D.pad_id) return nl_fail(WLK_ERR_ARG, "padding inside a sequence is not supported (one sentence per call)");
        stage[i] = (int)ids[i];
    }
    return WLK_OK;
}

int wlk_nllb_encode(wlk_nllb_session* s, const int64_t* src_ids, int32_t n) {
    if (!s || !src_ids) return nl_fail(WLK_ERR_ARG, "NULL argument");
    if (n < 1 || n > s->m->D.max_src) return nl_fail(WLK_ERR_CAPACITY, "source length out of range");
    return nl_guarded([&]() {
        std::vector<int> stage;
        if (int rc = nl_stage_tokens(s, src_ids, n, stage)) return rc;
        WLK_HIP(hipSetDevice(s->m->device));
        WLK_HIP(hipStreamSynchronize(s->stream));
        WLK_HIP(hipMemcpyAsync(s->tokens_dev, stage.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s->stream));
        WLK_HIP(hipStreamSynchronize(s->stream));
        s->src_len = n;
        nl_encode(s, n);
        s->encoded = true;
        s->self_len = 0;
        s->have_logits = false;
        return WLK_OK;
    });
}

int wlk_nllb_decode(wlk_nllb_session* s, const int64_t* tokens, int32_t n_rows, int32_t n_tok, int32_t first) {
    if (!s || !tokens) return nl_fail(WLK_ERR_ARG, "NULL argument");
    if (!s->encoded) return nl_fail(WLK_ERR_STATE, "wlk_nllb_decode before wlk_nllb_encode");
    if (n_rows != s->rows) return nl_fail(WLK_ERR_ARG, "n_rows must equal the session's row count");
    if (n_tok < 1) return nl_fail(WLK_ERR_ARG, "n_tok must be >= 1");
    return nl_guarded([&]() {
        if (first) s->self_len = 0;
        else if (s->self_len == 0) return nl_fail(WLK_ERR_STATE, "the first decode after an encode must set first=1");
        if (s->self_len + n_tok > s->m->D.max_tgt) return nl_fail(WLK_ERR_CAPACITY, "target context exceeded");
        std::vector<int> stage;
        if (int rc = nl_stage_tokens(s, tokens, n_rows * n_tok, stage)) return rc;
        stage.push_back(s->self_len);
        WLK_HIP(hipSetDevice(s->m->device));
        WLK_HIP(hipStreamSynchronize(s->stream));
        WLK_HIP(hipMemcpyAsync(s->tokens_dev, stage.data(), (size_t)n_rows * n_tok * sizeof(int), hipMemcpyHostToDevice, s->stream));
        WLK_HIP(hipMemcpyAsync(s->offset_dev, stage.data() + (size_t)n_rows * n_tok, sizeof(int), hipMemcpyHostToDevice, s->stream));
        WLK_HIP(hipStreamSynchronize(s->stream));           // `stage` is pageable
        nl_decode(s, n_tok);
        s->self_len += n_tok;
        s->have_logits = true;
        return WLK_OK;
    });
}

int wlk_nllb_step(wlk_nllb_session* s, const int64_t* tokens, int32_t n_rows, int32_t k, float* logprobs, int32_t* ids) {
    if (!s || !tokens || !logprobs || !ids) return nl_fail(WLK_ERR_ARG, "NULL argument");
    if (!s->encoded || s->self_len == 0) return nl_fail(WLK_ERR_STATE, "wlk_nllb_step before the decoder prompt (wlk_nllb_decode first=1)");
    if (n_rows != s->rows) return nl_fail(WLK_ERR_ARG, "n_rows must equal the session's row count");
    if (k < 1 || k > 8) return nl_fail(WLK_ERR_ARG, "k must be 1..8 (the top-k kernel's limit)");
    const wlk_nllb_dims& D = s->m->D;
    if (s->self_len + 1 > D.max_tgt) return nl_fail(WLK_ERR_CAPACITY, "target context exceeded");
    if (!gemv_applicable(n_rows, D.d_model)) return nl_fail(WLK_ERR_ARG, "wlk_nllb_step: too many rows for the single-token path");
    for (int r = 0; r < n_rows; ++r) {
        if (tokens[r] < 0 || tokens[r] >= D.vocab) return nl_fail(WLK_ERR_ARG, "token id out of range");
        if (tokens[r] == D.pad_id) return nl_fail(WLK_ERR_ARG, "padding inside a sequence is not supported");
    }
    return nl_guarded([&]() {
        WLK_HIP(hipSetDevice(s->m->device));
        WLK_HIP(hipStreamSynchronize(s->stream));           // the previous step's readers of the host block are done
        for (int r = 0; r < n_rows; ++r) s->step_host[r] = (int)tokens[r];
        s->step_host[n_rows] = s->self_len;
        hipGraphExec_t& exec = s->step_exec[s->kv_cur];
        if (!exec || s->step_exec_k[s->kv_cur] != k || s->step_exec_src[s->kv_cur] != s->src_len) {
            if (exec) { WLK_HIP(hipGraphExecDestroy(exec)); exec = nullptr; }
            s->step_k = k;
            hipGraph_t graph = nullptr;
            WLK_HIP(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
            try {
                nl_decode(s, 1, /*graph_step=*/true);
            } catch (...) {
                (void)hipStreamEndCapture(s->stream, &graph);
                if (graph) (void)hipGraphDestroy(graph);
                throw;
            }
            WLK_HIP(hipStreamEndCapture(s->stream, &graph));
            const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            WLK_HIP(e);
            s->step_exec_k[s->kv_cur] = k;
            s->step_exec_src[s->kv_cur] = s->src_len;
        }
        WLK_HIP(hipGraphLaunch(exec, s->stream));
        WLK_HIP(hipStreamSynchronize(s->stream));
        std::memcpy(logprobs, s->step_vals_host, (size_t)n_rows * k * sizeof(float));
        std::memcpy(ids, s->step_ids_host, (size_t)n_rows * k * sizeof(int));
        s->self_len += 1;
        s->have_logits = true;
        return WLK_OK;
    });
}

int wlk_nllb_kv_reorder(wlk_nllb_session* s, const int32_t* source_rows, int32_t n_rows) {
    if (!s || !source_rows) return nl_fail(WLK_ERR_ARG, "NULL argument");
    if (n_rows != s->rows) return nl_fail(WLK_ERR_ARG, "n_rows must equal the session's row count");
    bool identity = true;
    for (int i = 0; i < n_rows; ++i) {
        if (source_rows[i] < 0 || source_rows[i] >= n_rows) return nl_fail(WLK_ERR_ARG, "source row out of range");
        identity = identity && source_rows[i] == i;
    }
    if (identity || s->self_len == 0) return WLK_OK;
    return nl_guarded([&]() {
        const wlk_nllb_dims& D = s->m->D;
        WLK_HIP(hipSetDevice(s->m->device));
        WLK_HIP(hipStreamSynchronize(s->stream));
        WLK_HIP(hipMemcpyAsync(s->src_rows_dev, source_rows, (size_t)n_rows * sizeof(int), hipMemcpyHostToDevice, s->stream));
        WLK_HIP(hipStreamSynchronize(s->stream));
        const LaunchCtx c = s->ctx();
        const int nxt = s->kv_cur ^ 1;
        launch_kv_gather(c, s->kcache[s->kv_cur], s->kcache[nxt], s->src_rows_dev, n_rows, s->self_len, D.d_model, D.max_tgt, D.dec_layers);
        launch_kv_gather(c, s->vcache[s->kv_cur], s->vcache[nxt], s->src_rows_dev, n_rows, s->self_len, D.d_model, D.max_tgt, D.dec_layers);
        s->kv_cur = nxt;
        return WLK_OK;
    });
}

int wlk_nllb_topk(wlk_nllb_session* s, int32_t k, float* logprobs, int32_t* ids) {
    if (!s || !logprobs || !ids) return nl_fail(WLK_ERR_ARG, "NULL argument");
    if (!s->have_logits) return nl_fail(WLK_ERR_STATE, "wlk_nllb_topk before a decode");
    if (k < 1 || k > 8) return nl_fail(WLK_ERR_ARG, "k must be 1..8 (the top-k kernel's limit)");
    return nl_guarded([&]() {
        WLK_HIP(hipSetDevice(s->m->device));
        launch_logsoftmax_topk(s->ctx(), s->logits, s->m->D.vocab, s->rows, k, s->top_vals, s->top_ids, s->topk_scratch, nullptr,
                               nullptr, nullptr, 0);
        WLK_HIP(hipMemcpyAsync(logprobs, s->top_vals, (size_t)s->rows * k * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        WLK_HIP(hipMemcpyAsync(ids, s->top_ids, (size_t)s->rows * k * sizeof(int), hipMemcpyDeviceToHost, s->stream));
        WLK_HIP(hipStreamSynchronize(s->stream));
        return WLK_OK;
    });
}

int wlk_nllb_export(wlk_nllb_session* s, const char* what, float* host, uint64_t capacity, uint64_t* n_written) {
    if (!s || !what || !host || !n_written) return nl_fail(WLK_ERR_ARG, "NULL argument");
    return nl_guarded([&]() {
        const wlk_nllb_dims& D = s->m->D;
        const std::string w = what;
        const float* src = nullptr;
        uint64_t n = 0;
        if (w == "logits") {
            if (!s->have_logits) return nl_fail(WLK_ERR_STATE, "no logits yet");
            src = s->logits; n = (uint64_t)s->rows * D.vocab;
        } else if (w == "enc") {
            if (!s->encoded) return nl_fail(WLK_ERR_STATE, "not encoded");
            src = s->enc_out; n = (uint64_t)s->src_len * D.d_model;
        } else {
            return nl_fail(WLK_ERR_ARG, "unknown export " + w);
        }
        if (n > capacity) return nl_fail(WLK_ERR_CAPACITY, "export buffer too small");
        WLK_HIP(hipSetDevice(s->m->device));
        WLK_HIP(hipMemcpyAsync(host, src, n * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        WLK_HIP(hipStreamSynchronize(s->stream));
        *n_written = n;
        return WLK_OK;
    });
}

int wlk_nllb_sync(wlk_nllb_session* s) {
    if (!s) return nl_fail(WLK_ERR_ARG, "NULL argument");
    return nl_guarded([&]() {
        WLK_HIP(hipSetDevice(s->m->device));
        WLK_HIP(hipStreamSynchronize(s->stream));
        return WLK_OK;
    });
}

}  // extern "C"

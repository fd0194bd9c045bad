// C ABI of libwlk_hip.so (include/wlk_hip.h): weight arena, per-stream sessions, and the launch
// sequences for encode / decode / select.  Host-side orchestration only; the arithmetic lives in
// the kernel files next to this one.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/wlk_hip.h"
#include "common.h"
#include "internal.h"

namespace wlk {

namespace {
struct UtilStream {
    std::mutex mu;
    hipStream_t stream = nullptr;
};
UtilStream g_util[64];
UtilStream& util_of_current_device() {
    int dev = 0;
    WLK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) throw std::invalid_argument("device index out of range");
    return g_util[dev];
}
}  // namespace

void copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (bytes == 0) return;
    UtilStream& u = util_of_current_device();
    std::lock_guard<std::mutex> lk(u.mu);
    if (!u.stream) WLK_HIP(hipStreamCreateWithFlags(&u.stream, hipStreamNonBlocking));
    WLK_HIP(hipMemcpyAsync(dst, src, bytes, kind, u.stream));
    WLK_HIP(hipStreamSynchronize(u.stream));
}

void memset_sync(void* dst, int value, size_t bytes) {
    if (bytes == 0) return;
    UtilStream& u = util_of_current_device();
    std::lock_guard<std::mutex> lk(u.mu);
    if (!u.stream) WLK_HIP(hipStreamCreateWithFlags(&u.stream, hipStreamNonBlocking));
    WLK_HIP(hipMemsetAsync(dst, value, bytes, u.stream));
    WLK_HIP(hipStreamSynchronize(u.stream));
}

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
const std::string& get_last_error() { return g_last_error; }

// ------------------------------------------------------------------------------------------------
// profiler: HIP events around every launch on the session stream
// ------------------------------------------------------------------------------------------------
void prof_before(const LaunchCtx& ctx, const char* name, double flops, double bytes) {
    Profiler* p = ctx.prof;
    p->pending = p->get();
    p->pending_name = name;
    p->pending_flops = flops;
    p->pending_bytes = bytes;
    WLK_HIP(hipEventRecord(p->pending, ctx.stream));
}
void prof_after(const LaunchCtx& ctx) {
    Profiler* p = ctx.prof;
    hipEvent_t e = p->get();
    (void)hipEventRecord(e, ctx.stream);
    p->recs.push_back({p->pending_name, p->pending, e, p->pending_flops, p->pending_bytes});
}

// ------------------------------------------------------------------------------------------------
// packed weight arena layout
// ------------------------------------------------------------------------------------------------
static std::vector<TensorSlot> build_layout(const wlk_dims& D, uint64_t* total) {
    std::vector<TensorSlot> v;
    uint64_t off = 0;
    auto add = [&](const std::string& n, uint64_t numel) {
        v.push_back({n, off, numel});
        off += (numel + 63) / 64 * 64;  // 256-byte aligned slots
    };
    const uint64_t da = D.n_audio_state, dt = D.n_text_state;
    add("mel.filters", (uint64_t)D.n_mels * kNFreq);
    add("mel.window", kNFft);
    add("enc.conv1.w", da * 3 * D.n_mels);
    add("enc.conv1.b", da);
    add("enc.conv2.w", da * 3 * da);
    add("enc.conv2.b", da);
    add("enc.pos", (uint64_t)D.n_audio_ctx * da);
    for (int i = 0; i < D.n_audio_layer; ++i) {
        const std::string p = "enc." + std::to_string(i) + ".";
        add(p + "ln1.w", da); add(p + "ln1.b", da);
        add(p + "qkv.w", 3 * da * da); add(p + "qkv.b", 3 * da);
        add(p + "out.w", da * da); add(p + "out.b", da);
        add(p + "ln2.w", da); add(p + "ln2.b", da);
        add(p + "fc1.w", 4 * da * da); add(p + "fc1.b", 4 * da);
        add(p + "fc2.w", 4 * da * da); add(p + "fc2.b", da);
    }
    add("enc.ln_post.w", da); add("enc.ln_post.b", da);
    add("dec.tok_emb", (uint64_t)D.n_vocab * dt);
    add("dec.pos", (uint64_t)D.n_text_ctx * dt);
    for (int i = 0; i < D.n_text_layer; ++i) {
        const std::string p = "dec." + std::to_string(i) + ".";
        add(p + "ln1.w", dt); add(p + "ln1.b", dt);
        add(p + "qkv.w", 3 * dt * dt); add(p + "qkv.b", 3 * dt);
        add(p + "out.w", dt * dt); add(p + "out.b", dt);
        add(p + "lnx.w", dt); add(p + "lnx.b", dt);
        add(p + "xq.w", dt * dt); add(p + "xq.b", dt);
        add(p + "xkv.w", 2 * dt * da); add(p + "xkv.b", 2 * dt);
        add(p + "xout.w", dt * dt); add(p + "xout.b", dt);
        add(p + "ln2.w", dt); add(p + "ln2.b", dt);
        add(p + "fc1.w", 4 * dt * dt); add(p + "fc1.b", 4 * dt);
        add(p + "fc2.w", 4 * dt * dt); add(p + "fc2.b", dt);
    }
    add("dec.ln.w", dt); add("dec.ln.b", dt);
    if (total) *total = off;
    return v;
}

static int check_dims(const wlk_dims* d) {
    if (!d) return fail(WLK_ERR_ARG, "dims is NULL");
    if (d->n_audio_state != d->n_audio_head * kHeadDim || d->n_text_state != d->n_text_head * kHeadDim)
        return fail(WLK_ERR_ARG, "only 64-wide attention heads are supported");
    if (d->n_audio_state != d->n_text_state) return fail(WLK_ERR_ARG, "audio/text widths must match");
    if (d->n_mels % 4 != 0 || d->n_mels > 256) return fail(WLK_ERR_ARG, "n_mels must be a multiple of 4, <= 256");
    if (d->n_audio_ctx != 1500) return fail(WLK_ERR_ARG, "n_audio_ctx must be 1500 (30 s window)");
    if (d->n_text_ctx > 448) return fail(WLK_ERR_ARG, "n_text_ctx must be <= 448");
    if (d->n_audio_state > 1536) return fail(WLK_ERR_ARG, "model width must be <= 1536");
    return WLK_OK;
}

}  // namespace wlk

using namespace wlk;

static LayerW layer_weights(const wlk_model* m, const char* side, int i, bool cross) {
    const std::string p = std::string(side) + "." + std::to_string(i) + ".";
    LayerW L;
    L.ln1w = m->w(p + "ln1.w"); L.ln1b = m->w(p + "ln1.b");
    L.qkvw = m->w(p + "qkv.w"); L.qkvb = m->w(p + "qkv.b");
    L.outw = m->w(p + "out.w"); L.outb = m->w(p + "out.b");
    L.ln2w = m->w(p + "ln2.w"); L.ln2b = m->w(p + "ln2.b");
    L.fc1w = m->w(p + "fc1.w"); L.fc1b = m->w(p + "fc1.b");
    L.fc2w = m->w(p + "fc2.w"); L.fc2b = m->w(p + "fc2.b");
    if (cross) {
        L.lnxw = m->w(p + "lnx.w"); L.lnxb = m->w(p + "lnx.b");
        L.xqw = m->w(p + "xq.w"); L.xqb = m->w(p + "xq.b");
        L.xkvw = m->w(p + "xkv.w"); L.xkvb = m->w(p + "xkv.b");
        L.xoutw = m->w(p + "xout.w"); L.xoutb = m->w(p + "xout.b");
    }
    return L;
}

// ------------------------------------------------------------------------------------------------
// extern "C"
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* wlk_last_error(void) { return g_last_error.c_str(); }
int wlk_abi_version(void) { return 2; }   // 2: wlk_loop_params carries the teacher-forcing block

int wlk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int wlk_arena_floats(const wlk_dims* dims, uint64_t* n_floats) {
    if (int rc = check_dims(dims)) return rc;
    if (!n_floats) return fail(WLK_ERR_ARG, "n_floats is NULL");
    build_layout(*dims, n_floats);
    return WLK_OK;
}

int wlk_tensor_lookup(const wlk_dims* dims, const char* packed_name, uint64_t* offset_floats, uint64_t* numel) {
    if (int rc = check_dims(dims)) return rc;
    if (!packed_name) return fail(WLK_ERR_ARG, "name is NULL");
    for (const auto& s : build_layout(*dims, nullptr)) {
        if (s.name == packed_name) {
            if (offset_floats) *offset_floats = s.offset;
            if (numel) *numel = s.numel;
            return WLK_OK;
        }
    }
    return fail(WLK_ERR_ARG, std::string("unknown tensor ") + packed_name);
}

int wlk_tensor_name(const wlk_dims* dims, int index, const char** name) {
    if (int rc = check_dims(dims)) return rc;
    static thread_local std::string hold;
    auto v = build_layout(*dims, nullptr);
    if (index < 0 || index >= (int)v.size() || !name) return fail(WLK_ERR_ARG, "tensor index out of range");
    hold = v[index].name;
    *name = hold.c_str();
    return WLK_OK;
}

int wlk_model_create(const wlk_dims* dims, int device, float* arena_dev, wlk_model** out) {
    if (int rc = check_dims(dims)) return rc;
    if (!out) return fail(WLK_ERR_ARG, "out is NULL");
    return guarded([&]() {
        WLK_HIP(hipSetDevice(device));
        auto m = std::make_unique<wlk_model>();
        m->D = *dims;
        m->device = device;
        m->slots = build_layout(*dims, &m->arena_floats);
        for (const auto& s : m->slots) m->by_name[s.name] = &s;
        if (arena_dev) {
            m->arena = arena_dev;
        } else {
            m->arena = dev_alloc<float>(m->arena_floats);
            m->owns_arena = true;
            memset_sync(m->arena, 0, m->arena_floats * sizeof(float));
        }
        std::vector<double> tw(kNFft);
        for (int i = 0; i < kNFft; ++i) tw[i] = std::cos(2.0 * M_PI * (double)i / (double)kNFft);
        m->twiddle = dev_alloc<double>(kNFft);
        copy_sync(m->twiddle, tw.data(), kNFft * sizeof(double), hipMemcpyHostToDevice);
        m->filt_lo = dev_alloc<int>(dims->n_mels);
        m->filt_hi = dev_alloc<int>(dims->n_mels);
        m->head_rank = dev_alloc<int>((size_t)dims->n_text_layer * dims->n_text_head);
        m->layer_ranks = dev_alloc<int>((size_t)dims->n_text_layer * dims->n_text_head);
        m->layer_heads = dev_alloc<int>((size_t)dims->n_text_layer * dims->n_text_head);
        m->all_ranks = dev_alloc<int>((size_t)dims->n_text_layer * dims->n_text_head);
        {
            std::vector<int> iota((size_t)dims->n_text_layer * dims->n_text_head);
            for (size_t i = 0; i < iota.size(); ++i) iota[i] = (int)i;
            copy_sync(m->all_ranks, iota.data(), iota.size() * sizeof(int), hipMemcpyHostToDevice);
        }
        m->layer_rank_count.assign(dims->n_text_layer, 0);
        std::vector<int> none((size_t)dims->n_text_layer * dims->n_text_head, -1);
        copy_sync(m->head_rank, none.data(), none.size() * sizeof(int), hipMemcpyHostToDevice);
        *out = m.release();
        return WLK_OK;
    });
}

int wlk_model_arena(wlk_model* m, float** arena_dev, uint64_t* n_floats) {
    if (!m) return fail(WLK_ERR_ARG, "model is NULL");
    if (arena_dev) *arena_dev = m->arena;
    if (n_floats) *n_floats = m->arena_floats;
    return WLK_OK;
}

int wlk_model_upload(wlk_model* m, const char* packed_name, const float* host, uint64_t numel) {
    if (!m || !packed_name || !host) return fail(WLK_ERR_ARG, "NULL argument");
    // a finalized model is immutable: sessions size their alignment window from it, finalize copied the
    // cross-attention k|v weights, and captured step graphs bake the per-layer head counts in
    if (m->finalized) return fail(WLK_ERR_STATE, "model is finalized (immutable): upload weights before wlk_model_finalize");
    return guarded([&]() {
        auto it = m->by_name.find(packed_name);
        if (it == m->by_name.end()) return fail(WLK_ERR_ARG, std::string("unknown tensor ") + packed_name);
        if (it->second->numel != numel)
            return fail(WLK_ERR_ARG, std::string("size mismatch for ") + packed_name + ": expected " +
                                         std::to_string(it->second->numel) + ", got " + std::to_string(numel));
        WLK_HIP(hipSetDevice(m->device));
        copy_sync(m->arena + it->second->offset, host, numel * sizeof(float), hipMemcpyHostToDevice);
        return WLK_OK;
    });
}

int wlk_model_set_alignment_heads(wlk_model* m, const int32_t* pairs, int n_pairs) {
    if (!m || (n_pairs > 0 && !pairs) || n_pairs < 0) return fail(WLK_ERR_ARG, "bad alignment head list");
    if (m->finalized)
        return fail(WLK_ERR_STATE, "model is finalized (immutable): set alignment heads before wlk_model_finalize");
    return guarded([&]() {
        const int L = m->D.n_text_layer, H = m->D.n_text_head;
        std::vector<int> rank((size_t)L * H, -1);
        for (int i = 0; i < n_pairs; ++i) {
            const int l = pairs[2 * i], h = pairs[2 * i + 1];
            if (l < 0 || l >= L || h < 0 || h >= H) return fail(WLK_ERR_ARG, "alignment head out of range");
            // the kernels launch one side block per PAIR of a layer but look heads up in a table of DISTINCT heads: a
            // repeated (layer, head) would send a block to head 0 / rank -1 (an out-of-bounds window row)
            if (rank[(size_t)l * H + h] >= 0) return fail(WLK_ERR_ARG, "alignment head listed twice");
            rank[(size_t)l * H + h] = i;
        }
        WLK_HIP(hipSetDevice(m->device));
        copy_sync(m->head_rank, rank.data(), rank.size() * sizeof(int), hipMemcpyHostToDevice);
        std::vector<int> compact((size_t)L * H, -1);
        m->layer_rank_count.assign(L, 0);
        for (int i = 0; i < n_pairs; ++i) compact[(size_t)pairs[2 * i] * H + m->layer_rank_count[pairs[2 * i]]++] = i;
        copy_sync(m->layer_ranks, compact.data(), compact.size() * sizeof(int), hipMemcpyHostToDevice);
        std::vector<int> heads((size_t)L * H, 0);
        for (int l = 0; l < L; ++l) {
            int n = 0;
            for (int h = 0; h < H; ++h)
                if (rank[(size_t)l * H + h] >= 0) heads[(size_t)l * H + n++] = h;
        }
        copy_sync(m->layer_heads, heads.data(), heads.size() * sizeof(int), hipMemcpyHostToDevice);
        m->align_pairs.assign(pairs, pairs + 2 * n_pairs);
        m->n_align = n_pairs;
        return WLK_OK;
    });
}

int wlk_model_finalize(wlk_model* m) {
    if (!m) return fail(WLK_ERR_ARG, "model is NULL");
    return guarded([&]() {
        WLK_HIP(hipSetDevice(m->device));
        // sparse extent of every mel filter row, read back from the arena so that ranks that
        // received the weights by broadcast derive the same table
        const int nm = m->D.n_mels;
        std::vector<float> f((size_t)nm * kNFreq);
        copy_sync(f.data(), m->w("mel.filters"), f.size() * sizeof(float), hipMemcpyDeviceToHost);
        std::vector<int> lo(nm), hi(nm);
        for (int i = 0; i < nm; ++i) {
            int a = kNFreq, b = 0;
            for (int k = 0; k < kNFreq; ++k)
                if (f[(size_t)i * kNFreq + k] != 0.f) {
                    a = std::min(a, k);
                    b = k + 1;
                }
            if (b == 0) a = 0;
            lo[i] = a;
            hi[i] = b;
        }
        copy_sync(m->filt_lo, lo.data(), nm * sizeof(int), hipMemcpyHostToDevice);
        copy_sync(m->filt_hi, hi.data(), nm * sizeof(int), hipMemcpyHostToDevice);
        m->enc_layers.clear();
        m->dec_layers.clear();
        for (int i = 0; i < m->D.n_audio_layer; ++i) m->enc_layers.push_back(layer_weights(m, "enc", i, false));
        for (int i = 0; i < m->D.n_text_layer; ++i) m->dec_layers.push_back(layer_weights(m, "dec", i, true));
        {
            const size_t d = m->D.n_text_state, da = m->D.n_audio_state, Ld = m->D.n_text_layer;
            if (!m->xkv_all_w) m->xkv_all_w = dev_alloc<float>(Ld * 2 * d * da);
            if (!m->xkv_all_b) m->xkv_all_b = dev_alloc<float>(Ld * 2 * d);
            for (size_t i = 0; i < Ld; ++i) {
                copy_sync(m->xkv_all_w + i * 2 * d * da, m->dec_layers[i].xkvw, 2 * d * da * sizeof(float),
                                  hipMemcpyDeviceToDevice);
                copy_sync(m->xkv_all_b + i * 2 * d, m->dec_layers[i].xkvb, 2 * d * sizeof(float),
                                  hipMemcpyDeviceToDevice);
            }
        }
        {   // X3 images of the encoder's wide projection weights (gemm_x3.hip); shapes the wide kernel does not take keep
            // their fp32 kernels.  Built here, from the arena, so that ranks that got the weights by broadcast build the same.
            for (unsigned short* p : m->x3_owned) (void)hipFree(p);
            m->x3_owned.clear();
            m->xkv_all_w3 = nullptr;
            const int da = m->D.n_audio_state, T = m->D.n_audio_ctx;
            LaunchCtx c;
            c.stream = nullptr;
            hipStream_t st;
            WLK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            struct StreamGuard {      // a throwing pack / allocation must not leak the temporary stream
                hipStream_t s;
                ~StreamGuard() { (void)hipStreamDestroy(s); }
            } st_guard{st};
            c.stream = st;
            auto pack = [&](const float* w, int n_rows, int k = 0) -> unsigned short* {
                if (k == 0) k = da;
                unsigned short* p3 = dev_alloc<unsigned short>(x3_w_elems(n_rows, k));
                m->x3_owned.push_back(p3);
                launch_x3_pack_w(c, w, k, p3, n_rows, k);      // fragment-major: the wide kernel's waves load it straight into registers
                return p3;
            };
            // Round 6: the N = d projections (attention out, fc2) too where d >= 1024 (medium, large-v3): measured on MI355X
            // (scripts/x3_probe.py, profiles/r06_x3_narrow_probe.txt) at M = 1500: N = K = 1280 40.3 us against 52.4 for the fp32
            // k-pipe kernel, N 1280 / K 5120 147 against 187 - and N = 512 (base.en: 64 workgroups of 96 x 128) 18.9 against 11.3,
            // 53.6 against 33.9: gemm_x3_wide_applicable's N >= 1024 draws exactly that line.  Their A operands (attention
            // output, GELU output) are fp32 rows, packed into X3 by one elementwise launch each.  WLK_X3_NARROW=0 switches it off.
            static const bool narrow = [] { const char* e = getenv("WLK_X3_NARROW"); return !(e && e[0] == '0'); }();
            for (auto& L : m->enc_layers) {
                L.qkvw3 = gemm_x3_wide_applicable(T, 3 * da, da, da) ? pack(L.qkvw, 3 * da) : nullptr;
                L.fc1w3 = gemm_x3_wide_applicable(T, 4 * da, da, da) ? pack(L.fc1w, 4 * da) : nullptr;
                L.outw3 = narrow && gemm_x3_wide_applicable(T, da, da, da) ? pack(L.outw, da) : nullptr;
                L.fc2w3 = narrow && gemm_x3_wide_applicable(T, da, 4 * da, 4 * da) ? pack(L.fc2w, da, 4 * da) : nullptr;
            }
            const int n_xkv = m->D.n_text_layer * 2 * m->D.n_text_state;
            if (gemm_x3_wide_applicable(T, n_xkv, da, da)) m->xkv_all_w3 = pack(m->xkv_all_w, n_xkv);
            WLK_HIP(hipStreamSynchronize(st));
        }
        m->w_tok_emb = m->w("dec.tok_emb");
        m->w_dec_pos = m->w("dec.pos");
        m->w_ln_w = m->w("dec.ln.w");
        m->w_ln_b = m->w("dec.ln.b");
        m->finalized = true;
        return WLK_OK;
    });
}

int wlk_model_destroy(wlk_model* m) {
    if (!m) return WLK_OK;
    wlk_engine_destroy_for_model(m);
    (void)hipSetDevice(m->device);
    if (m->owns_arena) (void)hipFree(m->arena);
    if (m->xkv_all_w) (void)hipFree(m->xkv_all_w);
    if (m->xkv_all_b) (void)hipFree(m->xkv_all_b);
    for (unsigned short* p : m->x3_owned) (void)hipFree(p);
    (void)hipFree(m->twiddle);
    (void)hipFree(m->filt_lo);
    (void)hipFree(m->filt_hi);
    (void)hipFree(m->head_rank);
    (void)hipFree(m->layer_heads);
    (void)hipFree(m->layer_ranks);
    (void)hipFree(m->all_ranks);
    delete m;
    return WLK_OK;
}

// ------------------------------------------------------------------------------------------------
int wlk_session_create(wlk_model* m, int beam, int max_audio_samples, wlk_session** out) {
    if (!m || !out) return fail(WLK_ERR_ARG, "NULL argument");
    if (!m->finalized) return fail(WLK_ERR_STATE, "model not finalized");
    if (beam < 1 || beam > 7) return fail(WLK_ERR_ARG, "beam must be in [1, 7]");
    if (max_audio_samples < kHop) return fail(WLK_ERR_ARG, "max_audio_samples too small");
    return guarded([&]() {
        WLK_HIP(hipSetDevice(m->device));
        auto s = std::make_unique<wlk_session>();
        s->m = m;
        s->beam = beam;
        s->audio_cap = max_audio_samples;
        WLK_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        if (const char* e = std::getenv("WLK_NO_GRAPH")) s->use_graph = !(e[0] == '1');
        hipStream_t st = s->stream;
        const wlk_dims& D = m->D;
        const size_t d = D.n_audio_state, T = D.n_audio_ctx, ctx = D.n_text_ctx, V = D.n_vocab;
        s->audio[0] = dev_alloc_zero<float>(s->audio_cap, st);
        s->audio[1] = dev_alloc_zero<float>(s->audio_cap, st);
        s->frame_cap = s->audio_cap / kHop + 4;
        s->logmel = dev_alloc<float>((size_t)s->frame_cap * D.n_mels);
        s->frame_max = dev_alloc<float>(s->frame_cap);
        s->logmel_alt = dev_alloc<float>((size_t)s->frame_cap * D.n_mels);
        s->frame_max_alt = dev_alloc<float>(s->frame_cap);
        if (const char* e = std::getenv("WLK_MEL_INCREMENTAL")) s->mel_incremental = !(e[0] == '0');
        s->mel_t = dev_alloc_zero<float>((size_t)(kMelFrames + 2) * D.n_mels, st);
        s->x1p = dev_alloc_zero<float>((size_t)(kMelFrames + 1) * d, st);
        s->ex = dev_alloc<float>(T * d);
        s->eh = dev_alloc<float>(T * d);
        s->eqkv = dev_alloc<float>(T * 3 * d);
        s->eatt = dev_alloc<float>(T * d);
        s->emlp = dev_alloc<float>(T * 4 * d);
        s->enc_out = dev_alloc<float>(T * d);
        {   // X3 operand buffers (6 bytes per element) only where this model's encode chain takes the X3 path (the choice is
            // the model's: weights packed by wlk_model_finalize; WLK_X3=0 / WLK_X3_ATTN=0 leave them out)
            bool any_x3 = m->xkv_all_w3 != nullptr, qkv_x3 = false;
            bool fc2_x3 = false;
            for (const auto& L : m->enc_layers) {
                any_x3 |= L.qkvw3 || L.fc1w3 || L.outw3;
                qkv_x3 |= L.qkvw3 != nullptr;
                fc2_x3 |= L.fc2w3 != nullptr;
            }
            if (fc2_x3) s->emlp3 = dev_alloc<unsigned short>(T * 3 * 4 * d);
            const bool attn_x3 = qkv_x3 && enc_attention_x3_enabled() && d == (size_t)D.n_audio_head * 64 && T >= 64 && (2 * d) % 128 == 0;
            if (any_x3) s->eh3 = dev_alloc<unsigned short>(T * 3 * d);
            if (attn_x3) s->eqkv3 = dev_alloc_zero<unsigned short>(x3_attn_image_elems((int)T, (int)d), st);
            if (m->xkv_all_w3) s->enc_out3 = dev_alloc<unsigned short>(T * 3 * d);
        }
        s->cross_kv = dev_alloc<float>((size_t)D.n_text_layer * T * 2 * d);

        s->max_rows = beam * (int)ctx;
        const size_t R = s->max_rows;
        s->step_in = dev_alloc<int>(3 * R + 4);
        s->tokens_dev = s->step_in;
        s->dx = dev_alloc<float>(R * d);
        s->dh = dev_alloc<float>(R * d);
        s->dqkv = dev_alloc<float>(R * 3 * d);
        s->datt = dev_alloc<float>(R * d);
        s->dq = dev_alloc<float>(R * d);
        s->dmlp = dev_alloc<float>(R * 4 * d);
        const size_t cache = (size_t)D.n_text_layer * beam * ctx * d;
        s->kcache[0] = dev_alloc<float>(cache);
        s->vcache[0] = dev_alloc<float>(cache);
        s->hsel = dev_alloc<float>((size_t)2 * beam * d);
        // one allocation: the sot-row logits follow the last-row logits, so a beam-1 prefill projects both rows
        // in a single pass over the vocabulary weights
        s->logits_last = dev_alloc<float>((size_t)2 * beam * V);
        s->logits_sot = s->logits_last + (size_t)beam * V;
        s->ring_row = s->step_in + R;
        s->beam_of_row = s->step_in + 2 * R;
        s->d_offset = s->step_in + 3 * R;
        s->ring_rows = (int)ctx + kAlignWindow;
        if (m->n_align > 0) s->ring = dev_alloc<float>((size_t)m->n_align * beam * s->ring_rows * T);
        s->xsplit = dev_alloc<float>(cross_split_scratch_floats(8, D.n_text_head, (int)T));
        s->fsplit = dev_alloc<float>(flash_split_scratch_floats(s->max_rows, D.n_text_head, wlk_session::kFlashSplitsMax));
        // partial softmax states of the encoder attention's key splits (<= 8 per (query, head))
        s->esplit = dev_alloc<float>(flash_split_scratch_floats(D.n_audio_ctx, D.n_audio_head, 8));
        s->z = dev_alloc<float>((size_t)beam * std::max(m->n_align, 1) * T + (size_t)beam * 128);   // + the read-out's per-block (value, frame) pairs
        s->attn_last = dev_alloc_zero<float>((size_t)beam * T, st);
        s->adj_row = dev_alloc<int>(3 * wlk_session::kAdjCap);   // [rows n | ids n | deltas n] packed per call
        s->src_rows = dev_alloc<int>(8);
        s->top_vals = dev_alloc<float>((size_t)beam * 18);       // [log-probs B*8 | ids B*8 | frames B | no-speech B]: ONE D2H
        s->top_ids = reinterpret_cast<int*>(s->top_vals) + (size_t)beam * 8;
        s->frames = s->top_ids + (size_t)beam * 8;
        WLK_HIP(hipMalloc(&s->topk_scratch, topk_scratch_bytes(beam)));
        s->probs = dev_alloc<float>(beam);
        WLK_HIP(hipHostMalloc(&s->pinned, wlk_session::kPinnedBytes, hipHostMallocDefault));
        WLK_HIP(hipHostMalloc(&s->dec_stage, 65536, hipHostMallocDefault));
        WLK_HIP(hipHostMalloc(&s->audio_stage, wlk_session::kPinnedBytes, hipHostMallocDefault));
        WLK_HIP(hipEventCreateWithFlags(&s->audio_stage_ev, hipEventDisableTiming));
        WLK_HIP(hipEventCreateWithFlags(&s->dec_stage_ev, hipEventDisableTiming));
        if (beam == 1) {
            void* hp = nullptr;
            WLK_HIP(hipHostMalloc(&hp, 4096, hipHostMallocMapped | hipHostMallocCoherent));
            std::memset(hp, 0, 4096);
            s->step_host = static_cast<StepBlock*>(hp);
            s->result_host = reinterpret_cast<StepResult*>(static_cast<char*>(hp) + 2048);
            void* dp = nullptr;
            WLK_HIP(hipHostGetDevicePointer(&dp, hp, 0));
            s->step_host_dev = static_cast<StepBlock*>(dp);
            s->result_host_dev = reinterpret_cast<StepResult*>(static_cast<char*>(dp) + 2048);
            s->step_dev = reinterpret_cast<StepBlock*>(dev_alloc<int>(sizeof(StepBlock) / 4));
        }
        WLK_HIP(hipStreamSynchronize(st));
        *out = s.release();
        return WLK_OK;
    });
}

int wlk_session_destroy(wlk_session* s) {
    if (!s) return WLK_OK;
    if (s->step_count && getenv("WLK_STEP_TIMING"))
        fprintf(stderr, "[wlk] one-replay steps: %llu, mean %.1f us per step (graph launch call %.1f us)\n",
                (unsigned long long)s->step_count, s->step_ns / 1e3 / s->step_count, s->step_launch_ns / 1e3 / s->step_count);
    (void)wlk_engine_detach(s);
    (void)hipSetDevice(s->m->device);
    (void)hipStreamSynchronize(s->stream);
    float* fl[] = {s->audio[0], s->audio[1], s->logmel, s->frame_max, s->logmel_alt, s->frame_max_alt, s->mel_t, s->x1p, s->ex, s->eh, s->eqkv,
                   s->eatt, s->emlp, s->enc_out, s->cross_kv, s->dx, s->dh, s->dqkv, s->datt, s->dq, s->dmlp,
                   s->kcache[0], s->kcache[1], s->vcache[0], s->vcache[1], s->hsel, s->logits_last,
                   s->ring, s->z, s->attn_last, s->qk_debug, s->xsplit, s->fsplit, s->top_vals, s->probs};
    for (float* p : fl)
        if (p) (void)hipFree(p);
    int* il[] = {s->step_in, s->adj_row, s->src_rows};
    for (int* p : il)
        if (p) (void)hipFree(p);
    for (auto& e : s->step_exec)
        if (e) (void)hipGraphExecDestroy(e);
    if (s->topk_scratch) (void)hipFree(s->topk_scratch);
    if (s->pcm16_dev) (void)hipFree(s->pcm16_dev);
    if (s->eh3) (void)hipFree(s->eh3);
    if (s->eqkv3) (void)hipFree(s->eqkv3);
    if (s->enc_out3) (void)hipFree(s->enc_out3);
    if (s->emlp3) (void)hipFree(s->emlp3);
    if (s->wa_buf) (void)hipFree(s->wa_buf);
    if (s->esplit) (void)hipFree(s->esplit);
    if (s->pinned) (void)hipHostFree(s->pinned);
    if (s->dec_stage) (void)hipHostFree(s->dec_stage);
    if (s->audio_stage) (void)hipHostFree(s->audio_stage);
    if (s->pf_stream) {
        (void)hipStreamSynchronize(s->pf_stream);
        (void)hipStreamDestroy(s->pf_stream);
    }
    if (s->rules_mask) (void)hipFree(s->rules_mask);
    if (s->pick_out) (void)hipFree(s->pick_out);
    if (s->pf_table) (void)hipFree(s->pf_table);
    if (s->pf_progress) (void)hipFree(s->pf_progress);
    if (s->audio_stage_ev) (void)hipEventDestroy(s->audio_stage_ev);
    if (s->dec_stage_ev) (void)hipEventDestroy(s->dec_stage_ev);
    if (s->step_host) (void)hipHostFree(s->step_host);
    if (s->step_dev) (void)hipFree(s->step_dev);
    if (s->step_align_dev) (void)hipFree(s->step_align_dev);
    for (auto& e : s->fstep_exec)
        if (e) (void)hipGraphExecDestroy(e);
    for (auto& r : s->prof.recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : s->prof.pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(s->stream);
    delete s;
    return WLK_OK;
}

int wlk_session_set_debug(wlk_session* s, int on) {
    if (!s) return fail(WLK_ERR_ARG, "session is NULL");
    return guarded([&]() {
        WLK_HIP(hipSetDevice(s->m->device));
        s->debug = on != 0;
        if (s->debug && !s->qk_debug) {
            const wlk_dims& D = s->m->D;
            s->qk_debug = dev_alloc<float>((size_t)D.n_text_layer * s->max_rows * D.n_text_head * D.n_audio_ctx);
        }
        return WLK_OK;
    });
}

// A/B switch (WLK_DRAIN_STAGING=1): drain the stream around the staging blocks as the library did before the blocks had
// events of their own
static bool drain_staging() {
    static const bool on = [] {
        const char* e = getenv("WLK_DRAIN_STAGING");
        return e && e[0] == '1';
    }();
    return on;
}

// ---- audio ----------------------------------------------------------------------------------
int wlk_audio_append(wlk_session* s, const float* pcm_host, int n) {
    if (!s || (n > 0 && !pcm_host) || n < 0) return fail(WLK_ERR_ARG, "bad audio chunk");
    if (s->audio_len + n > s->audio_cap)
        return fail(WLK_ERR_CAPACITY, "audio buffer capacity exceeded (" + std::to_string(s->audio_len + n) + " > " +
                                          std::to_string(s->audio_cap) + " samples)");
    if (n == 0) return WLK_OK;
    return guarded([&]() {
        WLK_HIP(hipSetDevice(s->m->device));
        float* dst = s->audio[s->audio_cur] + s->audio_len;
        size_t done = 0;
        // stage through pinned memory so the copy is asynchronous and the caller's buffer is free on return
        while (done < (size_t)n) {
            const size_t chunk = std::min((size_t)n - done, wlk_session::kPinnedBytes / sizeof(float));
            if (s->audio_stage_used) WLK_HIP(hipEventSynchronize(s->audio_stage_ev));   // the block is single-entry
            std::memcpy(s->audio_stage, pcm_host + done, chunk * sizeof(float));
            WLK_HIP(hipMemcpyAsync(dst + done, s->audio_stage, chunk * sizeof(float), hipMemcpyHostToDevice, s->stream));
            WLK_HIP(hipEventRecord(s->audio_stage_ev, s->stream));
            s->audio_stage_used = true;
            done += chunk;
        }
        if (drain_staging()) WLK_HIP(hipStreamSynchronize(s->stream));
        s->audio_len += n;
        s->encoded = false;
        return WLK_OK;
    });
}

int wlk_audio_append_pcm16(wlk_session* s, const int16_t* pcm_host, int n) {
    if (!s || (n > 0 && !pcm_host) || n < 0) return fail(WLK_ERR_ARG, "bad audio chunk");
    if (s->audio_len + n > s->audio_cap)
        return fail(WLK_ERR_CAPACITY, "audio buffer capacity exceeded (" + std::to_string(s->audio_len + n) + " > " +
                                          std::to_string(s->audio_cap) + " samples)");
    if (n == 0) return WLK_OK;
    return guarded([&]() {
        WLK_HIP(hipSetDevice(s->m->device));
        constexpr size_t kMax = wlk_session::kPinnedBytes / sizeof(int16_t);
        if (!s->pcm16_dev) WLK_HIP(hipMalloc(reinterpret_cast<void**>(&s->pcm16_dev), kMax * sizeof(int16_t)));
        float* dst = s->audio[s->audio_cur] + s->audio_len;
        const LaunchCtx c = s->ctx();
        size_t done = 0;
        while (done < (size_t)n) {   // half the PCIe bytes of the fp32 path; widened on the device
            const size_t chunk = std::min((size_t)n - done, kMax);
            // the pinned block and the device staging buffer are single-entry: wait for the previous chunk's widening
            if (s->audio_stage_used) WLK_HIP(hipEventSynchronize(s->audio_stage_ev));
            std::memcpy(s->audio_stage, pcm_host + done, chunk * sizeof(int16_t));
            WLK_HIP(hipMemcpyAsync(s->pcm16_dev, s->audio_stage, chunk * sizeof(int16_t), hipMemcpyHostToDevice, s->stream));
            launch_pcm16_to_float(c, s->pcm16_dev, dst + done, (int)chunk);
            WLK_HIP(hipEventRecord(s->audio_stage_ev, s->stream));
            s->audio_stage_used = true;
            done += chunk;
        }
        if (drain_staging()) WLK_HIP(hipStreamSynchronize(s->stream));
        s->audio_len += n;
        s->encoded = false;
        return WLK_OK;
    });
}

int wlk_audio_append_zeros(wlk_session* s, int n) {
    if (!s || n < 0) return fail(WLK_ERR_ARG, "bad zero run");
    if (s->audio_len + n > s->audio_cap) return fail(WLK_ERR_CAPACITY, "audio buffer capacity exceeded");
    if (n == 0) return WLK_OK;
    return guarded([&]() {
        WLK_HIP(hipSetDevice(s->m->device));
        WLK_HIP(hipMemsetAsync(s->audio[s->audio_cur] + s->audio_len, 0, (size_t)n * sizeof(float), s->stream));
        s->audio_len += n;
        s->encoded = false;
        return WLK_OK;
    });
}

int wlk_audio_drop_front(wlk_session* s, int n) {
    if (!s || n < 0 || n > s->audio_len) return fail(WLK_ERR_ARG, "bad eviction length");
    if (n == 0) return WLK_OK;
    return guarded([&]() {
        WLK_HIP(hipSetDevice(s->m->device));
        const int keep = s->audio_len - n;
        if (keep > 0)
            WLK_HIP(hipMemcpyAsync(s->audio[s->audio_cur ^ 1], s->audio[s->audio_cur] + n, (size_t)keep * sizeof(float),
                                   hipMemcpyDeviceToDevice, s->stream));
        s->audio_cur ^= 1;
        s->audio_len = keep;
        s->encoded = false;
        // mel cache: whole evicted frames shift the cached rows; the two head frames (reflect padding at the new
        // start) are redone by the next encode.  Anything else invalidates the cache.
        const int shift = n / kHop;
        if (s->mel_incremental && n % kHop == 0 && s->mel_cached_samples > n && shift < s->mel_cached_frames) {
            const int rows = s->mel_cached_frames - shift;
            const int nm = s->m->D.n_mels;
            WLK_HIP(hipMemcpyAsync(s->logmel_alt, s->logmel + (size_t)shift * nm, (size_t)rows * nm * sizeof(float),
                                   hipMemcpyDeviceToDevice, s->stream));
            WLK_HIP(hipMemcpyAsync(s->frame_max_alt, s->frame_max + shift, (size_t)rows * sizeof(float),
                                   hipMemcpyDeviceToDevice, s->stream));
            std::swap(s->logmel, s->logmel_alt);
            std::swap(s->frame_max, s->frame_max_alt);
            s->mel_cached_samples -= n;
            s->mel_cached_frames = rows;
            s->mel_dirty_head = 2;
        } else {
            s->mel_cached_samples = 0;
        }
        return WLK_OK;
    });
}

int wlk_audio_clear(wlk_session* s) {
    if (!s) return fail(WLK_ERR_ARG, "session is NULL");
    s->audio_len = 0;
    s->encoded = false;
    s->mel_cached_samples = 0;
    return WLK_OK;
}

int wlk_audio_len(wlk_session* s, int* n) {
    if (!s || !n) return fail(WLK_ERR_ARG, "NULL argument");
    *n = s->audio_len;
    return WLK_OK;
}

// ---- encode ---------------------------------------------------------------------------------
// pre-LN MLP block: x += fc2(gelu(fc1(LN(x)))).  For decode steps (GEMV path) the LayerNorm is folded into the
// fc1 launch; for many rows it is its own kernel (folding it into the MFMA GEMM's tile staging was measured
// slower: every workgroup re-derives the row statistics)
static void transformer_mlp(const LaunchCtx& c, const LayerW& L, float* x, float* h, float* mlp, int rows, int d,
                            const char* t_ln, const char* t_fc1, const char* t_fc2) {
    GemmArgs g;
    g.lda = d; g.W = L.fc1w; g.bias = L.fc1b; g.C = mlp; g.ldc = 4 * d; g.M = rows; g.N = 4 * d; g.K = d;
    g.flags = kGemmGelu;
    if (gemv_applicable(rows, d)) {
        g.A = x; g.ln_gamma = L.ln2w; g.ln_beta = L.ln2b;
        launch_gemv(c, g, "dec_ln2_fc1");
    } else if (gemm_fuses_layernorm(rows, 4 * d, d)) {   // decoder prefill: see gemm_nt_f32_kwave16_kernel
        g.A = x; g.ln_gamma = L.ln2w; g.ln_beta = L.ln2b;
        launch_gemm(c, g, "dec_ln2_fc1");
    } else {
        launch_layernorm(c, x, d, L.ln2w, L.ln2b, h, d, rows, d, t_ln);
        g.A = h;
        launch_gemm(c, g, t_fc1);
    }
    GemmArgs g2;
    g2.A = mlp; g2.lda = 4 * d; g2.W = L.fc2w; g2.bias = L.fc2b; g2.C = x; g2.ldc = d; g2.M = rows; g2.N = d;
    g2.K = 4 * d; g2.flags = kGemmResidual; g2.R = x; g2.ldr = d;
    launch_linear(c, g2, t_fc2);
}

// One encode of every session of `group` (1..kMaxBatch sessions of one model) as ONE launch chain: the log-mel of each
// session (its own kernels: the frame counts differ), then every encoder operator once with grid.y = sessions - shared
// weights, per-session activation buffers through pointer tables.  A group of one is the plain per-session encode.
extern "C++" std::string wlk_encode_precheck(const wlk_session* s, const wlk_model* m) {
    if (!s) return "encode: session is NULL";
    if (s->m != m) return "encode: the session belongs to another model";
    const int N = s->audio_len;
    const int n_total = (N + kPadSamples) / kHop;
    int n_active = N > 0 ? (N + kNFft / 2 + kHop - 1) / kHop : 0;
    if (n_active > n_total) n_active = n_total;
    if (n_active > s->frame_cap) return "audio longer than the session's frame capacity";
    return std::string();
}

extern "C++" void wlk_encode_group(const std::vector<wlk_session*>& group, const LaunchCtx& c,
                                   std::vector<int>& content_out, bool mel_given) {
    const int B = (int)group.size();
    if (B < 1 || B > kMaxBatch) throw std::invalid_argument("encode: bad group size");
    wlk_model* m = group[0]->m;
    const wlk_dims& D = m->D;
    const int d = D.n_audio_state, T = D.n_audio_ctx;
    content_out.assign(B, 0);
    for (int i = 0; i < B; ++i) {
        wlk_session* s = group[i];
        if (s->m != m) throw std::invalid_argument("encode: sessions of different models in one group");
        if (mel_given) {           // the caller put a 3000-frame log-mel segment into mel_t (word_align.hip)
            content_out[i] = D.n_audio_ctx;
            continue;
        }
        const int N = s->audio_len;
        const int n_total = (N + kPadSamples) / kHop;               // stft frames minus the dropped last one
        int n_active = N > 0 ? (N + kNFft / 2 + kHop - 1) / kHop : 0;  // frames that see a non-zero sample
        if (n_active > n_total) n_active = n_total;
        if (n_active > s->frame_cap) throw std::length_error("audio longer than the session's frame capacity");
        content_out[i] = (n_total - kMelFrames) / 2;
        MelArgs ma;
        ma.audio = s->audio[s->audio_cur]; ma.n_samples = N;
        ma.window = m->w("mel.window"); ma.twiddle = m->twiddle; ma.filters = m->w("mel.filters");
        ma.filt_lo = m->filt_lo; ma.filt_hi = m->filt_hi; ma.n_mels = D.n_mels;
        ma.logmel = s->logmel; ma.frame_max = s->frame_max; ma.mel_t = s->mel_t;
        ma.n_active = n_active; ma.n_total = n_total;
        if (s->mel_incremental && s->mel_cached_samples > 0 && s->mel_cached_samples <= N && s->mel_cached_frames <= n_active) {
            // frame t reads samples [160 t - 200, 160 t + 200): untouched by the new samples iff 160 t + 199 < cached
            int first = s->mel_cached_samples >= kNFft / 2 ? (s->mel_cached_samples - kNFft / 2) / kHop + 1 : 0;
            first = std::min(first, s->mel_cached_frames);
            ma.first = first;
            ma.head = std::min(s->mel_dirty_head, first);
        }
        launch_mel(c, ma);
        s->mel_cached_samples = N;
        s->mel_cached_frames = n_active;
        s->mel_dirty_head = 0;
    }
    auto table = [&](auto in, auto out, auto res) {
        PtrTable z{};
        for (int i = 0; i < B; ++i) {
            z.in[i] = in(group[i]);
            z.out[i] = out(group[i]);
            z.res[i] = res(group[i]);
        }
        return z;
    };
    auto none = [](wlk_session*) -> const float* { return nullptr; };
    auto gemm = [&](GemmArgs g, const PtrTable& z, const char* tag) {
        const bool stackable = !gemm_takes_kwave(g.M, g.N, g.K) && (long)((g.N + 63) / 64) * ((g.M + 63) / 64) >= 64;
        if (stackable) {
            g.batch = B;
            g.z = z;
            launch_gemm(c, g, tag);
        } else {                     // narrow models (test shapes): the small-grid kernels take one session at a time
            for (int i = 0; i < B; ++i) {
                g.A = z.in[i]; g.C = z.out[i]; g.R = z.res[i];
                launch_gemm(c, g, tag);
            }
        }
    };
    {   // conv1 (k=3, pad=1) + GELU: rows of A overlap (lda = n_mels < K = 3 n_mels)
        GemmArgs g;
        g.lda = D.n_mels; g.W = m->w("enc.conv1.w"); g.bias = m->w("enc.conv1.b");
        g.ldc = d; g.M = kMelFrames; g.N = d; g.K = 3 * D.n_mels; g.flags = kGemmGelu;
        gemm(g, table([](wlk_session* s) { return (const float*)s->mel_t; }, [&](wlk_session* s) { return s->x1p + d; }, none),
             "enc_conv1");
    }
    {   // conv2 (k=3, stride=2, pad=1) + GELU + positional embedding
        const float* pos = m->w("enc.pos");
        GemmArgs g;
        g.lda = 2 * d; g.W = m->w("enc.conv2.w"); g.bias = m->w("enc.conv2.b");
        g.ldc = d; g.M = T; g.N = d; g.K = 3 * d; g.flags = kGemmGelu | kGemmResidual; g.ldr = d;
        gemm(g, table([](wlk_session* s) { return (const float*)s->x1p; }, [](wlk_session* s) { return s->ex; },
                      [&](wlk_session*) { return pos; }), "enc_conv2");
    }
    const float scale = std::pow((float)kHeadDim, -0.25f);
    const PtrTable z_ex_eh = table([](wlk_session* s) { return (const float*)s->ex; }, [](wlk_session* s) { return s->eh; }, none);
    const PtrTable z_eh_qkv = table([](wlk_session* s) { return (const float*)s->eh; }, [](wlk_session* s) { return s->eqkv; }, none);
    const PtrTable z_qkv_att = table([](wlk_session* s) { return (const float*)s->eqkv; }, [](wlk_session* s) { return s->eatt; },
                                     [](wlk_session* s) { return (const float*)s->esplit; });
    const PtrTable z_att_ex = table([](wlk_session* s) { return (const float*)s->eatt; }, [](wlk_session* s) { return s->ex; },
                                    [](wlk_session* s) { return (const float*)s->ex; });
    const PtrTable z_eh_mlp = table([](wlk_session* s) { return (const float*)s->eh; }, [](wlk_session* s) { return s->emlp; }, none);
    const PtrTable z_mlp_ex = table([](wlk_session* s) { return (const float*)s->emlp; }, [](wlk_session* s) { return s->ex; },
                                    [](wlk_session* s) { return (const float*)s->ex; });
    // Wide projections (qkv, fc1, cross-attention K|V) on the bf16 matrix cores at fp32 accuracy (gemm_x3.hip): their
    // LayerNorm writes the operand in the X3 format, the weights were packed by wlk_model_finalize.  Every session of a
    // group takes the same path (the choice depends on the model only).
    const PtrTable z_ex_eh3 = table([](wlk_session* s) { return (const float*)s->ex; },
                                    [](wlk_session* s) { return reinterpret_cast<float*>(s->eh3); }, none);
    const PtrTable z_eh3_qkv = table([](wlk_session* s) { return reinterpret_cast<const float*>(s->eh3); },
                                     [](wlk_session* s) { return s->eqkv; }, none);
    const PtrTable z_eh3_mlp = table([](wlk_session* s) { return reinterpret_cast<const float*>(s->eh3); },
                                     [](wlk_session* s) { return s->emlp; }, none);
    const PtrTable z_eh3_qkv3 = table([](wlk_session* s) { return reinterpret_cast<const float*>(s->eh3); },
                                      [](wlk_session* s) { return reinterpret_cast<float*>(s->eqkv3); }, none);
    const PtrTable z_qkv3_att = table([](wlk_session* s) { return reinterpret_cast<const float*>(s->eqkv3); },
                                      [](wlk_session* s) { return s->eatt; }, none);
    const PtrTable z_qkv3_eh3 = table([](wlk_session* s) { return reinterpret_cast<const float*>(s->eqkv3); },
                                      [](wlk_session* s) { return reinterpret_cast<float*>(s->eh3); }, none);
    const PtrTable z_eh3_ex_res = table([](wlk_session* s) { return reinterpret_cast<const float*>(s->eh3); }, [](wlk_session* s) { return s->ex; },
                                        [](wlk_session* s) { return (const float*)s->ex; });
    const PtrTable z_eh3_mlp3 = table([](wlk_session* s) { return reinterpret_cast<const float*>(s->eh3); },
                                      [](wlk_session* s) { return reinterpret_cast<float*>(s->emlp3); }, none);
    const PtrTable z_mlp3_ex_res = table([](wlk_session* s) { return reinterpret_cast<const float*>(s->emlp3); }, [](wlk_session* s) { return s->ex; },
                                         [](wlk_session* s) { return (const float*)s->ex; });
    const bool attn_x3 = enc_attention_x3_enabled() && d == D.n_audio_head * 64 && T >= 64 && (2 * d) % 128 == 0;
    auto gemm_x3 = [&](const GemmArgs& g, const unsigned short* w3, const PtrTable& z, const char* tag) {
        X3GemmArgs x;
        x.lda = g.lda; x.W3 = w3; x.bias = g.bias; x.ldc = g.ldc; x.ldr = g.ldr; x.M = g.M; x.N = g.N; x.K = g.K;
        x.flags = g.flags; x.scale = g.scale; x.scale_cols = g.scale_cols; x.scale_period = g.scale_period;
        x.batch = B;
        x.z = z;
        launch_gemm_x3(c, x, tag);
    };
    for (int i = 0; i < D.n_audio_layer; ++i) {
        const LayerW& L = m->enc_layers[i];
        GemmArgs g;
        g.lda = d; g.W = L.qkvw; g.bias = L.qkvb; g.ldc = 3 * d; g.M = T; g.N = 3 * d;
        g.K = d; g.flags = kGemmScaleCols; g.scale = scale; g.scale_cols = 2 * d;
        bool att_is_x3 = false;
        if (L.qkvw3 && attn_x3) {
            // q | k | v leave the projection as the X3 operand image of the bf16-MFMA attention (V transposed, keys in the
            // order a lane of that kernel holds its probabilities): no fp32 qkv is written at all
            launch_layernorm_x3_batched(c, z_ex_eh3, B, d, L.ln1w, L.ln1b, d, T, d, "enc_ln1");
            X3GemmArgs x;
            x.lda = d; x.W3 = L.qkvw3; x.bias = g.bias; x.M = T; x.N = 3 * d; x.K = d; x.flags = g.flags; x.scale = g.scale;
            x.scale_cols = g.scale_cols; x.batch = B; x.z = z_eh3_qkv3;
            x.x3_out = true; x.ldc3 = 2 * d; x.vt_col0 = 2 * d; x.vt_off = x3_attn_vt_off(T, d); x.vt_ld = x3_attn_vt_ld(T);
            launch_gemm_x3(c, x, "enc_qkv_x3");
            // (d >= 1024: the out projection runs on the X3 kernel, and the attention writes its rows in that format itself -
            // into the LayerNorm image, which is dead once the qkv projection has read it)
            att_is_x3 = L.outw3 != nullptr;
            launch_encoder_attention_x3(c, nullptr, 2L * d, x3_attn_vt_off(T, d), x3_attn_vt_ld(T), nullptr, d, T, d, D.n_audio_head,
                                        att_is_x3 ? &z_qkv3_eh3 : &z_qkv3_att, B, att_is_x3);
        } else {
            if (L.qkvw3) {
                launch_layernorm_x3_batched(c, z_ex_eh3, B, d, L.ln1w, L.ln1b, d, T, d, "enc_ln1");
                gemm_x3(g, L.qkvw3, z_eh3_qkv, "enc_qkv_x3");
            } else {
                launch_layernorm_batched(c, z_ex_eh, B, d, L.ln1w, L.ln1b, d, T, d, "enc_ln1");
                gemm(g, z_eh_qkv, "enc_qkv");
            }
            launch_encoder_attention_batched(c, z_qkv_att, B, T, d, D.n_audio_head);
        }
        GemmArgs o;
        o.lda = d; o.W = L.outw; o.bias = L.outb; o.ldc = d; o.M = T; o.N = d; o.K = d;
        o.flags = kGemmResidual; o.ldr = d;
        if (L.outw3) {
            // d >= 1024: the attention output goes through the X3 kernel too - packed into the (dead by now) LayerNorm image
            if (!att_is_x3)
                for (int i = 0; i < B; ++i) launch_x3_pack(c, group[i]->eatt, d, group[i]->eh3, d, T, d);
            gemm_x3(o, L.outw3, z_eh3_ex_res, "enc_out_x3");
        } else {
            gemm(o, z_att_ex, "enc_out");
        }
        GemmArgs f1;
        f1.lda = d; f1.W = L.fc1w; f1.bias = L.fc1b; f1.ldc = 4 * d; f1.M = T; f1.N = 4 * d; f1.K = d; f1.flags = kGemmGelu;
        const bool mlp_x3 = L.fc1w3 && L.fc2w3 && (4 * d) % 128 == 0;      // (128 = the X3 kernel's tile width: vt_col0 must be a multiple of it)      // fc1's GELU epilogue writes fc2's X3 operand itself
        if (mlp_x3) {
            launch_layernorm_x3_batched(c, z_ex_eh3, B, d, L.ln2w, L.ln2b, d, T, d, "enc_ln2");
            X3GemmArgs x;
            x.lda = d; x.W3 = L.fc1w3; x.bias = f1.bias; x.M = T; x.N = 4 * d; x.K = d; x.flags = kGemmGelu; x.batch = B; x.z = z_eh3_mlp3;
            x.x3_out = true; x.ldc3 = 4 * d; x.vt_col0 = 4 * d; x.vt_off = 0; x.vt_ld = (T + 31) / 32 * 32;
            launch_gemm_x3(c, x, "enc_fc1_x3");
        } else if (L.fc1w3) {
            launch_layernorm_x3_batched(c, z_ex_eh3, B, d, L.ln2w, L.ln2b, d, T, d, "enc_ln2");
            gemm_x3(f1, L.fc1w3, z_eh3_mlp, "enc_fc1_x3");
        } else {
            launch_layernorm_batched(c, z_ex_eh, B, d, L.ln2w, L.ln2b, d, T, d, "enc_ln2");
            gemm(f1, z_eh_mlp, "enc_fc1");
        }
        GemmArgs f2;
        f2.lda = 4 * d; f2.W = L.fc2w; f2.bias = L.fc2b; f2.ldc = d; f2.M = T; f2.N = d; f2.K = 4 * d;
        f2.flags = kGemmResidual; f2.ldr = d;
        if (L.fc2w3) {
            if (!mlp_x3)
                for (int i = 0; i < B; ++i) launch_x3_pack(c, group[i]->emlp, 4 * d, group[i]->emlp3, 4 * d, T, 4 * d);
            gemm_x3(f2, L.fc2w3, z_mlp3_ex_res, "enc_fc2_x3");
        } else {
            gemm(f2, z_mlp_ex, "enc_fc2");
        }
    }
    {   // final LayerNorm + cross-attention K (scaled) and V of every decoder layer in ONE GEMM: cross_kv is [T][L][k | v]
        GemmArgs g;
        g.lda = d; g.W = m->xkv_all_w; g.bias = m->xkv_all_b;
        g.ldc = (long)D.n_text_layer * 2 * d; g.M = T; g.N = D.n_text_layer * 2 * d; g.K = d;
        g.flags = kGemmScaleCols; g.scale = scale; g.scale_cols = d; g.scale_period = 2 * d;
        if (m->xkv_all_w3) {
            launch_layernorm_x3_batched(c, table([](wlk_session* s) { return (const float*)s->ex; },
                                                 [](wlk_session* s) { return reinterpret_cast<float*>(s->enc_out3); }, none),
                                        B, d, m->w("enc.ln_post.w"), m->w("enc.ln_post.b"), d, T, d, "enc_ln_post");
            gemm_x3(g, m->xkv_all_w3,
                    table([](wlk_session* s) { return reinterpret_cast<const float*>(s->enc_out3); }, [](wlk_session* s) { return s->cross_kv; }, none),
                    "dec_cross_kv_x3");
        } else {
            launch_layernorm_batched(c, table([](wlk_session* s) { return (const float*)s->ex; }, [](wlk_session* s) { return s->enc_out; }, none),
                                     B, d, m->w("enc.ln_post.w"), m->w("enc.ln_post.b"), d, T, d, "enc_ln_post");
            gemm(g, table([](wlk_session* s) { return (const float*)s->enc_out; }, [](wlk_session* s) { return s->cross_kv; }, none),
                 "dec_cross_kv");
        }
        for (int i = 0; i < B; ++i) group[i]->enc_out_is_x3 = m->xkv_all_w3 != nullptr;
    }
    for (int i = 0; i < B; ++i) {
        wlk_session* s = group[i];
        s->encoded = true;
        s->content_len = content_out[i];
        s->self_len = 0;
        s->n_steps = 0;
    }
}

int wlk_encode(wlk_session* s, int32_t* content_mel_len) {
    if (!s) return fail(WLK_ERR_ARG, "session is NULL");
    return guarded([&]() {
        WLK_HIP(hipSetDevice(s->m->device));
        int content = 0;
        struct Busy {   // the engine counts the sessions that are inside an encode or a decode loop right now
            wlk_session* s;
            explicit Busy(wlk_session* ss) : s(ss) { wlk_engine_loop_enter(s); }
            ~Busy() { wlk_engine_loop_exit(s); }
        } busy(s);
        if (wlk_engine_batches_encodes(s)) {
            // concurrent encodes of the sessions of this GPU are stacked into one launch chain (engine.hip)
            if (int rc = wlk_engine_encode(s, &content)) return rc;
        } else {
            std::vector<int> out;
            wlk_encode_group({s}, s->ctx(), out);
            content = out[0];
        }
        if (content_mel_len) *content_mel_len = content;
        return WLK_OK;
    });
}

// Key ranges of the decoder prefill's cross-attention (one workgroup per 32 prompt rows x head x range, partial softmax
// states folded by flash_merge_kernel in range order).  The count is part of the arithmetic (it groups the key sums): the
// same for a session alone and stacked with others.
int wlk_session::flash_splits() {
    static const int n = [] {
        const char* e = getenv("WLK_FLASH_SPLITS");
        const int v = e ? atoi(e) : 6;
        return v < 1 ? 1 : (v > kFlashSplitsMax ? kFlashSplitsMax : v);
    }();
    return n;
}

// ---- decode ---------------------------------------------------------------------------------
// Enqueue one decoder forward on the session stream.  Everything that changes from call to call
// (tokens, alignment-window row map, cache offset) is read from the pinned staging block through
// memcpy nodes / device scalars, so the single-token form of this sequence can be captured once
// into a hipGraph and replayed.
// The Infinity-Cache prefetcher beside a graph-replayed single-token step (decoder.hip: mall_prefetch_step_kernel).  OPT-IN
// (WLK_MALL_PREFETCH=1): measured a loss on large-v3 - step 1.49 -> 1.70 ms, 14.9 -> 14.2 audio-s/s, 510 / 510 decisions either
// way (profiles/r06k_mall_ab_large-v3.txt) - the second reader competes with the chain's own HBM reads instead of getting
// ahead of them.  WLK_MALL_LEAD = layers the prefetcher may run ahead of the chain's last mark (default 0).
static bool mall_prefetch_wanted(const wlk_model*) {
    static const bool on = [] {
        const char* e = getenv("WLK_MALL_PREFETCH");
        return e && e[0] == '1';
    }();
    return on;
}
static int mall_prefetch_lead() {
    static const int lead = [] {
        const char* e = getenv("WLK_MALL_LEAD");
        return e ? atoi(e) : 0;
    }();
    return lead;
}
static void mall_prefetch_prepare(wlk_session* s) {
    if (!mall_prefetch_wanted(s->m) || s->pf_stream) return;
    const wlk_model* m = s->m;
    const int L = m->D.n_text_layer, d = m->D.n_text_state;
    std::vector<TouchRanges> host((size_t)L);
    const unsigned dd = (unsigned)((size_t)d * d / 4);
    for (int i = 0; i < L; ++i) {
        const LayerW& W = m->dec_layers[i];
        host[(size_t)i] = TouchRanges{{W.qkvw, W.outw, W.xqw, W.xoutw, W.fc1w, W.fc2w}, {3 * dd, dd, dd, dd, 4 * dd, 4 * dd}};
    }
    WLK_HIP(hipStreamCreateWithFlags(&s->pf_stream, hipStreamNonBlocking));
    s->pf_table = dev_alloc<TouchRanges>((size_t)L);
    s->pf_progress = dev_alloc_zero<unsigned long long>(1, s->pf_stream);
    WLK_HIP(hipMemcpyAsync(s->pf_table, host.data(), host.size() * sizeof(TouchRanges), hipMemcpyHostToDevice, s->pf_stream));
    WLK_HIP(hipStreamSynchronize(s->pf_stream));
}

static void enqueue_decode(wlk_session* s, const LaunchCtx& c, int n_rows, int n_tok, bool first, int sot_index,
                           bool step_block = false, const AlignArgs* side_align = nullptr, int side_blocks = 0, int side_zf = 0) {
    wlk_model* m = s->m;
    const wlk_dims& D = m->D;
    const int ctx_len = D.n_text_ctx;
    const int R = n_rows * n_tok;
    const int d = D.n_text_state, T = D.n_audio_ctx, H = D.n_text_head, V = D.n_vocab;
    // one host->device copy: [tokens | alignment-window row | beam of row | cache offset]
    // (wlk_step_select: the embedding kernel reads the step's inputs from the host-coherent block itself)
    if (!step_block)
        WLK_HIP(hipMemcpyAsync(s->step_in, s->dec_stage, (size_t)(3 * s->max_rows + 1) * sizeof(int), hipMemcpyHostToDevice,
                               s->stream));
    // decode steps (<= 8 rows): LayerNorm and the KV-cache append are fused into the weight-streaming
    // GEMV launches; prefill keeps them as separate kernels in front of the MFMA GEMMs
    const bool fused = gemv_applicable(R, d) && n_tok == 1;

    if (step_block)
        launch_embed_step(c, s->step_host_dev, s->step_dev, s->tokens_dev, s->ring_row, s->beam_of_row, s->d_offset,
                          m->w_tok_emb, m->w_dec_pos, s->dx, d);
    else
        launch_embed(c, s->tokens_dev, m->w_tok_emb, m->w_dec_pos, s->dx, n_rows, n_tok, s->d_offset, d);
    const float scale = std::pow((float)kHeadDim, -0.25f);
    const size_t cache_layer = (size_t)s->beam * ctx_len * d;
    bool scores_dumped = false;   // prefill: raw alignment-head scores are waiting in the window rows
    const bool marks = s->pf_marks && s->pf_progress && step_block && fused;
    for (int i = 0; i < D.n_text_layer; ++i) {
        const LayerW& L = m->dec_layers[i];
        float* kc = s->kcache[s->kv_cur] + i * cache_layer;
        float* vc = s->vcache[s->kv_cur] + i * cache_layer;
        GemmArgs g;
        g.W = L.qkvw; g.bias = L.qkvb; g.C = s->dqkv; g.ldc = 3 * d; g.M = R; g.N = 3 * d;
        g.K = d; g.flags = kGemmScaleCols; g.scale = scale; g.scale_cols = 2 * d; g.lda = d;
        if (fused) {
            g.A = s->dx; g.ln_gamma = L.ln1w; g.ln_beta = L.ln1b;
            g.kcache = kc; g.vcache = vc; g.kv_pos = s->d_offset; g.kv_d = d; g.kv_ctx = ctx_len;
            launch_gemv(c, g, "dec_ln1_qkv_kv");
        } else if (gemm_fuses_layernorm(R, 3 * d, d)) {   // prompt rows: the 16 x 16 kernel normalises its A rows itself
            g.A = s->dx; g.ln_gamma = L.ln1w; g.ln_beta = L.ln1b;
            g.kcache = kc; g.vcache = vc; g.kv_pos = s->d_offset; g.kv_d = d; g.kv_ctx = ctx_len; g.kv_ntok = n_tok;
            launch_gemm(c, g, "dec_ln1_qkv");
        } else {
            launch_layernorm(c, s->dx, d, L.ln1w, L.ln1b, s->dh, d, R, d, "dec_ln1");
            g.A = s->dh;
            if (!gemv_applicable(R, d) && gemm_takes_kwave(R, 3 * d, d)) {   // k/v also land in the caches
                g.kcache = kc; g.vcache = vc; g.kv_pos = s->d_offset; g.kv_d = d; g.kv_ctx = ctx_len; g.kv_ntok = n_tok;
                launch_gemm(c, g, "dec_qkv");
            } else {
                launch_linear(c, g, "dec_qkv");
                launch_kv_append(c, s->dqkv, kc, vc, n_rows, n_tok, s->d_offset, d, ctx_len);
            }
        }
        launch_decoder_self_attention(c, s->dqkv, kc, vc, s->datt, n_rows, n_tok, s->d_offset, d, H, ctx_len,
                                      marks ? ProgressMark{&s->step_dev->seq, s->pf_progress, i} : ProgressMark{});
        GemmArgs o;
        o.A = s->datt; o.lda = d; o.W = L.outw; o.bias = L.outb; o.C = s->dx; o.ldc = d; o.M = R; o.N = d; o.K = d;
        o.flags = kGemmResidual; o.R = s->dx; o.ldr = d;
        launch_linear(c, o, "dec_out");

        GemmArgs q;
        q.lda = d; q.W = L.xqw; q.bias = L.xqb; q.C = s->dq; q.ldc = d; q.M = R; q.N = d; q.K = d;
        q.flags = kGemmScaleCols; q.scale = scale; q.scale_cols = d;
        // decode steps: the split cross-attention kernel derives its head's query values itself (same arithmetic) - one
        // launch less per layer
        // (the fold leaves s->dq unwritten, so it must never meet the flash branch below, which reads it: that branch is
        // taken for align_raw_scores - excluded here, not merely "never fed with R == 1 today")
        const bool fold_xq = fused && R == 1 && !s->debug && !s->align_raw_scores && cross_split_folds_query(d);   // one row: saves a launch; several rows would each re-stream Wq
        if (fold_xq) {
        } else if (fused) {
            q.A = s->dx; q.ln_gamma = L.lnxw; q.ln_beta = L.lnxb;
            launch_gemv(c, q, "dec_lnx_xq");
        } else if (gemm_fuses_layernorm(R, d, d)) {
            q.A = s->dx; q.ln_gamma = L.lnxw; q.ln_beta = L.lnxb;
            launch_gemm(c, q, "dec_lnx_xq");
        } else {
            launch_layernorm(c, s->dx, d, L.lnxw, L.lnxb, s->dh, d, R, d, "dec_lnx");
            q.A = s->dh;
            launch_linear(c, q, "dec_xq");
        }
        const int* ranks_l = m->n_align > 0 ? m->head_rank + (size_t)i * H : nullptr;
        bool merged_xout = false;
        GemmArgs xo_mg;   // carries the split-form operand description to the out projection when merged_xout
        if ((R > 8 || s->align_raw_scores) && !s->debug) {
            // prefill: MFMA flash kernel shares every K/V tile between 32 query rows
            if (fold_xq) throw std::logic_error("decode: the folded query projection left no q for the flash kernel");
            FlashArgs fa;
            fa.q = s->dq; fa.ldq = d;
            fa.k = s->cross_kv + (size_t)i * 2 * d; fa.v = fa.k + d; fa.ldkv = (long)D.n_text_layer * 2 * d;
            fa.out = s->datt; fa.ldo = d; fa.Tq = R; fa.Tk = T; fa.n_head = H;
            fa.head_rank = ranks_l; fa.ring = s->ring; fa.ring_row = s->ring_row;
            fa.beam_of_row = s->beam_of_row; fa.ring_rows = s->ring_rows; fa.n_beam = s->beam;
            if (((R + 31) / 32) * H < 256) {   // few query tiles: spread the 1500 keys over more workgroups
                fa.k_splits = wlk_session::flash_splits();
                fa.part_o = s->fsplit;
                fa.part_m = fa.part_o + (size_t)s->max_rows * H * fa.k_splits * 64;
                fa.part_l = fa.part_m + (size_t)s->max_rows * H * fa.k_splits;
            }
            launch_prefill_cross_attention(c, fa);
            scores_dumped = true;   // softmaxed in place behind the last layer, all alignment heads in one launch
        } else {
            CrossAttnArgs ca;
            ca.q = s->dq;
            ca.k = s->cross_kv + (size_t)i * 2 * d;
            ca.v = ca.k + d;
            ca.ldkv = (long)D.n_text_layer * 2 * d;
            ca.out = s->datt;
            ca.rows = R; ca.d = d; ca.n_head = H; ca.T = T;
            ca.head_rank = ranks_l;
            ca.ring = s->ring;
            ca.ring_row = s->ring_row;
            ca.beam_of_row = s->beam_of_row;
            ca.ring_rows = s->ring_rows;
            ca.n_beam = s->beam;
            ca.qk_debug = s->debug ? s->qk_debug + (size_t)i * s->max_rows * H * T : nullptr;
            if (fold_xq) {
                ca.xq_x = s->dx; ca.xq_w = L.xqw; ca.xq_b = L.xqb; ca.xq_gamma = L.lnxw; ca.xq_beta = L.lnxb; ca.xq_scale = scale;
            }
            if (R <= 8 && !s->debug) {
                float* sc = s->xsplit;
                float* pm = sc + (size_t)8 * H * T;
                float* pl = pm + (size_t)8 * H * 8;
                float* po = pl + (size_t)8 * H * 8;
                // beam-1 steps: the merge of the key splits is the A-operand load of the out projection below
                merged_xout = fused && R == 1 && gemv1_folds_merge(d);
                launch_decoder_cross_attention_split(c, ca, sc, pm, pl, po, !merged_xout);
                if (merged_xout) {
                    xo_mg.mg_pm = pm; xo_mg.mg_pl = pl; xo_mg.mg_po = po; xo_mg.mg_scores = sc;
                    xo_mg.mg_head_rank = ranks_l; xo_mg.mg_ring = s->ring; xo_mg.mg_ring_row = s->ring_row;
                    xo_mg.mg_side_heads = m->layer_heads + (size_t)i * H;
                    xo_mg.mg_beam_of_row = s->beam_of_row; xo_mg.mg_heads = H; xo_mg.mg_T = T;
                    xo_mg.mg_ring_rows = s->ring_rows; xo_mg.mg_n_beam = s->beam;
                    xo_mg.mg_side_blocks = ranks_l ? m->layer_rank_count[i] : 0;
                }
            } else {
                launch_decoder_cross_attention(c, ca);
            }
        }
        GemmArgs xo = xo_mg;
        xo.A = s->datt; xo.lda = d; xo.W = L.xoutw; xo.bias = L.xoutb; xo.C = s->dx; xo.ldc = d; xo.M = R; xo.N = d;
        xo.K = d; xo.flags = kGemmResidual; xo.R = s->dx; xo.ldr = d;
        launch_linear(c, xo, "dec_xout");
        transformer_mlp(c, L, s->dx, s->dh, s->dmlp, R, d, "dec_ln2", "dec_fc1", "dec_fc2");
    }
    if (scores_dumped && m->n_align > 0 && !s->align_raw_scores)     // find_alignment normalises the raw scores itself
        launch_ring_softmax(c, s->ring, s->ring_row, s->beam_of_row, m->all_ranks, m->n_align, R, s->ring_rows, s->beam, T);
    // final LayerNorm + vocabulary projection only for the rows the policy reads
    GemmArgs lg;
    lg.lda = d; lg.W = m->w_tok_emb; lg.C = s->logits_last; lg.ldc = V; lg.M = n_rows; lg.N = V; lg.K = d;
    if (side_align && !fused) throw std::logic_error("decode: side workgroups need the fused step");
    if (fused) {   // n_tok == 1: the last row of beam b is row b
        lg.A = s->dx; lg.ln_gamma = m->w_ln_w; lg.ln_beta = m->w_ln_b;
        // graph-replayed steps: the alignment window is complete behind the last layer's cross-attention, so its z-score
        // rides in this launch (gemv_f32_kernel's side workgroups) instead of in front of the step's last two
        lg.side_align = side_align; lg.side_blocks = side_blocks; lg.side_zf = side_zf;
        launch_gemv(c, lg, "dec_lnf_logits");
    } else if (first && n_rows == 1 && gemv_applicable(2, d) && getenv("WLK_NO_PREFILL_MERGE") == nullptr) {
        // beam-1 prefill: the last row and the sot row (no-speech probability) normalised by one launch (a negative
        // row stride walks from the last row back to the sot row) and projected by one M = 2 pass over the 51864 x d
        // weights; per-row arithmetic of both kernels does not depend on the row count
        launch_layernorm(c, s->dx + (size_t)(n_tok - 1) * d, (long)(sot_index - (n_tok - 1)) * d, m->w_ln_w, m->w_ln_b,
                         s->hsel, d, 2, d, "dec_ln_f");
        lg.A = s->hsel; lg.M = 2;   // C row 1 = logits_sot (same allocation, V floats further)
        launch_linear(c, lg, "dec_logits");
        return;
    } else {
        launch_layernorm(c, s->dx + (size_t)(n_tok - 1) * d, (long)n_tok * d, m->w_ln_w, m->w_ln_b, s->hsel, d,
                         n_rows, d, "dec_ln_f");
        lg.A = s->hsel;
        launch_linear(c, lg, "dec_logits");
    }
    if (first) {
        float* hs = s->hsel + (size_t)n_rows * d;
        launch_layernorm(c, s->dx + (size_t)sot_index * d, (long)n_tok * d, m->w_ln_w, m->w_ln_b, hs, d, n_rows, d,
                         "dec_ln_f");
        GemmArgs ls = lg;
        ls.ln_gamma = ls.ln_beta = nullptr;
        ls.A = hs; ls.C = s->logits_sot;
        launch_linear(c, ls, "dec_logits");
    }
}

// ---- stacked prefills (see internal.h) -----------------------------------------------------------------------------------
extern "C++" void wlk_prefill_ws_alloc(const wlk_model* m, wlk_prefill_ws& ws, int max_sessions, int session_rows) {
    const wlk_dims& D = m->D;
    const size_t d = D.n_text_state, V = D.n_vocab, H = D.n_text_head;
    ws.cap_session_rows = session_rows;
    ws.cap_rows = max_sessions * ((session_rows + 31) / 32 * 32);
    const size_t R = ws.cap_rows;
    ws.dx = dev_alloc<float>(R * d);
    ws.dh = dev_alloc<float>(R * d);
    ws.dqkv = dev_alloc<float>(R * 3 * d);
    ws.datt = dev_alloc<float>(R * d);
    ws.dq = dev_alloc<float>(R * d);
    ws.dmlp = dev_alloc<float>(R * 4 * d);
    ws.hsel = dev_alloc<float>((size_t)2 * max_sessions * d);
    ws.logits = dev_alloc<float>((size_t)8 * V);
    ws.part = dev_alloc<float>(flash_split_scratch_floats((int)R, (int)H, wlk_session::kFlashSplitsMax));
    ws.rows_dev = reinterpret_cast<StepRow*>(dev_alloc<char>(R * sizeof(StepRow)));
    ws.tiles_dev = reinterpret_cast<StepRow*>(dev_alloc<char>(R / 32 * sizeof(StepRow)));
    ws.ring_row_dev = dev_alloc<int>(R);
    ws.zeros_dev = dev_alloc<int>(R);
    memset_sync(ws.zeros_dev, 0, R * sizeof(int));
    ws.pinned_bytes = R * sizeof(StepRow) + R / 32 * sizeof(StepRow) + R * sizeof(int);
    WLK_HIP(hipHostMalloc(reinterpret_cast<void**>(&ws.pinned), ws.pinned_bytes, hipHostMallocDefault));
}

extern "C++" void wlk_prefill_ws_free(wlk_prefill_ws& ws) {
    void* dev[] = {ws.dx, ws.dh, ws.dqkv, ws.datt, ws.dq, ws.dmlp, ws.hsel, ws.logits, ws.part, ws.rows_dev, ws.tiles_dev,
                   ws.ring_row_dev, ws.zeros_dev};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    if (ws.pinned) (void)hipHostFree(ws.pinned);
    ws = wlk_prefill_ws{};
}

extern "C++" std::string wlk_prefill_precheck(const wlk_prefill_item& it, const wlk_prefill_ws& ws) {
    const wlk_session* s = it.s;
    if (!s || !it.tokens) return "prefill: NULL argument";
    const wlk_dims& D = s->m->D;
    const int P = it.n_tok, d = D.n_text_state;
    if (!s->encoded) return "wlk_decode before wlk_encode";
    if (s->beam != 1 || s->debug || s->prof_on || s->align_raw_scores) return "not a plain beam-1 session";
    if (it.sot_index < 0 || it.sot_index >= P) return "sot_index out of range";
    if (P > D.n_text_ctx) return "text context exceeded";
    for (int i = 0; i < P; ++i)
        if (it.tokens[i] < 0 || it.tokens[i] >= D.n_vocab) return "token id out of range";
    // the stack runs every GEMM on the k-wave kernel and the cross-attention with key splits: only prompts whose own
    // prefill would make exactly these choices ride in it (bit-identical results either way)
    if (P <= 8 || P > ws.cap_session_rows) return "prompt length outside the stacked range";
    if (!gemm_takes_kwave(P, 3 * d, d) || !gemm_takes_kwave(P, d, d) || !gemm_takes_kwave(P, 4 * d, d) ||
        !gemm_takes_kwave(P, d, 4 * d))
        return "prompt too long for the k-wave GEMMs";
    if (((P + 31) / 32) * D.n_text_head >= 256) return "prompt too long for the key-split cross-attention";
    if (!gemv_applicable(8, d)) return "model too wide for the stacked vocabulary projection";
    return std::string();
}

extern "C++" void wlk_prefill_group(const std::vector<wlk_prefill_item*>& items, const LaunchCtx& c, wlk_prefill_ws& ws) {
    if (items.empty()) return;
    wlk_model* m = items[0]->s->m;
    const wlk_dims& D = m->D;
    const int d = D.n_text_state, T = D.n_audio_ctx, H = D.n_text_head, V = D.n_vocab, L = D.n_text_layer;
    const int B = (int)items.size();
    // ---- tables: one StepRow per stacked row; a session's rows are padded to whole 32-row tiles with copies of its
    // row 0 (same token, position, destinations: every write of a copy repeats row 0's write with the same value)
    StepRow* rows_h = reinterpret_cast<StepRow*>(ws.pinned);
    std::vector<int> row0(B), padded(B);
    int R = 0;
    for (int i = 0; i < B; ++i) {
        row0[i] = R;
        padded[i] = (items[i]->n_tok + 31) / 32 * 32;
        R += padded[i];
    }
    if (R > ws.cap_rows) throw std::length_error("stacked prefill: more rows than the workspace holds");
    StepRow* tiles_h = rows_h + R;
    int* ring_row_h = reinterpret_cast<int*>(tiles_h + R / 32);
    for (int i = 0; i < B; ++i) {
        wlk_session* s = items[i]->s;
        for (int p = 0; p < padded[i]; ++p) {
            const int src = p < items[i]->n_tok ? p : 0;
            StepRow r{};
            r.kcache = s->kcache[s->kv_cur];
            r.vcache = s->vcache[s->kv_cur];
            r.cross_kv = s->cross_kv;
            r.ring = s->ring;
            r.token = (int)items[i]->tokens[src];
            r.offset = src;
            r.ring_row = src;
            rows_h[row0[i] + p] = r;
            ring_row_h[row0[i] + p] = src;
            if (p % 32 == 0) tiles_h[(row0[i] + p) / 32] = r;
        }
    }
    WLK_HIP(hipMemcpyAsync(ws.rows_dev, rows_h, (size_t)R * sizeof(StepRow), hipMemcpyHostToDevice, c.stream));
    WLK_HIP(hipMemcpyAsync(ws.tiles_dev, tiles_h, (size_t)(R / 32) * sizeof(StepRow), hipMemcpyHostToDevice, c.stream));
    WLK_HIP(hipMemcpyAsync(ws.ring_row_dev, ring_row_h, (size_t)R * sizeof(int), hipMemcpyHostToDevice, c.stream));

    // ---- the chain: enqueue_decode's prefill branch (rows > 8, not fused) with stacked rows
    launch_embed_rows(c, ws.rows_dev, m->w_tok_emb, m->w_dec_pos, ws.dx, R, d);
    const float scale = std::pow((float)kHeadDim, -0.25f);
    const size_t cache_layer = (size_t)D.n_text_ctx * d;       // beam 1
    auto linear = [&](const float* A, long lda, const float* W, const float* bias, float* C, long ldc, int N, int K, int flags,
                      const float* Rs, float sc, int sc_cols, const char* tag) {
        GemmArgs g;
        g.A = A; g.lda = lda; g.W = W; g.bias = bias; g.C = C; g.ldc = ldc; g.M = R; g.N = N; g.K = K;
        g.flags = flags; g.R = Rs; g.ldr = ldc; g.scale = sc; g.scale_cols = sc_cols;
        g.force_kwave = true;
        launch_gemm(c, g, tag);
    };
    // LayerNorm + projection: one launch where the 16 x 16 kernel takes it (the same rule as a session's own prefill, so
    // that a stacked row gets its solo arithmetic - which it would either way: the fused statistics are the kernel's)
    auto ln_linear = [&](const float* gamma, const float* beta, const float* W, const float* bias, float* C, long ldc, int N,
                         int flags, float sc, int sc_cols, const char* t_ln, const char* tag) {
        if (gemm_fuses_layernorm(R, N, d)) {
            GemmArgs g;
            g.A = ws.dx; g.lda = d; g.W = W; g.bias = bias; g.C = C; g.ldc = ldc; g.M = R; g.N = N; g.K = d;
            g.flags = flags; g.scale = sc; g.scale_cols = sc_cols; g.ln_gamma = gamma; g.ln_beta = beta;
            launch_gemm(c, g, tag);
        } else {
            launch_layernorm(c, ws.dx, d, gamma, beta, ws.dh, d, R, d, t_ln);
            linear(ws.dh, d, W, bias, C, ldc, N, d, flags, nullptr, sc, sc_cols, tag);
        }
    };
    for (int i = 0; i < L; ++i) {
        const LayerW& W = m->dec_layers[i];
        ln_linear(W.ln1w, W.ln1b, W.qkvw, W.qkvb, ws.dqkv, 3 * d, 3 * d, kGemmScaleCols, scale, 2 * d, "dec_ln1", "dec_qkv");
        launch_kv_append_rows(c, ws.dqkv, ws.rows_dev, (long)(i * cache_layer), R, d);
        launch_decoder_self_attention_rows(c, ws.dqkv, ws.rows_dev, (long)(i * cache_layer), ws.datt, R, d, H, D.n_text_ctx);
        linear(ws.datt, d, W.outw, W.outb, ws.dx, d, d, d, kGemmResidual, ws.dx, 1.f, 0, "dec_out");
        ln_linear(W.lnxw, W.lnxb, W.xqw, W.xqb, ws.dq, d, d, kGemmScaleCols, scale, d, "dec_lnx", "dec_xq");
        FlashArgs fa;
        fa.q = ws.dq; fa.ldq = d;
        fa.k = nullptr; fa.v = nullptr; fa.ldkv = (long)L * 2 * d;
        fa.tile_rows = ws.tiles_dev; fa.tile_kv_off = (long)i * 2 * d; fa.tile_v_off = d;
        fa.out = ws.datt; fa.ldo = d; fa.Tq = R; fa.Tk = T; fa.n_head = H;
        fa.head_rank = m->n_align > 0 ? m->head_rank + (size_t)i * H : nullptr;
        fa.ring = nullptr; fa.ring_row = ws.ring_row_dev; fa.beam_of_row = ws.zeros_dev;
        fa.ring_rows = items[0]->s->ring_rows; fa.n_beam = 1;
        fa.k_splits = wlk_session::flash_splits();
        fa.part_o = ws.part;
        fa.part_m = fa.part_o + (size_t)R * H * fa.k_splits * 64;
        fa.part_l = fa.part_m + (size_t)R * H * fa.k_splits;
        launch_prefill_cross_attention(c, fa);
        linear(ws.datt, d, W.xoutw, W.xoutb, ws.dx, d, d, d, kGemmResidual, ws.dx, 1.f, 0, "dec_xout");
        ln_linear(W.ln2w, W.ln2b, W.fc1w, W.fc1b, ws.dmlp, 4 * d, 4 * d, kGemmGelu, 1.f, 0, "dec_ln2", "dec_fc1");
        linear(ws.dmlp, 4 * d, W.fc2w, W.fc2b, ws.dx, d, d, 4 * d, kGemmResidual, ws.dx, 1.f, 0, "dec_fc2");
    }
    // ---- per session: the alignment heads' raw scores softmaxed in place; final LayerNorm of the last and the sot row
    for (int i = 0; i < B; ++i) {
        wlk_session* s = items[i]->s;
        const int P = items[i]->n_tok;
        if (m->n_align > 0)
            launch_ring_softmax(c, s->ring, ws.ring_row_dev + row0[i], ws.zeros_dev, m->all_ranks, m->n_align, P, s->ring_rows, 1, T);
        launch_layernorm(c, ws.dx + (size_t)(row0[i] + P - 1) * d, (long)(items[i]->sot_index - (P - 1)) * d, m->w_ln_w, m->w_ln_b,
                         ws.hsel + (size_t)2 * i * d, d, 2, d, "dec_ln_f");
    }
    // ---- vocabulary projection: [last, sot] rows of up to four sessions per pass over the embedding (the <= 8-row
    // weight-streaming kernel, whose per-row arithmetic does not depend on the row count), then out to the sessions
    for (int lo = 0; lo < B; lo += 4) {
        const int nb = std::min(4, B - lo);
        GemmArgs lg;
        lg.A = ws.hsel + (size_t)2 * lo * d; lg.lda = d; lg.W = m->w_tok_emb; lg.C = ws.logits; lg.ldc = V; lg.M = 2 * nb; lg.N = V;
        lg.K = d;
        launch_gemv(c, lg, "dec_logits");
        for (int i = 0; i < nb; ++i)
            WLK_HIP(hipMemcpyAsync(items[lo + i]->s->logits_last, ws.logits + (size_t)2 * i * V, (size_t)2 * V * sizeof(float),
                                   hipMemcpyDeviceToDevice, c.stream));
    }
    for (int i = 0; i < B; ++i) {
        wlk_session* s = items[i]->s;
        s->self_len = items[i]->n_tok;
        s->n_steps = 1;
        s->have_sot = true;
        s->prefill_rows = items[i]->n_tok;
        s->last_rows = 1;
        s->last_ntok = items[i]->n_tok;
    }
}

int wlk_decode(wlk_session* s, const int64_t* tokens, int n_rows, int n_tok, int first, int sot_index) {
    if (!s || !tokens) return fail(WLK_ERR_ARG, "NULL argument");
    if (!s->encoded) return fail(WLK_ERR_STATE, "wlk_decode before wlk_encode");
    if (n_rows != s->beam) return fail(WLK_ERR_ARG, "n_rows must equal the session's beam size");
    if (n_tok < 1) return fail(WLK_ERR_ARG, "n_tok must be >= 1");
    if (!first && n_tok != 1) return fail(WLK_ERR_ARG, "only one token per row may be fed after the first call");
    if (!first && s->n_steps == 0) return fail(WLK_ERR_STATE, "first decode of an infer must set first=1");
    if (first && (sot_index < 0 || sot_index >= n_tok)) return fail(WLK_ERR_ARG, "sot_index out of range");
    return guarded([&]() {
        wlk_model* m = s->m;
        const wlk_dims& D = m->D;
        WLK_HIP(hipSetDevice(m->device));
        const LaunchCtx c = s->ctx();
        if (first) {
            s->self_len = 0;
            s->n_steps = 0;
        }
        const int offset = s->self_len;
        const int ctx_len = D.n_text_ctx;
        if (offset + n_tok > ctx_len) return fail(WLK_ERR_CAPACITY, "text context exceeded");
        const int R = n_rows * n_tok;
        if ((size_t)(s->max_rows * 3 + 4) * sizeof(int) > 65536) return fail(WLK_ERR_CAPACITY, "too many rows");

        // staging block (pinned): tokens as int32 | alignment-window row of each query row | beam of row | offset
        // the previous call's copy out of this block is done (an event behind it, not a drained stream: the prefill
        // of an infer is enqueued while the encoder is still running)
        if (s->dec_stage_used) WLK_HIP(hipEventSynchronize(s->dec_stage_ev));
        if (drain_staging()) WLK_HIP(hipStreamSynchronize(s->stream));
        int* stage = static_cast<int*>(s->dec_stage);
        const int slot_row = first ? 0 : ctx_len + ((s->n_steps - 1) % kAlignWindow);
        const int MR = s->max_rows;
        for (int b = 0; b < n_rows; ++b)
            for (int p = 0; p < n_tok; ++p) {
                const int64_t t = tokens[(size_t)b * n_tok + p];
                if (t < 0 || t >= D.n_vocab) return fail(WLK_ERR_ARG, "token id out of range");
                stage[b * n_tok + p] = (int)t;
                stage[MR + b * n_tok + p] = first ? p : slot_row;
                stage[2 * MR + b * n_tok + p] = b;
            }
        stage[3 * MR] = offset;

        const bool graphable = !first && n_tok == 1 && gemv_applicable(R, D.n_text_state) && !s->debug &&
                               !s->prof_on && s->use_graph;
        if (graphable) {
            hipGraphExec_t& exec = s->step_exec[s->kv_cur];
            if (!exec) {
                hipGraph_t graph = nullptr;
                WLK_HIP(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
                try {
                    enqueue_decode(s, c, n_rows, n_tok, false, sot_index);
                } catch (...) {
                    (void)hipStreamEndCapture(s->stream, &graph);
                    if (graph) (void)hipGraphDestroy(graph);
                    throw;
                }
                WLK_HIP(hipStreamEndCapture(s->stream, &graph));
                WLK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                (void)hipGraphDestroy(graph);
            }
            WLK_HIP(hipGraphLaunch(exec, s->stream));
        } else {
            enqueue_decode(s, c, n_rows, n_tok, first != 0, sot_index);
            // timing probe only (scripts/eight_stream_probe.py): the prefill chain enqueued again - same inputs, same
            // buffers, same results - to measure what one prefill chain costs under load (the bound on what
            // stacking the sessions' prefills could save)
            static const int prefill_repeat = [] { const char* e = getenv("WLK_PROBE_PREFILL_REPEAT"); return e ? atoi(e) : 1; }();
            if (first)
                for (int rep = 1; rep < prefill_repeat; ++rep) enqueue_decode(s, c, n_rows, n_tok, true, sot_index);
        }
        WLK_HIP(hipEventRecord(s->dec_stage_ev, s->stream));
        s->dec_stage_used = true;
        s->have_sot = false;
        if (first) {
            s->have_sot = true;
            s->prefill_rows = n_tok;
        }
        s->self_len += n_tok;
        s->n_steps += 1;
        s->last_rows = n_rows;
        s->last_ntok = n_tok;
        return WLK_OK;
    });
}

int wlk_no_speech_prob(wlk_session* s, int no_speech_token, float* probs_host) {
    if (!s || !probs_host) return fail(WLK_ERR_ARG, "NULL argument");
    if (!s->have_sot) return fail(WLK_ERR_STATE, "no sot-row logits: call after the first decode of an infer");
    if (no_speech_token < 0 || no_speech_token >= s->m->D.n_vocab) return fail(WLK_ERR_ARG, "token out of range");
    return guarded([&]() {
        WLK_HIP(hipSetDevice(s->m->device));
        const LaunchCtx c = s->ctx();
        launch_token_prob(c, s->logits_sot, s->m->D.n_vocab, s->beam, no_speech_token, s->probs);
        WLK_HIP(hipMemcpyAsync(s->pinned, s->probs, s->beam * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        WLK_HIP(hipStreamSynchronize(s->stream));
        std::memcpy(probs_host, s->pinned, s->beam * sizeof(float));
        return WLK_OK;
    });
}

// ---- whisper's batch-decoder rules on the device (transcribe.py's greedy T = 0 loop; select.hip: rules_pick_kernel) ----------
int wlk_rules_set(wlk_session* s, const int32_t* suppressed, int n_suppressed, const int32_t* blank, int n_blank) {
    if (!s || n_suppressed < 0 || n_blank < 0 || (n_suppressed > 0 && !suppressed) || (n_blank > 0 && !blank))
        return fail(WLK_ERR_ARG, "invalid rule lists");
    const int V = s->m->D.n_vocab;
    for (int i = 0; i < n_suppressed; ++i)
        if (suppressed[i] < 0 || suppressed[i] >= V) return fail(WLK_ERR_ARG, "suppressed token out of range");
    for (int i = 0; i < n_blank; ++i)
        if (blank[i] < 0 || blank[i] >= V) return fail(WLK_ERR_ARG, "blank token out of range");
    return guarded([&]() {
        WLK_HIP(hipSetDevice(s->m->device));
        std::vector<unsigned char> host((size_t)V, 0);
        for (int i = 0; i < n_suppressed; ++i) host[(size_t)suppressed[i]] |= 1;
        for (int i = 0; i < n_blank; ++i) host[(size_t)blank[i]] |= 2;
        if (!s->rules_mask) {
            s->rules_mask = dev_alloc<unsigned char>((size_t)V);
            s->pick_out = dev_alloc<int>(2);
        }
        WLK_HIP(hipStreamSynchronize(s->stream));           // a pick of the previous rule set may still be reading the mask
        copy_sync(s->rules_mask, host.data(), host.size(), hipMemcpyHostToDevice);
        return WLK_OK;
    });
}

int wlk_pick_greedy(wlk_session* s, const wlk_pick_params* p, int32_t* token_host, float* logprob_host) {
    if (!s || !p || !token_host || !logprob_host) return fail(WLK_ERR_ARG, "NULL argument");
    if (s->beam != 1) return fail(WLK_ERR_ARG, "wlk_pick_greedy serves sessions of one row");
    if (s->n_steps == 0) return fail(WLK_ERR_STATE, "wlk_pick_greedy before wlk_decode");
    if (!s->rules_mask) return fail(WLK_ERR_STATE, "wlk_pick_greedy before wlk_rules_set");
    const int V = s->m->D.n_vocab;
    if (p->timestamp_begin < 0 || p->timestamp_begin > V || p->eot < 0 || p->eot >= V || p->ts_mode < 0 || p->ts_mode > 2)
        return fail(WLK_ERR_ARG, "rule parameters out of range");
    return guarded([&]() {
        WLK_HIP(hipSetDevice(s->m->device));
        const LaunchCtx c = s->ctx();
        const PickRules r{p->first_step, p->without_timestamps, p->timestamp_begin, p->eot, p->no_timestamps, p->ts_mode,
                          p->ts_bound, p->max_initial};
        launch_rules_pick(c, s->logits_last, V, s->rules_mask, r, s->pick_out, reinterpret_cast<float*>(s->pick_out + 1));
        WLK_HIP(hipMemcpyAsync(s->pinned, s->pick_out, 8, hipMemcpyDeviceToHost, s->stream));
        WLK_HIP(hipStreamSynchronize(s->stream));
        std::memcpy(token_host, s->pinned, 4);
        std::memcpy(logprob_host, static_cast<char*>(s->pinned) + 4, 4);
        return WLK_OK;
    });
}

// wlk_select, optionally with the no-speech probability of the sot row in the same read-back (first step of the
// library's decode loop: one synchronisation instead of two)
static int select_impl(wlk_session* s, const int32_t* adj_row, const int32_t* adj_ids, const float* adj_deltas, int n_adj,
                       int k, int content_mel_len, float* top_logprobs_host, int32_t* top_ids_host, int32_t* frames_host,
                       int no_speech_token, float* no_speech_host) {
    if (!s || !top_logprobs_host || !top_ids_host || !frames_host) return fail(WLK_ERR_ARG, "NULL argument");
    if (s->n_steps == 0) return fail(WLK_ERR_STATE, "wlk_select before wlk_decode");
    if (n_adj < 0 || n_adj > wlk_session::kAdjCap) return fail(WLK_ERR_ARG, "too many logit adjustments");
    if (n_adj > 0 && (!adj_row || !adj_ids || !adj_deltas)) return fail(WLK_ERR_ARG, "NULL adjustment arrays");
    if (k < 1 || k > 8) return fail(WLK_ERR_ARG, "k must be in [1, 8]");
    if (content_mel_len < 0) return fail(WLK_ERR_ARG, "content_mel_len out of range");
    // a buffer longer than 30 s yields content_mel_len > 1500; the reference only clips the attention slice
    // (attn[..., :content_mel_len] over 1500 columns, simul_whisper.py:432) and keeps decoding, the host keeps the
    // unclipped value for its frame-threshold test
    if (content_mel_len > s->m->D.n_audio_ctx) content_mel_len = s->m->D.n_audio_ctx;
    return guarded([&]() {
        wlk_model* m = s->m;
        const wlk_dims& D = m->D;
        WLK_HIP(hipSetDevice(m->device));
        const LaunchCtx c = s->ctx();
        const int B = s->beam, V = D.n_vocab;
        int* adj_rows_d = s->adj_row;
        int* adj_ids_d = s->adj_row + n_adj;
        float* adj_deltas_d = reinterpret_cast<float*>(s->adj_row + 2 * n_adj);
        if (n_adj > 0) {
            // the adjustment block has its own pinned region (offset 128 KiB) so it never races with the
            // decode staging block that a still-running graph replay may be reading
            char* st = static_cast<char*>(s->pinned) + 131072;
            std::memcpy(st, adj_row, n_adj * sizeof(int));
            std::memcpy(st + n_adj * 4, adj_ids, n_adj * sizeof(int));
            std::memcpy(st + n_adj * 8, adj_deltas, n_adj * sizeof(float));
            WLK_HIP(hipMemcpyAsync(s->adj_row, st, (size_t)n_adj * 12, hipMemcpyHostToDevice, s->stream));
        }
        AlignArgs a;
        a.ring = s->ring; a.n_align = m->n_align; a.n_beam = B; a.ring_rows = s->ring_rows; a.T = D.n_audio_ctx;
        a.prefill_rows = s->n_steps <= kAlignWindow ? s->prefill_rows : 0;
        a.n_single = std::min(s->n_steps - 1, kAlignWindow);
        a.single_base = D.n_text_ctx;
        a.newest_row = s->n_steps == 1 ? s->prefill_rows - 1 : D.n_text_ctx + ((s->n_steps - 2) % kAlignWindow);
        a.content_len = content_mel_len;
        a.z = s->z; a.attn_last = s->attn_last; a.frames = s->frames;
        // the no-speech probability of the sot rows is one more role of the read-out's second launch
        float* ns_dev = s->top_vals + (size_t)B * 17;
        if (!launch_select_fused(c, s->logits_last, V, B, k, s->top_vals, s->top_ids, s->topk_scratch, adj_rows_d,
                                 adj_ids_d, adj_deltas_d, n_adj, a, StepHostOut{}, no_speech_host ? s->logits_sot : nullptr,
                                 no_speech_token, ns_dev)) {
            if (no_speech_host) launch_token_prob(c, s->logits_sot, V, B, no_speech_token, ns_dev);
            launch_logsoftmax_topk(c, s->logits_last, V, B, k, s->top_vals, s->top_ids, s->topk_scratch, adj_rows_d,
                                   adj_ids_d, adj_deltas_d, n_adj);
            launch_alignatt(c, a);
        }

        // one readback of the packed result block [log-probs B*8 | ids B*8 | frames B (| no-speech B)]
        char* out = static_cast<char*>(s->pinned) + 65536;
        WLK_HIP(hipMemcpyAsync(out, s->top_vals, (size_t)B * (no_speech_host ? 18 : 17) * 4, hipMemcpyDeviceToHost, s->stream));
        WLK_HIP(hipStreamSynchronize(s->stream));
        if (no_speech_host) std::memcpy(no_speech_host, out + (size_t)B * 68, B * sizeof(float));
        for (int b = 0; b < B; ++b) {
            std::memcpy(top_logprobs_host + (size_t)b * k, out + (size_t)b * k * 4, k * sizeof(float));
            std::memcpy(top_ids_host + (size_t)b * k, out + (size_t)B * 32 + (size_t)b * k * 4, k * sizeof(int));
        }
        std::memcpy(frames_host, out + (size_t)B * 64, B * sizeof(int));
        return WLK_OK;
    });
}

int wlk_select(wlk_session* s, const int32_t* adj_row, const int32_t* adj_ids, const float* adj_deltas, int n_adj, int k,
               int content_mel_len, float* top_logprobs_host, int32_t* top_ids_host, int32_t* frames_host) {
    return select_impl(s, adj_row, adj_ids, adj_deltas, n_adj, k, content_mel_len, top_logprobs_host, top_ids_host,
                       frames_host, -1, nullptr);
}

extern "C++" int wlk_select_first(wlk_session* s, int no_speech_token, const int32_t* adj_row, const int32_t* adj_ids,
                                  const float* adj_deltas, int n_adj, int k, int content_mel_len, float* no_speech_host,
                                  float* top_logprobs_host, int32_t* top_ids_host, int32_t* frames_host) {
    if (!s || !no_speech_host) return fail(WLK_ERR_ARG, "NULL argument");
    if (!s->have_sot) return fail(WLK_ERR_STATE, "no sot-row logits: call after the first decode of an infer");
    if (no_speech_token < 0 || no_speech_token >= s->m->D.n_vocab) return fail(WLK_ERR_ARG, "token out of range");
    return select_impl(s, adj_row, adj_ids, adj_deltas, n_adj, k, content_mel_len, top_logprobs_host, top_ids_host,
                       frames_host, no_speech_token, no_speech_host);
}

// One single-token step + read-out of a beam-1 session as ONE graph replay: [embed (reads the host block)] [decoder
// layers] [logits] [top-k stage 1 + z-score] [top-k stage 2 + AlignAtt argmax (write the host result)].  Against
// wlk_decode + wlk_select this drops the two staging copies, the result copy and the hand-over between a graph replay
// and eagerly enqueued kernels; the arithmetic is the same kernels on the same data.
extern "C++" int wlk_step_select(wlk_session* s, int64_t token, const int32_t* adj_ids, const float* adj_deltas, int n_adj,
                                 int content_mel_len, float* top_logprobs2, int32_t* top_ids2, int32_t* frame) {
    static const bool enabled = [] {
        const char* e = getenv("WLK_FUSED_STEP");
        return !(e && e[0] == '0');
    }();
    wlk_model* m = s->m;
    const wlk_dims& D = m->D;
    if (!enabled || s->beam != 1 || !s->step_host || s->debug || s->prof_on || !s->use_graph || s->n_steps < 1 ||
        n_adj > kStepMaxAdj || !gemv_applicable(1, D.n_text_state) || !s->encoded)
        return 1;
    AlignArgs a;
    a.ring = nullptr; a.n_align = m->n_align; a.n_beam = 1; a.ring_rows = s->ring_rows; a.T = D.n_audio_ctx;
    a.prefill_rows = 0; a.n_single = 0; a.newest_row = 0; a.single_base = D.n_text_ctx; a.content_len = 0;
    a.z = s->z; a.attn_last = s->attn_last; a.frames = s->frames;
    if (!select_fused_applicable(1, 2, a)) return 1;
    if (token < 0 || token >= D.n_vocab) return fail(WLK_ERR_ARG, "token id out of range");
    if (s->self_len + 1 > D.n_text_ctx) return fail(WLK_ERR_CAPACITY, "text context exceeded");
    if (content_mel_len < 0) return fail(WLK_ERR_ARG, "content_mel_len out of range");
    if (content_mel_len > D.n_audio_ctx) content_mel_len = D.n_audio_ctx;
    return guarded([&]() {
        WLK_HIP(hipSetDevice(m->device));
        const LaunchCtx c = s->ctx();
        const int ctx_len = D.n_text_ctx;
        // the step's inputs (wlk_decode's staging block + wlk_select's arguments); the previous replay has delivered
        // its result, so its first kernel is long past reading this block
        StepBlock& b = *s->step_host;
        const int steps_after = s->n_steps + 1;
        b.row.kcache = s->kcache[s->kv_cur]; b.row.vcache = s->vcache[s->kv_cur];
        b.row.cross_kv = s->cross_kv; b.row.ring = s->ring;
        b.row.token = (int)token;
        b.row.offset = s->self_len;
        b.row.ring_row = ctx_len + ((s->n_steps - 1) % kAlignWindow);
        b.row.prefill_rows = steps_after <= kAlignWindow ? s->prefill_rows : 0;
        b.row.n_single = std::min(steps_after - 1, kAlignWindow);
        b.row.newest_row = ctx_len + ((steps_after - 2) % kAlignWindow);
        b.row.content_len = content_mel_len;
        b.n_adj = n_adj;
        for (int i = 0; i < n_adj; ++i) {
            b.adj_row[i] = -1;
            b.adj_ids[i] = adj_ids[i];
            b.adj_deltas[i] = adj_deltas[i];
        }
        const unsigned seq = ++s->step_seq ? s->step_seq : ++s->step_seq;   // never 0 (the blocks start zeroed)
        b.seq = seq;
        std::atomic_thread_fence(std::memory_order_release);

        const auto t_enter = std::chrono::steady_clock::now();
        hipGraphExec_t& exec = s->fstep_exec[s->kv_cur];
        if (!exec) {
            hipGraph_t graph = nullptr;
            mall_prefetch_prepare(s);
            const int zf_blocks = (a.T + 63) / 64;
            if (!s->step_align_dev) {      // the read-out's arguments as the side workgroups see them: everything that changes
                AlignArgs ad = a;          // from step to step comes through `rows` (the step's device block)
                ad.rows = &s->step_dev->row;
                s->step_align_dev = reinterpret_cast<AlignArgs*>(dev_alloc<char>(sizeof(AlignArgs)));
                WLK_HIP(hipMemcpyAsync(s->step_align_dev, &ad, sizeof(AlignArgs), hipMemcpyHostToDevice, s->stream));
            }
            WLK_HIP(hipStreamSynchronize(s->stream));
            WLK_HIP(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
            try {
                a.rows = &s->step_dev->row;
                a.part = s->z + (size_t)s->beam * std::max(m->n_align, 1) * D.n_audio_ctx;
                // the vocabulary projection's weights are a (V / 4 / 4 > 2048 workgroups) stream long enough to hide the z-score
                const bool early_z = select_early_z_enabled() && D.n_vocab >= 16384;
                s->pf_marks = true;
                if (early_z) enqueue_decode(s, c, 1, 1, false, 0, true, s->step_align_dev, zf_blocks * a.n_align * a.n_beam, zf_blocks);
                else enqueue_decode(s, c, 1, 1, false, 0, true);
                s->pf_marks = false;
                StepHostOut ho;
                ho.result = s->result_host_dev;
                ho.n_adj = &s->step_dev->n_adj;
                ho.seq = &s->step_dev->seq;
                if (!launch_select_fused(c, s->logits_last, D.n_vocab, 1, 2, s->top_vals, s->top_ids, s->topk_scratch,
                                         s->step_dev->adj_row, s->step_dev->adj_ids, s->step_dev->adj_deltas, 0, a, ho, nullptr, 0,
                                         nullptr, early_z))
                    throw std::runtime_error("fused step: read-out not available");
            } catch (...) {
                s->pf_marks = false;
                (void)hipStreamEndCapture(s->stream, &graph);
                if (graph) (void)hipGraphDestroy(graph);
                throw;
            }
            WLK_HIP(hipStreamEndCapture(s->stream, &graph));
            WLK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(graph);
        }
        WLK_HIP(hipGraphLaunch(exec, s->stream));
        if (s->pf_progress) {            // beside the replay: next layers' weights into the Infinity Cache, paced by the chain's marks
            LaunchCtx pc;
            pc.stream = s->pf_stream;
            launch_mall_prefetch_step(pc, s->pf_table, D.n_text_layer, seq, s->pf_progress, mall_prefetch_lead(), s->dh);
        }

        // the last kernel stores the two flags after the fields
        volatile StepResult* r = s->result_host;
        const auto t_start = std::chrono::steady_clock::now();
        s->step_launch_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_start - t_enter).count();
        try {
            wlk_wait_step_flags(s->stream, s->result_host, 1, seq);
        } catch (...) {
            // no result: whatever the replay did to the caches is unknown, so the session must be re-encoded (and
            // re-prefilled) before it decodes again - never continue at a guessed cache offset
            s->encoded = false;
            s->n_steps = 0;
            s->self_len = 0;
            throw;
        }
        // the step is in the caches and its result is here: only now does the bookkeeping advance
        s->have_sot = false;
        s->self_len += 1;
        s->n_steps += 1;
        s->last_rows = 1;
        s->last_ntok = 1;
        s->step_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_enter).count();
        s->step_count += 1;       // three clock reads per ~200 us step: always on (wlk_session_step_stats)
        top_logprobs2[0] = r->top_vals[0]; top_logprobs2[1] = r->top_vals[1];
        top_ids2[0] = r->top_ids[0]; top_ids2[1] = r->top_ids[1];
        *frame = r->frame;
        return WLK_OK;
    });
}

int wlk_kv_reorder(wlk_session* s, const int32_t* source_rows, int n_rows) {
    if (!s || !source_rows || n_rows != s->beam) return fail(WLK_ERR_ARG, "bad reorder request");
    bool identity = true;
    for (int i = 0; i < n_rows; ++i) {
        if (source_rows[i] < 0 || source_rows[i] >= n_rows) return fail(WLK_ERR_ARG, "source row out of range");
        identity &= source_rows[i] == i;
    }
    if (identity || s->self_len == 0) return WLK_OK;
    return guarded([&]() {
        const wlk_dims& D = s->m->D;
        WLK_HIP(hipSetDevice(s->m->device));
        const LaunchCtx c = s->ctx();
        const size_t cache = (size_t)D.n_text_layer * s->beam * D.n_text_ctx * D.n_text_state;
        if (!s->kcache[1]) {
            s->kcache[1] = dev_alloc<float>(cache);
            s->vcache[1] = dev_alloc<float>(cache);
        }
        WLK_HIP(hipStreamSynchronize(s->stream));
        std::memcpy(s->pinned, source_rows, n_rows * sizeof(int));
        WLK_HIP(hipMemcpyAsync(s->src_rows, s->pinned, n_rows * sizeof(int), hipMemcpyHostToDevice, s->stream));
        const int cur = s->kv_cur;
        launch_kv_gather(c, s->kcache[cur], s->kcache[cur ^ 1], s->src_rows, n_rows, s->self_len, D.n_text_state,
                         D.n_text_ctx, D.n_text_layer);
        launch_kv_gather(c, s->vcache[cur], s->vcache[cur ^ 1], s->src_rows, n_rows, s->self_len, D.n_text_state,
                         D.n_text_ctx, D.n_text_layer);
        s->kv_cur ^= 1;
        return WLK_OK;
    });
}

int wlk_sync(wlk_session* s) {
    if (!s) return fail(WLK_ERR_ARG, "session is NULL");
    return guarded([&]() {
        WLK_HIP(hipSetDevice(s->m->device));
        WLK_HIP(hipStreamSynchronize(s->stream));
        return WLK_OK;
    });
}

// ---- exports --------------------------------------------------------------------------------
static int copy_out(wlk_session* s, const float* dev, uint64_t n, float* host, uint64_t cap, uint64_t* n_written) {
    if (n_written) *n_written = n;
    if (n > cap) return fail(WLK_ERR_CAPACITY, "export buffer too small: need " + std::to_string(n) + " floats");
    WLK_HIP(hipStreamSynchronize(s->stream));
    copy_sync(host, dev, n * sizeof(float), hipMemcpyDeviceToHost);
    return WLK_OK;
}

int wlk_export(wlk_session* s, const char* what, float* host, uint64_t capacity, uint64_t* n_written) {
    if (!s || !what || !host) return fail(WLK_ERR_ARG, "NULL argument");
    return guarded([&]() {
        const wlk_dims& D = s->m->D;
        WLK_HIP(hipSetDevice(s->m->device));
        const std::string w(what);
        const uint64_t T = D.n_audio_ctx, d = D.n_text_state;
        if (w == "mel") {  // reference layout [n_mels][3000]
            const uint64_t n = (uint64_t)D.n_mels * kMelFrames;
            if (n_written) *n_written = n;
            if (n > capacity) return fail(WLK_ERR_CAPACITY, "export buffer too small");
            std::vector<float> tm(n);
            WLK_HIP(hipStreamSynchronize(s->stream));
            copy_sync(tm.data(), s->mel_t + D.n_mels, n * sizeof(float), hipMemcpyDeviceToHost);
            for (int t = 0; t < kMelFrames; ++t)
                for (int mm = 0; mm < D.n_mels; ++mm) host[(size_t)mm * kMelFrames + t] = tm[(size_t)t * D.n_mels + mm];
            return WLK_OK;
        }
        if (w == "enc") {
            if (s->enc_out_is_x3) {      // the encoder output exists as three bf16 planes: hi + mid + lo is the fp32 value, exactly
                launch_x3_unpack(s->ctx(), s->enc_out3, d, s->enc_out, d, (int)T, (int)d);
            }
            return copy_out(s, s->enc_out, T * d, host, capacity, n_written);
        }
        if (w == "logits_last") return copy_out(s, s->logits_last, (uint64_t)s->beam * D.n_vocab, host, capacity, n_written);
        if (w == "logits_sot") return copy_out(s, s->logits_sot, (uint64_t)s->beam * D.n_vocab, host, capacity, n_written);
        if (w == "attn_last") return copy_out(s, s->attn_last, (uint64_t)s->beam * T, host, capacity, n_written);
        if (w == "dec_x") return copy_out(s, s->dx, (uint64_t)s->last_rows * s->last_ntok * d, host, capacity, n_written);
        const auto colon = w.find(':');
        if (colon != std::string::npos) {
            const std::string kind = w.substr(0, colon);
            const int idx = std::atoi(w.c_str() + colon + 1);
            if (kind == "cross_qk") {
                if (!s->qk_debug) return fail(WLK_ERR_STATE, "cross_qk export needs a debug session");
                if (idx < 0 || idx >= D.n_text_layer) return fail(WLK_ERR_ARG, "layer out of range");
                const uint64_t rows = (uint64_t)s->last_rows * s->last_ntok;
                return copy_out(s, s->qk_debug + (size_t)idx * s->max_rows * D.n_text_head * T,
                                rows * D.n_text_head * T, host, capacity, n_written);
            }
            if (kind == "self_k" || kind == "self_v") {
                if (idx < 0 || idx >= D.n_text_layer) return fail(WLK_ERR_ARG, "layer out of range");
                const float* base = (kind == "self_k" ? s->kcache : s->vcache)[s->kv_cur] +
                                    (size_t)idx * s->beam * D.n_text_ctx * d;
                const uint64_t n = (uint64_t)s->beam * s->self_len * d;
                if (n_written) *n_written = n;
                if (n > capacity) return fail(WLK_ERR_CAPACITY, "export buffer too small");
                WLK_HIP(hipStreamSynchronize(s->stream));
                for (int b = 0; b < s->beam; ++b)
                    copy_sync(host + (size_t)b * s->self_len * d, base + (size_t)b * D.n_text_ctx * d,
                                      (size_t)s->self_len * d * sizeof(float), hipMemcpyDeviceToHost);
                return WLK_OK;
            }
            if (kind == "xattn_w") {  // beam 0 window rows of one alignment head: prefill rows then single slots
                if (idx < 0 || idx >= s->m->n_align) return fail(WLK_ERR_ARG, "alignment rank out of range");
                const int pre = s->n_steps <= kAlignWindow ? s->prefill_rows : 0;
                const int ns = std::min(s->n_steps - 1, kAlignWindow);
                const uint64_t n = (uint64_t)(pre + ns) * T;
                if (n_written) *n_written = n;
                if (n > capacity) return fail(WLK_ERR_CAPACITY, "export buffer too small");
                const float* base = s->ring + (size_t)idx * s->beam * s->ring_rows * T;
                WLK_HIP(hipStreamSynchronize(s->stream));
                if (pre) copy_sync(host, base, (size_t)pre * T * sizeof(float), hipMemcpyDeviceToHost);
                if (ns)
                    copy_sync(host + (size_t)pre * T, base + (size_t)D.n_text_ctx * T, (size_t)ns * T * sizeof(float),
                                      hipMemcpyDeviceToHost);
                return WLK_OK;
            }
        }
        return fail(WLK_ERR_ARG, "unknown export '" + w + "'");
    });
}

// ---- diarization front end: stand-alone log-mel extractor --------------------------------------

int wlk_melspec_create(int device, int n_fft, int win_length, int hop, int n_mels, const float* filters,
                       const float* window, float preemph, float log_guard, int max_samples, wlk_melspec** out) {
    if (!filters || !window || !out) return fail(WLK_ERR_ARG, "NULL argument");
    if (n_fft < 16 || n_fft > 512 || win_length < 1 || win_length > n_fft || hop < 1 || n_mels < 1 || n_mels > 256 ||
        max_samples < n_fft)
        return fail(WLK_ERR_ARG, "unsupported mel-spectrogram configuration");
    return guarded([&]() {
        WLK_HIP(hipSetDevice(device));
        auto m = std::make_unique<wlk_melspec>();
        m->device = device; m->n_fft = n_fft; m->win_length = win_length; m->hop = hop; m->n_mels = n_mels;
        m->preemph = preemph; m->log_guard = log_guard; m->cap = max_samples;
        const int nf = n_fft / 2 + 1;
        WLK_HIP(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
        m->window = dev_alloc<float>(win_length);
        m->filters = dev_alloc<float>((size_t)n_mels * nf);
        m->twiddle = dev_alloc<double>(n_fft);
        m->lo = dev_alloc<int>(n_mels);
        m->hi = dev_alloc<int>(n_mels);
        m->audio = dev_alloc<float>(max_samples);
        m->out = dev_alloc<float>((size_t)(max_samples / hop + 2) * n_mels);
        std::vector<double> tw(n_fft);
        for (int i = 0; i < n_fft; ++i) tw[i] = std::cos(2.0 * M_PI * (double)i / (double)n_fft);
        std::vector<int> lo(n_mels), hi(n_mels);
        for (int i = 0; i < n_mels; ++i) {
            int a = nf, b = 0;
            for (int k = 0; k < nf; ++k)
                if (filters[(size_t)i * nf + k] != 0.f) { a = std::min(a, k); b = k + 1; }
            if (b == 0) a = 0;
            lo[i] = a; hi[i] = b;
        }
        copy_sync(m->window, window, win_length * sizeof(float), hipMemcpyHostToDevice);
        copy_sync(m->filters, filters, (size_t)n_mels * nf * sizeof(float), hipMemcpyHostToDevice);
        copy_sync(m->twiddle, tw.data(), n_fft * sizeof(double), hipMemcpyHostToDevice);
        copy_sync(m->lo, lo.data(), n_mels * sizeof(int), hipMemcpyHostToDevice);
        copy_sync(m->hi, hi.data(), n_mels * sizeof(int), hipMemcpyHostToDevice);
        *out = m.release();
        return WLK_OK;
    });
}

int wlk_melspec_run(wlk_melspec* m, const float* pcm_host, int n, float* out_host, int capacity_frames, int* n_frames) {
    if (!m || !pcm_host || !out_host || !n_frames) return fail(WLK_ERR_ARG, "NULL argument");
    if (n < 1 || n > m->cap) return fail(WLK_ERR_CAPACITY, "chunk does not fit the extractor's buffer");
    return guarded([&]() {
        std::lock_guard<std::mutex> lock(m->mu);
        WLK_HIP(hipSetDevice(m->device));
        // FilterbankFeatures.get_seq_len with centre padding: floor((n + 2*(n_fft/2) - n_fft) / hop) + 1
        const int frames = n / m->hop + 1;
        *n_frames = frames;
        if (frames > capacity_frames) return fail(WLK_ERR_CAPACITY, "output buffer too small");
        WLK_HIP(hipMemcpyAsync(m->audio, pcm_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice, m->stream));
        MelSpecArgs a;
        a.audio = m->audio; a.n_samples = n; a.window = m->window; a.twiddle = m->twiddle; a.filters = m->filters;
        a.filt_lo = m->lo; a.filt_hi = m->hi; a.out = m->out; a.n_fft = m->n_fft; a.win_length = m->win_length;
        a.hop = m->hop; a.n_mels = m->n_mels; a.preemph = m->preemph; a.log_guard = m->log_guard;
        LaunchCtx c{m->stream, nullptr};
        launch_melspec(c, a, frames);
        WLK_HIP(hipMemcpyAsync(out_host, m->out, (size_t)frames * m->n_mels * sizeof(float), hipMemcpyDeviceToHost, m->stream));
        WLK_HIP(hipStreamSynchronize(m->stream));
        return WLK_OK;
    });
}

int wlk_melspec_destroy(wlk_melspec* m) {
    if (!m) return WLK_OK;
    (void)hipSetDevice(m->device);
    float* fl[] = {m->window, m->filters, m->audio, m->out};
    for (float* p : fl)
        if (p) (void)hipFree(p);
    if (m->twiddle) (void)hipFree(m->twiddle);
    if (m->lo) (void)hipFree(m->lo);
    if (m->hi) (void)hipFree(m->hi);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
    return WLK_OK;
}

// ---- profiling ------------------------------------------------------------------------------
int wlk_prof_begin(wlk_session* s) {
    if (!s) return fail(WLK_ERR_ARG, "session is NULL");
    for (auto& r : s->prof.recs) {
        s->prof.pool.push_back(r.a);
        s->prof.pool.push_back(r.b);
    }
    s->prof.recs.clear();
    s->prof_on = true;
    return WLK_OK;
}

int wlk_session_step_stats(wlk_session* s, uint64_t* steps, uint64_t* wall_ns, uint64_t* launch_ns) {
    if (!s || !steps || !wall_ns || !launch_ns) return fail(WLK_ERR_ARG, "NULL argument");
    *steps = s->step_count;
    *wall_ns = s->step_ns;
    *launch_ns = s->step_launch_ns;
    return WLK_OK;
}

int wlk_prof_end(wlk_session* s, int cap, const char** names, float* total_ms, int32_t* launches, double* flops,
                 double* bytes, int32_t* n_out) {
    if (!s || !n_out) return fail(WLK_ERR_ARG, "NULL argument");
    return guarded([&]() {
        WLK_HIP(hipSetDevice(s->m->device));
        WLK_HIP(hipStreamSynchronize(s->stream));
        s->prof_on = false;
        std::vector<const char*> order;
        std::map<std::string, std::pair<double, int>> acc;
        std::map<std::string, std::pair<double, double>> work;
        std::map<std::string, const char*> cname;
        for (auto& r : s->prof.recs) {
            float ms = 0.f;
            WLK_HIP(hipEventElapsedTime(&ms, r.a, r.b));
            work[r.name].first += r.flops;
            work[r.name].second += r.bytes;
            auto it = acc.find(r.name);
            if (it == acc.end()) {
                acc[r.name] = {ms, 1};
                cname[r.name] = r.name;
                order.push_back(r.name);
            } else {
                it->second.first += ms;
                it->second.second += 1;
            }
        }
        int n = 0;
        for (const char* nm : order) {
            if (n >= cap) break;
            if (names) names[n] = cname[nm];
            if (total_ms) total_ms[n] = (float)acc[nm].first;
            if (launches) launches[n] = acc[nm].second;
            if (flops) flops[n] = work[nm].first;
            if (bytes) bytes[n] = work[nm].second;
            ++n;
        }
        *n_out = n;
        return WLK_OK;
    });
}

}  // extern "C"

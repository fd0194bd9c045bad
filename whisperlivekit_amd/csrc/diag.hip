// Kernel-level diagnostics behind the C ABI: run ONE kernel on host data and hand the result back,
// so the GPU parity tests can compare each building block with a plain fp32 reference.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/wlk_hip.h"
#include "common.h"
#include "internal.h"
#include "wave_ops.h"

using namespace wlk;

namespace {
thread_local std::string g_diag_error;
struct DevBuf {
    float* p = nullptr;
    explicit DevBuf(size_t n, const float* host = nullptr) {
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(n, 1) * sizeof(float)));
        if (host) WLK_HIP(hipMemcpy(p, host, n * sizeof(float), hipMemcpyHostToDevice));
    }
    ~DevBuf() { (void)hipFree(p); }
};
template <typename F>
int run(F&& f) {
    try {
        f();
        WLK_HIP(hipDeviceSynchronize());
        return WLK_OK;
    } catch (const std::exception& e) {
        g_diag_error = e.what();
        return WLK_ERR_HIP;
    }
}
}  // namespace


// wave_ops.h against the __shfl_xor loops it replaces: out[0..9][lane] = VALU-butterfly results, ref[0..9][lane] = the
// shuffle forms, on the same 64 floats (rows: sum, max, row16 sum 1-2-4-8, xor 1, 2, 4, 8, 16, 32, arg-max index)
__global__ __launch_bounds__(64) void wave_ops_probe_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            float* __restrict__ ref) {
    const int lane = threadIdx.x;
    const float v = in[lane];
    out[0 * 64 + lane] = wave_sum(v);
    out[1 * 64 + lane] = wave_max(v);
    out[2 * 64 + lane] = row16_sum_1248(v);
    out[3 * 64 + lane] = wave_xor<1>(v);
    out[4 * 64 + lane] = wave_xor<2>(v);
    out[5 * 64 + lane] = wave_xor<4>(v);
    out[6 * 64 + lane] = wave_xor<8>(v);
    out[7 * 64 + lane] = wave_xor<16>(v);
    out[8 * 64 + lane] = wave_xor<32>(v);
    {
        float bv = v;
        int bi = lane;
        wave_argmax(bv, bi);
        out[9 * 64 + lane] = (float)bi;
    }
    float s = v, m = v, r = v;
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    r += __shfl_xor(r, 1, 64);
    r += __shfl_xor(r, 2, 64);
    r += __shfl_xor(r, 4, 64);
    r += __shfl_xor(r, 8, 64);
    ref[0 * 64 + lane] = s;
    ref[1 * 64 + lane] = m;
    ref[2 * 64 + lane] = r;
    ref[3 * 64 + lane] = __shfl_xor(v, 1, 64);
    ref[4 * 64 + lane] = __shfl_xor(v, 2, 64);
    ref[5 * 64 + lane] = __shfl_xor(v, 4, 64);
    ref[6 * 64 + lane] = __shfl_xor(v, 8, 64);
    ref[7 * 64 + lane] = __shfl_xor(v, 16, 64);
    ref[8 * 64 + lane] = __shfl_xor(v, 32, 64);
    {
        float bv = v;
        int bi = lane;
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        ref[9 * 64 + lane] = (float)bi;
    }
}

extern "C" {

const char* wlk_diag_last_error(void) { return g_diag_error.c_str(); }

int wlk_diag_linear(const float* a, int64_t lda, int64_t a_floats, const float* w, const float* bias, const float* r,
                    int64_t ldr, int m, int n, int k, int flags, float scale, int scale_cols, int force_gemv, float* c) {
    return run([&]() {
        DevBuf A(a_floats, a), W((size_t)n * k, w), B(n, bias), R(r ? (size_t)m * ldr : 1, r), Cc((size_t)m * n);
        GemmArgs g;
        g.A = A.p; g.lda = lda; g.W = W.p; g.bias = bias ? B.p : nullptr; g.C = Cc.p; g.ldc = n;
        g.R = r ? R.p : nullptr; g.ldr = ldr; g.M = m; g.N = n; g.K = k; g.flags = flags; g.scale = scale;
        g.scale_cols = scale_cols;
        LaunchCtx ctx;
        g.force_kwave = force_gemv == 2;
        g.force_kernel = (force_gemv >= 2 && force_gemv <= 4) || (force_gemv >= 6 && force_gemv <= 8) ? force_gemv : 0;
        if (force_gemv == 1) launch_gemv(ctx, g, "diag_gemv");
        else if (force_gemv >= 5 && force_gemv <= 8) launch_gemm_kp(ctx, g, "diag_gemm_kp");      // 6 / 7: 16 x 16 / 32 x 32 tiles below 512 rows
        else launch_gemm(ctx, g, "diag_gemm");
        WLK_HIP(hipDeviceSynchronize());
        WLK_HIP(hipMemcpy(c, Cc.p, (size_t)m * n * sizeof(float), hipMemcpyDeviceToHost));
    });
}

/* average microseconds per launch of `reps` back-to-back launches of one linear layer (device-resident operands,
 * HIP events around the whole train, one warm-up launch first): the kernel-tuning probe */
int wlk_diag_linear_time(int m, int n, int k, int flags, int force, int reps, float* us_per_launch) {
    return run([&]() {
        std::vector<float> ha((size_t)m * k), hw((size_t)n * k);
        unsigned seed = 12345u;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : ha) v = rnd();
        for (auto& v : hw) v = rnd() * 0.05f;
        DevBuf A((size_t)m * k, ha.data()), W((size_t)n * k, hw.data()), B(n), R((size_t)m * n), Cc((size_t)m * n);
        WLK_HIP(hipMemset(B.p, 0, n * sizeof(float)));
        WLK_HIP(hipMemset(R.p, 0, (size_t)m * n * sizeof(float)));
        GemmArgs g;
        g.A = A.p; g.lda = k; g.W = W.p; g.bias = B.p; g.C = Cc.p; g.ldc = n; g.R = (flags & kGemmResidual) ? R.p : nullptr;
        g.ldr = n; g.M = m; g.N = n; g.K = k; g.flags = flags; g.scale = 0.5f; g.scale_cols = n / 2;
        g.force_kwave = force == 2;
        g.force_kernel = (force >= 2 && force <= 4) || (force >= 6 && force <= 8) || force >= 500 ? force : 0;      // >= 500: kp family with the tile (force - 500) / 10 x % 10
        long long* dbg = nullptr;
        const bool want_clock = getenv("WLK_GEMM_CLOCKS") != nullptr;
        if (want_clock) {
            WLK_HIP(hipMalloc(reinterpret_cast<void**>(&dbg), 4 * 8192 * sizeof(long long)));
            WLK_HIP(hipMemset(dbg, 0, 4 * 8192 * sizeof(long long)));
            g.dbg_clock = dbg;
        }
        hipStream_t st;
        WLK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        LaunchCtx ctx{st, nullptr};
        auto go = [&]() {
            if (force == 1) launch_gemv(ctx, g, "diag");
            else if ((force >= 5 && force <= 8) || force >= 500) launch_gemm_kp(ctx, g, "diag");
            else launch_gemm(ctx, g, "diag");
        };
        go();
        WLK_HIP(hipStreamSynchronize(st));
        hipEvent_t e0, e1;
        WLK_HIP(hipEventCreate(&e0));
        WLK_HIP(hipEventCreate(&e1));
        WLK_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) go();
        WLK_HIP(hipEventRecord(e1, st));
        WLK_HIP(hipStreamSynchronize(st));
        float ms = 0.f;
        WLK_HIP(hipEventElapsedTime(&ms, e0, e1));
        *us_per_launch = 1e3f * ms / (float)reps;
        if (want_clock) {
            std::vector<long long> h(4 * 8192);
            WLK_HIP(hipMemcpy(h.data(), dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
            long long t0 = 0, t1 = 0;
            double pro = 0, loop = 0, epi = 0;
            int n = 0;
            for (int i = 0; i < 8192; ++i) {
                if (!h[4 * i + 3]) continue;
                if (!n || h[4 * i] < t0) t0 = h[4 * i];
                if (!n || h[4 * i + 3] > t1) t1 = h[4 * i + 3];
                pro += h[4 * i + 1] - h[4 * i]; loop += h[4 * i + 2] - h[4 * i + 1]; epi += h[4 * i + 3] - h[4 * i + 2];
                ++n;
            }
            if (n) fprintf(stderr, "[clocks] %d workgroups: prologue %.0f, loop %.0f, fold+epilogue %.0f ticks (mean); first start -> last end %lld ticks\n",
                           n, pro / n, loop / n, epi / n, t1 - t0);
            (void)hipFree(dbg);
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipStreamDestroy(st);
    });
}

int wlk_diag_linear_ln(const float* a, const float* w, const float* bias, const float* gamma, const float* beta, int m,
                       int n, int k, int force_gemv, float* c) {
    return run([&]() {
        DevBuf A((size_t)m * k, a), W((size_t)n * k, w), B(n, bias), G(k, gamma), Bt(k, beta), Cc((size_t)m * n);
        GemmArgs g;
        g.A = A.p; g.lda = k; g.W = W.p; g.bias = B.p; g.C = Cc.p; g.ldc = n; g.M = m; g.N = n; g.K = k;
        g.ln_gamma = G.p; g.ln_beta = Bt.p;
        LaunchCtx ctx;
        (void)force_gemv;
        if (m <= 8) launch_gemv(ctx, g, "diag_gemv_ln");     // decode steps: the weight-streaming kernels
        else launch_gemm(ctx, g, "diag_gemm_ln");            // prompt rows: the 16 x 16 kernel (throws where it does not apply)
        WLK_HIP(hipDeviceSynchronize());
        WLK_HIP(hipMemcpy(c, Cc.p, (size_t)m * n * sizeof(float), hipMemcpyDeviceToHost));
    });
}

int wlk_diag_layernorm(const float* x, const float* gamma, const float* beta, int rows, int d, float* y) {
    return run([&]() {
        DevBuf X((size_t)rows * d, x), G(d, gamma), Bt(d, beta), Y((size_t)rows * d);
        LaunchCtx ctx;
        launch_layernorm(ctx, X.p, d, G.p, Bt.p, Y.p, d, rows, d, "diag_ln");
        WLK_HIP(hipDeviceSynchronize());
        WLK_HIP(hipMemcpy(y, Y.p, (size_t)rows * d * sizeof(float), hipMemcpyDeviceToHost));
    });
}

int wlk_diag_prefill_stack(wlk_session** sessions, const int64_t* tokens, const int32_t* n_tok, const int32_t* sot_index, int32_t n,
                           int32_t* taken) {
    if (!sessions || !tokens || !n_tok || !sot_index || !taken || n < 1 || n > kMaxBatch) {
        set_last_error("prefill_stack: bad arguments");
        return WLK_ERR_ARG;
    }
    return run([&]() {
        wlk_model* m = sessions[0]->m;
        WLK_HIP(hipSetDevice(m->device));
        wlk_prefill_ws ws;
        wlk_prefill_ws_alloc(m, ws, kMaxBatch, std::min(256, (int)m->D.n_text_ctx));
        hipStream_t st = nullptr;
        try {
            WLK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            std::vector<wlk_prefill_item> items(n);
            std::vector<wlk_prefill_item*> stack;
            const int64_t* t = tokens;
            for (int i = 0; i < n; ++i) {
                items[i].s = sessions[i];
                items[i].tokens = t;
                items[i].n_tok = n_tok[i];
                items[i].sot_index = sot_index[i];
                t += n_tok[i];
                WLK_HIP(hipStreamSynchronize(sessions[i]->stream));
                taken[i] = sessions[i]->m == m && wlk_prefill_precheck(items[i], ws).empty() ? 1 : 0;
                if (taken[i]) stack.push_back(&items[i]);
            }
            wlk_prefill_group(stack, LaunchCtx{st, nullptr}, ws);
            WLK_HIP(hipStreamSynchronize(st));
        } catch (...) {
            if (st) (void)hipStreamDestroy(st);
            wlk_prefill_ws_free(ws);
            throw;
        }
        (void)hipStreamDestroy(st);
        wlk_prefill_ws_free(ws);
    });
}

int wlk_diag_encoder_attention_time(int t, int d, int n_head, int k_splits, int reps, float* us_per_launch) {

    return run([&]() {
        std::vector<float> h((size_t)t * 3 * d);
        unsigned seed = 777u;
        for (auto& v : h) { seed = seed * 1664525u + 1013904223u; v = (((seed >> 8) & 0xffff) / 65536.0f - 0.5f) * 1.5f; }
        DevBuf Q((size_t)t * 3 * d, h.data()), O((size_t)t * d), S(flash_split_scratch_floats(t, n_head, 8));

        hipStream_t st;
        WLK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        LaunchCtx ctx{st, nullptr};
        auto go = [&]() { launch_encoder_attention(ctx, Q.p, O.p, t, d, n_head, k_splits, S.p); };
        long long* dbg = nullptr;
        if (getenv("WLK_ATTN_CLOCKS")) {
            WLK_HIP(hipMalloc(reinterpret_cast<void**>(&dbg), 8 * 4096 * sizeof(long long)));
            WLK_HIP(hipMemset(dbg, 0, 8 * 4096 * sizeof(long long)));
            g_attn_dbg_clock = dbg;
        }
        go();
        WLK_HIP(hipStreamSynchronize(st));
        hipEvent_t e0, e1;
        WLK_HIP(hipEventCreate(&e0));
        WLK_HIP(hipEventCreate(&e1));
        WLK_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) go();
        WLK_HIP(hipEventRecord(e1, st));
        WLK_HIP(hipStreamSynchronize(st));
        float ms = 0.f;
        WLK_HIP(hipEventElapsedTime(&ms, e0, e1));
        *us_per_launch = 1e3f * ms / (float)reps;
        if (dbg) {
            g_attn_dbg_clock = nullptr;
            std::vector<long long> h(8 * 4096);
            WLK_HIP(hipMemcpy(h.data(), dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
            double pro = 0, loop = 0, mer = 0, wk = 0, wv = 0, sm = 0;
            long long t0 = 0, t1 = 0;
            int n = 0;
            for (int i = 0; i < 4096; ++i) {
                if (!h[8 * i + 7]) continue;
                if (!n || h[8 * i] < t0) t0 = h[8 * i];
                if (!n || h[8 * i + 3] > t1) t1 = h[8 * i + 3];
                pro += h[8 * i + 1] - h[8 * i]; loop += h[8 * i + 2] - h[8 * i + 1]; mer += h[8 * i + 3] - h[8 * i + 2];
                wk += h[8 * i + 4]; wv += h[8 * i + 5]; sm += h[8 * i + 6];
                ++n;
            }
            if (n) fprintf(stderr, "[attn clocks] %d workgroups (wave 0): prologue %.0f, key loop %.0f (K waits %.0f, V waits %.0f, softmax %.0f), "
                           "merge %.0f ticks (mean); first start -> last end %lld ticks\n", n, pro / n, loop / n, wk / n, wv / n, sm / n, mer / n, t1 - t0);
            (void)hipFree(dbg);
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipStreamDestroy(st);
    });
}

int wlk_diag_encoder_attention(const float* qkv, int t, int d, int n_head, float* out) {
    return run([&]() {
        DevBuf Q((size_t)t * 3 * d, qkv), O((size_t)t * d), S(flash_split_scratch_floats(t, n_head, 8));
        LaunchCtx ctx;
        launch_encoder_attention(ctx, Q.p, O.p, t, d, n_head, 1, S.p);
        WLK_HIP(hipDeviceSynchronize());
        WLK_HIP(hipMemcpy(out, O.p, (size_t)t * d * sizeof(float), hipMemcpyDeviceToHost));
    });
}

int wlk_diag_env_refresh(void) {
    wlk::refresh_env_switches();
    return 0;
}

int wlk_diag_wave_ops(const float* in64, float* out640, float* ref640) {
    return run([&]() {
        DevBuf I(64, in64), O(640), R(640);
        hipLaunchKernelGGL(wave_ops_probe_kernel, dim3(1), dim3(64), 0, nullptr, I.p, O.p, R.p);
        WLK_HIP(hipGetLastError());
        WLK_HIP(hipDeviceSynchronize());
        WLK_HIP(hipMemcpy(out640, O.p, 640 * sizeof(float), hipMemcpyDeviceToHost));
        WLK_HIP(hipMemcpy(ref640, R.p, 640 * sizeof(float), hipMemcpyDeviceToHost));
    });
}

/* C[m, n] = epilogue(A[m, k] . W[n, k]^T + bias) through the X3 path (gemm_x3.hip): both operands are packed into three
 * bf16 planes on the device, the wide bf16-MFMA kernel runs, the fp32 result comes back */
int wlk_diag_linear_x3(const float* a, const float* w, const float* bias, int m, int n, int k, int flags, float scale,
                       int scale_cols, float* c) {
    return run([&]() {
        DevBuf A((size_t)m * k, a), W((size_t)n * k, w), B(n, bias), Cc((size_t)m * n);
        unsigned short *a3 = nullptr, *w3 = nullptr;
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&a3), (size_t)m * 3 * k * sizeof(unsigned short)));
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&w3), x3_w_elems(n, k) * sizeof(unsigned short)));
        LaunchCtx ctx;
        launch_x3_pack(ctx, A.p, k, a3, k, m, k);
        launch_x3_pack_w(ctx, W.p, k, w3, n, k);
        X3GemmArgs g;
        g.A3 = a3; g.lda = k; g.W3 = w3; g.bias = bias ? B.p : nullptr; g.C = Cc.p; g.ldc = n; g.M = m; g.N = n; g.K = k;
        g.flags = flags; g.scale = scale; g.scale_cols = scale_cols;
        launch_gemm_x3(ctx, g, "diag_x3");
        WLK_HIP(hipDeviceSynchronize());
        WLK_HIP(hipMemcpy(c, Cc.p, (size_t)m * n * sizeof(float), hipMemcpyDeviceToHost));
        (void)hipFree(a3);
        (void)hipFree(w3);
    });
}

/* kernel-tuning probe: average microseconds per launch of the X3 wide GEMM (device-resident pseudo-random operands) */
int wlk_diag_linear_x3_time(int m, int n, int k, int flags, int reps, float* us_per_launch) {
    return run([&]() {
        std::vector<float> ha((size_t)m * k), hw((size_t)n * k);
        unsigned seed = 12345u;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : ha) v = rnd();
        for (auto& v : hw) v = rnd() * 0.05f;
        DevBuf A((size_t)m * k, ha.data()), W((size_t)n * k, hw.data()), B(n), Cc((size_t)m * n);
        WLK_HIP(hipMemset(B.p, 0, n * sizeof(float)));
        unsigned short *a3 = nullptr, *w3 = nullptr;
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&a3), (size_t)m * 3 * k * sizeof(unsigned short)));
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&w3), x3_w_elems(n, k) * sizeof(unsigned short)));
        hipStream_t st;
        WLK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        LaunchCtx ctx{st, nullptr};
        launch_x3_pack(ctx, A.p, k, a3, k, m, k);
        launch_x3_pack_w(ctx, W.p, k, w3, n, k);
        X3GemmArgs g;
        g.A3 = a3; g.lda = k; g.W3 = w3; g.bias = B.p; g.C = Cc.p; g.ldc = n; g.M = m; g.N = n; g.K = k; g.flags = flags;
        g.scale = 0.5f; g.scale_cols = n / 2;
        launch_gemm_x3(ctx, g, "diag");
        WLK_HIP(hipStreamSynchronize(st));
        hipEvent_t e0, e1;
        WLK_HIP(hipEventCreate(&e0));
        WLK_HIP(hipEventCreate(&e1));
        WLK_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) launch_gemm_x3(ctx, g, "diag");
        WLK_HIP(hipEventRecord(e1, st));
        WLK_HIP(hipStreamSynchronize(st));
        float ms = 0.f;
        WLK_HIP(hipEventElapsedTime(&ms, e0, e1));
        *us_per_launch = 1e3f * ms / (float)reps;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipStreamDestroy(st);
        (void)hipFree(a3);
        (void)hipFree(w3);
    });
}

/* LayerNorm with the result in the X3 format, unpacked again: y must equal wlk_diag_layernorm's y bit for bit */
int wlk_diag_layernorm_x3(const float* x, const float* gamma, const float* beta, int rows, int d, float* y) {
    return run([&]() {
        DevBuf X((size_t)rows * d, x), G(d, gamma), Bt(d, beta), Y((size_t)rows * d);
        unsigned short* y3 = nullptr;
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&y3), (size_t)rows * 3 * d * sizeof(unsigned short)));
        LaunchCtx ctx;
        launch_layernorm_x3(ctx, X.p, d, G.p, Bt.p, y3, d, rows, d, "diag_ln_x3");
        launch_x3_unpack(ctx, y3, d, Y.p, d, rows, d);
        WLK_HIP(hipDeviceSynchronize());
        WLK_HIP(hipMemcpy(y, Y.p, (size_t)rows * d * sizeof(float), hipMemcpyDeviceToHost));
        (void)hipFree(y3);
    });
}

/* encoder self-attention through the X3 path (attention_x3.hip): qkv [t, 3d] (q and k pre-scaled) is packed into the
 * operand image on the device, out [t, d] comes back; compare with wlk_diag_encoder_attention */
int wlk_diag_encoder_attention_x3(const float* qkv, int t, int d, int n_head, float* out) {
    return run([&]() {
        DevBuf Q((size_t)t * 3 * d, qkv), O((size_t)t * d);
        unsigned short* img = nullptr;
        const size_t n = x3_attn_image_elems(t, d);
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&img), n * sizeof(unsigned short)));
        LaunchCtx ctx;
        launch_x3_pack_qkv(ctx, Q.p, img, t, d, x3_attn_vt_off(t, d), x3_attn_vt_ld(t));
        launch_encoder_attention_x3(ctx, img, 2L * d, x3_attn_vt_off(t, d), x3_attn_vt_ld(t), O.p, d, t, d, n_head, nullptr, 0);
        WLK_HIP(hipDeviceSynchronize());
        WLK_HIP(hipMemcpy(out, O.p, (size_t)t * d * sizeof(float), hipMemcpyDeviceToHost));
        (void)hipFree(img);
    });
}

/* The PRODUCTION operand route of the X3 attention against the diagnostic one: x [t][d] . w [3 d][d]^T (+ bias, q | k columns
 * scaled) through gemm_x3 with the X3 epilogue (q | k as X3 rows, V transposed in the lane order of the attention kernel) ->
 * enc_attention_x3 -> out_epilogue; the same projection with an fp32 result -> x3_pack_qkv -> attention -> out_packed.
 * The two must agree bit for bit (same fp32 values split into the same planes, the same image). */
int wlk_diag_qkv_x3_attention(const float* x, const float* w, const float* bias, int t, int d, int n_head, float scale,
                              float* out_epilogue, float* out_packed) {
    return run([&]() {
        DevBuf X((size_t)t * d, x), W((size_t)3 * d * d, w), B((size_t)3 * d, bias), QKV((size_t)t * 3 * d), O((size_t)t * d);
        unsigned short *x3 = nullptr, *w3 = nullptr, *img = nullptr;
        const size_t n_img = x3_attn_image_elems(t, d);
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&x3), (size_t)t * 3 * d * sizeof(unsigned short)));
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&w3), x3_w_elems(3 * d, d) * sizeof(unsigned short)));
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&img), n_img * sizeof(unsigned short)));
        LaunchCtx ctx;
        launch_x3_pack(ctx, X.p, d, x3, d, t, d);
        launch_x3_pack_w(ctx, W.p, d, w3, 3 * d, d);
        X3GemmArgs g;
        g.A3 = x3; g.lda = d; g.W3 = w3; g.bias = B.p; g.M = t; g.N = 3 * d; g.K = d;
        g.flags = kGemmScaleCols; g.scale = scale; g.scale_cols = 2 * d;
        for (int route = 0; route < 2; ++route) {
            WLK_HIP(hipMemset(img, 0, n_img * sizeof(unsigned short)));     // the session's image starts zeroed as well
            X3GemmArgs r = g;
            if (route == 0) {
                r.x3_out = true; r.C3 = img; r.ldc3 = 2 * d; r.vt_col0 = 2 * d; r.vt_off = x3_attn_vt_off(t, d); r.vt_ld = x3_attn_vt_ld(t);
                launch_gemm_x3(ctx, r, "diag_qkv_x3");
            } else {
                r.C = QKV.p; r.ldc = 3 * d;
                launch_gemm_x3(ctx, r, "diag_qkv");
                launch_x3_pack_qkv(ctx, QKV.p, img, t, d, x3_attn_vt_off(t, d), x3_attn_vt_ld(t));
            }
            launch_encoder_attention_x3(ctx, img, 2L * d, x3_attn_vt_off(t, d), x3_attn_vt_ld(t), O.p, d, t, d, n_head, nullptr, 0);
            WLK_HIP(hipDeviceSynchronize());
            WLK_HIP(hipMemcpy(route == 0 ? out_epilogue : out_packed, O.p, (size_t)t * d * sizeof(float), hipMemcpyDeviceToHost));
        }
        (void)hipFree(x3);
        (void)hipFree(w3);
        (void)hipFree(img);
    });
}

/* timing probe of the same: average microseconds per launch */
int wlk_diag_encoder_attention_x3_time(int t, int d, int n_head, int reps, float* us_per_launch) {
    return run([&]() {
        std::vector<float> h((size_t)t * 3 * d);
        unsigned seed = 777u;
        for (auto& v : h) { seed = seed * 1664525u + 1013904223u; v = (((seed >> 8) & 0xffff) / 65536.0f - 0.5f) * 1.5f; }
        DevBuf Q((size_t)t * 3 * d, h.data()), O((size_t)t * d);
        unsigned short* img = nullptr;
        WLK_HIP(hipMalloc(reinterpret_cast<void**>(&img), x3_attn_image_elems(t, d) * sizeof(unsigned short)));
        hipStream_t st;
        WLK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        LaunchCtx ctx{st, nullptr};
        launch_x3_pack_qkv(ctx, Q.p, img, t, d, x3_attn_vt_off(t, d), x3_attn_vt_ld(t));
        auto go = [&]() { launch_encoder_attention_x3(ctx, img, 2L * d, x3_attn_vt_off(t, d), x3_attn_vt_ld(t), O.p, d, t, d, n_head, nullptr, 0); };
        go();
        WLK_HIP(hipStreamSynchronize(st));
        hipEvent_t e0, e1;
        WLK_HIP(hipEventCreate(&e0));
        WLK_HIP(hipEventCreate(&e1));
        WLK_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) go();
        WLK_HIP(hipEventRecord(e1, st));
        WLK_HIP(hipStreamSynchronize(st));
        float ms = 0.f;
        WLK_HIP(hipEventElapsedTime(&ms, e0, e1));
        *us_per_launch = 1e3f * ms / (float)reps;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipStreamDestroy(st);
        (void)hipFree(img);
    });
}

}  // extern "C"

"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
(1) plain fp32 torch references of each kernel, (2) the committed golden fixtures produced by the
reference itself, (3) the CPU oracle on the same seeded inputs, (4) the reference's golden streams
end to end (bit-exact token ids / frames / timestamps).

Tolerances: north-star says mel/logits within 1e-3 (fp32); the asserts below are tighter where the
measured error allows and every measured error is also written to gpurun_out/parity_report.json."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

import helpers as H
from whisperlivekit_amd import _lib, synth
from whisperlivekit_amd.dims import ALIGNMENT_HEADS, MODEL_DIMS
from whisperlivekit_amd.melbank import mel_filterbank

pytestmark = pytest.mark.gpu

REPORT = {}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def report(key, **vals):
    REPORT[key] = {k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in vals.items()}
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report.json"), "w") as fh:
        json.dump(REPORT, fh, indent=1, sort_keys=True)


@pytest.fixture(scope="module")
def lib():
    lib = _lib.load()
    assert lib.wlk_device_count() > 0, "no MI355X visible"
    return lib


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------------------------------------
# kernels in isolation
# ---------------------------------------------------------------------------------------------
GEMM_CASES = [
    # M, N, K, lda, flags, tag
    (1500, 512, 512, 512, 0, "square"),
    (1500, 1536, 512, 512, 4, "qkv_scale"),
    (1500, 2048, 512, 512, 1, "fc1_gelu"),
    (1500, 512, 2048, 2048, 2, "fc2_resid"),
    (3000, 384, 240, 80, 1, "conv1_overlap"),
    (1500, 128, 384, 256, 3, "conv2_stride"),
    (37, 200, 132, 132, 0, "ragged"),
    (1, 64, 4, 4, 0, "tiny"),
    (70, 51864, 128, 128, 0, "vocab_tail"),
]


@pytest.mark.parametrize("M,N,K,lda,flags,tag", GEMM_CASES)
def test_gemm_matches_torch(lib, M, N, K, lda, flags, tag):
    rng = np.random.default_rng(1)
    a_floats = (M - 1) * lda + K
    a = rng.standard_normal(a_floats).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    r = rng.standard_normal((M, N)).astype(np.float32)
    scale, scale_cols = 0.3535533905932738, N // 3
    c = np.empty((M, N), np.float32)
    rc = lib.wlk_diag_linear(vp(a), lda, a_floats, vp(w), vp(bias), vp(r) if flags & 2 else None, N, M, N, K, flags,
                             scale, scale_cols, 0, vp(c))
    assert rc == 0, lib.wlk_diag_last_error()
    A = torch.from_numpy(np.lib.stride_tricks.as_strided(a, (M, K), (lda * 4, 4)).copy())
    ref = A.double() @ torch.from_numpy(w).double().T + torch.from_numpy(bias).double()
    if flags & 4:
        ref[:, :scale_cols] *= np.float32(scale).astype(np.float64)
    if flags & 1:
        ref = torch.nn.functional.gelu(ref)
    if flags & 2:
        ref = ref + torch.from_numpy(r).double()
    err = float((torch.from_numpy(c).double() - ref).abs().max())
    report(f"gemm_{tag}", max_abs_err=err, ref_abs_max=float(ref.abs().max()))
    assert err <= 2e-5 * max(1.0, float(ref.abs().max()))


KSPLIT_CASES = [
    # M, N, K, lda, flags: every tile shape of the one-tile-per-CU k-split kernel (3x1, 3x3, 3x4, three rounds of 3x4, the
    # large-v3 picks), ragged M / N tails, overlapping A rows (the stride-2 convolution), every epilogue
    (1500, 512, 512, 512, 2), (1500, 512, 2048, 2048, 2), (1500, 512, 1536, 1024, 3), (1500, 1536, 512, 512, 4),
    (1500, 2048, 512, 512, 1), (1500, 6144, 512, 512, 4), (1500, 1280, 1280, 1280, 2), (1500, 3840, 1280, 1280, 4),
    (1500, 5120, 1280, 1280, 1), (1500, 1280, 5120, 5120, 2), (1500, 384, 384, 384, 2), (1500, 1152, 384, 384, 4),
    (1500, 1536, 384, 384, 1), (1500, 384, 1536, 1536, 2), (1500, 768, 768, 768, 16), (1500, 2304, 768, 768, 8),
    (1501, 520, 256, 256, 3), (777, 200, 320, 320, 2), (515, 96, 1024, 1024, 0),
]


@pytest.mark.parametrize("M,N,K,lda,flags", KSPLIT_CASES)
def test_ksplit_gemm_matches_torch_and_is_deterministic(lib, M, N, K, lda, flags):
    """Encoder-sized problems take the one-tile-per-CU kernel (four waves split K, partial tiles added in wave order):
    vs fp64, run-to-run identical, and within rounding of the 64x64 kernel it replaces on these shapes."""
    rng = np.random.default_rng(M + N + K)
    a_floats = (M - 1) * lda + K
    a = rng.standard_normal(a_floats).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    r = rng.standard_normal((M, N)).astype(np.float32)
    outs = []
    for mode in (4, 4, 3):       # 4 = k-split (must apply), 3 = the 64x64 kernel
        c = np.full((M, N), np.nan, np.float32)
        rc = lib.wlk_diag_linear(vp(a), lda, a_floats, vp(w), vp(bias), vp(r) if flags & 2 else None, N, M, N, K, flags,
                                 0.5, N // 2, mode, vp(c))
        assert rc == 0, lib.wlk_diag_last_error()
        outs.append(c)
    assert np.array_equal(outs[0], outs[1])
    A = torch.from_numpy(np.lib.stride_tricks.as_strided(a, (M, K), (lda * 4, 4)).copy())
    ref = A.double() @ torch.from_numpy(w).double().T + torch.from_numpy(bias).double()
    if flags & 4:
        ref[:, : N // 2] *= 0.5
    if flags & 1:
        ref = torch.nn.functional.gelu(ref)
    if flags & 8:
        ref = torch.relu(ref)
    if flags & 16:
        ref = ref * torch.sigmoid(ref)
    if flags & 2:
        ref = ref + torch.from_numpy(r).double()
    scale = max(1.0, float(ref.abs().max()))
    err = float((torch.from_numpy(outs[0]).double() - ref).abs().max())
    err_classic = float((torch.from_numpy(outs[2]).double() - ref).abs().max())
    report(f"gemm_ksplit_{M}x{N}x{K}", max_abs_err=err, max_abs_err_64x64_kernel=err_classic, ref_abs_max=scale)
    assert err <= 2e-5 * scale
    assert float(np.abs(outs[0] - outs[2]).max()) <= 2e-5 * scale


@pytest.mark.parametrize("M,N,K,flags", [(60, 512, 512, 2), (60, 512, 2048, 2), (9, 1536, 512, 4), (149, 2048, 512, 1),
                                         (401, 512, 2048, 2), (401, 2048, 512, 16), (401, 1536, 512, 0),
                                         (401, 192, 768, 2 | 8), (1500, 512, 2048, 3), (1500, 512, 512, 2),
                                         (33, 70, 1284, 0)])
def test_kwave_gemm_matches_torch_and_is_deterministic(lib, M, N, K, flags):
    """Grids that under-fill the chip (decoder prefill, Sortformer) let the four waves of a workgroup split K of one
    32x32 tile; the partial tiles are added in wave order, so the result is run-to-run identical."""
    rng = np.random.default_rng(7)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    r = rng.standard_normal((M, N)).astype(np.float32)
    outs = []
    for mode in (2, 2, 0):
        c = np.empty((M, N), np.float32)
        rc = lib.wlk_diag_linear(vp(a), K, M * K, vp(w), vp(bias), vp(r) if flags & 2 else None, N, M, N, K, flags,
                                 0.5, N // 2, mode, vp(c))
        assert rc == 0, lib.wlk_diag_last_error()
        outs.append(c)
    assert np.array_equal(outs[0], outs[1])
    ref = torch.from_numpy(a).double() @ torch.from_numpy(w).double().T + torch.from_numpy(bias).double()
    if flags & 4:
        ref[:, : N // 2] *= 0.5
    if flags & 1:
        ref = torch.nn.functional.gelu(ref)
    if flags & 8:
        ref = torch.relu(ref)
    if flags & 16:
        ref = ref * torch.sigmoid(ref)
    if flags & 2:
        ref = ref + torch.from_numpy(r).double()
    scale = max(1.0, float(ref.abs().max()))
    err = float((torch.from_numpy(outs[0]).double() - ref).abs().max())
    report(f"gemm_kwave_{M}x{N}x{K}", max_abs_err=err, vs_unsplit=float(np.abs(outs[0] - outs[2]).max()))
    assert err <= 2e-5 * scale
    assert np.abs(outs[0] - outs[2]).max() <= 2e-5 * scale


@pytest.mark.parametrize("M,N,K", [(1, 512, 512), (1, 2048, 512), (1, 512, 2048), (2, 1536, 512), (3, 130, 384),
                                   (4, 51864, 128), (8, 257, 512), (1, 51864, 512)])
def test_gemv_matches_torch(lib, M, N, K):
    rng = np.random.default_rng(2)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    r = rng.standard_normal((M, N)).astype(np.float32)
    c = np.empty((M, N), np.float32)
    rc = lib.wlk_diag_linear(vp(a), K, M * K, vp(w), vp(bias), vp(r), N, M, N, K, 1 | 2 | 4, 0.5, N // 2, 1, vp(c))
    assert rc == 0, lib.wlk_diag_last_error()
    ref = torch.from_numpy(a).double() @ torch.from_numpy(w).double().T + torch.from_numpy(bias).double()
    ref[:, :N // 2] *= 0.5
    ref = torch.nn.functional.gelu(ref) + torch.from_numpy(r).double()
    err = float((torch.from_numpy(c).double() - ref).abs().max())
    report(f"gemv_{M}x{N}x{K}", max_abs_err=err)
    assert err <= 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("M,N,K,gemv", [(1, 1536, 512, 1), (3, 512, 128, 1), (8, 2048, 512, 1), (2, 51864, 384, 1)])
def test_fused_layernorm_linear_matches_torch(lib, M, N, K, gemv):
    rng = np.random.default_rng(6)
    a = (rng.standard_normal((M, K)) * 2 + 0.5).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias, gamma, beta = (rng.standard_normal(n).astype(np.float32) for n in (N, K, K))
    c = np.empty((M, N), np.float32)
    assert lib.wlk_diag_linear_ln(vp(a), vp(w), vp(bias), vp(gamma), vp(beta), M, N, K, gemv, vp(c)) == 0, \
        lib.wlk_diag_last_error()
    h = torch.nn.functional.layer_norm(torch.from_numpy(a).double(), (K,), torch.from_numpy(gamma).double(),
                                       torch.from_numpy(beta).double())
    ref = h @ torch.from_numpy(w).double().T + torch.from_numpy(bias).double()
    err = float((torch.from_numpy(c).double() - ref).abs().max())
    report(f"ln_linear_{M}x{N}x{K}_{'gemv' if gemv else 'gemm'}", max_abs_err=err)
    assert err <= 3e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("rows,d", [(1500, 512), (7, 128), (3, 384), (1, 1280)])
def test_layernorm_matches_torch(lib, rows, d):
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((rows, d)) * 3 + 1).astype(np.float32)
    g = rng.standard_normal(d).astype(np.float32)
    b = rng.standard_normal(d).astype(np.float32)
    y = np.empty_like(x)
    assert lib.wlk_diag_layernorm(vp(x), vp(g), vp(b), rows, d, vp(y)) == 0, lib.wlk_diag_last_error()
    ref = torch.nn.functional.layer_norm(torch.from_numpy(x), (d,), torch.from_numpy(g), torch.from_numpy(b)).numpy()
    err = float(np.abs(y - ref).max())
    report(f"layernorm_{rows}x{d}", max_abs_err=err)
    assert err <= 1e-5


@pytest.mark.parametrize("T,H", [(1500, 8), (1500, 2), (100, 6), (33, 1)])
def test_encoder_attention_matches_torch(lib, T, H):
    d = 64 * H
    rng = np.random.default_rng(4)
    qkv = rng.standard_normal((T, 3 * d)).astype(np.float32)
    qkv[:, :2 * d] *= 0.6
    qkv[5, :d] *= 6.0                       # a spiky query row: exercises the online-softmax rescale
    out = np.empty((T, d), np.float32)
    assert lib.wlk_diag_encoder_attention(vp(qkv), T, d, H, vp(out)) == 0, lib.wlk_diag_last_error()
    t = torch.from_numpy(qkv).double()
    q, k, v = (t[:, i * d:(i + 1) * d].view(T, H, 64).permute(1, 0, 2) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).permute(1, 0, 2).reshape(T, d)
    err = float((torch.from_numpy(out).double() - ref).abs().max())
    report(f"enc_attention_T{T}_H{H}", max_abs_err=err)
    assert err <= 2e-5


# ---------------------------------------------------------------------------------------------
# model-level parity
# ---------------------------------------------------------------------------------------------
_models = {}


def hip_model(name, seed=0):
    from whisperlivekit_amd.engine import HipWhisperModel
    if (name, seed) not in _models:
        _models[(name, seed)] = HipWhisperModel.synthetic(name, seed)
    return _models[(name, seed)]


MEL_META = H.golden_json("mel.json")


def mel128_model():
    from whisperlivekit_amd.dims import ModelDims
    from whisperlivekit_amd.engine import HipWhisperModel
    if "mel128" not in _models:
        dims = ModelDims(128, 1500, 128, 2, 1, 51866, 448, 128, 2, 1)
        _models["mel128"] = HipWhisperModel.from_state_dict(dims, synth.synth_state_dict(dims, 5), [(0, 0)])
    return _models["mel128"]


@pytest.mark.parametrize("key", sorted(MEL_META))
def test_mel_matches_reference_golden(key):
    meta = MEL_META[key]
    gold = H.golden_npz("mel.npz")
    if meta["n_mels"] == 128:
        # the 128-bin front end of the large-v3 family (m128_2s, m128_30s) without a 6 GB model: the log-mel does not depend on
        # the weights, so a micro-width model with n_mels = 128 carries it (round 5 skipped these two on the GPU)
        sess = mel128_model().new_session()
    else:
        sess = hip_model("micro.en").new_session()
    sess.append(H.mel_case_audio(meta))
    cml = sess.encode()
    assert cml == meta["content_mel_len"]
    mel = sess.export("mel").reshape(meta["n_mels"], 3000)
    worst = 0.0
    for lo, hi, ref in H.expand_mel_golden(gold[key], meta):
        worst = max(worst, float(np.abs(mel[:, lo:hi] - ref).max()))
    tail_err = 0.0
    if meta["tail_value"] is not None and meta["keep"] < 3000:
        tail_err = float(np.abs(mel[:, meta["keep"]:] - np.float32(meta["tail_value"])).max())
    report(f"mel_{key}", max_abs_err=worst, tail_err=tail_err)
    sess.close()
    assert worst <= 1e-3 and tail_err <= 1e-6          # north star: mel within 1e-3


@pytest.mark.parametrize("name", ["micro.en", "tiny.en", "base.en", "small.en", "medium.en", "large-v3"])
def test_encoder_decoder_match_reference_golden(name):
    if not os.path.exists(os.path.join(H.GOLDEN, f"numerics_{name}.npz")):
        pytest.skip(f"numerics_{name}.npz not generated")
    gold = H.golden_npz(f"numerics_{name}.npz")
    dims = MODEL_DIMS[name]
    sess = hip_model(name).new_session()
    sess.set_debug(True)
    sess.append(synth.to_pcm16_roundtrip(synth.speech_like(3.2, 11)))
    sess.encode()
    enc = sess.export("enc").reshape(1500, dims.n_audio_state)
    e_enc = float(np.abs(enc[::50] - gold["enc_rows"]).max())
    report(f"enc_{name}", max_abs_err=e_enc, abs_mean=float(np.abs(enc).mean()))
    feeds = [gold["tokens"], np.array([[31000]]), np.array([[46]])]
    worst_logit, worst_qk = 0.0, 0.0
    top_ok = True
    for si, feed in enumerate(feeds):
        sess.decode(feed, first=(si == 0), sot_index=3)
        logits = sess.export("logits_last").reshape(-1)
        worst_logit = max(worst_logit, float(np.abs(logits[H.PROBE_IDS] - gold[f"s{si}_probe"]).max()))
        top_ok &= (np.argsort(-logits, kind="stable")[:16].tolist() == gold[f"s{si}_top_ids"].tolist())
        lse = float(torch.logsumexp(torch.from_numpy(logits), -1))
        worst_logit = max(worst_logit, abs(lse - float(gold[f"s{si}_lse"])))
        if si == 0:
            sot = sess.export("logits_sot").reshape(-1)
            worst_logit = max(worst_logit, float(np.abs(sot[H.PROBE_IDS] - gold["s0_row3_probe"]).max()))
        rows = feed.shape[1]
        for (l, h) in ALIGNMENT_HEADS[name]:
            qk = sess.export(f"cross_qk:{l}").reshape(rows, dims.n_text_head, 1500)
            worst_qk = max(worst_qk, float(np.abs(qk[:, h, ::3] - gold[f"s{si}_qk_l{l}h{h}"]).max()))
    report(f"decoder_{name}", logits_max_abs_err=worst_logit, cross_qk_max_abs_err=worst_qk, top16_equal=bool(top_ok))
    sess.close()
    assert e_enc <= 1e-3 and worst_logit <= 1e-3 and worst_qk <= 1e-3 and top_ok


def test_mel_128_against_oracle():
    """128-bin front end (large-v3 family) without a 6 GB model: a micro-width model with n_mels=128."""
    from oracle import whisper_oracle as wo
    from whisperlivekit_amd.dims import ModelDims
    from whisperlivekit_amd.engine import HipWhisperModel
    dims = ModelDims(128, 1500, 128, 2, 1, 51866, 448, 128, 2, 1)
    sd = synth.synth_state_dict(dims, 5)
    model = HipWhisperModel.from_state_dict(dims, sd, [(0, 0)])
    sess = model.new_session()
    audio = synth.to_pcm16_roundtrip(synth.speech_like(2.0, 1))
    sess.append(audio)
    cml = sess.encode()
    mel = sess.export("mel").reshape(128, 3000)
    ref, rcml = wo.encoder_input_from_audio(torch.from_numpy(audio), torch.from_numpy(np.array(mel_filterbank(128))))
    gold = H.golden_npz("mel.npz")["m128_2s"]
    err = float(np.abs(mel - ref[0].numpy()).max())
    gerr = float(np.abs(mel[:, :gold.shape[1]] - gold).max())
    with torch.no_grad():
        enc_ref = wo.encoder_forward(wo.to_torch_state_dict(sd), dims, ref)
    enc = sess.export("enc").reshape(1500, 128)
    eerr = float(np.abs(enc - enc_ref[0].numpy()).max())
    report("mel128", max_abs_err=err, golden_err=gerr, enc_err=eerr)
    sess.close(); model.close()
    assert cml == rcml and err <= 1e-3 and gerr <= 1e-3 and eerr <= 1e-3


@pytest.mark.parametrize("name,beam", [("micro.en", 1), ("micro.en", 3), ("tiny.en", 1)])
def test_select_and_alignment_against_oracle(name, beam):
    """One infer's worth of decode steps: top-k log-probs, no-speech prob, the AlignAtt row and
    the attended frame against the oracle on identical tokens."""
    from oracle import whisper_oracle as wo
    dims = MODEL_DIMS[name]
    sd = H.oracle_sd(name)
    audio = synth.to_pcm16_roundtrip(synth.speech_like(5.0, 7))
    sess = hip_model(name).new_session(beam=beam)
    sess.append(audio)
    cml = sess.encode()
    with torch.no_grad():
        mel, rcml = wo.encoder_input_from_audio(torch.from_numpy(audio), torch.from_numpy(np.array(mel_filterbank(dims.n_mels))))
        enc = wo.encoder_forward(sd, dims, mel)
    assert cml == rcml
    cache = wo.DecoderCache(dims.n_text_layer)
    rng = np.random.default_rng(5)
    toks = np.tile(np.array([[50257, 50362] + rng.integers(300, 40000, 9).tolist()]), (beam, 1))
    toks[:, -1] += np.arange(beam)           # beams differ in the last token
    kept = []
    worst = dict(lp=0.0, attn=0.0, nsp=0.0)
    frames_equal = ids_equal = True
    for step in range(20):                   # > 16 steps: the prefill rows leave the window
        feed = toks if step == 0 else toks[:, -1:]
        sess.decode(feed, first=(step == 0), sot_index=0)
        with torch.no_grad():
            logits, cross = wo.decoder_forward(sd, dims, torch.from_numpy(feed), enc, cache)
        kept = (kept + [cross])[-16:]
        if step == 0:
            nsp = sess.no_speech_prob(50361)
            ref = logits[:, 0].float().softmax(-1)[:, 50361].numpy()
            worst["nsp"] = max(worst["nsp"], float(np.abs(nsp - ref).max() / ref.max()))
        last = logits[:, -1].clone()
        adj_ids = [50256, 220, 50257, 50362, int(toks[0, -1])]
        adj_d = [-np.inf, -np.inf, -np.inf, -np.inf, -2.0]
        for t, dl in zip(adj_ids, adj_d):
            last[:, t] += dl
        lp_ref, id_ref = torch.log_softmax(last, -1).topk(beam + 1)
        lp, ids, frames = sess.select([-1] * 5, adj_ids, adj_d, beam + 1, cml)
        attn_ref = wo.alignatt_attention(kept, ALIGNMENT_HEADS[name], dims.n_text_layer, cml, beam)
        attn = sess.export("attn_last").reshape(beam, 1500)[:, :cml]
        worst["lp"] = max(worst["lp"], float(np.abs(lp - lp_ref.numpy()).max()))
        worst["attn"] = max(worst["attn"], float(np.abs(attn - attn_ref[:, -1].numpy()).max()))
        ids_equal &= ids.tolist() == id_ref.tolist()
        frames_equal &= frames.tolist() == attn_ref[:, -1].argmax(-1).tolist()
        nxt = id_ref[:, 0].numpy()
        toks = np.concatenate([toks, nxt[:, None]], axis=1)
    report(f"select_{name}_beam{beam}", **worst, ids_equal=bool(ids_equal), frames_equal=bool(frames_equal))
    sess.close()
    assert worst["lp"] <= 1e-3 and worst["attn"] <= 1e-3 and worst["nsp"] <= 1e-3 and ids_equal and frames_equal


# ---------------------------------------------------------------------------------------------
# end to end: the reference's golden streams through the real backend (bit-exact ids/frames/words)
# ---------------------------------------------------------------------------------------------
from test_oracle_golden import STREAMS, check_stream_against_golden, replay_stream  # noqa: E402


def make_hip_processor(model_name, cfg_over, seed=0):
    from test_policy_golden import RecordingProcessor
    from whisperlivekit_amd.backend import HipSimulStreamingASR
    asr = HipSimulStreamingASR(model_name, hip_model=hip_model(model_name, seed), **H.asr_kwargs(cfg_over))
    return RecordingProcessor(asr)


# GPU-only goldens: the exact workload bench.py times (base.en, 30 s, seeds 0..7) and config 3 at FULL depth
# (large-v3: 32 + 32 layers, 1280 wide, 128 mels) - generated by the unmodified reference on CPU, too slow for the
# CPU oracle suite (tests/test_oracle_golden.py replays a prefix of one of them)
# bench_large-v3_30s_s8 (round 6): config 3 at BASELINE.json's own length, the audio seed on which the seeded large-v3 weights
# commit 47 words over 9 calls (scripts/lv3_seed_scan.py) - what bench.py's `large_v3` leg times
GPU_STREAMS = [f"bench_base_30s_s{i}" for i in range(8)] + ["large_v3_2s", "bench_large-v3_30s_s8"]


def with_teacher(make, teacher):
    """Processor factory whose hooks force the decisions in ``teacher`` to the reference's side (H.run_resynced)."""
    def mk(model_name, cfg_over, seed=0):
        proc = make(model_name, cfg_over, seed)
        proc.model.teacher = dict(teacher) or None
        return proc
    return mk


@pytest.mark.parametrize("case", STREAMS + GPU_STREAMS)
def test_stream_matches_reference_golden(case):
    """Every decision of the stream against the reference's.  A divergence inside an fp32 tie (reference margin < TIE_EPS)
    does not end the comparison: the stream is replayed with the reference's choice forced at that step and the rest is
    compared too (H.run_resynced) - the report lists the ties, nothing stays unchecked."""
    if not H.golden_exists(f"stream_{case}.json"):
        pytest.skip(f"golden stream {case} not generated")
    g0 = H.golden_json(f"stream_{case}.json")
    open_procs = []

    def run(teacher):
        g, proc, got = replay_stream(case, with_teacher(make_hip_processor, teacher))
        open_procs.append(proc)
        return g, proc, got

    def compare(res):
        g, proc, got = res
        return check_stream_against_golden(g, proc.trace, got, tol=1e-3, allow_ties=True)

    try:
        (g, proc, got), ties = H.run_resynced(run, g0, compare)
        n_steps = sum(len(r["steps"]) for r in proc.trace)
        agree = total = 0
        for rec, ref in zip(proc.trace, g["calls"]):
            for st, rs in zip(rec["steps"], ref["steps"]):
                if rs.get("token") is not None and "token" in st:
                    total += 1
                    agree += int(st["token"] == rs["token"] and st.get("frame") == rs["frame"])
        report(f"stream_{case}", decode_steps=n_steps, compared=total, agree=agree, tie_divergences=ties,
               forced_decisions=len(ties), unchecked_calls=0, last_error=repr(getattr(proc, "last_error", None)))
        assert agree == total and len(proc.trace) == len(g["calls"])
    finally:
        for p in open_procs:
            p.close()


def make_hip_loop_processor(model_name, cfg_over, seed=0):
    from whisperlivekit_amd import policy as P
    from whisperlivekit_amd.backend import HipSimulStreamingASR, HipSimulStreamingOnlineProcessor
    asr = HipSimulStreamingASR(model_name, hip_model=hip_model(model_name, seed), **H.asr_kwargs(cfg_over))

    class P2(HipSimulStreamingOnlineProcessor):
        def new_speaker(self, speaker, start):
            return super().new_speaker(P.ChangeSpeaker(speaker=speaker, start=start))

    proc = P2(asr)
    proc.model.decision_log = []
    assert proc.model.device_loop_available()
    return proc


@pytest.mark.parametrize("case", [c for c in STREAMS + GPU_STREAMS if "beam" not in c])
def test_stream_through_library_decode_loop(case):
    """the same golden streams with the per-token loop inside the library (wlk_decode_until_stop):
    identical decisions, words and end state; no Python between tokens.  fp32 ties are re-synchronised through
    wlk_loop_params.force_* and the rest of the stream is compared as well."""
    from test_policy_golden import check_loop_stream
    if not H.golden_exists(f"stream_{case}.json"):
        pytest.skip(f"golden stream {case} not generated")
    g0 = H.golden_json(f"stream_{case}.json")
    open_procs = []

    def run(teacher):
        g, proc, got = replay_stream(case, with_teacher(make_hip_loop_processor, teacher))
        open_procs.append(proc)
        emitted = [[(t.start, t.end, t.text) for t in toks] for ev, toks, _ in got if ev["kind"] == "chunk"]
        return g, proc, got, H.compare_decisions(g, proc.model.decision_log, emitted)

    def compare(res):
        r = res[3]
        assert r["mismatch"] is None, r
        return r["tie_divergence"]

    try:
        (g, proc, got, r), ties = H.run_resynced(run, g0, compare)
        report(f"loop_stream_{case}", **{k: (list(v) if isinstance(v, list) else v) for k, v in r.items()},
               tie_divergences=ties, unchecked_calls=len(g["calls"]) - r["calls"],
               last_error=repr(getattr(proc, "last_error", None)))
        assert r["calls"] == len(g["calls"]) and r["identical"] == r["decisions"] and r["words_identical"], r
        check_loop_stream(g, proc, got)
    finally:
        for p in open_procs:
            p.close()


@pytest.mark.parametrize("case", ["micro_12s", "base_4s", "bench_base_30s_s0"])
def test_early_zscore_beside_the_vocabulary_projection_changes_nothing(case, monkeypatch):
    """A graph-replayed step runs the AlignAtt z-score as side workgroups of the vocabulary projection's launch, the medians /
    arg-max beside the top-k slice pass, and ends in the 64-thread fold alone (select.hip: select_stage1_early_kernel).  Same
    device functions in other launches: every token, attended frame and summed log-probability of a stream must be the bits
    of the two-launch read-out behind the logits (WLK_EARLY_Z=0, read when a session captures its step graph)."""
    if not H.golden_exists(f"stream_{case}.json"):
        pytest.skip(f"golden stream {case} not generated")
    runs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("WLK_EARLY_Z", flag)
        g, proc, got = replay_stream(case, make_hip_loop_processor)
        try:
            words = [[(t.start, t.end, t.text) for t in toks] for ev, toks, _ in got if ev["kind"] == "chunk"]
            runs.append((proc.model.decision_log, words))
            assert sum(len(d[1]) for d in proc.model.decision_log) > 0
        finally:
            proc.close()
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1]


def test_barrier_free_single_row_gemv_is_bit_identical_to_the_staged_kernel(tmp_path):
    """gemv1_f32_kernel (beam-1 decode steps: no LDS, no barrier) keeps the staged kernel's reduction and fmaf
    order, so switching it off (WLK_NO_GEMV1, read once per process) must not change a single bit."""
    import subprocess
    import sys
    code = r"""
import ctypes as C, sys, numpy as np
from whisperlivekit_amd import _lib
lib = _lib.load()
vp = lambda a: a.ctypes.data_as(C.c_void_p)
rng = np.random.default_rng(5)
out = []
for (N, K, flags, ln) in [(512, 512, 2, 0), (2048, 512, 1, 1), (512, 2048, 2, 0), (1536, 512, 4, 1), (384, 384, 0, 1),
                          (1280, 1280, 2, 0), (130, 768, 0, 1), (3840, 1280, 0, 1), (5120, 1280, 1, 1), (768, 3072, 2, 0),
                          (1024, 4096, 2, 0), (1280, 5120, 2, 0), (1100, 1536, 0, 0), (2304, 768, 0, 1)]:
    a = rng.standard_normal((1, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    r = rng.standard_normal((1, N)).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    be = (0.1 * rng.standard_normal(K)).astype(np.float32)
    c = np.empty((1, N), np.float32)
    if ln:
        rc = lib.wlk_diag_linear_ln(vp(a), vp(w), vp(b), vp(g), vp(be), 1, N, K, 1, vp(c))
    else:
        rc = lib.wlk_diag_linear(vp(a), K, K, vp(w), vp(b), vp(r) if flags & 2 else None, N, 1, N, K, flags, 0.5, N // 2, 1, vp(c))
    assert rc == 0
    out.append(c.copy())
np.save(sys.argv[1], np.concatenate([o.reshape(-1) for o in out]))
"""
    import os
    outs = []
    for tag, extra in (("new", {}), ("old", {"WLK_NO_GEMV1": "1"})):
        path = str(tmp_path / f"{tag}.npy")
        env = dict(os.environ, **extra)
        env["PYTHONPATH"] = os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env.get("PYTHONPATH", "")])
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=300)
        outs.append(np.load(path))
    assert outs[0].shape == outs[1].shape and np.array_equal(outs[0], outs[1])


def test_prefill_gemm_fuses_the_layernorm():
    """gemm_nt_f32_kwave16_kernel<NPL> (decoder prefill: ln1 -> qkv, lnx -> xq, ln2 -> fc1 in one launch) derives the row
    statistics the way layernorm_kernel does and normalises with the same expression: LayerNorm launch + projection
    launch, bit for bit, at every Whisper width and for ragged row counts."""
    lib = _lib.load()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rng = np.random.default_rng(23)
    for (M, N, K) in [(60, 1536, 512), (60, 512, 512), (61, 2048, 512), (9, 512, 512), (128, 1152, 384), (33, 768, 768),
                      (17, 3072, 1024), (60, 1280, 1280)]:
        a = (rng.standard_normal((M, K)) * 3 + 0.5).astype(np.float32)
        w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        g = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
        be = (0.1 * rng.standard_normal(K)).astype(np.float32)
        fused = np.empty((M, N), np.float32)
        assert lib.wlk_diag_linear_ln(vp(a), vp(w), vp(b), vp(g), vp(be), M, N, K, 0, vp(fused)) == 0, (lib.wlk_diag_last_error(), M, N, K)
        y = np.empty((M, K), np.float32)
        assert lib.wlk_diag_layernorm(vp(a), vp(g), vp(be), M, K, vp(y)) == 0
        two = np.empty((M, N), np.float32)
        assert lib.wlk_diag_linear(vp(y), K, M * K, vp(w), vp(b), None, N, M, N, K, 0, 1.0, 0, 0, vp(two)) == 0
        assert np.array_equal(fused, two), (M, N, K, float(np.abs(fused - two).max()))


def test_valu_wave_butterflies_equal_the_shuffle_loops():
    """csrc/wave_ops.h: sum / max / 16-lane sum / xor exchanges / arg-max through DPP and v_permlane{16,32}_swap against
    the __shfl_xor loops they replace in every latency-bound kernel - bit for bit, on values whose sums round."""
    lib = _lib.load()
    rng = np.random.default_rng(11)
    for trial in range(8):
        x = (rng.standard_normal(64) * 10.0 ** rng.integers(-3, 4, 64)).astype(np.float32)
        if trial == 7:
            x[:] = np.float32(1.5)                      # all ties: the arg-max must name lane 0
        out = np.empty((10, 64), np.float32)
        ref = np.empty((10, 64), np.float32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        assert lib.wlk_diag_wave_ops(vp(x), vp(out), vp(ref)) == 0, lib.wlk_diag_last_error()
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), np.argwhere(out.view(np.uint32) != ref.view(np.uint32))
        assert np.array_equal(out[8], np.roll(x, 32)) and np.array_equal(out[3], x.reshape(32, 2)[:, ::-1].reshape(-1))
        assert out[9, 0] == float(np.argmax(x))


def test_pcm16_upload_equals_float_upload():
    """wlk_audio_append_pcm16 == convert_pcm_to_float (audio_processor.py:416-418) + wlk_audio_append, bit for bit
    (the widening is an exact power-of-two scale), including a chunk larger than the pinned staging buffer."""
    from whisperlivekit_amd import synth
    m = hip_model("micro.en")
    audio = synth.speech_like(40.0, 3)
    pcm = np.clip(np.round(audio * 32768.0), -32768, 32767).astype(np.int16)
    mels = []
    for mode in ("float", "pcm16"):
        sess = m.new_session(beam=1, max_audio_seconds=64.0)
        for lo, hi in ((0, 1), (1, 8000), (8000, 8000 + 600001)):
            if mode == "float":
                sess.append(pcm[lo:hi].astype(np.float32) / 32768.0)
            else:
                sess.append_pcm16(pcm[lo:hi])
        assert sess.audio_len == 608001
        sess.drop_front(608001 - 480000)
        cml = sess.encode()
        mels.append((cml, sess.export("mel").copy()))
        sess.close()
    assert mels[0][0] == mels[1][0] == 1500
    assert np.array_equal(mels[0][1], mels[1][1])


def test_hipgraph_replay_equals_eager_launches(monkeypatch):
    """The captured decode-step graph and the eager launch sequence are the same arithmetic: identical
    tokens, frames and log-prob sums, bit for bit, over a whole stream."""
    def run():
        g, proc, got = replay_stream("micro_12s", make_hip_processor)
        trace = [(r["content_mel_len"], [(s.get("token"), s.get("frame"), s.get("sum_logprob")) for s in r["steps"]])
                 for r in proc.trace]
        words = [[(t.start, t.end, t.text) for t in toks] for _, toks, _ in got]
        proc.close()
        return trace, words
    with_graph = run()
    monkeypatch.setenv("WLK_NO_GRAPH", "1")
    eager = run()
    assert with_graph == eager


@pytest.mark.parametrize("case", ["micro_12s", "tiny_6s", "base_4s", "micro_cif", "large_v3_2s"])
def test_merge_folded_into_out_projection_is_bit_identical(monkeypatch, case):
    """Beam-1 steps: the merge of the 8 key splits of the cross-attention is the A-operand load of the out-projection
    GEMV (and spare workgroups write the alignment rows) instead of a kernel of its own - same arithmetic, same order.
    Up to d = 512 every wave merges its own slice (gemv1_f32_kernel<MG>), beyond it the workgroup merges once through LDS
    (gemv1_mgl_f32_kernel: large-v3)."""
    if not H.golden_exists(f"stream_{case}.json"):
        pytest.skip(f"golden stream {case} not generated")
    def run():
        g, proc, got = replay_stream(case, make_hip_processor)
        trace = [(r["content_mel_len"], [(s.get("token"), s.get("frame"), s.get("sum_logprob")) for s in r["steps"]])
                 for r in proc.trace]
        words = [[(t.start, t.end, t.text) for t in toks] for _, toks, _ in got]
        proc.close()
        return trace, words
    folded = run()
    monkeypatch.setenv("WLK_NO_MERGE_FOLD", "1")
    separate = run()
    assert folded == separate
    assert sum(len(steps) for _, steps in folded[0]) > (5 if case == "large_v3_2s" else 20)


@pytest.mark.parametrize("case", ["micro_12s", "base_4s", "micro_prompt"])
def test_prefill_layernorm_fused_into_the_projection_is_bit_identical(monkeypatch, case):
    """WLK_PREFILL_LN_FUSE=1: the decoder prefill's three LayerNorm launches per layer run inside the 16 x 16 projection
    kernel - same statistics, same expression, so whole streams keep every token, frame and log-prob sum."""
    def run():
        g, proc, got = replay_stream(case, make_hip_processor)
        trace = [(r["content_mel_len"], [(s.get("token"), s.get("frame"), s.get("sum_logprob")) for s in r["steps"]])
                 for r in proc.trace]
        words = [[(t.start, t.end, t.text) for t in toks] for _, toks, _ in got]
        proc.close()
        return trace, words
    from whisperlivekit_amd import _lib
    separate = run()
    monkeypatch.setenv("WLK_PREFILL_LN_FUSE", "1")
    _lib.load().wlk_diag_env_refresh()          # the library reads the switch once and caches it
    try:
        fused = run()
    finally:
        monkeypatch.delenv("WLK_PREFILL_LN_FUSE")
        _lib.load().wlk_diag_env_refresh()
    assert fused == separate
    assert sum(len(steps) for _, steps in fused[0]) > 10


def test_transcribe_wav_script_streams_a_file(tmp_path):
    """scripts/transcribe_wav.py: s16le WAV -> insert_pcm16_chunk / process_iter loop; same words as the float path."""
    import sys
    import wave
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import transcribe_wav
    audio = synth.speech_like(6.0, 0)
    pcm = np.clip(np.round(audio * 32768.0), -32768, 32767).astype(np.int16)
    path = str(tmp_path / "a.wav")
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
    words = transcribe_wav.main([path, "--synthetic", "micro.en"])
    from whisperlivekit_amd.backend import HipSimulStreamingASR, HipSimulStreamingOnlineProcessor
    proc = HipSimulStreamingOnlineProcessor(HipSimulStreamingASR("micro.en", hip_model=hip_model("micro.en")))
    ref = []
    as_float = pcm.astype(np.float32) / 32768.0
    for lo in range(0, len(pcm), 8000):
        proc.insert_audio_chunk(as_float[lo:lo + 8000], min(lo + 8000, len(pcm)) / 16000)
        ref += proc.process_iter()[0]
    ref += proc.process_iter(is_last=True)[0]
    proc.close()
    assert [(t.start, t.end, t.text) for t in words] == [(t.start, t.end, t.text) for t in ref]


def test_diarization_melspec_against_oracle():
    """a12 front end: 128-bin log-mel of 1 s chunks (NeMo FilterbankFeatures config) vs the torch restatement."""
    from oracle.sortformer_oracle import nemo_log_mel
    from whisperlivekit_amd.diarization import HipMelSpectrogram
    mel = HipMelSpectrogram()
    filters = np.array(mel_filterbank(128, 16000, 512))
    worst = 0.0
    for seed, n in ((0, 16000), (1, 16000), (2, 12345), (3, 400)):
        pcm = synth.to_pcm16_roundtrip(synth.speech_like(1.0, seed))[:n]
        got = mel(pcm)
        ref = nemo_log_mel(pcm, filters)
        assert got.shape == ref.shape == (n // 160 + 1, 128)
        worst = max(worst, float(np.abs(got - ref).max()))
    silence = mel(np.zeros(16000, np.float32))
    report("diar_melspec", max_abs_err=worst, silence_value=float(silence[5, 5]))
    mel.close()
    assert worst <= 1e-3
    # 100 valid frames of log(2^-24) and the zero-filled frame behind them (FilterbankFeatures.get_seq_len = len // hop)
    assert np.allclose(silence[:100], np.log(np.float32(2.0 ** -24)), atol=1e-5) and np.all(silence[100] == 0.0)


def test_large_v3_shapes_smoke():
    """Config 3 (large-v3: d=1280, 20 heads, 32+32 layers, 128 mels, 51866 tokens) with a 2-layer stand-in
    of the same widths: exercises the 128-mel conv1, the 1280-wide GEMM/GEMV/LayerNorm paths and the
    5120-wide MLP against the oracle."""
    from oracle import whisper_oracle as wo
    from whisperlivekit_amd.dims import ModelDims
    from whisperlivekit_amd.engine import HipWhisperModel
    dims = ModelDims(128, 1500, 1280, 20, 2, 51866, 448, 1280, 20, 2)
    sd = synth.synth_state_dict(dims, 9)
    model = HipWhisperModel.from_state_dict(dims, sd, [(1, 3), (1, 17)])
    sess = model.new_session()
    audio = synth.to_pcm16_roundtrip(synth.speech_like(3.0, 4))
    sess.append(audio)
    cml = sess.encode()
    sdt = wo.to_torch_state_dict(sd)
    with torch.no_grad():
        mel, rcml = wo.encoder_input_from_audio(torch.from_numpy(audio), torch.from_numpy(np.array(mel_filterbank(128))))
        enc = wo.encoder_forward(sdt, dims, mel)
    e_enc = float(np.abs(sess.export("enc").reshape(1500, 1280) - enc[0].numpy()).max())
    toks = np.array([[50258, 50259, 50360, 50364, 1000, 2000, 3000, 4000, 5000, 6000]])
    cache = wo.DecoderCache(dims.n_text_layer)
    worst = 0.0
    for step in range(3):
        feed = toks if step == 0 else toks[:, -1:]
        sess.decode(feed, first=(step == 0), sot_index=0)
        with torch.no_grad():
            logits, _ = wo.decoder_forward(sdt, dims, torch.from_numpy(feed), enc, cache)
        got = sess.export("logits_last").reshape(-1)
        worst = max(worst, float(np.abs(got - logits[0, -1].numpy()).max()))
        toks = np.concatenate([toks, [[int(got.argmax())]]], axis=1)
        assert int(got.argmax()) == int(logits[0, -1].argmax())
    report("large_v3_widths", enc_err=e_enc, logits_err=worst)
    sess.close(); model.close()
    assert cml == rcml and e_enc <= 1e-3 and worst <= 1e-3


def test_incremental_mel_is_bit_identical_to_full_recompute(monkeypatch):
    """SURVEY 8f rank 2 (mel half): per-frame log-mel results stay in the session and only the frames the new chunk
    touches are recomputed (+ the two reflect-padded head frames after an eviction of whole frames; anything else
    invalidates the cache).  Against a session that recomputes everything (WLK_MEL_INCREMENTAL=0): identical mel and
    encoder output, bit for bit, through appends of ragged sizes, zero runs, frame-aligned and unaligned evictions,
    a clear, and a buffer longer than 30 s."""
    m = hip_model("micro.en")
    rng = np.random.default_rng(3)
    audio = synth.to_pcm16_roundtrip(synth.speech_like(70.0, 9))
    monkeypatch.setenv("WLK_MEL_INCREMENTAL", "1")
    inc = m.new_session(beam=1, max_audio_seconds=64.0, batched=False)
    monkeypatch.setenv("WLK_MEL_INCREMENTAL", "0")
    full = m.new_session(beam=1, max_audio_seconds=64.0, batched=False)
    pos = 0
    script = [("a", 8000), ("a", 8000), ("a", 123), ("a", 1), ("z", 4000), ("a", 16000), ("d", 8000), ("a", 8000), ("d", 160),
              ("a", 333), ("d", 777), ("a", 8000), ("a", 480000), ("d", 16000), ("a", 8000), ("d", 8000), ("a", 8000),
              ("c", 0), ("a", 5000), ("a", 8000), ("d", 3200), ("a", 8000)]
    n_enc = 0
    for op, n in script:
        for s in (inc, full):
            if op == "a":
                s.append(audio[pos:pos + n])
            elif op == "z":
                s.append_zeros(n)
            elif op == "d":
                s.drop_front(n)
            else:
                s.clear_audio()
        if op == "a":
            pos += n
        if inc.audio_len == 0:
            continue
        c1, c2 = inc.encode(), full.encode()
        assert c1 == c2
        assert np.array_equal(inc.export("mel"), full.export("mel")), (op, n, inc.audio_len)
        n_enc += 1
    assert np.array_equal(inc.export("enc"), full.export("enc"))
    assert n_enc >= 20
    inc.close(); full.close()

"""a12 on the GPU: the Sortformer network through the C ABI (wlk_sf_*) against the torch-CPU restatement in
oracle/sortformer_oracle.py on the same seeded weights and inputs, and (round 5) against known answers from the independent
ports of NeMo's modules in `transformers` - ParakeetFeatureExtractor (log-mel), ParakeetEncoder (sub-sampling stem,
FastConformer) - in tests/golden/sortformer_hf_kat.npz.  Still unpinned with respect to NeMo itself (no NeMo / checkpoint
offline): the wiring of the Transformer blocks + sigmoid head and the speaker-cache update; there HIP path == oracle is
what is checked.  fp32 tolerance stated per test."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import sortformer_oracle as so
from whisperlivekit_amd import _lib
from whisperlivekit_amd import sortformer as sf
from whisperlivekit_amd.diarization import HipSortformerDiarizationOnline
from whisperlivekit_amd.synth import speech_like

pytestmark = pytest.mark.gpu


def as_torch(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def oracle_dims(d):
    return so.SortformerDims(n_mels=d.n_mels, fc_d_model=d.fc_d_model, fc_layers=d.fc_layers, fc_heads=d.fc_heads,
                             conv_kernel=d.conv_kernel, sub_channels=d.sub_channels, tf_d_model=d.tf_d_model,
                             tf_layers=d.tf_layers, tf_heads=d.tf_heads, tf_inner=d.tf_inner, n_spk=d.n_spk)


def logmel_like(rng, n):
    return (rng.standard_normal((n, 128)) * 2.5 - 9.0).astype(np.float32)


@pytest.fixture(scope="module")
def shallow():
    dims = sf.SortformerDims(fc_layers=2, tf_layers=2)
    sd = sf.synth_sortformer_state_dict(dims, 11)
    m = sf.HipSortformerModel(dims, sd)
    yield dims, as_torch(sd), m
    m.close()


@pytest.fixture(scope="module")
def full():
    dims = sf.SortformerDims()
    sd = sf.synth_sortformer_state_dict(dims, 12)
    m = sf.HipSortformerModel(dims, sd)
    yield dims, as_torch(sd), m
    m.close()


@pytest.mark.parametrize("n_feat", [200, 101, 9, 8])
def test_subsampling_stem_matches_oracle(shallow, n_feat):
    """ConvSubsampling (conv0, 2 x depthwise+pointwise, Linear over (freq, channel)); tolerance 1e-4 relative to the
    embedding scale (K = 4096 fp32 accumulation in a different order than torch's)."""
    dims, tsd, m = shallow
    feats = logmel_like(np.random.default_rng(n_feat), n_feat)
    chunk, preds = m.step(feats, None)
    ref = so.pre_encode(tsd, oracle_dims(dims), torch.from_numpy(feats)).numpy()
    sub = lambda n: (n - 1) // 2 + 1
    assert chunk.shape == ref.shape == (sub(sub(sub(n_feat))), 512)
    assert np.abs(chunk - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    assert preds.shape == (ref.shape[0], 4) and np.all((preds >= 0) & (preds <= 1))


@pytest.mark.parametrize("n_ctx,n_feat", [(300, 200), (0, 200), (1, 0), (37, 0), (376, 200)])
def test_network_matches_oracle_two_blocks(shallow, n_ctx, n_feat):
    """2 Conformer + 2 Transformer blocks at full width: Conformer output, Transformer output and activities.
    Tolerances: 2e-4 absolute on LayerNorm-ed activations (O(1)), 1e-4 on the sigmoid outputs."""
    dims, tsd, m = shallow
    rng = np.random.default_rng(100 + n_ctx)
    ctx = (rng.standard_normal((n_ctx, 512)) * 0.3).astype(np.float32) if n_ctx else None
    feats = logmel_like(rng, n_feat) if n_feat else None
    chunk, preds = m.step(feats, ctx)
    od = oracle_dims(dims)
    parts = ([torch.from_numpy(ctx)] if n_ctx else []) + ([so.pre_encode(tsd, od, torch.from_numpy(feats))] if n_feat else [])
    embs = torch.cat(parts, 0)
    T = embs.shape[0]
    x = embs * math.sqrt(512)
    pos = so.rel_positional_encoding(T, 512)
    for i in range(dims.fc_layers):
        x = so.conformer_layer(tsd, f"encoder.layers.{i}.", od, x, pos)
    fc = m.export("fc_out")
    assert fc.shape == (T, 512)
    assert np.abs(fc - x.numpy()).max() <= 2e-4
    ref = so.forward_embeddings(tsd, od, embs).numpy()
    assert preds.shape == ref.shape
    assert np.abs(preds - ref).max() <= 1e-4
    assert ref.std() > 0.01


def _vp(a):
    import ctypes as C
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("N,K,flags", [(2048, 512, 16), (512, 2048, 2), (1536, 512, 0), (512, 512, 2), (1024, 512, 0), (192, 768, 2),
                                       (192, 512, 0), (256, 256, 8), (576, 192, 4), (768, 192, 8), (192, 192, 2)])
def test_kp_family_rows_do_not_depend_on_what_is_stacked_under_them(N, K, flags):
    """Round 6: every projection of the Sortformer goes through launch_gemm_kp - 32 x 32 k-wave tiles below 512 rows,
    one-tile-per-CU k-pipe tiles from there on, ONE per-element arithmetic (wave w sums k = 32 t + 8 w .. + 7 of every slab,
    partials folded in wave order; K = 192 the same with a half-empty last slab, on k-wave tiles at every row count) - so the rows of one session come out bit for bit the
    same alone (M = 291, 401, 37) and stacked with other sessions' rows (M = 2 392: a k-pipe tile).  Also against float64."""
    lib = _lib.load()
    rng = np.random.default_rng(N + K)
    M = 2392
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(N)).astype(np.float32)
    r = rng.standard_normal((M, N)).astype(np.float32)

    def run(rows, force=5):
        c = np.empty((rows, N), np.float32)
        rc = lib.wlk_diag_linear(_vp(a), K, rows * K, _vp(w), _vp(bias), _vp(r) if flags & 2 else None, N, rows, N, K, flags,
                                 0.5, N // 2, force, _vp(c))
        assert rc == 0, lib.wlk_diag_last_error()
        return c

    big = run(M)
    ref = a.astype(np.float64) @ w.astype(np.float64).T + bias
    if flags & 4:
        ref[:, : N // 2] *= 0.5
    if flags & 8:
        ref = np.maximum(ref, 0)
    if flags & 16:
        ref = ref / (1 + np.exp(-ref))
    if flags & 2:
        ref = ref + r
    assert np.abs(big - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    for rows in (291, 401, 37, 600):
        assert np.array_equal(run(rows).view(np.uint32), big[:rows].view(np.uint32)), (rows, N, K)
    if K % 128 == 0:          # below 512 rows the family has two tile shapes (16 x 16: force 6, 32 x 32: force 7): same bits
        for rows in (291, 37):
            for force in (6, 7, 8):      # 8: the 32 x 32 tiles with one LDS slab buffer (three to four workgroups per CU)
                assert np.array_equal(run(rows, force).view(np.uint32), big[:rows].view(np.uint32)), (rows, N, K, force)


def test_concurrent_steps_on_one_model_equal_serial_steps(shallow):
    """Round 5: a step runs in one of the model's workspaces on that workspace's stream, so sessions that share the model
    (config 4) overlap instead of queueing behind one set of buffers.  Round 6: steps that wait for the model at the same time
    run as ONE stacked launch chain (rows of up to eight sessions one after the other, ragged lengths, sessions without a
    feature chunk among them).  Eight threads x five steps of different shapes on one model: every result bit-identical to
    the same step run alone, and the steps did get stacked."""
    import threading
    dims, _tsd, m = shallow
    rng = np.random.default_rng(77)
    jobs = []
    for t in range(8):
        for k in range(5):
            n_feat = int(rng.choice([0, 9, 101, 200]))
            n_ctx = int(rng.integers(1 if n_feat == 0 else 0, 300))
            feats = logmel_like(np.random.default_rng(1000 * t + k), n_feat) if n_feat else None
            ctx = (0.3 * np.random.default_rng(2000 * t + k).standard_normal((n_ctx, 512))).astype(np.float32) if n_ctx else None
            jobs.append((t, feats, ctx))
    serial = [m.step(f, c) for _t, f, c in jobs]
    before = m.stats()
    got = [None] * len(jobs)
    errors = []

    def worker(t):
        try:
            for i, (jt, f, c) in enumerate(jobs):
                if jt == t:
                    got[i] = m.step(f, c)
        except Exception as e:            # pragma: no cover
            errors.append(e)
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [th.start() for th in threads]
    [th.join() for th in threads]
    assert not errors, errors
    for (chunk_s, preds_s), (chunk_g, preds_g) in zip(serial, got):
        assert np.array_equal(chunk_s.view(np.uint32), chunk_g.view(np.uint32))
        assert np.array_equal(preds_s.view(np.uint32), preds_g.view(np.uint32))
    after = m.stats()
    steps, sessions = after["stacked_steps"] - before["stacked_steps"], after["session_steps"] - before["session_steps"]
    assert sessions == len(jobs) and steps < sessions, (before, after)      # several sessions per launch chain


def test_stacked_step_of_eight_sessions_equals_eight_steps_alone(full):
    """Full depth (17 + 18 blocks), eight sessions released together so that they land in one or two stacked chains of
    M ~ 2 400 rows (the k-pipe tiles), each with its own context length: activities and chunk embeddings bit-identical to
    the sessions' steps alone, and <= 1e-4 from the torch oracle for one of them."""
    import threading
    dims, tsd, m = full
    rng = np.random.default_rng(5)
    jobs = []
    for t in range(8):
        n_ctx = int(rng.integers(200, 377))
        jobs.append((logmel_like(np.random.default_rng(300 + t), 200), (0.3 * rng.standard_normal((n_ctx, 512))).astype(np.float32)))
    serial = [m.step(f, c) for f, c in jobs]
    before = m.stats()
    gate = threading.Barrier(8)
    got = [None] * 8

    def worker(t):
        gate.wait()
        got[t] = m.step(*jobs[t])
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [th.start() for th in threads]
    [th.join() for th in threads]
    after = m.stats()
    assert after["stacked_steps"] - before["stacked_steps"] <= 4, (before, after)
    for (cs, ps), (cg, pg) in zip(serial, got):
        assert np.array_equal(cs.view(np.uint32), cg.view(np.uint32)) and np.array_equal(ps.view(np.uint32), pg.view(np.uint32))
    od = oracle_dims(dims)
    feats, ctx = jobs[3]
    embs = torch.cat([torch.from_numpy(ctx), so.pre_encode(tsd, od, torch.from_numpy(feats))], 0)
    with torch.no_grad():
        ref = so.forward_embeddings(tsd, od, embs).numpy()
    assert np.abs(got[3][1] - ref).max() <= 1e-4


def test_front_end_inside_the_step_equals_the_three_separate_calls(shallow):
    """Round 6: wlk_sf_step_pcm computes the chunk's log-mel rows on the step's own stream, behind the rows kept from the previous
    chunk (one launch chain, one synchronisation instead of extractor call + step).  A 6 s stream through
    HipSortformerDiarizationOnline on the fused path and on the three-call path (features, concatenate, forward_streaming_step):
    every chunk's activities, the kept feature rows and the streaming state bit-identical; so are the segments."""
    dims, _tsd, m = shallow
    audio = speech_like(6.0, 3).astype(np.float32)

    class ThreeCalls:                      # the same model without the fused entry point
        n_spk, params = m.n_spk, m.params
        features, new_state, forward_streaming_step = m.features, m.new_state, m.forward_streaming_step

    fused, plain = HipSortformerDiarizationOnline(m), HipSortformerDiarizationOnline(ThreeCalls())
    for lo in range(0, len(audio) - 15999, 16000):
        for o in (fused, plain):
            o.insert_audio_chunk(audio[lo:lo + 16000])
        sa, sb = fused.diarize_sync(), plain.diarize_sync()
        assert [(x.speaker, x.start, x.end) for x in sa] == [(x.speaker, x.start, x.end) for x in sb]
        assert np.array_equal(fused._previous_chunk_features.view(np.uint32), plain._previous_chunk_features.view(np.uint32))
        assert fused._previous_chunk_features.shape == (101, 128) and not fused._previous_chunk_features[100].any()
        assert np.array_equal(fused.total_preds.view(np.uint32), plain.total_preds.view(np.uint32))
        for name in ("spkcache", "spkcache_preds", "fifo", "fifo_preds", "mean_sil_emb"):
            assert np.array_equal(getattr(fused.streaming_state, name), getattr(plain.streaming_state, name)), name
    assert fused.total_preds.shape[0] > 50
    with pytest.raises(_lib.WlkError):      # an audio chunk longer than the step's buffer
        m.step_pcm(np.zeros(64001, np.float32), None, None)


def test_capacity_and_argument_errors(shallow):
    dims, _, m = shallow
    with pytest.raises(_lib.WlkError):
        m.step(logmel_like(np.random.default_rng(0), m.max_feat_frames + 1), None)
    with pytest.raises(_lib.WlkError):
        m.step(None, np.zeros((m.max_frames + 1, 512), np.float32))
    with pytest.raises(_lib.WlkError):
        m.step(None, None)


def test_full_depth_streaming_session_teacher_forced(full):
    """The whole a12 pass: 1 s chunks of audio -> HIP log-mel -> 17 + 18 blocks -> speaker-cache update -> segments,
    18 chunks (FIFO overflow at chunk ~9, first cache compression at ~15).  Every step's activities are compared
    with the oracle run on the SAME context (the HIP session's own speaker cache / FIFO: teacher forcing, so that a
    frame-selection flip cannot cascade); tolerance 2e-3 absolute on sigmoid outputs after 35 blocks."""
    dims, tsd, m = full
    od = oracle_dims(dims)
    steps = []
    orig = m.step_pcm

    def recording_step(pcm, prev, ctx):          # the session's fused call: log-mel of the chunk + stem + network
        feats, chunk, preds = orig(pcm, prev, ctx)
        total = feats if prev is None else np.concatenate([prev, feats], axis=0)
        steps.append((total.copy(), None if ctx is None else ctx.copy(), chunk.copy(), preds.copy()))
        return feats, chunk, preds

    m.step_pcm = recording_step
    try:
        online = HipSortformerDiarizationOnline(m)
        audio = speech_like(18.0, seed=5)
        segs = []
        for i in range(0, len(audio), 8000):
            online.insert_audio_chunk(audio[i: i + 8000])
            segs += online.diarize_sync()
    finally:
        m.step_pcm = orig
    assert len(steps) == 18
    st = online.streaming_state
    assert st.spkcache_len == 188 and 0 < st.fifo_len <= 188
    worst = 0.0
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    for k, (feats, ctx, chunk, preds) in enumerate(steps):
        assert feats.shape == ((101, 128) if k == 0 else (200, 128))
        emb_c = so.pre_encode(tsd, od, torch.from_numpy(feats))
        assert np.abs(chunk - emb_c.numpy()).max() <= 1e-4 * max(1.0, float(emb_c.abs().max()))
        embs = torch.cat(([torch.from_numpy(ctx)] if ctx is not None else []) + [torch.from_numpy(chunk)], 0)
        ref = so.forward_embeddings(tsd, od, embs).numpy()
        worst = max(worst, float(np.abs(preds - ref).max()))
    assert worst <= 2e-3, worst
    assert online.total_preds.shape == (12 + 17 * 23, 4)
    assert segs and all(s.end >= s.start for s in segs)
    assert abs(segs[-1].end - 18.0) < 0.05 and segs[0].start == 0.0
    assert all(abs(a.end - b.start) < 1e-6 for a, b in zip(segs, segs[1:]))


HF_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sortformer_hf_kat.npz")


@pytest.fixture(scope="module")
def hf_weights_model():
    """The weights scripts/gen_golden_sortformer_hf.py loaded into transformers' ParakeetEncoder: seed 0, full geometry."""
    dims = sf.SortformerDims()
    m = sf.HipSortformerModel(dims, sf.synth_sortformer_state_dict(dims, 0))
    yield m
    m.close()


def test_log_mel_against_parakeet_feature_extractor(hf_weights_model):
    """wlk_melspec_run == transformers' port of NeMo's FilterbankFeatures (<= 1e-3; the valid-frame rule and the zero
    fill behind the valid frames included), un-normalised and through the extractor's own per-feature normalisation."""
    from test_sortformer_hf_golden import FEATURE_CASES, feature_case_pcm, per_feature_normalized
    g = np.load(HF_GOLDEN)
    worst = worst_n = 0.0
    for name, seconds, seed, cut in FEATURE_CASES:
        pcm = feature_case_pcm(seconds, seed, cut)
        valid = int(g[f"feat_{name}_valid"])
        got = hf_weights_model.features(pcm)
        raw = g[f"feat_{name}_raw"]
        assert got.shape == raw.shape
        assert np.all(got[valid:] == 0.0)
        worst = max(worst, float(np.abs(got[:valid] - raw[:valid]).max()))
        if valid > 1:
            worst_n = max(worst_n, float(np.abs(per_feature_normalized(got, valid) - g[f"feat_{name}_normalized"]).max()))
    assert worst <= 1e-3 and worst_n <= 1e-3, (worst, worst_n)


def test_stem_and_fastconformer_against_parakeet_encoder(hf_weights_model):
    """wlk_sf_step == transformers' port of NeMo's FastConformer on the same seeded weights (<= 1e-3 on O(1) values):
    the dw-striding stem on the diarizer's second-chunk input (99 + 101 feature frames), the 17 Conformer blocks over
    the chunk alone, and over [95 context embeddings | chunk] (the [speaker cache | FIFO | chunk] layout)."""
    g = np.load(HF_GOLDEN)
    m = hf_weights_model
    chunk, _ = m.step(g["stem_in"], None)
    assert chunk.shape == g["stem_out"].shape
    e_stem = float(np.abs(chunk - g["stem_out"]).max())
    fc = m.export("fc_out")
    assert fc.shape == g["stack_chunk_out"].shape
    e_stack = float(np.abs(fc - g["stack_chunk_out"]).max())
    m.step(g["stem_in"], g["ctx_embs"])
    fc = m.export("fc_out")
    assert fc.shape == g["stack_ctx_out"].shape
    e_ctx = float(np.abs(fc - g["stack_ctx_out"]).max())
    assert e_stem <= 1e-4 and e_stack <= 1e-3 and e_ctx <= 1e-3, (e_stem, e_stack, e_ctx)


NEMO_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sortformer_nemo.npz")


@pytest.mark.skipif(not (os.path.exists(NEMO_GOLDEN) and os.environ.get("WLK_SORTFORMER_MODEL_PATH")),
                    reason="needs tests/golden/sortformer_nemo.npz (scripts/gen_golden_sortformer.py, run where NeMo is "
                           "installed) and the same .nemo checkpoint in WLK_SORTFORMER_MODEL_PATH")
def test_against_nemo_golden():
    """a12 PINNED: the HIP diarizer on the real diar_streaming_sortformer_4spk-v2 weights against what NeMo itself
    computed (dither 0) for the reference's two-speaker fixture signal - features, pre-encode, Conformer / Transformer
    outputs and speaker activities of every streaming step, and the emitted segments."""
    g = np.load(NEMO_GOLDEN, allow_pickle=True)
    model = sf.HipSortformerModel.from_checkpoint(os.environ["WLK_SORTFORMER_MODEL_PATH"])
    online = HipSortformerDiarizationOnline(model)
    signal = g["signal"]
    seen = []
    inner = model.step

    def step(feats, ctx):
        out = inner(feats, ctx)
        seen.append(dict(feats=feats, chunk=out[0], preds=out[1], fc=model.export("fc_out"), tf=model.export("tf_out")))
        return out

    model.step = step
    segments = []
    for lo in range(0, len(signal), 8000):
        online.insert_audio_chunk(signal[lo:lo + 8000])
        segments += online.diarize_sync()
    n = int(g["n_calls"])
    assert len(seen) == n
    worst = dict(feats=0.0, pre=0.0, fc=0.0, tf=0.0)
    for i in range(n):
        worst["feats"] = max(worst["feats"], float(np.abs(seen[i]["feats"] - g[f"c{i}_feats"]).max()))
        worst["pre"] = max(worst["pre"], float(np.abs(seen[i]["chunk"] - g[f"c{i}_pre_encode"][0]).max()))
        fc = g[f"c{i}_fc_out"]
        fc = fc[0].T if fc.ndim == 3 and fc.shape[1] == model.dims.fc_d_model else fc.reshape(-1, model.dims.fc_d_model)
        worst["fc"] = max(worst["fc"], float(np.abs(seen[i]["fc"][: fc.shape[0]] - fc).max()))
        tf = g[f"c{i}_tf_out"].reshape(-1, model.dims.tf_d_model)
        worst["tf"] = max(worst["tf"], float(np.abs(seen[i]["tf"][: tf.shape[0]] - tf).max()))
        assert online.streaming_state.spkcache_len >= 0
    got_preds = np.asarray(online.total_preds)
    ref_preds = g[f"c{n - 1}_total_preds"]
    assert got_preds.shape == ref_preds.shape
    assert np.abs(got_preds - ref_preds).max() <= 2e-3
    assert worst["feats"] <= 1e-3 and worst["pre"] <= 2e-3 and worst["fc"] <= 5e-3 and worst["tf"] <= 5e-3, worst
    want = g["segments"]
    assert [(round(s.start, 2), round(s.end, 2), int(s.speaker)) for s in segments] == \
           [(round(a, 2), round(b, 2), int(c)) for a, b, c in want]
    model.close()

"""LocalAgreement's batch Whisper: whisperlivekit_amd/transcribe.py against runs of the reference's own
`transcribe()` (tests/golden/transcribe_kat.json.gz, scripts/gen_golden_transcribe.py: seeded micro Whisper, real
vocabulary, seeded audio).

The golden holds, besides the result dictionary, every choice the reference's GreedyDecoder made (token per row and its
log-probability).  The test follows those choices step by step: at temperature 0 its own arg-max has to be the
reference's token unless the two logits are within TIE_EPS of each other; at a temperature > 0 the drawn token is taken
from the record (torch's generator is not part of the contract) and has to have the reference's log-probability.  Any
difference in the fallback decisions, prompts, window positions or segment cuts puts the replay out of step and fails.

CPU: the host logic over the oracle-backed stand-in sessions (tests/oracle_session.py).  GPU: the same checks over the
HIP library (wlk_log_mel, wlk_encode_mel, wlk_decode, wlk_kv_reorder, wlk_find_alignment)."""
import os

import numpy as np
import pytest
import torch

import helpers as H
from whisperlivekit_amd import synth, transcribe as TR

KAT = H.golden_json("transcribe_kat.json")
TIE_EPS = H.TIE_EPS
LOGPROB_ATOL = 5e-4


@pytest.fixture()
def real_vocab(tmp_path, monkeypatch):
    monkeypatch.setenv("WLK_VOCAB_DIR", H.real_vocab_dir(tmp_path))
    monkeypatch.setenv("WLK_SYNTHETIC_VOCAB", "0")


def make_audio(spec):
    if spec["kind"] == "speech_like":
        return synth.speech_like(spec["seconds"], seed=spec["seed"])
    if spec["kind"] == "white_noise":
        return synth.white_noise(spec["seconds"], seed=spec["seed"])
    return np.zeros(int(spec["seconds"] * 16000), np.float32)


class Replay:
    """Stands in for transcribe.choose: answers with the reference's recorded choice after checking it against the
    logits this implementation produced."""

    def __init__(self, calls):
        self.queue = [(c["temperature"], step, lps) for c in calls if c["beam"] is None
                      for step, lps in zip(c["steps"], c["logprobs"])]
        self.at = 0
        self.ties = 0
        self.worst_logprob = 0.0

    def __call__(self, logits, temperature):
        assert self.at < len(self.queue), "more greedy steps than the reference took"
        t, tokens, lps = self.queue[self.at]
        assert t == pytest.approx(temperature), f"step {self.at}: temperature {temperature}, the reference was at {t}"
        assert logits.shape[0] == len(tokens), f"step {self.at}: {logits.shape[0]} rows, the reference had {len(tokens)}"
        logprobs = torch.log_softmax(logits.float(), dim=-1)
        for r, (tok, lp) in enumerate(zip(tokens, lps)):
            if lp is None:            # the row had already ended: its choice is overridden by <|endoftext|>
                continue
            if temperature == 0:
                mine = int(logits[r].argmax())
                if mine != tok:
                    margin = float(logits[r, mine] - logits[r, tok])
                    assert margin <= TIE_EPS * max(1.0, abs(float(logits[r, mine]))), \
                        f"step {self.at} row {r}: arg-max {mine}, the reference chose {tok} (margin {margin:.3e})"
                    self.ties += 1
            err = abs(float(logprobs[r, tok]) - lp)
            self.worst_logprob = max(self.worst_logprob, err)
            assert err <= LOGPROB_ATOL, f"step {self.at} row {r}: log-prob of {tok} off by {err:.2e}"
        self.at += 1
        return torch.tensor(tokens, dtype=torch.int64)


def compare_result(got, want):
    assert got["language"] == want["language"]
    assert got["text"] == want["text"]
    assert len(got["segments"]) == len(want["segments"])
    for g, w in zip(got["segments"], want["segments"]):
        assert set(g) == set(w), (sorted(g), sorted(w))
        for key in ("id", "seek", "text", "tokens", "temperature", "compression_ratio"):
            assert g[key] == w[key], (w["id"], key, g[key], w[key])
        for key in ("start", "end"):
            assert g[key] == pytest.approx(w[key], abs=1e-9), (w["id"], key, g[key], w[key])
        assert g["avg_logprob"] == pytest.approx(w["avg_logprob"], abs=LOGPROB_ATOL)
        assert g["no_speech_prob"] == pytest.approx(w["no_speech_prob"], rel=2e-3, abs=1e-8)
        if "words" in w:
            assert [x["word"] for x in g["words"]] == [x["word"] for x in w["words"]], w["id"]
            for a, b in zip(g["words"], w["words"]):
                assert (a["start"], a["end"]) == (b["start"], b["end"]), (w["id"], a, b)
                assert a["probability"] == pytest.approx(b["probability"], rel=2e-3, abs=1e-7)


def check_case(case, model, monkeypatch):
    replay = Replay(case["calls"])
    monkeypatch.setattr(TR, "choose", replay)
    # the replay follows the reference's choices through transcribe.choose, i.e. through the HOST form of the logit rules;
    # the device form (wlk_pick_greedy) is checked against it in tests/test_gpu_transcribe_rules.py
    monkeypatch.setenv("WLK_TRANSCRIBE_DEVICE_RULES", "0")
    kwargs = dict(case["kwargs"])
    if isinstance(kwargs.get("temperature"), list):
        kwargs["temperature"] = tuple(kwargs["temperature"])
    got = TR.transcribe(model, make_audio(case["audio"]), **kwargs)
    want = case["result"]
    assert replay.at == len(replay.queue), f"{len(replay.queue) - replay.at} recorded greedy steps were not reached"
    compare_result(got, want)
    return replay


def test_categorical_draw_is_torchs():
    """The sampling step of the fallback temperatures: same tokens and the same generator state afterwards as
    torch.distributions.Categorical(logits=...).sample() (decoding.py:277), masked entries included."""
    rng = np.random.default_rng(0)
    for seed in range(120):
        x = torch.from_numpy((rng.standard_normal((1 + seed % 3, 51864)) * 3).astype(np.float32))
        x[0, 100:40000] = float("-inf")
        t = (0.2, 0.4, 0.6, 0.8, 1.0)[seed % 5]
        torch.manual_seed(seed)
        want = torch.distributions.Categorical(logits=x / t).sample()
        after_want = torch.rand(1)
        torch.manual_seed(seed)
        got = TR.choose(x.clone(), t)
        after_got = torch.rand(1)
        assert got.tolist() == want.tolist(), seed
        assert float(after_got) == float(after_want), seed


@pytest.mark.parametrize("case", KAT, ids=lambda c: c["name"])
def test_transcribe_host_logic_over_the_oracle(case, real_vocab, monkeypatch):
    from oracle_session import OracleModel
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    check_case(case, OracleModel(case["model"], 0), monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("case", KAT, ids=lambda c: c["name"])
def test_hip_transcribe_matches_the_reference(case, real_vocab, monkeypatch):
    from whisperlivekit_amd.engine import HipWhisperModel
    model = HipWhisperModel.synthetic(case["model"], 0, device=0)
    try:
        replay = check_case(case, model, monkeypatch)
        print(f"{case['name']}: {replay.at} greedy steps followed, {replay.ties} arg-max ties, "
              f"worst log-prob error {replay.worst_logprob:.2e}")
    finally:
        TR.release_sessions(model)
        model.close()


@pytest.mark.gpu
def test_hip_log_mel_matches_the_oracle():
    """wlk_log_mel on whole recordings (ragged lengths, with and without the 30 s padding, a padding shorter than the
    reflection) against the oracle's restatement of whisper/audio.py:110-157."""
    from oracle import whisper_oracle as wo
    from whisperlivekit_amd.engine import HipWhisperModel
    from whisperlivekit_amd.melbank import mel_filterbank
    model = HipWhisperModel.synthetic("micro.en", 0, device=0)
    sess = model.new_session(beam=1, batched=False)
    filters = torch.from_numpy(np.array(mel_filterbank(80)))
    try:
        for seconds, padding, seed in ((41.0, 480000, 1), (0.31, 480000, 2), (3.0, 0, 3), (1.234, 100, 4), (30.0, 480000, 5)):
            audio = synth.speech_like(seconds, seed=seed)
            got = sess.log_mel(audio, padding=padding)
            want = wo.log_mel_spectrogram(torch.from_numpy(audio), filters, padding=padding).numpy()
            assert got.shape == want.shape, (seconds, padding, got.shape, want.shape)
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-4, err_msg=f"{seconds} s, padding {padding}")
    finally:
        sess.close()
        model.close()


# ---- the wrapper under the reference's own LocalAgreement policy (build container only) --------------------------------
def _reference_available():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import ref_stubs
    return ref_stubs


@pytest.mark.reference
def test_wrapper_under_the_reference_local_agreement_policy(real_vocab, monkeypatch):
    """The reference's OnlineASRProcessor (local_agreement/online_asr.py, unmodified) driven once over the reference's
    WhisperASR with the seeded micro Whisper and once over HipWhisperASR (oracle-backed model on CPU): the same committed
    words with the same times after every chunk.  Greedy only (temperature 0): the draws of the fallback temperatures are
    covered by the replayed goldens above."""
    ref_stubs = _reference_available()
    if not ref_stubs.reference_available():
        pytest.skip("reference tree not present")
    ref_stubs.install(synthetic_vocab=False)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import gen_golden
    from oracle_session import OracleModel
    from whisperlivekit.local_agreement.backends import WhisperASR
    from whisperlivekit.local_agreement.online_asr import OnlineASRProcessor
    from whisperlivekit_amd.local_agreement import HipWhisperASR

    # the reference's tokenizer is built once per process on whichever vocabulary the first test installed: follow it
    from whisperlivekit.whisper.tokenizer import get_tokenizer as ref_get_tokenizer
    from whisperlivekit_amd.tokenizer import SyntheticEncoding
    if isinstance(ref_get_tokenizer(False, language="en", task="transcribe").encoding._e, SyntheticEncoding):
        monkeypatch.setenv("WLK_SYNTHETIC_VOCAB", "1")

    class RefASR(WhisperASR):
        def load_model(self, *a, **k):
            return gen_golden.build_reference_model("micro.en", 0)

    def configure(asr):
        asr.transcribe_kargs = {"temperature": 0.0, "vad": True}
        asr.tokenizer, asr.confidence_validation = None, False
        asr.buffer_trimming, asr.buffer_trimming_sec = "segment", 6
        return asr

    ref = OnlineASRProcessor(configure(RefASR(lan="en", model_size="micro.en")), logfile=None)
    mine = OnlineASRProcessor(configure(HipWhisperASR(lan="en", hip_model=OracleModel("micro.en", 0))), logfile=None)
    audio = synth.speech_like(14.0, seed=21)
    committed = 0
    for lo in range(0, len(audio), 32000):
        chunk = audio[lo:lo + 32000]
        outs = []
        for proc in (ref, mine):
            proc.insert_audio_chunk(chunk)
            tokens, upto = proc.process_iter()
            outs.append(([(t.start, t.end, t.text) for t in tokens], upto, len(proc.audio_buffer), proc.buffer_time_offset))
        assert outs[0] == outs[1], f"after {lo / 16000 + 2:.0f} s"
        committed += len(outs[0][0])
    assert committed > 0


@pytest.mark.gpu
def test_hip_transcribe_is_reentrant_on_a_shared_model(real_vocab):
    """Four threads transcribe different recordings on ONE model at once (LocalAgreement serves every connection from one
    ASR object): each result equals the one the same call gives alone."""
    import threading
    from whisperlivekit_amd.engine import HipWhisperModel
    model = HipWhisperModel.synthetic("micro.en", 0, device=0)
    kw = dict(language="en", temperature=0.0, word_timestamps=True, logprob_threshold=None, compression_ratio_threshold=None)
    audios = [synth.speech_like(6.0 + i, seed=40 + i) for i in range(4)]
    try:
        alone = [TR.transcribe(model, a, **kw) for a in audios]
        got, errors = [None] * 4, []

        def work(i):
            try:
                for _ in range(2):
                    got[i] = TR.transcribe(model, audios[i], **kw)
            except Exception as e:      # noqa: BLE001
                errors.append(e)
        threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        assert got == alone
    finally:
        TR.release_sessions(model)
        model.close()


def test_sessions_of_ended_threads_are_released(real_vocab):
    """The batch path keeps its sessions per calling thread; the entry of a thread that has ended is closed and dropped the
    next time another thread asks for its own."""
    import threading
    from oracle_session import OracleModel
    model = OracleModel("micro.en", 0)
    closed = []
    created = []

    def make(beam=1, max_audio_seconds=64.0, batched=None, _orig=model.new_session):
        s = _orig(beam, max_audio_seconds, batched)
        s.close = lambda s=s: closed.append(s)
        created.append(s)
        return s
    model.new_session = make
    t = threading.Thread(target=lambda: TR._rows_of(model).get(1))
    t.start(); t.join()
    assert len(created) == 1 and not closed
    TR._rows_of(model).get(1)                     # the main thread's first request sweeps the dead thread's entry
    assert closed == created[:1] and len(created) == 2
    assert list(model.__dict__["_batch_rows"]) == [threading.get_ident()]
    TR.release_sessions(model)
    assert closed == created


def _allowed_by_pick_params(p, suppressed, blank, n_vocab):
    """The token set wlk_pick_greedy's kernel keeps (csrc/select.hip: rules_pick_kernel's `allowed`), restated over the
    wlk_pick_params fields: what the C ABI promises for a given parameter block."""
    v = np.arange(n_vocab)
    ok = np.ones(n_vocab, bool)
    ok[list(suppressed)] = False
    if p["first_step"]:
        ok[list(blank)] = False
    if p["without_timestamps"]:
        return ok
    tb = p["timestamp_begin"]
    if p["no_timestamps"] >= 0:
        ok[p["no_timestamps"]] = False
    if p["ts_mode"] == 1:
        ok &= v < tb
    if p["ts_mode"] == 2:
        ok &= v >= p["eot"]
    ok &= ~((v >= tb) & (v < p["ts_bound"]))
    if p["first_step"]:
        ok &= v >= tb
        if p["max_initial"] >= 0:
            ok &= v < tb + p["max_initial"] + 1
    return ok


@pytest.mark.parametrize("options", [dict(), dict(without_timestamps=True), dict(suppress_blank=False, suppress_tokens=""),
                                     dict(max_initial_timestamp=None), dict(prompt="hello there", suppress_tokens="1,2,-1")])
def test_pick_params_describe_the_mask_of_the_host_rules(options, real_vocab):
    """The host half of the device rules: for token histories on every branch of ApplyTimestampRules, the parameter block
    `_pick_state` hands to wlk_pick_greedy describes exactly the set of tokens `_apply_rules` (the form followed against the
    reference's recorded choices above) leaves finite - before the data-dependent "timestamps outweigh text" rule, which the
    kernel applies to the logits itself."""
    from oracle_session import OracleModel
    model = OracleModel("micro", 0)
    dec = TR._WindowDecoder(model, TR.DecodingOptions(language="en", temperature=0.0, **options))
    V = model.dims.n_vocab
    tb, eot = dec.tok.timestamp_begin, dec.tok.eot
    rng = np.random.default_rng(5)
    text = lambda n: [int(t) for t in rng.integers(300, 20000, n)]
    histories = [[], [tb], [tb + 3], [tb, *text(3)], [tb, *text(2), tb + 40], [tb, *text(2), tb + 40, tb + 40],
                 [tb, *text(1), tb + 40, tb + 40, *text(2)], [tb, *text(4), tb + 1499], [tb, *text(2), tb + 700, tb + 700, eot],
                 text(5), [tb + 1500], [tb, *text(2), tb + 1500, tb + 1500]]
    for hist in histories:
        tokens = np.asarray([list(dec.initial) + hist], np.int64)
        # logits under which the data-dependent rule stays silent: text far above the timestamps
        logits = np.zeros((1, V), np.float32)
        logits[0, :tb] = 50.0
        dec._apply_rules(logits, tokens)
        want = np.isfinite(logits[0])
        got = _allowed_by_pick_params(dec._pick_state(tokens), dec.suppressed or [], dec.blank_ids or [], V)
        assert np.array_equal(got, want), (options, hist, np.flatnonzero(got != want)[:8])

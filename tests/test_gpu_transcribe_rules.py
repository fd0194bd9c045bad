"""whisper's per-step logit rules on the device (wlk_rules_set / wlk_pick_greedy, csrc/select.hip: rules_pick_kernel) against the
host form in whisperlivekit_amd/transcribe.py (`_WindowDecoder._apply_rules` + arg-max, itself followed step by step against
the reference's recorded choices in tests/test_transcribe.py): SuppressBlank, SuppressTokens, ApplyTimestampRules
(whisper/decoding.py:417-499) and GreedyDecoder.update at temperature 0 (:270-287)."""
import numpy as np
import pytest
import torch

import helpers as H
from whisperlivekit_amd import synth, transcribe as TR

pytestmark = pytest.mark.gpu
KAT = H.golden_json("transcribe_kat.json")


@pytest.fixture()
def real_vocab(tmp_path, monkeypatch):
    monkeypatch.setenv("WLK_VOCAB_DIR", H.real_vocab_dir(tmp_path))
    monkeypatch.setenv("WLK_SYNTHETIC_VOCAB", "0")


def _histories(dec, rng):
    """Sampled-token histories that walk every branch of ApplyTimestampRules: nothing sampled yet, an opening timestamp, a
    closed pair, text behind a pair, text only, a timestamp close to the end of the window, <|endoftext|>."""
    tb, eot = dec.tok.timestamp_begin, dec.tok.eot
    text = lambda n: [int(t) for t in rng.integers(300, 20000, n)]
    return [[], [tb], [tb + 3], [tb, *text(3)], [tb, *text(2), tb + 40], [tb, *text(2), tb + 40, tb + 40],
            [tb, *text(1), tb + 40, tb + 40, *text(2)], [tb, *text(4), tb + 1499], [tb, *text(2), tb + 700, tb + 700, eot],
            text(5), [tb + 1500], [tb, *text(2), tb + 1500, tb + 1500]]


@pytest.mark.parametrize("options", [dict(), dict(without_timestamps=True), dict(suppress_blank=False, suppress_tokens=""),
                                     dict(max_initial_timestamp=None), dict(prompt="hello there", suppress_tokens="1,2,-1")])
def test_device_pick_equals_the_host_rules_on_every_branch(options, real_vocab):
    from whisperlivekit_amd.engine import HipWhisperModel
    model = HipWhisperModel.synthetic("micro", 3, device=0)
    rng = np.random.default_rng(11)
    try:
        dec = TR._WindowDecoder(model, TR.DecodingOptions(language="en", temperature=0.0, **options))
        V = model.dims.n_vocab
        s = TR._rows_of(model).get(1)
        s.encode_mel(TR.pad_or_trim(s.log_mel(synth.speech_like(7.0, seed=5))))
        s.set_rules(dec.suppressed or [], dec.blank_ids or [])
        n = 0
        for hist in _histories(dec, rng):
            tokens = np.asarray([list(dec.initial) + hist], np.int64)
            s.decode(tokens, first=True, sot_index=dec.sot_index)
            got_tok, got_lp = s.pick_greedy(**dec._pick_state(tokens))
            logits = TR._logits(s, 1, V).copy()
            logprobs = dec._apply_rules(logits, tokens)             # masks `logits` in place, as the reference does
            want_tok = int(torch.from_numpy(logits).argmax(dim=-1)[0])
            assert np.isfinite(logprobs[0, want_tok]), hist
            assert got_tok == want_tok, (options, hist, got_tok, want_tok)
            assert got_lp == pytest.approx(float(logprobs[0, want_tok]), abs=2e-6), (options, hist)
            n += 1
        assert n == 12
    finally:
        TR.release_sessions(model)
        model.close()


@pytest.mark.parametrize("case", [c for c in KAT if all(call["temperature"] == 0 and call["beam"] is None for call in c["calls"])],
                         ids=lambda c: c["name"])
def test_transcribe_with_device_rules_gives_the_reference_result(case, real_vocab, monkeypatch):
    """The recordings of tests/golden/transcribe_kat.json.gz that the reference decoded greedily at temperature 0 throughout:
    with the rules on the device the result dictionary is the reference's (tokens, times, words; log-probabilities <= 5e-4)."""
    from test_transcribe import compare_result, make_audio
    from whisperlivekit_amd.engine import HipWhisperModel
    monkeypatch.setenv("WLK_TRANSCRIBE_DEVICE_RULES", "1")
    calls = []
    real = TR.choose
    monkeypatch.setattr(TR, "choose", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    model = HipWhisperModel.synthetic(case["model"], 0, device=0)
    try:
        kwargs = dict(case["kwargs"])
        if isinstance(kwargs.get("temperature"), list):
            kwargs["temperature"] = tuple(kwargs["temperature"])
        got = TR.transcribe(model, make_audio(case["audio"]), **kwargs)
        assert not calls, "the host rules ran although the decode was greedy at temperature 0"
        compare_result(got, case["result"])
    finally:
        TR.release_sessions(model)
        model.close()

"""The known-answer scenarios the REFERENCE's own tests pin for this path, applied to this
implementation: the a16 output guards of the online processor (reference tests/test_backend_deep_bugs.py:185-301),
the a9 word-timestamp and pending-UTF-8 helpers (:302-391), and warm-up-must-raise
(tests/test_silent_backend_guard.py:62-83).  Only inputs and expected values are taken from there."""
from types import SimpleNamespace

import numpy as np
import pytest

from whisperlivekit_amd import policy as P
from whisperlivekit_amd.backend import HipSimulStreamingOnlineProcessor


class ScriptedModel:
    """Stands in for the AlignAtt object: hands out prepared word batches."""

    def __init__(self, batches):
        self.batches = list(batches)
        self.cfg = SimpleNamespace(language="en")
        self.refresh_calls = []
        self.global_time_offset = 0.0

    def infer(self, is_last=False):
        return self.batches.pop(0) if self.batches else []

    def refresh_segment(self, complete=False):
        self.refresh_calls.append(complete)


def processor_over(model, end=0.0):
    p = object.__new__(HipSimulStreamingOnlineProcessor)
    p.asr = SimpleNamespace()
    p.model = model
    p.end = end
    p.buffer = []
    p._last_committed_end = 0.0
    p._recent_words = []
    return p


def tok(s, e, text):
    return P.ASRToken(start=s, end=e, text=text)


def test_rewound_words_behind_the_committed_time_are_dropped():
    model = ScriptedModel([[tok(10.00, 10.10, " hello"), tok(10.20, 10.30, " world")],
                           [tok(9.50, 9.60, " stale"), tok(10.35, 10.45, " again"), tok(10.25, 10.35, " stale2"),
                            tok(10.50, 10.60, " now")]])
    p = processor_over(model, end=11.0)
    first, _ = p.process_iter()
    second, _ = p.process_iter()
    assert [t.text for t in first] == [" hello", " world"]
    assert [t.text for t in second] == [" again", " now"]
    assert p._last_committed_end == pytest.approx(10.60)
    assert model.refresh_calls == []


def test_minor_timestamp_jitter_inside_a_batch_is_kept():
    model = ScriptedModel([[tok(1.00, 1.10, " concord"), tok(1.60, 1.70, " returned"), tok(2.20, 2.30, " its"),
                            tok(2.18, 2.28, " place"), tok(2.50, 2.60, " amidst")]])
    p = processor_over(model, end=3.0)
    tokens, _ = p.process_iter()
    assert [t.text for t in tokens] == [" concord", " returned", " its", " place", " amidst"]
    assert model.refresh_calls == []


def test_segment_is_reset_when_every_word_rewinds_far_behind():
    model = ScriptedModel([[tok(186.0, 186.1, " old"), tok(187.0, 187.1, " text")]])
    p = processor_over(model, end=195.0)
    p._last_committed_end = 191.2
    tokens, upto = p.process_iter()
    assert tokens == [] and upto == 195.0
    assert model.refresh_calls == [True]
    assert model.global_time_offset == 195.0
    assert p.buffer == []


def test_repetition_loop_resets_before_anything_is_emitted():
    phrase = [" Det", " ar", " en", " ny", " kriska", " klimat"]
    model = ScriptedModel([[tok(i * 0.2, i * 0.2 + 0.1, w) for i, w in enumerate(phrase * 3)]])
    p = processor_over(model, end=42.0)
    emitted, upto = p.process_iter()
    assert emitted == [] and upto == 42.0
    assert model.refresh_calls == [True]
    assert model.global_time_offset == 42.0
    assert p._last_committed_end == 0.0


def test_repetition_is_detected_across_small_batches():
    phrase = [" Det", " ar", " en", " ny", " kriska", " klimat"]
    batches = [[tok(r * 2.0 + i * 0.2, r * 2.0 + i * 0.2 + 0.1, w) for i, w in enumerate(phrase)] for r in range(3)]
    model = ScriptedModel(batches)
    p = processor_over(model, end=10.0)
    first, _ = p.process_iter()
    second, _ = p.process_iter()
    third, _ = p.process_iter()
    assert len(first) == 6 and len(second) == 6 and third == []
    assert model.refresh_calls == [True]


def timestamp_tester(offset=0.0):
    a = object.__new__(P.AlignAttPolicy)
    a.state = SimpleNamespace(speaker=2, detected_language="en", global_time_offset=offset, pending_incomplete_tokens=[],
                              pending_incomplete_token_timestamps=[], pending_retries=0)
    return a


def test_word_end_is_the_next_words_start_with_the_global_offset():
    words = timestamp_tester(10.0)._build_timestamped_words([" hello", " world"], [[101, 102], [103]], [0.50, 0.70, 1.20])
    assert words[0].start == pytest.approx(10.50) and words[0].end == pytest.approx(11.20)
    assert words[1].start == pytest.approx(11.20) and words[1].end == pytest.approx(11.30)
    assert words[0].end <= words[1].start
    assert words[0].speaker == 2 and words[0].detected_language == "en"


def test_final_multi_token_word_ends_after_its_last_token():
    words = timestamp_tester()._build_timestamped_words([" longer"], [[201, 202]], [2.00, 2.34])
    assert words[0].start == pytest.approx(2.00) and words[0].end == pytest.approx(2.44)


def test_single_token_word_keeps_a_short_end():
    words = timestamp_tester()._build_timestamped_words([" word"], [[301]], [4.00])
    assert words[0].start == pytest.approx(4.00) and words[0].end == pytest.approx(4.10)


def test_pending_utf8_tokens_keep_their_original_timestamps():
    a = timestamp_tester()
    a._handle_pending_tokens([" caf�"], [[401, 402]], [3.20, 3.24])
    assert a.state.pending_incomplete_tokens == [401, 402]
    assert a.state.pending_incomplete_token_timestamps == [3.20, 3.24]
    merged, times = a._prepend_pending_tokens([403, 404], [4.00, 4.12])
    assert merged == [401, 402, 403, 404] and times == [3.20, 3.24, 4.00, 4.12]
    words = a._build_timestamped_words([" cafe", " next"], [[401, 402, 403], [404]], times)
    assert words[0].start == pytest.approx(3.20) and words[0].end == pytest.approx(4.12)


def test_warmup_raises_when_inference_is_broken():
    class Broken(P.AlignAttPolicy):
        def __init__(self):
            pass

        def insert_audio(self, audio=None):
            raise RuntimeError("Tensor for argument weight is on cpu but expected on mps")

        def infer(self, is_last=False):
            pass

        def refresh_segment(self, complete=False):
            pass

    with pytest.raises(RuntimeError, match="refusing to serve"):
        Broken().warmup(np.zeros(16000, dtype=np.float32))


def test_new_speaker_returns_the_tail_tokens_and_position():
    """reference tests/test_asr_coalescing_boundaries.py:265-290"""
    class Model:
        speaker = -1
        global_time_offset = 0.0

        def refresh_segment(self, complete=False):
            self.refreshed = complete

    token = tok(0.0, 2.5, "tail")
    p = object.__new__(HipSimulStreamingOnlineProcessor)
    p.asr = SimpleNamespace()
    p.process_iter = lambda is_last=False: ([token], 2.5)
    p.model = Model()
    p._last_committed_end = 0.0
    p._recent_words = ["old"]
    tokens, upto = p.new_speaker(P.ChangeSpeaker(speaker=3, start=3))
    assert tokens == [token] and upto == 2.5
    assert p.model.refreshed is True and p.model.speaker == 3 and p.model.global_time_offset == 3
    assert p._recent_words == []


def test_silence_shorter_than_five_seconds_becomes_zeros_longer_resets_the_segment():
    """simul_whisper/backend.py:77-93 (exercised end to end by the micro_events golden stream; here the contract)"""
    class Model:
        global_time_offset = 0.0

        def __init__(self):
            self.inserted, self.refreshed = [], []

        def insert_audio(self, a=None):
            self.inserted.append(len(a))

        def refresh_segment(self, complete=False):
            self.refreshed.append(complete)

    p = object.__new__(HipSimulStreamingOnlineProcessor)
    p.model, p.end, p._last_committed_end, p._recent_words = Model(), 10.0, 9.0, ["w"]
    p.end_silence(1.5, 10.0)
    assert p.model.inserted == [24000] and p.model.refreshed == [] and p.end == pytest.approx(11.5)
    p.end_silence(6.0, 11.5)
    assert p.model.refreshed == [True] and p.model.global_time_offset == pytest.approx(17.5)
    assert p._last_committed_end == pytest.approx(17.5) and p._recent_words == [] and p.end == pytest.approx(17.5)

"""a12 glue: the speaker-channel cap and arg-max / run-length post-processing, pinned with the
known-answer vectors of the reference's own tests (tests/test_sortformer_max_speakers.py:79-118,
:181-215, :240-247 there), plus the 1 s chunking / 99-frame left context / offset contract."""
import numpy as np
import pytest

from whisperlivekit_amd.diarization import (HipSortformerDiarizationOnline, SortformerStreamingParams,
                                            resolve_max_speakers)


class FakeBackend:
    n_spk = 4
    params = SortformerStreamingParams()

    def __init__(self, preds_per_chunk=None):
        self.calls = []
        self.preds = preds_per_chunk

    def features(self, pcm):
        return np.full((len(pcm) // 160 + 1, 128), float(len(self.calls)), np.float32)

    def new_state(self):
        return {}

    def forward_streaming_step(self, features, state, left_offset, right_offset):
        self.calls.append((features.shape, float(features[0, 0]), float(features[-1, 0]), left_offset, right_offset))
        return self.preds[len(self.calls) - 1]


def processor(predictions, max_speakers, chunk_index=0):
    online = HipSortformerDiarizationOnline(FakeBackend(), max_speakers=max_speakers)
    online.total_preds = np.asarray(predictions, np.float32)
    online._chunk_index = chunk_index
    return online


def tuples(segments):
    return [(int(s.speaker), s.start, s.end) for s in segments]


KAT = [[0.90, 0.10, 0.20, 0.05], [0.10, 0.80, 0.99, 0.05], [0.85, 0.10, 0.99, 0.05], [0.80, 0.10, 0.95, 0.05]]


def test_two_speaker_cap_keeps_first_arrival_ordered_channels():
    assert tuples(processor(KAT, 2)._process_predictions()) == [(0, 0.0, 0.25), (1, 0.25, 0.5), (0, 0.5, 1.0)]


def test_default_matches_argmax_across_all_checkpoint_channels():
    assert resolve_max_speakers(None, 4) == 4
    assert tuples(processor(KAT, 4)._process_predictions()) == [(0, 0.0, 0.25), (2, 0.25, 1.0)]


def test_cap_does_not_remap_retained_channel_at_chunk_boundary():
    online = processor([[0.90, 0.10, 0.05, 0.05], [0.80, 0.20, 0.05, 0.05], [0.10, 0.90, 0.05, 0.05],
                        [0.20, 0.80, 0.05, 0.05]], 2)
    assert tuples(online._process_predictions()) == [(0, 0.0, 0.5), (1, 0.5, 1.0)]
    online.total_preds = np.asarray([[0.10, 0.90, 0.99, 0.05], [0.20, 0.80, 0.99, 0.05], [0.90, 0.10, 0.99, 0.05],
                                     [0.80, 0.20, 0.99, 0.05]], np.float32)
    online._chunk_index = 1
    assert tuples(online._process_predictions()) == [(1, 1.0, 1.5), (0, 1.5, 2.0)]


def test_checkpoint_limit_is_validated():
    assert resolve_max_speakers(2, 4) == 2
    with pytest.raises(ValueError, match="between 1 and 2"):
        resolve_max_speakers(3, 2)
    for bad in (0, 5, -1, 1.5, True):
        with pytest.raises(ValueError):
            resolve_max_speakers(bad, 4)
    with pytest.raises(RuntimeError):
        p = processor(KAT, 4)
        p.total_preds = np.zeros((4, 2), np.float32)
        p._process_predictions()


def test_chunking_left_context_and_offsets():
    """1.0 s chunks; frames of the previous chunk are prepended (99 of them); offsets 0/8 then 8/8."""
    chunk_preds = [np.tile(np.array([[0.9, 0.1, 0, 0]], np.float32), (12, 1)),
                   np.tile(np.array([[0.1, 0.9, 0, 0]], np.float32), (12, 1)),
                   np.tile(np.array([[0.9, 0.1, 0, 0]], np.float32), (12, 1))]
    backend = FakeBackend(chunk_preds)
    online = HipSortformerDiarizationOnline(backend)
    assert online.chunk_duration_seconds == pytest.approx(1.0)
    assert online.diarize_sync() == []                         # nothing buffered
    online.insert_audio_chunk(np.zeros(8000, np.float32))
    assert online.diarize_sync() == []                         # < 1 s
    online.insert_audio_chunk(np.zeros(8000 + 16000 + 100, np.float32))
    first = online.diarize_sync()
    second = online.diarize_sync()
    assert online.diarize_sync() == []                         # 100 samples left
    assert len(online.buffer_audio) == 100
    assert backend.calls[0] == ((101, 128), 0.0, 0.0, 0, 8)
    assert backend.calls[1] == ((200, 128), 0.0, 1.0, 8, 8)    # 99 old frames (value 0) + 101 new (value 1)
    assert tuples(first) == [(0, 0.0, 1.0)] and tuples(second) == [(1, 1.0, 2.0)]
    online.insert_silence(2.5)
    online.insert_audio_chunk(np.zeros(16000, np.float32))
    assert tuples(online.diarize_sync()) == [(0, 4.5, 5.5)]

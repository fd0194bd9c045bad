"""The reference's UNMODIFIED AudioProcessor / TestHarness over this repository's hooks, on CPU: the HIP session is replaced by
the oracle-backed stand-in (tests/fake_session.py), everything above it - the reference's PCM framing, queues, to_thread
transcription worker, SessionMetrics, TokensAlignment, the reference's own SimulStreamingOnlineProcessor and AlignAttBase.infer -
is the real thing.  What it pins without a GPU: the harness of tests/ref_pipeline.py (engine instance, the three routed
factories, lock-step feeding) drives the pipeline so that it commits exactly the golden stream's words.  The GPU form with a
real HipWhisperModel / Sortformer / VAC is tests/test_gpu_pipeline.py."""
import asyncio
import logging
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
import ref_pipeline as RP  # noqa: E402

pytestmark = pytest.mark.skipif(not RP.reference_available(), reason="no WhisperLiveKit tree (WLK_REFERENCE_ROOT, /root/reference or oracle/_ref)")


def golden_words(case, n_chunks=None):
    g = H.golden_json(f"stream_{case}.json")
    evs = [ev for ev in g["events"] if ev["kind"] == "chunk"][:n_chunks]
    return [(round(s, 2), round(e, 2), x) for ev in evs for s, e, x, _sp in ev["tokens"]]


def fake_model(name, seed=0):
    from fake_session import FakeHipModel
    from whisperlivekit_amd.dims import ALIGNMENT_HEADS, MODEL_DIMS
    return FakeHipModel(MODEL_DIMS[name], H.oracle_sd(name, seed), ALIGNMENT_HEADS[name])


def test_audio_processor_commits_the_golden_words_over_the_routed_backend(caplog):
    case, n_chunks = "micro_12s", 24
    audio = H.stream_audio(case)[: n_chunks * 8000]
    engine = RP.make_engine(RP.make_asr("micro.en", fake_model("micro.en")))
    with caplog.at_level(logging.WARNING, logger="whisperlivekit"):
        run = asyncio.run(RP.run_session(engine, RP.pcm16_bytes(audio)))
    from whisperlivekit.audio_processor import AudioProcessor
    from whisperlivekit.metrics_collector import SessionMetrics
    assert isinstance(run.metrics, SessionMetrics)
    got = [(round(float(t.start), 2), round(float(t.end), 2), t.text) for t in run.tokens]
    want = golden_words(case, n_chunks)
    assert len(want) > 3 and got == want
    # one process_iter per fed chunk (lock-step), counted by the reference's own metrics; the EOF flush adds its call(s)
    assert run.metrics.n_chunks_received == n_chunks and run.metrics.n_transcription_calls >= n_chunks
    assert len(run.metrics.transcription_durations) == run.metrics.n_transcription_calls
    assert run.metrics.n_tokens_produced >= len(got) and len(run.final_tokens) >= len(run.tokens)
    assert [c[4] for c in run.calls] == [[(round(s, 2), round(e, 2), x) for s, e, x, _ in ev["tokens"]]
                                         for ev in H.golden_json(f"stream_{case}.json")["events"][:n_chunks]]
    assert not [r for r in caplog.records if "silent" in r.getMessage().lower() or "Exception in" in r.getMessage()], \
        [r.getMessage() for r in caplog.records]
    assert run.front, "the reference's results_formatter yielded nothing"
    assert AudioProcessor.transcription_processor.__module__ == "whisperlivekit.audio_processor"

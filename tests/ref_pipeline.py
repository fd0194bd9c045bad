"""The drop-in under the reference's OWN session pipeline (north_star: "audio_processor.py ... untouched").

TEST INFRASTRUCTURE (used by tests/test_gpu_pipeline.py and bench.py's `pipeline` leg).  Everything that runs the session is
the reference's unmodified code, imported from /root/reference, WLK_REFERENCE_ROOT or the archive staged under oracle/_ref/:

* ``AudioProcessor`` (whisperlivekit/audio_processor.py) - PCM framing (:1099-1169), the VAC gate (:1171-1233), the
  transcription worker that calls ``process_iter`` through ``asyncio.to_thread`` (:543-551, :647-827), the diarization worker
  that awaits ``diarize()`` on the event loop (:853-885), ``SessionMetrics`` (metrics_collector.py:16-62),
  ``results_formatter`` / ``TokensAlignment``;
* ``TestHarness`` (whisperlivekit/test_harness.py:467-632) - engine cache, ``feed_pcm``, ``finish``, ``.metrics``.

What is harness-side, and why it does not touch the reference's files:

* the ENGINE: ``TranscriptionEngine.__init__`` loads checkpoints from disk / the network (core.py:86-343).  Here an instance is
  made with ``object.__new__`` and given the five attributes ``AudioProcessor.__init__`` reads (``args`` built by the
  reference's own ``WhisperLiveKitConfig.from_kwargs``, ``asr``, ``diarization_model``, ``translation_model``,
  ``vac_session``) - the way the reference's tests build fake engines (tests/test_asr_coalescing_boundaries.py:105-129);
* the ROUTING: the three factories ``AudioProcessor.__init__`` calls (core.py:395-493, imported by name into
  audio_processor.py) are replaced for the duration of the construction by the branches INTEGRATION.md shows a maintainer
  adding: ``online_factory`` -> the reference's ``SimulStreamingOnlineProcessor`` with ``_create_alignatt`` routed to the HIP
  hooks (backend.reference_online_processor_class), ``online_diarization_factory`` -> ``HipSortformerDiarizationOnline``,
  and the VAC iterator -> ``HipFixedVADIterator`` over the HIP Silero model.
"""
from __future__ import annotations

import asyncio
import contextlib
import os
import sys
import time
from argparse import Namespace
from dataclasses import asdict
from types import SimpleNamespace
from typing import List, Optional
from unittest import mock

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import ref_stubs  # noqa: E402


def reference_available() -> bool:
    return ref_stubs.reference_available()


def install(synthetic_vocab: bool = True):
    ref_stubs.install(synthetic_vocab=synthetic_vocab)


def pcm16_bytes(audio: np.ndarray) -> bytes:
    """float32 in [-1, 1) that went through synth.to_pcm16_roundtrip -> the s16le bytes a client would send
    (AudioProcessor.convert_pcm_to_float, audio_processor.py:416-418, turns them back into the same floats)."""
    return np.round(np.asarray(audio, np.float64) * 32768.0).astype(np.int16).tobytes()


def make_asr(model_name: str, hip_model, **cfg_over):
    """The shared ASR object the reference's SimulStreamingOnlineProcessor reads (``cfg``, ``hip_model`` for the routed
    ``_create_alignatt``, and ``sep`` which AudioProcessor.__init__ takes from ``transcription.asr``)."""
    install()
    from whisperlivekit.simul_whisper.config import AlignAttConfig
    kw = dict(tokenizer_is_multilingual=not model_name.endswith(".en"), segment_length=0.5, frame_threshold=25, language="en",
              audio_max_len=30.0, audio_min_len=0.0, cif_ckpt_path=None, decoder_type="beam", beam_size=1, task="transcribe",
              never_fire=False, init_prompt=None, max_context_tokens=None, static_init_prompt=None)
    kw.update(cfg_over)
    return SimpleNamespace(cfg=AlignAttConfig(**kw), hip_model=hip_model, shared_model=hip_model, use_full_mlx=False,
                           mlx_encoder=None, fw_encoder=None, tokenizer=None, sep=" ")


def make_engine(asr, diarization_model=None, vac: bool = False, translation_model=None, **cfg):
    """A TranscriptionEngine INSTANCE of the reference's own class without running its loader; ``args`` is what the
    reference derives from its own config dataclass."""
    install()
    from whisperlivekit.config import WhisperLiveKitConfig
    from whisperlivekit.core import TranscriptionEngine
    conf = dict(pcm_input=True, vac=vac, min_chunk_size=0.5, vac_chunk_size=0.5, backend_policy="simulstreaming", backend="hip",
                diarization=diarization_model is not None, diarization_backend="sortformer", transcription=asr is not None,
                lan="en", target_language="")
    conf.update(cfg)
    config = WhisperLiveKitConfig.from_kwargs(**conf)
    eng = object.__new__(TranscriptionEngine)
    eng.config = config
    eng.args = Namespace(**asdict(config))
    eng.asr = asr
    eng.tokenizer = None
    eng.diarization = None
    eng.diarization_model = diarization_model
    eng.translation_model = translation_model      # a whisperlivekit_amd.translation.HipNllbTranslationModel (config 5) or None
    eng.vac_session = None
    return eng


_routes = dict(depth=0, stack=None)


@contextlib.contextmanager
def hip_routes(vad_weights=None):
    """The three branches a maintainer adds, applied to the names audio_processor.py imported.  Re-entrant
    (concurrent sessions of one event loop construct their AudioProcessors under it): the outermost entry installs the
    routes, the last exit removes them."""
    install()
    import whisperlivekit.audio_processor as ap
    from whisperlivekit_amd.backend import reference_online_processor_class
    from whisperlivekit_amd.diarization import HipSortformerDiarizationOnline
    if _routes["depth"] == 0:
        routed = reference_online_processor_class()

        def online_factory(args, asr, language=None):
            return routed(asr)

        def online_diarization_factory(args, backend):
            return HipSortformerDiarizationOnline(shared_model=backend, max_speakers=getattr(args, "sortformer_max_speakers", None))

        def online_translation_factory(args, translation_model):
            # core.py:483-493 builds nllw.OnlineTranslation(model, [source], [target]); the HIP session takes the same two codes
            # (the test tokenizer spells them as NLLB does: eng_Latn, fra_Latn, ...)
            from whisperlivekit_amd.translation import online_translation_factory as hip_factory
            return hip_factory(translation_model, args.lan, args.target_language)

        patches = [mock.patch.object(ap, "online_factory", online_factory),
                   mock.patch.object(ap, "online_diarization_factory", online_diarization_factory),
                   mock.patch.object(ap, "online_translation_factory", online_translation_factory)]
        if vad_weights is not None:
            from whisperlivekit_amd import vad as V

            def load_jit_vad():
                return V.HipSileroVAD(vad_weights)

            patches += [mock.patch.object(ap, "load_jit_vad", load_jit_vad),
                        mock.patch.object(ap, "FixedVADIterator", V.HipFixedVADIterator)]
        st = contextlib.ExitStack()
        for p in patches:
            st.enter_context(p)
        _routes["stack"] = st
    _routes["depth"] += 1
    try:
        yield
    finally:
        _routes["depth"] -= 1
        if _routes["depth"] == 0:
            _routes["stack"].close()
            _routes["stack"] = None


class PipelineRun:
    """What one session produced."""

    def __init__(self):
        self.tokens: List = []               # committed ASRTokens while the stream was fed (state.tokens before EOF)
        self.final_tokens: List = []         # ... including what the end-of-stream flush committed
        self.calls: List = []                # per fed chunk: (call wall s, 0.0, [token end times], stream time, [(start, end, text)]),
                                             #   word times rounded to 10 ms like the golden streams' comparison
        self.metrics = None                  # the reference's SessionMetrics
        self.front: List = []                # FrontData updates the formatter yielded
        self.translations: List = []         # validated Translation pieces the translation worker published (state.new_translation)
        self.translation_calls = 0           # device translations the session ran (HipOnlineTranslation.translations)
        self.diar_frames = 0                 # activity frames the session's diarizer produced
        self.end_attributed_speaker = 0.0    # State.end_attributed_speaker: how far the diarization worker got
        self.wall_s = 0.0
        self.audio_s = 0.0


async def run_session(engine, pcm: bytes, chunk_s: float = 0.5, lockstep: bool = True, vad_weights=None, drain_s: float = 20.0,
                      harness: bool = True) -> PipelineRun:
    """Feed ``pcm`` (s16le, 16 kHz mono) through the reference's TestHarness -> AudioProcessor in ``chunk_s`` messages.
    ``lockstep``: wait after every message until the transcription worker has consumed it (one ``process_iter`` per chunk,
    the chunking of the golden streams; without it the worker coalesces whatever queued up - audio_processor.py:66-91)."""
    install()
    import whisperlivekit.test_harness as th
    from whisperlivekit.audio_processor import AudioProcessor
    out = PipelineRun()
    bps = 16000 * 2
    step = int(chunk_s * bps)
    with hip_routes(vad_weights):
        if harness:
            h = th.TestHarness(transcription_engine=engine)
            # TestHarness caches engines by its keyword arguments and builds missing ones with the loader: seed the cache
            th._engine_cache[tuple(sorted(h._engine_kwargs.items()))] = engine
            await h.__aenter__()
            proc = h._processor
        else:
            h = None
            proc = AudioProcessor(transcription_engine=engine)
            gen = await proc.create_tasks()

            async def collect():
                async for fd in gen:
                    out.front.append(fd)
            collector = asyncio.create_task(collect())
    assert type(proc) is AudioProcessor
    t0 = time.perf_counter()
    try:
        n_before = 0
        for k, lo in enumerate(range(0, len(pcm), step)):
            msg = pcm[lo:lo + step]
            calls_before = proc.metrics.n_transcription_calls
            a = time.perf_counter()
            if h is not None:
                await h.feed_pcm(msg, speed=0, chunk_duration=chunk_s)
            else:
                await proc.process_audio(msg)
            if lockstep and proc.transcription_queue is not None:
                # consumed = the worker has published this chunk's call: it extends state.tokens and moves
                # end_transcription_processed to the chunk's stream time under one lock (audio_processor.py:783-796), AFTER
                # SessionMetrics counted the call - so the position, not the counter, is what to wait for
                deadline = time.perf_counter() + drain_s
                target = (k + 1) * chunk_s - 1e-6
                while (proc.state.end_transcription_processed < target or proc.metrics.n_transcription_calls == calls_before
                       or not proc.transcription_queue.empty()) and time.perf_counter() < deadline:
                    await asyncio.sleep(0.0005)
            dt = time.perf_counter() - a
            toks = list(proc.state.tokens)
            new = [t for t in toks[n_before:] if hasattr(t, "text")]
            n_before = len(toks)
            call_s = proc.metrics.transcription_durations[-1] if proc.metrics.transcription_durations else dt
            out.calls.append((call_s, 0.0, [float(t.end) for t in new], (k + 1) * chunk_s, [(round(float(t.start), 2), round(float(t.end), 2), t.text) for t in new]))
        out.tokens = [t for t in proc.state.tokens if hasattr(t, "text")]
        out.wall_s = time.perf_counter() - t0
        out.audio_s = len(pcm) / bps
        # end of stream: the reference's own flush (process_audio(b"") -> SENTINEL -> _finish_transcription)
        if h is not None:
            await h._processor.process_audio(b"")
        else:
            await proc.process_audio(b"")
        deadline = time.perf_counter() + drain_s
        while not proc._processing_tasks_done() and time.perf_counter() < deadline:
            await asyncio.sleep(0.005)
        out.final_tokens = [t for t in proc.state.tokens if hasattr(t, "text")]
        out.metrics = proc.metrics
        if proc.translation is not None:
            out.translations = list(getattr(proc.tokens_alignment, "all_translation_segments", [])) + list(proc.state.new_translation)
            out.translation_calls = int(getattr(proc.translation, "translations", 0))
        if proc.diarization is not None:
            out.diar_frames = int(proc.diarization.total_preds.shape[0])
            out.end_attributed_speaker = float(proc.state.end_attributed_speaker)
        if h is not None:
            out.front = list(h.history)
    finally:
        if h is not None:
            await h.__aexit__(None, None, None)
        else:
            await proc.cleanup()
            collector.cancel()
    return out

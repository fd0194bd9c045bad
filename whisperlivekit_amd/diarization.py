"""Streaming Sortformer diarization on the HIP backend - session glue and feature front end.

Mirror of the diarization duck type AudioProcessor consumes (``insert_audio_chunk``, ``insert_silence``,
``async diarize()``, ``close()``, a ``buffer_audio`` attribute; audio_processor.py:853-885, :1081) as
implemented by the reference's ``SortformerDiarizationOnline``
(whisperlivekit/diarization/sortformer_backend.py:151-371):

* 1.0 s chunking of the incoming PCM (chunk_len 10 x subsampling 10 x 10 ms stride, :190-194, :261-267);
* 128-bin log-mel features of each chunk (:273-275) - on the GPU through ``wlk_melspec_*``;
* the last 99 feature frames of the previous chunk are prepended (:279-285), left/right offsets 8/8 (:290-291);
* the model's ``forward_streaming_step`` is called through a small backend protocol (``SortformerBackend``);
* arg-max over the first ``max_speakers`` channels -> run-length -> ``SpeakerSegment`` (:313-363).

The Sortformer network itself (NeMo ``SortformerEncLabelModel``: FastConformer + Transformer + streaming
speaker cache) is third-party code that is NOT part of the reference tree; its HIP port is the
``SortformerBackend`` implementation.  Parity: the feature front end and the FastConformer are pinned by the independent
ports of those NeMo modules in `transformers` (tests/golden/sortformer_hf_kat.npz); the Transformer blocks' wiring, the
sigmoid head and the speaker-cache update are restatements (no NeMo, no weights offline).
"""
from __future__ import annotations

import ctypes as C
import threading
from dataclasses import dataclass
from typing import List, Optional, Protocol, Tuple

import numpy as np

from . import _lib
from .melbank import mel_filterbank


@dataclass
class SpeakerSegment:
    """whisperlivekit/timed_objects.py:88-93"""
    start: Optional[float] = 0
    end: Optional[float] = 0
    speaker: Optional[int] = -1


@dataclass(frozen=True)
class SortformerStreamingParams:
    """The streaming configuration the reference forces on the model (sortformer_backend.py:120-128)."""
    chunk_len: int = 10
    subsampling_factor: int = 10
    chunk_right_context: int = 0
    chunk_left_context: int = 10
    spkcache_len: int = 188
    fifo_len: int = 188
    spkcache_update_period: int = 144
    window_stride: float = 0.01

    @property
    def chunk_duration_seconds(self) -> float:
        return self.chunk_len * self.subsampling_factor * self.window_stride


def resolve_max_speakers(max_speakers: Optional[int], model_speakers: int) -> int:
    """sortformer_backend.py:134-148: validate a caller-declared speaker cap against the checkpoint."""
    if model_speakers < 1:
        raise ValueError("The Sortformer checkpoint exposes no speaker channels.")
    if max_speakers is None:
        return model_speakers
    if isinstance(max_speakers, bool) or not isinstance(max_speakers, int):
        raise ValueError("max_speakers must be an integer.")
    if not 1 <= max_speakers <= model_speakers:
        raise ValueError(f"max_speakers must be between 1 and {model_speakers} for the loaded Sortformer checkpoint.")
    return max_speakers


def hann_symmetric(n: int) -> np.ndarray:
    """torch.hann_window(n, periodic=False), the window NeMo's FilterbankFeatures uses."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / (n - 1))).astype(np.float32)


class HipMelSpectrogram:
    """128-bin log-mel features on the GPU (wlk_melspec_*): NeMo FilterbankFeatures with window 25 ms,
    stride 10 ms, n_fft 512, pre-emphasis 0.97, log(x + 2^-24), normalize "NA".  ``dither`` (NeMo adds
    1e-5 * N(0,1) because the reference never puts its preprocessor in eval mode) is off by default so that
    the features are reproducible; pass a numpy Generator to get it.

    Frame count: the centred STFT yields ``len // hop + 1`` frames and ``get_features`` returns all of them, but
    ``FilterbankFeatures.get_seq_len`` of the NeMo the reference requires (nemo-toolkit >= 3, pyproject.toml:80-85)
    counts ``len // hop`` VALID frames and fills the rest with pad_value 0 - the rule transformers' port of the module
    (``ParakeetFeatureExtractor``: ``features_lengths``, ``input_features *= mask``) implements and
    tests/golden/sortformer_hf_kat.npz pins.  So a 1.0 s chunk is 100 log-mel frames plus one all-zero frame, and the
    reference hands all 101 to the network (sortformer_backend.py:279-296).  ``seq_len_plus_one=True`` gives the
    pre-2.0 NeMo rule (``len // hop + 1`` valid frames, nothing zeroed)."""

    def __init__(self, device: int = 0, sample_rate: int = 16000, n_mels: int = 128, n_fft: int = 512,
                 window_size: float = 0.025, window_stride: float = 0.01, preemph: float = 0.97,
                 max_seconds: float = 4.0, seq_len_plus_one: bool = False):
        self.lib = _lib.load()
        self.n_mels, self.n_fft = n_mels, n_fft
        self.seq_len_plus_one = seq_len_plus_one
        self.win_length = int(window_size * sample_rate)
        self.hop = int(window_stride * sample_rate)
        filters = np.ascontiguousarray(mel_filterbank(n_mels, sample_rate, n_fft))
        window = hann_symmetric(self.win_length)
        self._h = C.c_void_p()
        _lib.check(self.lib.wlk_melspec_create(device, n_fft, self.win_length, self.hop, n_mels,
                                               filters.ctypes.data_as(C.c_void_p), window.ctypes.data_as(C.c_void_p),
                                               preemph, 2.0 ** -24, int(max_seconds * sample_rate), C.byref(self._h)))

    def __call__(self, pcm: np.ndarray, dither: Optional[np.random.Generator] = None) -> np.ndarray:
        """-> [n_frames, n_mels] float32 (time-major), n_frames = len(pcm) // hop + 1; frames from ``len(pcm) // hop``
        on are zero unless ``seq_len_plus_one`` (see the class docstring)."""
        a = np.ascontiguousarray(pcm, dtype=np.float32).reshape(-1)
        if dither is not None:
            a = a + np.float32(1e-5) * dither.standard_normal(a.shape[0], dtype=np.float32)
        cap = a.shape[0] // self.hop + 2
        out = np.empty((cap, self.n_mels), np.float32)
        n = C.c_int()
        _lib.check(self.lib.wlk_melspec_run(self._h, a.ctypes.data_as(C.c_void_p), a.shape[0],
                                            out.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        if not self.seq_len_plus_one:
            out[a.shape[0] // self.hop: n.value] = 0.0
        return out[: n.value]

    def close(self):
        if self._h:
            self.lib.wlk_melspec_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class SortformerBackend(Protocol):
    """What the session needs from the network (NeMo's SortformerEncLabelModel in the reference)."""
    n_spk: int
    params: SortformerStreamingParams

    def features(self, pcm: np.ndarray) -> np.ndarray:
        """[n_frames, 128] log-mel of one chunk (sortformer_backend.py:273-275)."""

    def new_state(self):
        """Fresh streaming state (spkcache / fifo / silence profile; sortformer_backend.py:212-234)."""

    def forward_streaming_step(self, features: np.ndarray, state, left_offset: int, right_offset: int) -> np.ndarray:
        """Run one chunk [T, 128]; returns this chunk's speaker activities [frames, n_spk] in [0, 1] and
        updates ``state`` in place (NeMo forward_streaming_step, sortformer_backend.py:293-300)."""


class HipSortformerDiarizationOnline:
    """Per-session streaming diarizer (SortformerDiarizationOnline, sortformer_backend.py:151-371)."""

    def __init__(self, shared_model: SortformerBackend, sample_rate: int = 16000, max_speakers: Optional[int] = None):
        self.sample_rate = sample_rate
        self.diarization_segments: List[SpeakerSegment] = []
        self.buffer_audio = np.array([], dtype=np.float32)
        self.segment_lock = threading.Lock()
        self.global_time_offset = 0.0
        self.model = shared_model
        self.max_speakers = resolve_max_speakers(max_speakers, int(shared_model.n_spk))
        self.chunk_duration_seconds = shared_model.params.chunk_duration_seconds
        self.streaming_state = shared_model.new_state()
        self.total_preds = np.zeros((0, int(shared_model.n_spk)), np.float32)
        self._previous_chunk_features: Optional[np.ndarray] = None
        self._chunk_index = 0
        self._len_prediction: Optional[int] = None

    # -- duck type ----------------------------------------------------------------------------------
    def insert_silence(self, silence_duration: Optional[float]):
        with self.segment_lock:
            self.global_time_offset += silence_duration

    def insert_audio_chunk(self, pcm_array: np.ndarray):
        self.buffer_audio = np.concatenate([self.buffer_audio, np.asarray(pcm_array, dtype=np.float32).copy()])

    async def diarize(self):
        return self.diarize_sync()

    def diarize_sync(self) -> List[SpeakerSegment]:
        """One 1.0 s chunk if that much audio is buffered (sortformer_backend.py:253-311)."""
        threshold = int(self.chunk_duration_seconds * self.sample_rate)
        if len(self.buffer_audio) < threshold:
            return []
        audio = self.buffer_audio[:threshold]
        self.buffer_audio = self.buffer_audio[threshold:]
        left_offset = 8 if self._chunk_index > 0 else 0                  # :290-291
        fused = getattr(self.model, "forward_streaming_step_pcm", None)
        if fused is not None:
            # the HIP model takes the audio itself: log-mel, stem and network in ONE launch chain / ONE synchronisation
            # (same kernels, same values as the three calls below)
            prev = self._previous_chunk_features[-99:] if self._previous_chunk_features is not None else None   # :279-283
            chunk_preds, feats = fused(audio, prev, self.streaming_state, left_offset, 8)
            self._previous_chunk_features = feats
        else:
            feats = self.model.features(audio)                            # [frames, 128]
            if self._previous_chunk_features is not None:
                total = np.concatenate([self._previous_chunk_features[-99:], feats], axis=0)   # :279-283
            else:
                total = feats
            self._previous_chunk_features = feats
            chunk_preds = self.model.forward_streaming_step(total, self.streaming_state, left_offset, 8)
        self.total_preds = np.concatenate([self.total_preds, np.asarray(chunk_preds, np.float32)], axis=0)
        keep = max(1024, 4 * (self._len_prediction or 256))              # :305-307
        if self.total_preds.shape[0] > keep:
            self.total_preds = self.total_preds[-keep:]
        new_segments = self._process_predictions()
        self._chunk_index += 1
        return new_segments

    def _process_predictions(self) -> List[SpeakerSegment]:
        """Arg-max over the first ``max_speakers`` arrival-ordered channels, run-length encode the last
        chunk's frames into segments (sortformer_backend.py:313-363)."""
        preds = np.asarray(self.total_preds)
        if preds.ndim == 3:
            preds = preds[0]
        if preds.shape[1] < self.max_speakers:
            raise RuntimeError("Sortformer returned fewer speaker channels than configured.")
        active = np.argmax(preds[:, : self.max_speakers], axis=1)
        if not len(active):
            return []
        if self._len_prediction is None:
            self._len_prediction = len(active)
        frame_duration = self.chunk_duration_seconds / self._len_prediction
        current = active[-self._len_prediction:]
        out: List[SpeakerSegment] = []
        with self.segment_lock:
            base = self._chunk_index * self.chunk_duration_seconds + self.global_time_offset
            spk = current[0]
            start = round(base, 2)
            for idx, s in enumerate(current):
                now = round(base + idx * frame_duration, 2)
                if s != spk:
                    out.append(SpeakerSegment(speaker=spk, start=start, end=now))
                    start, spk = now, s
            out.append(SpeakerSegment(speaker=spk, start=start, end=round(base + len(current) * frame_duration, 2)))
        return out

    def get_segments(self) -> List[SpeakerSegment]:
        with self.segment_lock:
            return self.diarization_segments.copy()

    def close(self):
        with self.segment_lock:
            self.diarization_segments.clear()

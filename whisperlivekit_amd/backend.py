"""Session-level plugin surface ("boundary A") on the HIP backend.

``HipSimulStreamingASR`` is the shared, per-GPU object (the role of SimulStreamingASR,
whisperlivekit/simul_whisper/backend.py:293-570): it owns the packed weights and the AlignAtt config.
``HipSimulStreamingOnlineProcessor`` is the per-session object AudioProcessor drives
(SimulStreamingOnlineProcessor, backend.py:38-290): ``insert_audio_chunk``, ``process_iter``,
``get_buffer``, ``start_silence``, ``end_silence``, ``new_speaker``, ``warmup``, with the reference's
output guards (stale / rewound words, repetition loops).

With WhisperLiveKit installed, :func:`reference_online_processor_class` returns a subclass of the
reference's own processor whose only override is ``_create_alignatt`` - the <=10-line routing hook
INTEGRATION.md describes; everything else (guards included) is then the reference's code.
"""
from __future__ import annotations

import logging
import re
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import policy as P
from .align_att import HipAlignAttStandalone, make_alignatt_class
from .dims import default_alignment_heads, ALIGNMENT_HEADS, MODEL_DIMS, ModelDims
from .engine import HipWhisperModel

logger = logging.getLogger(__name__)

MIN_DURATION_REAL_SILENCE = 5           # backend.py:36
_WORDS = re.compile(r"[^\W_]+(?:'[^\W_]+)*", re.UNICODE)


def build_config(model_name: str, *, lan: str = "en", frame_threshold: int = 25, beams: int = 1,
                 audio_max_len: float = 30.0, audio_min_len: float = 0.0, min_chunk_size: float = 0.1,
                 cif_ckpt_path: Optional[str] = None, never_fire: bool = False,
                 init_prompt: Optional[str] = None, static_init_prompt: Optional[str] = None,
                 max_context_tokens: Optional[int] = None, direct_english_translation: bool = False,
                 nonspeech_prob: float = 0.5, decoder_type: str = "beam") -> P.AlignAttConfig:
    """AlignAttConfig exactly as the engine fills it (backend.py:369-384; defaults from
    whisperlivekit/config.py:101-113).  ``decoder_type`` is hard-wired to "beam" there."""
    return P.AlignAttConfig(
        tokenizer_is_multilingual=not model_name.endswith(".en"), segment_length=min_chunk_size,
        frame_threshold=frame_threshold, language=lan, audio_max_len=audio_max_len,
        audio_min_len=audio_min_len, cif_ckpt_path=cif_ckpt_path, decoder_type=decoder_type,
        beam_size=beams, task="translate" if direct_english_translation else "transcribe",
        never_fire=never_fire, init_prompt=init_prompt, max_context_tokens=max_context_tokens,
        static_init_prompt=static_init_prompt, nonspeech_prob=nonspeech_prob)


def load_openai_checkpoint(path: str):
    """``.pt`` checkpoints in the openai layout: {"dims": {...}, "model_state_dict": {...}}
    (the first branch of load_model, whisper/__init__.py:520-560).  Returns (ModelDims, state_dict)."""
    import torch
    ck = torch.load(path, map_location="cpu", weights_only=True)
    if "dims" not in ck or "model_state_dict" not in ck:
        raise ValueError(f"{path}: not an openai-whisper checkpoint (expected 'dims' and 'model_state_dict')")
    return ModelDims(**ck["dims"]), ck["model_state_dict"]


class HipSimulStreamingASR:
    """Shared model owner for one GPU.  ``sep`` and the attribute names mirror what
    AudioProcessor / online_factory read from the reference object (audio_processor.py:203,
    core.py:246-271)."""
    sep = ""

    def __init__(self, model_size: str = "base.en", *, device: int = 0, model_path: Optional[str] = None,
                 state_dict=None, dims: Optional[ModelDims] = None, alignment_heads=None,
                 synthetic_seed: Optional[int] = None, hip_model: Optional[HipWhisperModel] = None,
                 custom_alignment_heads: Optional[Sequence[Tuple[int, int]]] = None,
                 hw_queues: Optional[int] = None, lora_path: Optional[str] = None, **cfg_kwargs):
        if hw_queues is not None:     # explicit deployment knob (process-wide, see _lib.configure_hw_queues); default: off
            from . import _lib
            _lib.configure_hw_queues(hw_queues)
        self.model_name = model_size
        self.cfg = build_config(model_size, **cfg_kwargs)
        self.tokenizer = None
        self.use_full_mlx = False
        self.mlx_encoder = self.fw_encoder = self.mlx_model = None
        self.fast_encoder = False
        heads = custom_alignment_heads or alignment_heads
        if lora_path is not None and (model_path is None or hip_model is not None):
            # the adapter is merged while the checkpoint is read (checkpoint.load_whisper_checkpoint); a ready model, a state dict
            # or synthetic weights would silently serve the un-adapted model (the reference merges it for every load)
            raise ValueError("lora_path needs model_path (the adapter is merged at checkpoint load)")
        if hip_model is not None:
            self.hip_model = hip_model
        elif model_path is not None:
            # any LOCAL checkpoint load_model takes (whisper/__init__.py:466-596): file or directory, .pt / .bin /
            # .safetensors / shards, openai / HuggingFace / MLX names, an optional LoRA adapter merged in
            from .checkpoint import load_whisper_checkpoint
            d, sd, own_heads = load_whisper_checkpoint(model_path, lora_path)
            # Alignment heads of a PATH load, as the reference resolves them (whisper/__init__.py:546-552, model.py:353-361;
            # simul_whisper/backend.py:530-539 hands load_model the path, not the size name): custom heads, else the pairs the
            # checkpoint carries (MLX exports), else Whisper.__init__'s default - every head of the upper half of the decoder
            # layers.  The per-size tables (dims.ALIGNMENT_HEADS) belong to the official NAMES the reference downloads.
            self.hip_model = HipWhisperModel.from_state_dict(d, sd, heads or own_heads or default_alignment_heads(d), device)
        elif state_dict is not None:
            d = dims or MODEL_DIMS[model_size]
            self.hip_model = HipWhisperModel.from_state_dict(
                d, state_dict, heads or ALIGNMENT_HEADS.get(model_size), device)
        elif synthetic_seed is not None:
            self.hip_model = HipWhisperModel.synthetic(model_size, synthetic_seed, device)
        else:
            raise ValueError("give one of hip_model, model_path, state_dict or synthetic_seed")
        self.shared_model = self.hip_model      # the name the reference processor reads

    def transcribe(self, audio):                # backend.py:566-570: warm-up happens at load time
        pass

    def warmup(self, audio: np.ndarray) -> None:
        """Run one throw-away call so the first real chunk does not pay lazy initialisation;
        failure must propagate (tests/test_silent_backend_guard.py in the reference)."""
        model = HipAlignAttStandalone(cfg=self.cfg, hip_model=self.hip_model)
        try:
            model.warmup(audio)
        finally:
            model.close()


def has_repetition_loop(words: List[str], min_words: int = 12) -> bool:
    """Three detectors over the recent word history (backend.py:128-177): a run of >=8 identical
    words; the tail n-gram (n=2..8) repeated back-to-back >=3 times over >=12 words; or one
    n-gram occurring >=4 times and covering >=55 % of the history."""
    if len(words) < min_words:
        return False
    run = 1
    for prev, cur in zip(words, words[1:]):
        run = run + 1 if cur == prev else 1
        if run >= 8:
            return True
    widest = min(8, len(words) // 2)
    for n in range(2, widest + 1):
        reps, end = 1, len(words)
        while end - 2 * n >= 0 and words[end - n:end] == words[end - 2 * n:end - n]:
            reps += 1
            end -= n
        if reps >= 3 and reps * n >= min_words:
            return True
    for n in range(2, widest + 1):
        seen = {}
        for i in range(len(words) - n + 1):
            key = tuple(words[i:i + n])
            seen[key] = seen.get(key, 0) + 1
        top = max(seen.values()) if seen else 0
        if top >= 4 and top * n >= min_words and top * n / len(words) >= 0.55:
            return True
    return False


class HipSimulStreamingOnlineProcessor:
    """Per-session online processor (duck type consumed at audio_processor.py:668-758)."""
    SAMPLING_RATE = 16000
    _COMMITTED_EPSILON = 0.05
    _INTRA_BATCH_REWIND_SECONDS = 0.75
    _REWIND_RESET_SECONDS = 1.0
    _RECENT_WORD_HISTORY = 80
    _MIN_REPETITION_WORDS = 12

    def __init__(self, asr: HipSimulStreamingASR, logfile=None):
        self.asr = asr
        self.logfile = logfile
        self.end = 0.0
        self.buffer: list = []
        self.model = self._create_alignatt()
        self._last_committed_end = 0.0
        self._recent_words: List[str] = []
        if asr.tokenizer:
            self.model.tokenizer = asr.tokenizer
            self.model.state.tokenizer = asr.tokenizer

    def _create_alignatt(self):
        return HipAlignAttStandalone(cfg=self.asr.cfg, hip_model=self.asr.hip_model)

    # -- audio / events ---------------------------------------------------------------------------
    def insert_audio_chunk(self, audio: np.ndarray, audio_stream_end_time: float):
        self.end = audio_stream_end_time
        self.model.insert_audio(np.asarray(audio, dtype=np.float32))

    def insert_pcm16_chunk(self, pcm, audio_stream_end_time: float):
        """The same insert for s16le PCM as it arrives from the client (bytes or an int16 array): what
        AudioProcessor.convert_pcm_to_float (audio_processor.py:416-418) + insert_audio_chunk do, with the
        int16 -> fp32 / 32768 widening done on the GPU."""
        self.end = audio_stream_end_time
        a = np.frombuffer(pcm, dtype=np.int16) if isinstance(pcm, (bytes, bytearray, memoryview)) else np.asarray(pcm)
        if a.dtype != np.int16:
            raise TypeError("insert_pcm16_chunk expects s16le bytes or an int16 array")
        self.model.insert_audio(a)

    def start_silence(self):
        return self.process_iter(is_last=True)

    def end_silence(self, silence_duration: float, offset: float):
        """Short gaps become zeros in the audio buffer; >= 5 s starts a new segment (backend.py:77-93)."""
        self.end += silence_duration
        if silence_duration < MIN_DURATION_REAL_SILENCE:
            gap = int(16000 * silence_duration)
            if gap > 0:
                on_device = getattr(self.model, "insert_silence", None)
                if on_device is not None:
                    on_device(gap)                  # zeros are written on the device; nothing crosses PCIe
                else:
                    self.model.insert_audio(np.zeros(gap, dtype=np.float32))
            return
        self.model.refresh_segment(complete=True)
        self.model.global_time_offset = silence_duration + offset
        self._last_committed_end = max(self._last_committed_end, self.model.global_time_offset)
        self._recent_words = []

    def new_speaker(self, change_speaker) -> Tuple[list, float]:
        tokens, upto = self.process_iter(is_last=True)
        self.model.refresh_segment(complete=True)
        self.model.speaker = change_speaker.speaker
        self.model.global_time_offset = change_speaker.start
        self._last_committed_end = max(self._last_committed_end, change_speaker.start)
        self._recent_words = []
        return tokens or [], upto

    def get_buffer(self):
        text = "".join(t.text for t in self.buffer)
        if self.buffer:
            return P.ASRToken(start=self.buffer[0].start, end=self.buffer[-1].end, text=text)
        return P.ASRToken(start=None, end=None, text=text)

    # -- guards -------------------------------------------------------------------------------------
    @staticmethod
    def _spoken(tokens) -> List[str]:
        return [w for t in tokens for w in _WORDS.findall((t.text or "").casefold())]

    def _filter_stable_words(self, tokens):
        """Drop words that end at or before the committed time (+50 ms) and words that jump back
        more than 0.75 s inside one batch (backend.py:186-225)."""
        kept = []
        horizon = self._last_committed_end
        for t in tokens:
            start = float(t.start or 0.0)
            end = float(t.end or start)
            if end < start or end <= self._last_committed_end + self._COMMITTED_EPSILON:
                continue
            if kept and horizon - end > self._INTRA_BATCH_REWIND_SECONDS:
                continue
            kept.append(t)
            horizon = max(horizon, end)
        return kept

    def _reset_after_unstable_output(self, reason: str):
        logger.warning("[SimulStreaming guard] %s; resetting current segment", reason)
        self.model.refresh_segment(complete=True)
        self.model.global_time_offset = max(self._last_committed_end, self.end)
        self.buffer = []
        self._recent_words = []

    # -- the call AudioProcessor times ------------------------------------------------------------------
    def process_iter(self, is_last: bool = False) -> Tuple[list, float]:
        try:
            words = self.model.infer(is_last=is_last)
            if not words:
                return [], self.end
            if self.model.cfg.language == "auto" and words[0].detected_language is None:
                self.buffer.extend(words)
                return [], self.end
            stable = self._filter_stable_words(words)
            if not stable:
                newest = max(float(t.end or 0.0) for t in words)
                if self._last_committed_end - newest > self._REWIND_RESET_SECONDS:
                    self._reset_after_unstable_output(
                        f"all emitted words rewound behind committed time {self._last_committed_end:.2f}s")
                self.buffer = []
                return [], self.end
            if has_repetition_loop(self._recent_words + self._spoken(stable), self._MIN_REPETITION_WORDS):
                self._reset_after_unstable_output("repetition loop detected")
                return [], self.end
            self.buffer = []
            self._last_committed_end = max(self._last_committed_end, max(float(t.end or 0.0) for t in stable))
            self._recent_words = (self._recent_words + self._spoken(stable))[-self._RECENT_WORD_HISTORY:]
            return stable, self.end
        except Exception as e:  # the reference swallows everything here (backend.py:266-268)
            logger.exception("SimulStreaming processing error: %s", e)
            self.last_error = e
            return [], self.end

    def warmup(self, audio, init_prompt: str = ""):
        try:
            self.model.insert_audio(np.asarray(audio, dtype=np.float32))
            self.model.infer(True)
            self.model.refresh_segment(complete=True)
        except Exception as e:
            logger.exception("SimulStreaming warmup failed: %s", e)

    def close(self):
        self.model.close()


def reference_online_processor_class():
    """The drop-in form: the REFERENCE's SimulStreamingOnlineProcessor with one method replaced.
    Requires WhisperLiveKit to be importable; the ASR object handed to it must expose ``hip_model``
    (HipSimulStreamingASR does) next to the attributes the reference already reads."""
    from whisperlivekit.simul_whisper.backend import SimulStreamingOnlineProcessor  # type: ignore
    hip_alignatt = make_alignatt_class()

    class HipRoutedOnlineProcessor(SimulStreamingOnlineProcessor):
        def _create_alignatt(self):
            return hip_alignatt(cfg=self.asr.cfg, hip_model=self.asr.hip_model)

    return HipRoutedOnlineProcessor

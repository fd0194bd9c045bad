"""LocalAgreement's batch ASR wrapper on the HIP library: the role of ``WhisperASR`` (whisperlivekit/local_agreement/
backends.py:39-98) under the reference's ``OnlineASRProcessor`` (local_agreement/online_asr.py:94-427), which calls
``transcribe(audio_buffer, init_prompt=...)``, ``ts_words(result)``, ``segments_end_ts(result)`` and reads ``sep``.

The policy (HypothesisBuffer, buffer trimming) is host logic of the reference and is not restated here: this object plugs
into it.  ``transcribe`` is :func:`whisperlivekit_amd.transcribe.transcribe` with the arguments the reference wrapper
passes (language, ``initial_prompt``, ``condition_on_previous_text=True``, ``word_timestamps=True`` and the user's
``transcribe_kargs`` minus the VAD switches).  Attributes the reference's ``backend_factory`` sets afterwards
(``tokenizer``, ``confidence_validation``, ``buffer_trimming``, ``buffer_trimming_sec``, ``backend_choice``) are plain
attributes here as well.
"""
from __future__ import annotations

import logging
import sys
from typing import List, Optional

from .backend import load_openai_checkpoint
from .dims import ALIGNMENT_HEADS, MODEL_DIMS
from .engine import HipWhisperModel
from .policy import ASRToken
from .transcribe import transcribe as hip_transcribe

logger = logging.getLogger(__name__)


class HipWhisperASR:
    sep = " "         # words of `transcribe` are joined with a space (backends.py:16-17, 41)

    def __init__(self, lan: str, model_size: Optional[str] = None, cache_dir: Optional[str] = None,
                 model_dir: Optional[str] = None, lora_path: Optional[str] = None, logfile=sys.stderr, *, device: int = 0,
                 hip_model: Optional[HipWhisperModel] = None, state_dict=None, synthetic_seed: Optional[int] = None):
        self.logfile = logfile
        self.transcribe_kargs = {}
        self.lora_path = lora_path
        self.original_language = None if lan == "auto" else lan
        self.tokenizer = None
        self.confidence_validation = False
        self.buffer_trimming = "segment"
        self.buffer_trimming_sec = 15
        self.backend_choice = "whisper"
        if lora_path is not None and model_dir is None:
            raise ValueError("lora_path needs model_dir: the adapter is merged into the checkpoint's weights at load time "
                             "(whisper/__init__.py:337-391)")
        if hip_model is not None:
            self.model = hip_model
        else:
            self.model = self.load_model(model_size, cache_dir, model_dir, device=device, state_dict=state_dict,
                                         synthetic_seed=synthetic_seed)

    def load_model(self, model_size=None, cache_dir=None, model_dir=None, *, device: int = 0, state_dict=None,
                   synthetic_seed: Optional[int] = None) -> HipWhisperModel:
        """A local checkpoint in any layout the reference's load_model takes (``model_dir`` = file or directory,
        backends.py:44-61; openai / HuggingFace / MLX names, safetensors, shards, a LoRA adapter merged in:
        checkpoint.load_whisper_checkpoint), a state_dict for a named size, or seeded random weights of a named size (parity /
        timing runs: there is no checkpoint where this is built)."""
        if model_dir is not None:
            from .checkpoint import load_whisper_checkpoint
            dims, sd, own_heads = load_whisper_checkpoint(str(model_dir), self.lora_path)
            return HipWhisperModel.from_state_dict(dims, sd, own_heads or ALIGNMENT_HEADS.get(model_size), device)
        if model_size is None:
            raise ValueError("Either model_size or model_dir must be set for HipWhisperASR")
        if state_dict is not None:
            return HipWhisperModel.from_state_dict(MODEL_DIMS[model_size], state_dict, ALIGNMENT_HEADS.get(model_size), device)
        if synthetic_seed is not None:
            return HipWhisperModel.synthetic(model_size, synthetic_seed, device)
        raise ValueError("no weights: give model_dir (a .pt checkpoint), state_dict or synthetic_seed")

    def transcribe(self, audio, init_prompt: str = "") -> dict:
        options = dict(self.transcribe_kargs)
        options.pop("vad", None)
        options.pop("vad_filter", None)
        return hip_transcribe(self.model, audio, language=self.original_language or None, initial_prompt=init_prompt,
                              condition_on_previous_text=True, word_timestamps=True, **options)

    def ts_words(self, result: dict) -> List[ASRToken]:
        return [ASRToken(w["start"], w["end"], w["word"], probability=w.get("probability"))
                for segment in result["segments"] for w in segment["words"]]

    def segments_end_ts(self, result: dict) -> List[float]:
        return [segment["end"] for segment in result["segments"]]

    def use_vad(self):
        logger.warning("VAD is not currently supported for the Whisper backend and will be ignored.")

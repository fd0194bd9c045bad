"""Model dimension tables and alignment-head lists for the Whisper family.

The numbers restate facts of the reference:
  * ``ModelDimensions`` fields            -> whisperlivekit/whisper/model.py:26-37
  * per-model alignment heads (decoded)   -> whisperlivekit/whisper/__init__.py:39-54
    (the reference stores them as base85+gzip boolean masks; we keep the decoded
    ``(decoder_layer, head)`` pairs, in the row-major order ``mask.to_sparse().indices()``
    yields, because that order defines the "alignment head rank" used by AlignAtt,
    whisperlivekit/simul_whisper/simul_whisper.py:151-159).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict, List, Tuple

# fixed audio front-end constants, whisperlivekit/whisper/audio.py:13-22
SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
N_SAMPLES = 480000          # 30 s
N_FRAMES = 3000             # mel frames fed to the encoder
N_FREQ = N_FFT // 2 + 1     # 201
TOKENS_PER_SECOND = 50      # encoder positions per second (hop*2)


@dataclass(frozen=True)
class ModelDims:
    n_mels: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_vocab: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int

    def as_tuple(self) -> Tuple[int, ...]:
        return tuple(asdict(self).values())

    @property
    def is_multilingual(self) -> bool:
        return self.n_vocab >= 51865

    @property
    def num_languages(self) -> int:
        return self.n_vocab - 51765 - int(self.is_multilingual)


def _d(mels, a_state, a_head, a_layer, vocab, t_state, t_head, t_layer) -> ModelDims:
    return ModelDims(mels, 1500, a_state, a_head, a_layer, vocab, 448, t_state, t_head, t_layer)


MODEL_DIMS: Dict[str, ModelDims] = {
    # "micro" is not a released checkpoint: it is the smallest shape that keeps every
    # structural constant of the path (1500 audio positions, 448 text positions, 64-wide
    # heads, the .en vocabulary) and is used by the parity tests to stay fast on CPU.
    "micro.en": _d(80, 128, 2, 2, 51864, 128, 2, 2),
    "micro": _d(80, 128, 2, 2, 51865, 128, 2, 2),      # multilingual twin (language auto-detect tests)
    "tiny.en": _d(80, 384, 6, 4, 51864, 384, 6, 4),
    "tiny": _d(80, 384, 6, 4, 51865, 384, 6, 4),
    "base.en": _d(80, 512, 8, 6, 51864, 512, 8, 6),
    "base": _d(80, 512, 8, 6, 51865, 512, 8, 6),
    "small.en": _d(80, 768, 12, 12, 51864, 768, 12, 12),
    "small": _d(80, 768, 12, 12, 51865, 768, 12, 12),
    "medium.en": _d(80, 1024, 16, 24, 51864, 1024, 16, 24),
    "medium": _d(80, 1024, 16, 24, 51865, 1024, 16, 24),
    "large-v1": _d(80, 1280, 20, 32, 51865, 1280, 20, 32),
    "large-v2": _d(80, 1280, 20, 32, 51865, 1280, 20, 32),
    "large-v3": _d(128, 1280, 20, 32, 51866, 1280, 20, 32),
    "large-v3-turbo": _d(128, 1280, 20, 32, 51866, 1280, 20, 4),
}
# the reference's aliases (whisper/__init__.py:20-35: "large" and "turbo" point at the v3 checkpoints)
MODEL_DIMS["large"] = MODEL_DIMS["large-v3"]
MODEL_DIMS["turbo"] = MODEL_DIMS["large-v3-turbo"]

ALIGNMENT_HEADS: Dict[str, List[Tuple[int, int]]] = {
    "micro.en": [(1, 0), (1, 1)],
    "micro": [(1, 0), (1, 1)],
    "tiny.en": [(1, 0), (2, 0), (2, 5), (3, 0), (3, 1), (3, 2), (3, 3), (3, 4)],
    "tiny": [(2, 2), (3, 0), (3, 2), (3, 3), (3, 4), (3, 5)],
    "base.en": [(3, 3), (4, 7), (5, 1), (5, 5), (5, 7)],
    "base": [(3, 1), (4, 2), (4, 3), (4, 7), (5, 1), (5, 2), (5, 4), (5, 6)],
    "small.en": [(6, 6), (7, 0), (7, 3), (7, 8), (8, 2), (8, 5), (8, 7), (9, 0), (9, 4), (9, 8),
                 (9, 10), (10, 0), (10, 1), (10, 2), (10, 3), (10, 6), (10, 11), (11, 2), (11, 4)],
    "small": [(5, 3), (5, 9), (8, 0), (8, 4), (8, 7), (8, 8), (9, 0), (9, 7), (9, 9), (10, 5)],
    "medium.en": [(11, 4), (14, 1), (14, 12), (14, 14), (15, 4), (16, 0), (16, 4), (16, 9),
                  (17, 12), (17, 14), (18, 7), (18, 10), (18, 15), (20, 0), (20, 3), (20, 9),
                  (20, 14), (21, 12)],
    "medium": [(13, 15), (15, 4), (15, 15), (16, 1), (20, 0), (23, 4)],
    "large-v1": [(9, 19), (11, 2), (11, 4), (11, 17), (22, 7), (22, 11), (22, 17), (23, 2), (23, 15)],
    "large-v2": [(10, 12), (13, 17), (16, 11), (16, 12), (16, 13), (17, 15), (17, 16), (18, 4),
                 (18, 11), (18, 19), (19, 11), (21, 2), (21, 3), (22, 3), (22, 9), (22, 12),
                 (23, 5), (23, 7), (23, 13), (25, 5), (26, 1), (26, 12), (27, 15)],
    "large-v3": [(7, 0), (10, 17), (12, 18), (13, 12), (16, 1), (17, 14), (19, 11), (21, 4),
                 (24, 1), (25, 6)],
    "large-v3-turbo": [(2, 4), (2, 11), (3, 3), (3, 6), (3, 11), (3, 14)],
}
ALIGNMENT_HEADS["large"] = ALIGNMENT_HEADS["large-v3"]
ALIGNMENT_HEADS["turbo"] = ALIGNMENT_HEADS["large-v3-turbo"]


def default_alignment_heads(dims: ModelDims) -> List[Tuple[int, int]]:
    """Every head of the upper half of the decoder (whisper/model.py:352-356)."""
    return [(l, h) for l in range(dims.n_text_layer // 2, dims.n_text_layer)
            for h in range(dims.n_text_head)]

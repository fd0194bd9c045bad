"""Config 5 session glue: the per-session translation object ``AudioProcessor.translation_processor`` drives
(whisperlivekit/audio_processor.py:887-920), over the NLLB / M2M-100 network of :mod:`whisperlivekit_amd.nllb`.

What the reference does: ``TranscriptionEngine`` loads ONE shared model through the third-party ``nllw`` package
(``core.py:320-329``: ``nllw.load_model([source], nllb_backend=, nllb_size=)``) and every session gets its own
``nllw.OnlineTranslation(model, [source], [target])`` (``core.py:483-493``; per-session target languages:
``translation.py:17-47``).  ``nllw`` is NOT in the reference tree and no test there pins its numerics or its policy.
What IS in the tree is the contract those objects must meet - the four calls at ``audio_processor.py:903-911`` - and a
second implementation of that contract, ``AlignAttTranslationClient`` (``translation_alignatt.py:99-181``, "duck-typed
contract (mirrors nllw.OnlineTranslation)"):

* ``insert_tokens(items)``: the ASR's newly committed ``ASRToken`` s (``HypothesisTail`` items only for backends that
  ask for them with ``wants_hypothesis_tail`` - this one does not);
* ``process() -> (Translation | None, TimedText)``: newly VALIDATED target text (appended to ``state.new_translation``,
  append-only on screen) and the current unstable buffer (replaces ``state.new_translation_buffer``);
* ``validate_buffer_and_reset() -> (Translation, TimedText)``: at a silence start / speaker change the open buffer is
  validated as it stands and the session starts a fresh segment;
* ``insert_silence(duration)``.

The policy between those calls (WHEN to re-translate WHICH source prefix, and how much of a hypothesis to validate) is
``nllw``'s own and is restated here FROM THE CALL CONTRACT ONLY, as the standard local-agreement rule of simultaneous
translation (the rule the reference itself uses for ASR hypotheses, ``local_agreement/online_asr.py``): the open source
segment (committed words since the last sentence end) is re-translated whenever it grew; target words on which two
successive hypotheses agree are validated, the rest of the newest hypothesis is the buffer; a source word that carries
sentence punctuation (``TimedText.has_punctuation``, timed_objects.py:28-29) closes the segment - its final translation
is validated whole and the next segment starts with an empty history.  Timestamps are the source words': a validated
piece spans from where the last one ended to the end of the newest source word it was produced from.

Device work per ``process()``: one encoder pass over the open segment and one greedy (or beam) decode -
``nllb.generate`` / ``nllb.beam_search`` with ``forced_bos_token_id`` = the target language code, exactly the calls
``transformers``' NLLB recipe makes.  Tokenisation is the caller's: any object with the ``transformers`` tokenizer
surface this module uses (``src_lang`` attribute, ``__call__(text).input_ids``, ``convert_tokens_to_ids(lang)``,
``decode(ids, skip_special_tokens=True)``) - ``transformers.NllbTokenizer`` over the checkpoint's
``sentencepiece.bpe.model`` in deployment, a seeded stand-in in the tests (no SentencePiece model exists offline).
No CPU fallback: the model is a :class:`whisperlivekit_amd.nllb.HipNllbModel`.
"""
from __future__ import annotations

import logging
import threading
from dataclasses import dataclass, field
from typing import Any, List, Optional, Sequence, Tuple

from . import nllb

logger = logging.getLogger(__name__)

PUNCTUATION_MARKS = {".", "!", "?", "。", "！", "？"}        # timed_objects.py:4


@dataclass
class TimedText:
    """whisperlivekit/timed_objects.py:19-44 (the fields the translation path reads / writes)."""
    start: Optional[float] = 0
    end: Optional[float] = 0
    text: Optional[str] = ""
    speaker: Optional[int] = -1
    detected_language: Optional[str] = None

    def has_punctuation(self) -> bool:
        return any(ch in PUNCTUATION_MARKS for ch in (self.text or "").strip())

    def __bool__(self) -> bool:
        return bool(self.text)


@dataclass
class Translation(TimedText):
    """timed_objects.py:96-97"""


def _has_punctuation(item: Any) -> bool:
    fn = getattr(item, "has_punctuation", None)
    if callable(fn):
        return bool(fn())
    return any(ch in PUNCTUATION_MARKS for ch in (getattr(item, "text", "") or "").strip())


def _common_prefix(a: Sequence[str], b: Sequence[str]) -> int:
    n = 0
    for x, y in zip(a, b):
        if x != y:
            break
        n += 1
    return n


class HipNllbTranslationModel:
    """The server-wide handle ``nllw.load_model`` returns in the reference (``TranscriptionEngine.translation_model``,
    core.py:320-329): one network per GPU shared by every session, plus the tokenizer and the decoding options.  Device
    sessions are 1-row and cheap; each ``HipOnlineTranslation`` owns one, so sessions never share decoder caches."""

    def __init__(self, model: nllb.HipNllbModel, tokenizer: Any, num_beams: int = 1, max_new_tokens: int = 199,
                 max_source_tokens: int = 200):
        self.model, self.tokenizer = model, tokenizer
        self.num_beams, self.max_new_tokens, self.max_source_tokens = int(num_beams), int(max_new_tokens), int(max_source_tokens)
        # a tokenizer with a mutable src_lang (transformers' NllbTokenizer) is shared by all sessions
        self.tokenizer_lock = threading.Lock()

    def language_id(self, code: str) -> int:
        tid = self.tokenizer.convert_tokens_to_ids(code)
        unk = getattr(self.tokenizer, "unk_token_id", None)
        if tid is None or (unk is not None and tid == unk):
            raise ValueError(f"unknown NLLB language code {code!r}")          # translation.py:40-46 catches ValueError
        return int(tid)

    def encode(self, text: str, src_lang: str) -> List[int]:
        with self.tokenizer_lock:
            self.tokenizer.src_lang = src_lang
            ids = list(self.tokenizer(text).input_ids)
        if len(ids) > self.max_source_tokens:                 # keep the language code (first) and </s> (last)
            ids = ids[:1] + ids[-(self.max_source_tokens - 1):]
        return ids

    def decode(self, ids: Sequence[int]) -> str:
        return self.tokenizer.decode(list(ids), skip_special_tokens=True)

    def new_session(self, source_language: str, target_language: str) -> "HipOnlineTranslation":
        return HipOnlineTranslation(self, [source_language], [target_language])


@dataclass
class _Segment:
    tokens: List[Any] = field(default_factory=list)          # source words (ASRToken-like) of the open sentence

    @property
    def start(self) -> Optional[float]:
        return self.tokens[0].start if self.tokens else None

    @property
    def end(self) -> Optional[float]:
        return self.tokens[-1].end if self.tokens else None

    def text(self) -> str:
        # ASR words carry their own leading spaces (simul_whisper) or not (LocalAgreement: asr.sep): normalise
        return " ".join((t.text or "").strip() for t in self.tokens if (t.text or "").strip())


class HipOnlineTranslation:
    """``nllw.OnlineTranslation(model, [source], [target])`` for one session (constructed at core.py:490-493 /
    translation.py:36-39); duck type of audio_processor.py:903-911.  One call in flight per session
    (``translation_processor`` awaits ``to_thread(self.translation.process)``); different sessions run concurrently on
    their own device sessions."""

    wants_hypothesis_tail = False          # audio_processor.py only queues HypothesisTail items to backends that ask

    def __init__(self, translation_model: HipNllbTranslationModel, source_languages: Sequence[str],
                 target_languages: Sequence[str]):
        if not source_languages or not target_languages:
            raise ValueError("source and target language lists must not be empty")
        self.shared = translation_model
        self.source_language, self.target_language = source_languages[0], target_languages[0]
        self.target_id = translation_model.language_id(self.target_language)       # ValueError for an unknown code
        translation_model.language_id(self.source_language)
        self.session = translation_model.model.new_session(rows=max(1, translation_model.num_beams))
        self._segment = _Segment()
        self._closed: List[_Segment] = []          # sentences that ended (punctuation) and await their final translation
        self._validated_words: List[str] = []      # target words of the open segment already handed out
        self._previous: List[str] = []             # previous hypothesis of the open segment (target words)
        self._buffer_words: List[str] = []
        self._dirty = False
        self._last_end: Optional[float] = None     # end time of the last validated piece
        self._silence = 0.0
        self.translations = 0                      # device translations run (bench / tests)

    # ---- duck type --------------------------------------------------------------------------------------------------
    def insert_tokens(self, items: List[Any]) -> None:
        for item in items:
            if type(item).__name__ == "HypothesisTail" or not hasattr(item, "text") or not hasattr(item, "end"):
                continue
            if not (item.text or "").strip():
                continue
            self._segment.tokens.append(item)
            self._dirty = True
            if _has_punctuation(item):
                self._closed.append(self._segment)
                self._segment = _Segment()

    def process(self) -> Tuple[Optional[Translation], TimedText]:
        pieces: List[str] = []
        end: Optional[float] = None
        start = self._piece_start(self._closed[0].start if self._closed else self._segment.start)
        for seg in self._closed:                      # finished sentences: their last translation is final
            words = self._translate(seg)
            # validated text is append-only: only what lies behind it is new.  "Behind it" is by CONTENT, not by count - a
            # final hypothesis that rewrote or shortened the validated prefix continues from where the two still agree, so no
            # word is dropped or printed twice (the open-sentence path below makes the same check)
            keep = _common_prefix(words, self._validated_words)
            if keep != len(self._validated_words):
                logger.debug("final translation rewrites %d validated word(s)", len(self._validated_words) - keep)
            pieces += words[keep:]
            end = seg.end
            self._validated_words, self._previous, self._buffer_words = [], [], []
        self._closed = []
        if self._segment.tokens and self._dirty:      # the open sentence: local agreement of two successive hypotheses
            hyp = self._translate(self._segment)
            agreed = _common_prefix(hyp, self._previous)
            n_val = len(self._validated_words)
            if hyp[:n_val] != self._validated_words:
                # the new hypothesis rewrites validated text: on screen that text is append-only, so it stays; nothing
                # new is validated until the hypotheses settle behind it
                agreed = 0
            if agreed > n_val:
                pieces += hyp[n_val:agreed]
                self._validated_words = hyp[:agreed]
                end = self._segment.end
            self._previous = hyp
            self._buffer_words = hyp[len(self._validated_words):] if hyp[:len(self._validated_words)] == self._validated_words else []
        self._dirty = False
        new = None
        if pieces:
            new = Translation(start=start, end=end if end is not None else start, text=" ".join(pieces))
            self._last_end = new.end
        return new, self._buffer()

    def validate_buffer_and_reset(self) -> Tuple[Translation, TimedText]:
        """Silence start / speaker change (audio_processor.py:903-908): what is on screen as the buffer becomes validated
        text, pending sentences are translated now, and the next words start a fresh segment."""
        pending, buffer_words = None, list(self._buffer_words)
        if self._closed or (self._segment.tokens and self._dirty):
            pending, _ = self.process()
            buffer_words = list(self._buffer_words)
        start = self._piece_start(self._segment.start)
        text = " ".join(([pending.text] if pending else []) + buffer_words)
        end = self._segment.end if (buffer_words and self._segment.end is not None) else (pending.end if pending else start)
        validated = Translation(start=pending.start if pending else start, end=end, text=text)
        if validated.text:
            self._last_end = validated.end
        self._segment = _Segment()
        self._validated_words, self._previous, self._buffer_words = [], [], []
        self._dirty = False
        return validated, TimedText()

    def insert_silence(self, duration: Optional[float]) -> None:
        """audio_processor.py:909-911: the ASR tokens that follow already carry the shifted times; kept for the record."""
        self._silence += float(duration or 0.0)

    def close(self) -> None:
        self.session.close()

    # ---- internals --------------------------------------------------------------------------------------------------
    def _piece_start(self, fallback: Optional[float]) -> float:
        if self._last_end is not None:
            return self._last_end
        return fallback if fallback is not None else 0.0

    def _buffer(self) -> TimedText:
        if not self._buffer_words:
            return TimedText()
        return TimedText(start=self._piece_start(self._segment.start), end=self._segment.end, text=" ".join(self._buffer_words))

    def _translate(self, seg: _Segment) -> List[str]:
        m = self.shared
        src = m.encode(seg.text(), self.source_language)
        if m.num_beams > 1:
            out = nllb.beam_search(self.session, src, self.target_id, num_beams=m.num_beams, max_new_tokens=m.max_new_tokens)
        else:
            out = nllb.generate(self.session, src, self.target_id, max_new_tokens=m.max_new_tokens)
        self.translations += 1
        return m.decode(out).split()


def online_translation_factory(translation_model: HipNllbTranslationModel, source_language: str, target_language: str,
                               fallback_target: Optional[str] = None) -> HipOnlineTranslation:
    """core.py:483-493 / translation.py:17-47 for this backend: a session object for ``target_language``; an unknown
    per-session target falls back to the server-wide one (``fallback_target``) like translation.py:40-47."""
    try:
        return HipOnlineTranslation(translation_model, [source_language], [target_language])
    except ValueError:
        if fallback_target is None or fallback_target == target_language:
            raise
        return HipOnlineTranslation(translation_model, [source_language], [fallback_target])

"""Word-timestamp alignment pieces of LocalAgreement's batch Whisper on the HIP library.

Mirrors the reference names of ``whisperlivekit/whisper/timing.py``: ``dtw`` (:141-152) returns the warping path as
``[text_indices, time_indices]`` exactly as ``dtw_cpu`` + ``backtrace`` do (:58-105).  The recurrence runs on the GPU
(``wlk_dtw``, csrc/dtw.hip); walking the trace back is a few hundred integer steps and stays on the host.  There is no
CPU fallback: without the library / a GPU this raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

MAX_ROWS = 1024


def backtrace(trace: np.ndarray) -> np.ndarray:
    """timing.py:58-79: from the last cell back to the origin; row 0 only moves left, column 0 only moves up."""
    steps = np.array(trace, dtype=np.int8, copy=True)
    steps[0, :] = 2
    steps[:, 0] = 1
    i, j = steps.shape[0] - 1, steps.shape[1] - 1
    path = []
    while i > 0 or j > 0:
        path.append((i - 1, j - 1))
        code = steps[i, j]
        if code == 0:
            i, j = i - 1, j - 1
        elif code == 1:
            i -= 1
        elif code == 2:
            j -= 1
        else:
            raise ValueError(f"unexpected trace code {code} at ({i}, {j})")
    return np.array(path[::-1], dtype=np.int64).T.reshape(2, -1)


def dtw_trace(x: np.ndarray, device: int = 0) -> np.ndarray:
    """The (N + 1) x (M + 1) step-code array of ``dtw_cpu`` for the cost matrix ``x`` [N tokens, M frames]."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.ndim != 2 or x.shape[0] < 1 or x.shape[1] < 1:
        raise ValueError(f"dtw: expected a non-empty 2-D cost matrix, got shape {x.shape}")
    n, m = x.shape
    trace = np.empty((n + 1, m + 1), dtype=np.int8)
    lib = _lib.load()
    _lib.check(lib.wlk_dtw(int(device), x.ctypes.data_as(C.POINTER(C.c_float)), n, m,
                           trace.ctypes.data_as(C.POINTER(C.c_int8))))
    return trace


def dtw(x, device: int = 0) -> np.ndarray:
    """timing.py:141-152: ``text_indices, time_indices = dtw(-matrix)``."""
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return backtrace(dtw_trace(np.asarray(x), device))


# ---- from a warping path to words (host logic of find_alignment / add_word_timestamps) ---------------------------------
# Everything below is integer / float bookkeeping on a few hundred words per window; it stays in Python next to the
# reference's own.  Pinned by tests/golden/word_timing_kat.json (scripts/gen_golden_word_timing.py runs the reference).
from dataclasses import dataclass, field  # noqa: E402
from typing import List, Sequence  # noqa: E402

TOKENS_PER_SECOND = 50            # whisper/audio.py: 20 ms per encoder position
HOP_LENGTH, SAMPLE_RATE = 160, 16000
SENTENCE_END_MARKS = ".。!！?？"
PREPEND_PUNCTUATIONS = "\"'“¿([{-"
APPEND_PUNCTUATIONS = "\"'.。,，!！?？:：”)]}、"


@dataclass
class WordTiming:
    """timing.py:155-161 (same field names: the segments built from it are read by the reference's callers)."""
    word: str
    tokens: List[int] = field(default_factory=list)
    start: float = 0.0
    end: float = 0.0
    probability: float = 0.0


def word_timings(text_indices, time_indices, words: Sequence[str], word_tokens: Sequence[Sequence[int]],
                 text_token_probs: Sequence[float]) -> List[WordTiming]:
    """The tail of find_alignment (timing.py:220-243): the frame at which the path first enters a token is that
    token's start; a word runs from its first token's start to the next word's first token's start.

    ``words`` / ``word_tokens`` come from ``tokenizer.split_to_word_tokens(text_tokens + [eot])`` (the last entry is
    the eot pseudo-word and only closes the last real word)."""
    if len(word_tokens) <= 1:
        return []
    text_indices = np.asarray(text_indices)
    time_indices = np.asarray(time_indices)
    first_of_token = np.ones(len(text_indices), dtype=bool)
    first_of_token[1:] = text_indices[1:] != text_indices[:-1]
    token_start = time_indices[first_of_token] / TOKENS_PER_SECOND
    bounds = np.concatenate([[0], np.cumsum([len(t) for t in word_tokens[:-1]])]).astype(np.int64)
    out = []
    for w, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
        out.append(WordTiming(words[w], list(word_tokens[w]), token_start[lo], token_start[hi],
                              np.mean(text_token_probs[lo:hi])))
    return out


def find_alignment(session, tokenizer, text_tokens: Sequence[int], mel, num_frames: int, *, medfilt_width: int = 7,
                   qk_scale: float = 1.0) -> List[WordTiming]:
    """whisper/timing.py:163-243 with the model replaced by a HIP session (engine.HipSession, beam 1): the decoder pass,
    token probabilities, attention normalisation and the DTW recurrence run on the GPU (wlk_find_alignment); the path
    is walked back and cut into words here.  ``mel`` is the [n_mels, 3000] segment the reference passes (None: the
    session is already encoded, e.g. from its own audio).  ``tokenizer`` needs ``sot_sequence``, ``no_timestamps``,
    ``eot`` and ``split_to_word_tokens`` (the reference's Tokenizer or this package's)."""
    if medfilt_width != 7:
        raise ValueError("find_alignment: the device kernel implements the reference's default median width 7")
    text_tokens = [int(t) for t in text_tokens]
    if len(text_tokens) == 0:
        return []
    if mel is not None:
        if hasattr(mel, "detach"):
            mel = mel.detach().cpu().numpy()
        session.encode_mel(np.asarray(mel))
    sot = [int(t) for t in tokenizer.sot_sequence]
    tokens = [*sot, int(tokenizer.no_timestamps), *text_tokens, int(tokenizer.eot)]
    trace, probs, _ = session.find_alignment(tokens, len(sot), int(tokenizer.eot), int(num_frames), qk_scale)
    text_indices, time_indices = backtrace(trace)
    words, word_tokens = tokenizer.split_to_word_tokens(text_tokens + [int(tokenizer.eot)])
    return word_timings(text_indices, time_indices, words, word_tokens, probs.tolist())


def merge_punctuations(alignment: List[WordTiming], prepended: str = PREPEND_PUNCTUATIONS,
                       appended: str = APPEND_PUNCTUATIONS) -> None:
    """timing.py:245-276, in place: opening punctuation joins the word after it (scanning backwards, so runs chain),
    closing punctuation joins the word before it; emptied entries stay in the list with word == ""."""
    target = len(alignment) - 1
    for i in range(len(alignment) - 2, -1, -1):
        cur, nxt = alignment[i], alignment[target]
        if cur.word.startswith(" ") and cur.word.strip() in prepended:
            nxt.word, nxt.tokens = cur.word + nxt.word, cur.tokens + nxt.tokens
            cur.word, cur.tokens = "", []
        else:
            target = i
    target = 0
    for j in range(1, len(alignment)):
        prv, cur = alignment[target], alignment[j]
        if not prv.word.endswith(" ") and cur.word in appended:
            prv.word, prv.tokens = prv.word + cur.word, prv.tokens + cur.tokens
            cur.word, cur.tokens = "", []
        else:
            target = j


def attach_words(segments: List[dict], alignment: List[WordTiming], text_tokens_per_segment: Sequence[Sequence[int]], *,
                 last_speech_timestamp: float, prepend_punctuations: str = PREPEND_PUNCTUATIONS,
                 append_punctuations: str = APPEND_PUNCTUATIONS) -> None:
    """add_word_timestamps after its find_alignment call (timing.py:295-388): clamp implausibly long words around
    sentence ends, merge punctuation, deal the words out to the segments by token count, reconcile word and segment
    boundaries.  Mutates ``alignment`` and ``segments`` (adds "words", may move "start" / "end")."""
    if not segments:
        return
    durations = np.array([t.end - t.start for t in alignment])
    durations = durations[durations.nonzero()]
    median_duration = min(0.7, float(np.median(durations))) if len(durations) else 0.0
    max_duration = median_duration * 2
    if len(durations):
        for prev, cur in zip(alignment[:-1], alignment[1:]):
            if cur.end - cur.start > max_duration:
                if cur.word in SENTENCE_END_MARKS:
                    cur.end = cur.start + max_duration
                elif prev.word in SENTENCE_END_MARKS:
                    cur.start = cur.end - max_duration
    merge_punctuations(alignment, prepend_punctuations, append_punctuations)

    offset = segments[0]["seek"] * HOP_LENGTH / SAMPLE_RATE      # same operation order as the reference (rounding)
    cursor = 0
    for segment, seg_tokens in zip(segments, text_tokens_per_segment):
        taken, words = 0, []
        while cursor < len(alignment) and taken < len(seg_tokens):
            t = alignment[cursor]
            if t.word:
                words.append(dict(word=t.word, start=round(offset + t.start, 2), end=round(offset + t.end, 2),
                                  probability=t.probability))
            taken += len(t.tokens)
            cursor += 1
        if words:
            first, last = words[0], words[-1]
            # the first (and second) word after a pause must not be longer than twice the median word
            after_pause = first["end"] - last_speech_timestamp > median_duration * 4
            too_long = first["end"] - first["start"] > max_duration or (
                len(words) > 1 and words[1]["end"] - first["start"] > max_duration * 2)
            if after_pause and too_long:
                if len(words) > 1 and words[1]["end"] - words[1]["start"] > max_duration:
                    boundary = max(words[1]["end"] / 2, words[1]["end"] - max_duration)
                    first["end"] = words[1]["start"] = boundary
                first["start"] = max(0, first["end"] - max_duration)
            # prefer the segment-level timestamps where the outer words are too long
            if segment["start"] < first["end"] and segment["start"] - 0.5 > first["start"]:
                first["start"] = max(0, min(first["end"] - median_duration, segment["start"]))
            else:
                segment["start"] = first["start"]
            if segment["end"] > last["start"] and segment["end"] + 0.5 < last["end"]:
                last["end"] = max(last["start"] + median_duration, segment["end"])
            else:
                segment["end"] = last["end"]
            last_speech_timestamp = segment["end"]
        segment["words"] = words

"""LocalAgreement's batch Whisper on the HIP library: ``transcribe()`` = the 30-second window loop with
temperature fallback of ``whisperlivekit/whisper/transcribe.py:21-494``, ``decode()`` = one window through the
``DecodingTask`` of ``whisperlivekit/whisper/decoding.py:502-784`` (prompt / prefix assembly, blank / annotation / timestamp
rules, greedy, sampling and beam search, ranking), ``detect_language()`` = ``decoding.py:19-77``.

What runs where: the log-mel of the whole recording (``wlk_log_mel``), the encoder and cross-K/V of each window
(``wlk_encode_mel``), every decoder step with its KV cache (``wlk_decode`` / ``wlk_kv_reorder``), the no-speech probability
and the device half of the word alignment (``wlk_find_alignment``) are HIP kernels.  The per-step logit rules are a handful
of masked fills and one log-softmax over the vocabulary row the step produced; they run on that row on the host
(single-threaded numpy; the sampling draw takes its random numbers from torch's global generator exactly as the reference's
``Categorical.sample()`` does, so a seeded run consumes the generator as the reference does).  There is no CPU model path: without the library / a GPU the first call raises.

Same keyword names, defaults and result dictionary as the reference; ``fp16`` is accepted and ignored (the path computes
in fp32, as the reference does on its CPU).  Decoding a file name (ffmpeg) is outside the path: ``audio`` is samples.
"""
from __future__ import annotations

import os
import threading
import warnings
import zlib
from dataclasses import dataclass, field, replace
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import timing as T
from .engine import HipSession, HipWhisperModel
from .policy import BeamUpdate
from .tokenizer import WhisperTokenizer, get_tokenizer

SAMPLE_RATE, HOP_LENGTH, CHUNK_LENGTH = 16000, 160, 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE          # 480000
N_FRAMES = N_SAMPLES // HOP_LENGTH              # 3000
FRAMES_PER_SECOND = SAMPLE_RATE // HOP_LENGTH   # 100
NEG_INF = float("-inf")


# ---- options / result (decoding.py:80-127) ----------------------------------------------------------------------------------
@dataclass(frozen=True)
class DecodingOptions:
    task: str = "transcribe"
    language: Optional[str] = None
    temperature: float = 0.0
    sample_len: Optional[int] = None
    best_of: Optional[int] = None
    beam_size: Optional[int] = None
    patience: Optional[float] = None
    length_penalty: Optional[float] = None
    prompt: Optional[Union[str, List[int]]] = None
    prefix: Optional[Union[str, List[int]]] = None
    suppress_tokens: Optional[Union[str, Iterable[int]]] = "-1"
    suppress_blank: bool = True
    without_timestamps: bool = False
    max_initial_timestamp: Optional[float] = 1.0
    fp16: bool = False            # accepted for signature compatibility; the HIP path is fp32


@dataclass(frozen=True)
class DecodingResult:
    audio_features: object = None             # stays on the device (the session's cross-K/V); kept for field parity
    language: str = "en"
    language_probs: Optional[Dict[str, float]] = None
    tokens: List[int] = field(default_factory=list)
    text: str = ""
    avg_logprob: float = float("nan")
    no_speech_prob: float = float("nan")
    temperature: float = float("nan")
    compression_ratio: float = float("nan")


def compression_ratio(text: str) -> float:
    """utils.py:45-47."""
    raw = text.encode("utf-8")
    return len(raw) / len(zlib.compress(raw))


def pad_or_trim(mel: np.ndarray, length: int = N_FRAMES) -> np.ndarray:
    """audio.py:65-88 on the last axis: cut, or fill with ZEROS (not with the mel's silence value)."""
    if mel.shape[-1] > length:
        mel = mel[..., :length]
    if mel.shape[-1] < length:
        mel = np.pad(mel, [(0, 0)] * (mel.ndim - 1) + [(0, length - mel.shape[-1])])
    return np.ascontiguousarray(mel, dtype=np.float32)


def choose(logits: torch.Tensor, temperature: float) -> torch.Tensor:
    """GreedyDecoder's choice (decoding.py:274-277) on the filtered logits [rows, vocabulary]: arg-max at temperature 0,
    else one categorical draw per row from torch's global generator.  Module-level so that the parity tests can follow
    the reference's recorded choices step by step."""
    if temperature == 0:
        return torch.from_numpy(logits.numpy().argmax(axis=-1))
    return _categorical_draw(logits.numpy() / np.float32(temperature))


def _categorical_draw(scaled: np.ndarray) -> torch.Tensor:
    """``Categorical(logits=scaled).sample()`` with the same consumption of torch's global generator and the same result:
    a single draw per row goes through ``torch.multinomial``'s one-sample path, argmax(p / q) with q ~ Exp(1) drawn per
    vocabulary entry.  Only the exponentials come from torch (that is the generator's part); the softmax, the division and
    the arg-max are single-threaded numpy - torch hands each of these 50k-element rows to its whole intra-op pool (34 ms
    per draw on a 128-thread host).  tests/test_transcribe.py::test_categorical_draw_is_torchs pins the equivalence."""
    m = scaled.max(axis=-1, keepdims=True)
    e = np.exp(scaled - m)
    p = e / e.sum(axis=-1, keepdims=True, dtype=np.float32)
    q = torch.empty(p.shape, dtype=torch.float32).exponential_(1)
    return torch.from_numpy((p / q.numpy()).argmax(axis=-1))


# ---- the model-side handle --------------------------------------------------------------------------------------------------
class _Rows:
    """Sessions of one HipWhisperModel for the batch path, by row count (greedy / sampling: 1 row; best_of / beam: n).
    One set per calling thread: the reference's `transcribe` is re-entrant on a shared model (LocalAgreement serves every
    connection from one WhisperASR, audio_processor.py runs them in worker threads), so concurrent calls must not share
    device state."""

    def __init__(self, model: HipWhisperModel):
        self.model = model
        self.sessions: Dict[int, HipSession] = {}

    def get(self, rows: int) -> HipSession:
        s = self.sessions.get(rows)
        if s is None:
            # its own stream, no batch engine: the audio ring is not used (windows arrive as mel segments)
            s = self.sessions[rows] = self.model.new_session(beam=rows, max_audio_seconds=1.0, batched=False)
        return s


_ROWS_LOCK = threading.Lock()


def _rows_of(model: HipWhisperModel) -> _Rows:
    key = threading.get_ident()
    stale = []
    with _ROWS_LOCK:
        per_thread = model.__dict__.setdefault("_batch_rows", {})
        r = per_thread.get(key)
        if r is None:
            alive = {t.ident for t in threading.enumerate()}
            stale = [per_thread.pop(k) for k in list(per_thread) if k not in alive]     # threads that have ended
            r = per_thread[key] = _Rows(model)
    for rows in stale:
        for sess in rows.sessions.values():
            sess.close()
    return r


def release_sessions(model: HipWhisperModel) -> None:
    """Close the batch-path sessions every thread created on `model` (call before closing the model)."""
    with _ROWS_LOCK:
        per_thread = model.__dict__.pop("_batch_rows", {})
    for rows in per_thread.values():
        for sess in rows.sessions.values():
            sess.close()


def _tokenizer_for(model: HipWhisperModel, language: Optional[str], task: Optional[str]) -> WhisperTokenizer:
    return get_tokenizer(model.is_multilingual, num_languages=model.num_languages, language=language, task=task)


def _logits(session: HipSession, rows: int, n_vocab: int, what: str = "logits_last") -> np.ndarray:
    session.sync()
    return session.export(what, rows * n_vocab).reshape(rows, n_vocab)


def _logsumexp(x: np.ndarray) -> np.float32:
    m = x.max()
    if not np.isfinite(m):
        return np.float32(m)
    return np.float32(m + np.log(np.exp(x - m).sum(dtype=np.float32)))


def _log_softmax(x: np.ndarray) -> np.ndarray:
    """F.log_softmax(logits.float(), dim=-1) row by row, single-threaded."""
    m = x.max(axis=-1, keepdims=True)
    z = x - m
    return z - np.log(np.exp(z).sum(axis=-1, keepdims=True, dtype=np.float32))


# ---- language id (decoding.py:19-77) ----------------------------------------------------------------------------------------
def detect_language(model: HipWhisperModel, mel: Optional[np.ndarray], tokenizer: Optional[WhisperTokenizer] = None,
                    *, session: Optional[HipSession] = None) -> Tuple[int, Dict[str, float]]:
    """-> (most probable language token, {code: probability}).  ``mel`` None: the session is already encoded."""
    if tokenizer is None:
        tokenizer = _tokenizer_for(model, None, None)
    if tokenizer.language is None or (tokenizer.sot + 1 + tokenizer.all_language_codes.index(tokenizer.language)
                                      not in tokenizer.sot_sequence):
        raise ValueError("This model doesn't have language tokens so it can't perform lang id")
    s = session or _rows_of(model).get(1)
    if mel is not None:
        s.encode_mel(pad_or_trim(np.asarray(mel)))
    rows = s.beam
    s.decode(np.full((rows, 1), tokenizer.sot, np.int64), first=True, sot_index=0)
    logits = torch.from_numpy(_logits(s, rows, model.dims.n_vocab)[0])
    keep = torch.zeros(logits.shape[-1], dtype=torch.bool)
    keep[list(tokenizer.all_language_tokens)] = True
    logits[~keep] = NEG_INF
    token = int(logits.argmax())
    probs = logits.softmax(dim=-1)
    return token, {c: float(probs[t]) for t, c in zip(tokenizer.all_language_tokens, tokenizer.all_language_codes)}


# ---- one 30 s window (decoding.py:502-784) ----------------------------------------------------------------------------------
class _WindowDecoder:
    def __init__(self, model: HipWhisperModel, options: DecodingOptions):
        self.model = model
        self.o = options
        self._check(options)
        self.tok = _tokenizer_for(model, options.language or "en", options.task)
        d = model.dims
        self.n_group = options.beam_size or options.best_of or 1
        self.n_ctx = d.n_text_ctx
        self.sample_len = options.sample_len or d.n_text_ctx // 2
        self.sot_sequence = (self.tok.sot_sequence_including_notimestamps if options.without_timestamps
                             else self.tok.sot_sequence)
        self.initial = self._initial_tokens()
        self.sample_begin = len(self.initial)
        self.sot_index = self.initial.index(self.tok.sot)
        self.blank_ids = [*self.tok.encode(" "), self.tok.eot] if options.suppress_blank else None
        self.suppressed = self._suppressed() if options.suppress_tokens else None
        self.max_initial_ts = None
        if not options.without_timestamps and options.max_initial_timestamp:
            self.max_initial_ts = round(options.max_initial_timestamp / (CHUNK_LENGTH / d.n_audio_ctx))

    @staticmethod
    def _check(o: DecodingOptions) -> None:
        if o.beam_size is not None and o.best_of is not None:
            raise ValueError("beam_size and best_of can't be given together")
        if o.temperature == 0 and o.best_of is not None:
            raise ValueError("best_of with greedy sampling (T=0) is not compatible")
        if o.patience is not None and o.beam_size is None:
            raise ValueError("patience requires beam_size to be given")
        if o.length_penalty is not None and not 0 <= o.length_penalty <= 1:
            raise ValueError("length_penalty (alpha) should be a value between 0 and 1")

    def _as_ids(self, text_or_ids) -> List[int]:
        return self.tok.encode(" " + text_or_ids.strip()) if isinstance(text_or_ids, str) else list(text_or_ids)

    def _initial_tokens(self) -> List[int]:
        """[<|startofprev|>, last n_ctx/2 - 1 prompt ids] + sot sequence + prefix (decoding.py:581-607)."""
        ids = list(self.sot_sequence)
        if self.o.prefix:
            prefix = self._as_ids(self.o.prefix)
            if self.sample_len is not None:
                room = self.n_ctx // 2 - self.sample_len
                prefix = prefix[-room:]          # room == 0 keeps everything, as in the reference
            ids = ids + prefix
        if self.o.prompt:
            prompt = self._as_ids(self.o.prompt)
            ids = [self.tok.sot_prev] + prompt[-(self.n_ctx // 2 - 1):] + ids
        return ids

    def _suppressed(self) -> List[int]:
        """decoding.py:609-636: "-1" = the tokenizer's annotation symbols; always the task / sot family and <|nospeech|>."""
        sup = self.o.suppress_tokens
        if isinstance(sup, str):
            sup = [int(t) for t in sup.split(",")]
        sup = list(sup)
        if -1 in sup:
            sup = [t for t in sup if t >= 0] + list(self.tok.non_speech_tokens)
        t = self.tok
        sup += [t.transcribe, t.translate, t.sot, t.sot_prev, t.sot_lm]
        if t.no_speech is not None:
            sup.append(t.no_speech)
        return sorted(set(sup))

    # -- logit rules (decoding.py:417-499), in the reference's order ------------------------------------------------------
    def _apply_rules(self, logits: np.ndarray, tokens: np.ndarray) -> np.ndarray:
        """In place on the [rows, vocabulary] logits; -> log_softmax of the result.  numpy on purpose: these are
        reductions over one 50k-element row per step, and torch's CPU operators hand such a row to the whole intra-op
        thread pool (3 ms per call on a 128-thread host against 0.1 ms single-threaded)."""
        tb, eot = self.tok.timestamp_begin, self.tok.eot
        first_step = tokens.shape[1] == self.sample_begin
        if self.blank_ids is not None and first_step:
            logits[:, self.blank_ids] = NEG_INF
        if self.suppressed is not None:
            logits[:, self.suppressed] = NEG_INF
        if self.o.without_timestamps:
            return _log_softmax(logits)
        if self.tok.no_timestamps is not None:
            logits[:, self.tok.no_timestamps] = NEG_INF
        for k in range(tokens.shape[0]):
            sampled = tokens[k, self.sample_begin:]
            last_ts = len(sampled) >= 1 and sampled[-1] >= tb
            before_last_ts = len(sampled) < 2 or sampled[-2] >= tb
            if last_ts:
                if before_last_ts:
                    logits[k, tb:] = NEG_INF            # a closed pair: text (or <|endoftext|>) must follow
                else:
                    logits[k, :eot] = NEG_INF           # an opening timestamp needs its partner (or the end)
            stamps = sampled[sampled >= tb]
            if len(stamps) > 0:
                # never backwards; and strictly forwards unless this closes a segment (no zero-length loops)
                bound = int(stamps[-1]) if (last_ts and not before_last_ts) else int(stamps[-1]) + 1
                logits[k, tb:bound] = NEG_INF
        if first_step:
            logits[:, :tb] = NEG_INF                    # a window opens with a timestamp ...
            if self.max_initial_ts is not None:
                logits[:, tb + self.max_initial_ts + 1:] = NEG_INF      # ... no later than max_initial_timestamp
        logprobs = _log_softmax(logits)
        changed = False
        for k in range(tokens.shape[0]):
            if _logsumexp(logprobs[k, tb:]) > logprobs[k, :tb].max():
                logits[k, :tb] = NEG_INF                # timestamps as a group outweigh every text token
                changed = True
        return _log_softmax(logits) if changed else logprobs

    def _pick_state(self, tokens: np.ndarray) -> dict:
        """What ApplyTimestampRules (decoding.py:441-499) reads from the sampled tokens of the one sequence, as the fields of
        wlk_pick_params: the device applies the rules to the logits row, the history stays here."""
        tb = self.tok.timestamp_begin
        sampled = tokens[0, self.sample_begin:]
        last_ts = len(sampled) >= 1 and sampled[-1] >= tb
        before_last_ts = len(sampled) < 2 or sampled[-2] >= tb
        stamps = sampled[sampled >= tb]
        bound = tb
        if len(stamps) > 0:
            bound = int(stamps[-1]) if (last_ts and not before_last_ts) else int(stamps[-1]) + 1
        return dict(first_step=tokens.shape[1] == self.sample_begin, without_timestamps=bool(self.o.without_timestamps),
                    timestamp_begin=tb, eot=self.tok.eot,
                    no_timestamps=-1 if self.tok.no_timestamps is None else int(self.tok.no_timestamps),
                    ts_mode=0 if not last_ts else (1 if before_last_ts else 2), ts_bound=bound,
                    max_initial=-1 if self.max_initial_ts is None else int(self.max_initial_ts))

    # -- the loop ------------------------------------------------------------------------------------------------------
    def run(self, mel_segment: Optional[np.ndarray], session: Optional[HipSession] = None) -> DecodingResult:
        o, tok, V = self.o, self.tok, self.model.dims.n_vocab
        rows = self.n_group
        s = session or _rows_of(self.model).get(rows)
        if mel_segment is not None:
            s.encode_mel(pad_or_trim(np.asarray(mel_segment)))
        initial = list(self.initial)
        language, language_probs = o.language, None
        if o.language is None or o.task == "lang_id":
            lang_token, language_probs = detect_language(self.model, None, tok, session=s)
            language = max(language_probs, key=language_probs.get)
            if o.language is None:
                initial[self.sot_index + 1] = lang_token
            if o.task == "lang_id":
                return DecodingResult(language=language, language_probs=language_probs)

        tokens = np.tile(np.asarray(initial, np.int64), (rows, 1))
        sum_logprobs = np.zeros(rows, np.float32)
        no_speech = float("nan")
        beam = BeamUpdate(o.beam_size, tok.eot, o.patience or 1.0) if o.beam_size is not None else None
        if beam is not None and beam.max_candidates <= 0:
            raise ValueError(f"Invalid beam size ({o.beam_size}) or patience ({o.patience})")
        # greedy decoding of one sequence at temperature 0: rules, argmax and log-probability on the device, 8 bytes back per
        # step (WLK_TRANSCRIBE_DEVICE_RULES=0: the host path below, which sampling and beam search always take)
        device_rules = (beam is None and rows == 1 and o.temperature == 0 and hasattr(s, "pick_greedy")
                        and os.environ.get("WLK_TRANSCRIBE_DEVICE_RULES", "1") != "0")
        if device_rules:
            s.set_rules(self.suppressed or [], self.blank_ids or [])
        for i in range(self.sample_len):
            s.decode(tokens if i == 0 else tokens[:, -1:], first=(i == 0), sot_index=self.sot_index)
            if i == 0 and tok.no_speech is not None:
                no_speech = float(s.no_speech_prob(tok.no_speech)[0])
            if device_rules:
                nxt_id, picked = s.pick_greedy(**self._pick_state(tokens))
                if tokens[0, -1] != tok.eot:
                    sum_logprobs[0] += np.float32(picked)
                else:
                    nxt_id = tok.eot
                tokens = np.concatenate([tokens, np.asarray([[nxt_id]], np.int64)], axis=1)
                if nxt_id == tok.eot or tokens.shape[-1] > self.n_ctx:
                    break
                continue
            logits = _logits(s, rows, V)
            logprobs = self._apply_rules(logits, tokens)
            if beam is not None:
                top_lp, top_id = torch.from_numpy(logprobs).topk(o.beam_size + 1, dim=-1)
                tokens, done, sources = beam.update(tokens, top_lp.numpy(), top_id.numpy(), sum_logprobs)
                s.kv_reorder(sources)
            else:
                nxt = choose(torch.from_numpy(logits), o.temperature)
                picked = logprobs[np.arange(rows), nxt.numpy()]
                live = tokens[:, -1] != tok.eot
                sum_logprobs += picked * live
                nxt = np.where(live, nxt.numpy(), tok.eot)
                tokens = np.concatenate([tokens, nxt[:, None].astype(np.int64)], axis=1)
                done = bool((tokens[:, -1] == tok.eot).all())
            if done or tokens.shape[-1] > self.n_ctx:
                break

        # candidates of the group (decoding.py:289-292 / :378-399), cut between the first sampled token and <|endoftext|>
        if beam is not None:
            finished = beam.finished[0]
            if len(finished) < o.beam_size:
                for j in np.argsort(sum_logprobs)[::-1]:
                    finished[tuple(tokens[j].tolist() + [tok.eot])] = float(sum_logprobs[j])
                    if len(finished) >= o.beam_size:
                        break
            candidates = [list(seq) for seq in finished]
            sums = [float(v) for v in finished.values()]
        else:
            candidates = [row.tolist() + [tok.eot] for row in tokens]
            sums = [float(v) for v in sum_logprobs]
        candidates = [c[self.sample_begin:c.index(tok.eot)] for c in candidates]
        best = self._rank(candidates, sums)
        out = candidates[best]
        text = tok.decode(out).strip()
        return DecodingResult(language=language, language_probs=language_probs, tokens=out, text=text,
                              avg_logprob=sums[best] / (len(out) + 1), no_speech_prob=no_speech,
                              temperature=o.temperature, compression_ratio=compression_ratio(text))

    def _rank(self, candidates: List[List[int]], sums: List[float]) -> int:
        """MaximumLikelihoodRanker (decoding.py:184-207): log-probability over length, or over ((5 + length) / 6) ** alpha."""
        def norm(n):
            return n if self.o.length_penalty is None else ((5 + n) / 6) ** self.o.length_penalty
        return int(np.argmax([lp / norm(len(c)) for c, lp in zip(candidates, sums)]))


def decode(model: HipWhisperModel, mel: Optional[np.ndarray], options: DecodingOptions = DecodingOptions(), *,
           session: Optional[HipSession] = None, **kwargs) -> DecodingResult:
    """decoding.py:787-820 for one 30 s segment [n_mels, 3000] (None: the session already holds the encoded window)."""
    if kwargs:
        options = replace(options, **kwargs)
    return _WindowDecoder(model, options).run(mel, session)


# ---- the window loop (transcribe.py:21-494) ---------------------------------------------------------------------------------
_PUNCTUATION = "\"'“¿([{-\"'.。,，!！?？:：”)]}、"


def _word_anomaly(word: dict) -> float:
    """transcribe.py:287-297: improbable, very short or very long words score."""
    duration = word["end"] - word["start"]
    score = 1.0 if word.get("probability", 0.0) < 0.15 else 0.0
    if duration < 0.133:
        score += (0.133 - duration) * 15
    if duration > 2.0:
        score += duration - 2.0
    return score


def _segment_is_anomaly(segment: Optional[dict]) -> bool:
    if segment is None or not segment["words"]:
        return False
    words = [w for w in segment["words"] if w["word"] not in _PUNCTUATION][:8]
    score = sum(_word_anomaly(w) for w in words)
    return score >= 3 or score + 0.01 >= len(words)


def _first_with_words(segments: Sequence[dict]) -> Optional[dict]:
    return next((s for s in segments if s["words"]), None)


def _last_word_end(segments: Sequence[dict]) -> Optional[float]:
    """utils.py:78-82."""
    for seg in reversed(segments):
        for w in reversed(seg["words"]):
            return w["end"]
    return segments[-1]["end"] if segments else None


def transcribe(model: HipWhisperModel, audio, *, verbose: Optional[bool] = None,
               temperature: Union[float, Tuple[float, ...]] = (0.0, 0.2, 0.4, 0.6, 0.8, 1.0),
               compression_ratio_threshold: Optional[float] = 2.4, logprob_threshold: Optional[float] = -1.0,
               no_speech_threshold: Optional[float] = 0.6, condition_on_previous_text: bool = True,
               initial_prompt: Optional[str] = None, carry_initial_prompt: bool = False, word_timestamps: bool = False,
               prepend_punctuations: str = T.PREPEND_PUNCTUATIONS, append_punctuations: str = T.APPEND_PUNCTUATIONS,
               clip_timestamps: Union[str, List[float]] = "0", hallucination_silence_threshold: Optional[float] = None,
               **decode_options) -> dict:
    """-> {"text", "segments": [{"id", "seek", "start", "end", "text", "tokens", "temperature", "avg_logprob",
    "compression_ratio", "no_speech_prob", ("words")}], "language"}."""
    if isinstance(audio, str):
        raise TypeError("transcribe: pass samples (float32, 16 kHz); decoding a file is outside the accelerated path")
    if hasattr(audio, "detach"):
        audio = audio.detach().cpu().numpy()
    audio = np.ascontiguousarray(audio, dtype=np.float32).reshape(-1)
    decode_options = dict(decode_options)
    decode_options["fp16"] = False
    dims = model.dims
    session = _rows_of(model).get(1)

    mel = session.log_mel(audio, padding=N_SAMPLES)            # 30 s of silence appended, for slicing
    content_frames = mel.shape[-1] - N_FRAMES
    content_duration = float(content_frames * HOP_LENGTH / SAMPLE_RATE)

    if decode_options.get("language") is None:
        if not model.is_multilingual:
            decode_options["language"] = "en"
        else:
            if verbose:
                print("Detecting language using up to the first 30 seconds. Use `--language` to specify the language")
            _, probs = detect_language(model, pad_or_trim(mel), session=session)
            decode_options["language"] = max(probs, key=probs.get)
            if verbose is not None:
                print(f"Detected language: {decode_options['language']}")
    language: str = decode_options["language"]
    task: str = decode_options.get("task", "transcribe")
    tokenizer = _tokenizer_for(model, language, task)

    if isinstance(clip_timestamps, str):
        clip_timestamps = [float(ts) for ts in (clip_timestamps.split(",") if clip_timestamps else [])]
    points = [round(ts * FRAMES_PER_SECOND) for ts in clip_timestamps]
    if not points:
        points.append(0)
    if len(points) % 2 == 1:
        points.append(content_frames)
    clips = list(zip(points[::2], points[1::2]))

    if word_timestamps and task == "translate":
        warnings.warn("Word-level timestamps on translations may not be reliable.")

    temperatures = [temperature] if isinstance(temperature, (int, float)) else list(temperature)

    def decode_with_fallback(segment: np.ndarray) -> DecodingResult:
        result = None
        for n, t in enumerate(temperatures):
            kw = dict(decode_options)
            if t > 0:
                kw.pop("beam_size", None)         # sampling: no beam
                kw.pop("patience", None)
            else:
                kw.pop("best_of", None)           # greedy / beam: no best_of
            opts = DecodingOptions(**kw, temperature=t)
            rows = opts.beam_size or opts.best_of or 1
            sess = _rows_of(model).get(rows)
            # the encoder output of this window stays in the session across the temperatures
            result = _WindowDecoder(model, opts).run(segment if (n == 0 or sess is not encoded_in[0]) else None, sess)
            encoded_in[0] = sess
            retry = False
            if compression_ratio_threshold is not None and result.compression_ratio > compression_ratio_threshold:
                retry = True                      # too repetitive
            if logprob_threshold is not None and result.avg_logprob < logprob_threshold:
                retry = True                      # too improbable
            if (no_speech_threshold is not None and result.no_speech_prob > no_speech_threshold
                    and logprob_threshold is not None and result.avg_logprob < logprob_threshold):
                retry = False                     # silence: improbable text is expected
            if not retry:
                break
        return result

    encoded_in: List[Optional[HipSession]] = [None]
    input_stride = N_FRAMES // dims.n_audio_ctx                      # mel frames per encoder position: 2
    assert N_FRAMES % dims.n_audio_ctx == 0
    time_precision = input_stride * HOP_LENGTH / SAMPLE_RATE         # 0.02 s
    all_tokens: List[int] = []
    all_segments: List[dict] = []
    prompt_reset_since = 0
    remaining_prompt_length = dims.n_text_ctx // 2 - 1
    if initial_prompt is not None:
        initial_prompt_tokens = tokenizer.encode(" " + initial_prompt.strip())
        all_tokens.extend(initial_prompt_tokens)
        remaining_prompt_length -= len(initial_prompt_tokens)
    else:
        initial_prompt_tokens = []

    ts_begin = tokenizer.timestamp_begin
    last_speech_timestamp = 0.0
    clip_idx = 0
    seek = clips[0][0]
    while clip_idx < len(clips):
        clip_start, clip_end = clips[clip_idx]
        if seek < clip_start:
            seek = clip_start
        if seek >= clip_end:
            clip_idx += 1
            if clip_idx < len(clips):
                seek = clips[clip_idx][0]
            continue
        time_offset = float(seek * HOP_LENGTH / SAMPLE_RATE)
        window_end_time = float((seek + N_FRAMES) * HOP_LENGTH / SAMPLE_RATE)
        segment_size = min(N_FRAMES, content_frames - seek, clip_end - seek)
        segment_duration = segment_size * HOP_LENGTH / SAMPLE_RATE
        mel_segment = pad_or_trim(mel[:, seek:seek + segment_size])

        if carry_initial_prompt:
            ignored = max(len(initial_prompt_tokens), prompt_reset_since)
            decode_options["prompt"] = initial_prompt_tokens + all_tokens[ignored:][-remaining_prompt_length:]
        else:
            decode_options["prompt"] = all_tokens[prompt_reset_since:]

        encoded_in[0] = None
        result = decode_with_fallback(mel_segment)
        tokens = np.asarray(result.tokens, dtype=np.int64)

        if no_speech_threshold is not None:
            skip = result.no_speech_prob > no_speech_threshold
            if logprob_threshold is not None and result.avg_logprob > logprob_threshold:
                skip = False                      # confident text despite the no-speech probability
            if skip:
                seek += segment_size
                continue

        previous_seek = seek
        current: List[dict] = []

        def new_segment(start: float, end: float, ids: np.ndarray) -> dict:
            ids = [int(t) for t in ids]
            return {"seek": seek, "start": start, "end": end,
                    "text": tokenizer.decode([t for t in ids if t < tokenizer.eot]), "tokens": ids,
                    "temperature": result.temperature, "avg_logprob": result.avg_logprob,
                    "compression_ratio": result.compression_ratio, "no_speech_prob": result.no_speech_prob}

        is_ts = tokens >= ts_begin
        single_timestamp_ending = is_ts[-2:].tolist() == [False, True]
        pair_ends = (np.where(is_ts[:-1] & is_ts[1:])[0] + 1).tolist()
        if pair_ends:
            # consecutive timestamp tokens close one segment and open the next
            if single_timestamp_ending:
                pair_ends.append(len(tokens))
            lo = 0
            for hi in pair_ends:
                piece = tokens[lo:hi]
                current.append(new_segment(time_offset + (int(piece[0]) - ts_begin) * time_precision,
                                           time_offset + (int(piece[-1]) - ts_begin) * time_precision, piece))
                lo = hi
            if single_timestamp_ending:
                seek += segment_size              # nothing is spoken after the last timestamp
            else:
                seek += (int(tokens[lo - 1]) - ts_begin) * input_stride     # redo the unfinished tail
        else:
            duration = segment_duration
            stamps = tokens[is_ts]
            if len(stamps) > 0 and int(stamps[-1]) != ts_begin:
                duration = (int(stamps[-1]) - ts_begin) * time_precision
            current.append(new_segment(time_offset, time_offset + duration, tokens))
            seek += segment_size

        if word_timestamps:
            _add_word_timestamps(current, session if encoded_in[0] is session else None, model, tokenizer, mel_segment,
                                 segment_size, prepend_punctuations, append_punctuations, last_speech_timestamp)
            if not single_timestamp_ending:
                end = _last_word_end(current)
                if end is not None and end > time_offset:
                    seek = round(end * FRAMES_PER_SECOND)

            if hallucination_silence_threshold is not None:
                threshold = hallucination_silence_threshold
                if not single_timestamp_ending:
                    end = _last_word_end(current)
                    if end is not None and end > time_offset:
                        seek = (round(end * FRAMES_PER_SECOND) if window_end_time - end > threshold
                                else previous_seek + segment_size)
                # a doubtful first segment: skip the silence before it and decode again from there
                first = _first_with_words(current)
                if first is not None and _segment_is_anomaly(first):
                    gap = first["start"] - time_offset
                    if gap > threshold:
                        seek = previous_seek + round(gap * FRAMES_PER_SECOND)
                        continue
                # a doubtful segment with silence (or more doubt) on both sides: drop it and what follows
                hal_last_end = last_speech_timestamp
                for si, segment in enumerate(current):
                    if not segment["words"]:
                        continue
                    if _segment_is_anomaly(segment):
                        following = _first_with_words(current[si + 1:])
                        hal_next_start = (following["words"][0]["start"] if following is not None
                                          else time_offset + segment_duration)
                        silence_before = (segment["start"] - hal_last_end > threshold or segment["start"] < threshold
                                          or segment["start"] - time_offset < 2.0)
                        silence_after = (hal_next_start - segment["end"] > threshold or _segment_is_anomaly(following)
                                         or window_end_time - segment["end"] < 2.0)
                        if silence_before and silence_after:
                            seek = round(max(time_offset + 1, segment["start"]) * FRAMES_PER_SECOND)
                            if content_duration - segment["end"] < threshold:
                                seek = content_frames
                            current[si:] = []
                            break
                    hal_last_end = segment["end"]

            end = _last_word_end(current)
            if end is not None:
                last_speech_timestamp = end

        if verbose:
            for segment in current:
                print(f"[{_stamp(segment['start'])} --> {_stamp(segment['end'])}] {segment['text']}")

        for segment in current:                   # instantaneous or empty segments keep their slot, without content
            if segment["start"] == segment["end"] or segment["text"].strip() == "":
                segment["text"] = ""
                segment["tokens"] = []
                segment["words"] = []

        all_segments.extend({"id": i, **segment} for i, segment in enumerate(current, start=len(all_segments)))
        all_tokens.extend(t for segment in current for t in segment["tokens"])
        if not condition_on_previous_text or result.temperature > 0.5:
            prompt_reset_since = len(all_tokens)  # text sampled at a high temperature is not fed back

    return dict(text=tokenizer.decode(all_tokens[len(initial_prompt_tokens):]), segments=all_segments, language=language)


def _stamp(seconds: float) -> str:
    ms = round(seconds * 1000.0)
    h, ms = divmod(ms, 3_600_000)
    m, ms = divmod(ms, 60_000)
    s, ms = divmod(ms, 1_000)
    return (f"{h:02d}:" if h else "") + f"{m:02d}:{s:02d}.{ms:03d}"


def _add_word_timestamps(segments: List[dict], encoded: Optional[HipSession], model: HipWhisperModel,
                         tokenizer: WhisperTokenizer, mel_segment: np.ndarray, num_frames: int, prepend: str, append: str,
                         last_speech_timestamp: float) -> None:
    """timing.py:279-388: one alignment pass over the text of all segments of the window, dealt back to them."""
    if not segments:
        return
    per_segment = [[t for t in seg["tokens"] if t < tokenizer.eot] for seg in segments]
    text_tokens = [t for ids in per_segment for t in ids]
    if encoded is None:                           # the window was decoded in a several-row session: encode it for the pass
        encoded = _rows_of(model).get(1)
        alignment = T.find_alignment(encoded, tokenizer, text_tokens, mel_segment, num_frames)
    else:                                         # the 1-row session still holds this window's encoder output
        alignment = T.find_alignment(encoded, tokenizer, text_tokens, None, num_frames)
    T.attach_words(segments, alignment, per_segment, last_speech_timestamp=last_speech_timestamp,
                   prepend_punctuations=prepend, append_punctuations=append)

"""Stand-alone AlignAtt streaming policy and its data types.

When WhisperLiveKit itself is importable the HIP backend plugs in UNDER the reference's own
``AlignAttBase`` (whisperlivekit/simul_whisper/align_att_base.py) and none of this file is used:
the policy stays the reference's code, untouched.  On machines without the reference package
(the GPU box, the benchmark, the parity tests) this module provides the same template - the
decode loop with its stop rules, context trimming, word splitting and timestamping - written
against the same hook names, so the one hook implementation in ``align_att.py`` serves both.

Behaviour follows align_att_base.py line by line in *what* it does (cited below); it is pinned
against outputs of the reference by tests/test_policy_golden.py.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import tokenizer as wtok

logger = logging.getLogger(__name__)

DEC_PAD = 50257            # align_att_base.py:9
REPLACEMENT = "�"
MIN_WORD_DURATION = 0.02   # align_att_base.py:388-389
FALLBACK_WORD_DURATION = 0.10


@dataclass
class ASRToken:
    """Output type of the path (whisperlivekit/timed_objects.py:21-55)."""
    start: Optional[float] = 0
    end: Optional[float] = 0
    text: Optional[str] = ""
    speaker: Optional[int] = -1
    detected_language: Optional[str] = None
    probability: Optional[float] = None

    def with_offset(self, offset: float) -> "ASRToken":
        return ASRToken(self.start + offset, self.end + offset, self.text, self.speaker,
                        detected_language=self.detected_language, probability=self.probability)

    def is_silence(self) -> bool:
        return False

    def __bool__(self) -> bool:
        return bool(self.text)


@dataclass
class ChangeSpeaker:       # timed_objects.py:227-229
    speaker: int
    start: float


@dataclass
class AlignAttConfig:
    """simul_whisper/config.py:5-23 with the values the engine passes (backend.py:369-384)."""
    eval_data_path: str = "tmp"
    segment_length: float = 1.0
    frame_threshold: int = 4
    rewind_threshold: int = 200
    audio_max_len: float = 20.0
    cif_ckpt_path: Optional[str] = ""
    never_fire: bool = False
    language: str = "zh"
    nonspeech_prob: float = 0.5
    audio_min_len: float = 1.0
    decoder_type: str = "greedy"
    beam_size: int = 5
    task: str = "transcribe"
    tokenizer_is_multilingual: bool = False
    init_prompt: Optional[str] = None
    static_init_prompt: Optional[str] = None
    max_context_tokens: Optional[int] = None


class TextContext:
    """Prompt context kept as TEXT and re-encoded on use (simul_whisper/token_buffer.py:5-95)."""

    def __init__(self, text: str = "", tokenizer=None, prefix_token_ids: Sequence[int] = ()):
        self.text = text
        self.tokenizer = tokenizer
        self.prefix_token_ids = list(prefix_token_ids)
        self.pending_token_ids: List[int] = []

    def is_empty(self) -> bool:
        return self.text is None or self.text == ""

    def as_text(self) -> str:
        return self.text

    def as_token_ids(self) -> List[int]:
        return self.prefix_token_ids + self.tokenizer.encode(self.text)

    def trim_words(self, num: int = 1, after: int = 0) -> int:
        ids = self.tokenizer.encode(self.text[after:])
        words, groups = self.tokenizer.split_to_word_tokens(ids)
        if not words:
            return 0
        self.text = self.text[:after] + "".join(words[num:])
        return sum(len(g) for g in groups[:num])

    def append_token_ids(self, token_ids: Sequence[int]) -> None:
        everything = self.pending_token_ids + list(token_ids)
        text = self.tokenizer.decode(everything)
        if REPLACEMENT not in text:
            self.text += text
            self.pending_token_ids = []
            return
        if len(everything) > 1:
            head = self.tokenizer.decode(everything[:-1])
            if REPLACEMENT not in head:
                self.text += head
                self.pending_token_ids = [everything[-1]]
                return
        self.pending_token_ids = everything


@dataclass
class StreamState:
    """Host-side half of the reference's DecoderState (simul_whisper/decoder_state.py:7-91); the
    tensor-valued half (KV caches, audio, cross-attention window) lives in the HIP session."""
    tokenizer: Any = None
    detected_language: Optional[str] = None
    tokens: List[np.ndarray] = field(default_factory=list)
    initial_tokens: Optional[np.ndarray] = None
    initial_token_length: int = 0
    sot_index: int = 0
    align_source: Dict[int, List[Tuple[int, int]]] = field(default_factory=dict)
    num_align_heads: int = 0
    segments: List[Any] = field(default_factory=list)
    context: Any = None
    pending_incomplete_tokens: List[int] = field(default_factory=list)
    pending_incomplete_token_timestamps: List[float] = field(default_factory=list)
    pending_retries: int = 0
    global_time_offset: float = 0.0
    cumulative_time_offset: float = 0.0
    first_timestamp: Optional[float] = None
    last_attend_frame: int = 0
    speaker: int = -1
    log_segments: int = 0
    always_fire: bool = False
    never_fire: bool = False
    cif_weight: Optional[np.ndarray] = None
    cif_bias: float = 0.0
    decoder_type: str = "greedy"
    suppress_ids: Tuple[int, ...] = ()
    kv_cache: Dict[str, Any] = field(default_factory=dict)   # kept for interface parity; always empty
    on_clean_cache: Any = None

    def clean_cache(self):
        """decoder_state.py:51-59: forget per-infer decoder state (the next decode re-prefills)."""
        self.kv_cache.clear()
        if self.on_clean_cache is not None:
            self.on_clean_cache()


class AlignAttPolicy:
    """The AlignAtt template method over abstract tensor hooks (align_att_base.py:13-322).
    Subclasses provide the hooks listed at align_att_base.py:541-649."""

    # -- properties -------------------------------------------------------------------------
    @property
    def speaker(self):
        return self.state.speaker

    @speaker.setter
    def speaker(self, value):
        self.state.speaker = value

    @property
    def global_time_offset(self):
        return self.state.global_time_offset

    @global_time_offset.setter
    def global_time_offset(self, value):
        self.state.global_time_offset = value

    # -- construction -----------------------------------------------------------------------
    def _base_init(self, cfg, model):
        self.model = model
        self.cfg = cfg
        self.tokenizer_is_multilingual = cfg.tokenizer_is_multilingual
        self.max_text_len = model.dims.n_text_ctx
        self.num_decoder_layers = model.dims.n_text_layer
        self.max_context_tokens = (self.max_text_len if cfg.max_context_tokens is None
                                   else cfg.max_context_tokens)
        self.task = cfg.task

    def _init_state_common(self, cfg):
        self.create_tokenizer(cfg.language if cfg.language != "auto" else None)
        st = self.state
        st.detected_language = cfg.language if cfg.language != "auto" else None
        st.global_time_offset = 0.0
        st.last_attend_frame = -cfg.rewind_threshold
        st.speaker = -1

    def create_tokenizer(self, language=None):
        self.tokenizer = wtok.get_tokenizer(
            multilingual=self.tokenizer_is_multilingual, language=language,
            num_languages=self.model.num_languages, task=self.task)
        self.state.tokenizer = self.tokenizer

    # -- shared helpers -----------------------------------------------------------------------
    def warmup(self, audio):
        """align_att_base.py:75-89: a failing warm-up must abort start-up, not be swallowed."""
        try:
            self.insert_audio(audio)
            self.infer(is_last=True)
            self.refresh_segment(complete=True)
        except Exception as e:
            logger.exception("Model warmup failed: %s", e)
            raise RuntimeError(
                "Model warmup inference failed; refusing to serve since "
                f"sessions would produce empty output. Cause: {e}") from e

    def segments_len(self) -> float:
        return sum(s.shape[0] for s in self.state.segments) / 16000

    def trim_context(self):
        """Drop leading context words until the prompt fits (align_att_base.py:100-113)."""
        ctx = self.state.context
        n_ctx = len(ctx.as_token_ids()) - len(ctx.prefix_token_ids)
        total = sum(t.shape[1] for t in self.state.tokens) + n_ctx
        keep = 0 if self.cfg.static_init_prompt is None else len(self.cfg.static_init_prompt)
        while n_ctx > self.max_context_tokens or total > self.max_text_len - 20:
            dropped = ctx.trim_words(after=keep)
            total -= dropped
            n_ctx -= dropped
            if dropped == 0:
                break

    def refresh_segment(self, complete=False):
        """New segment: fresh tokens/context, keep at most the last two audio chunks
        (align_att_base.py:115-132)."""
        st = self.state
        self.init_tokens()
        st.last_attend_frame = -self.cfg.rewind_threshold
        st.cumulative_time_offset = 0.0
        self.init_context()
        if not complete and len(st.segments) > 2:
            st.segments = st.segments[-2:]
        else:
            st.segments = []
        st.log_segments += 1
        st.pending_incomplete_tokens = []
        st.pending_incomplete_token_timestamps = []
        st.pending_retries = 0

    def _clean_cache(self):
        self.state.clean_cache()

    def _detect_language_if_needed(self, encoder_feature):
        """align_att_base.py:153-170: language id once >= 2 s of speech have been seen."""
        st = self.state
        if self.cfg.language != "auto" or st.detected_language is not None or not st.first_timestamp:
            return
        if self.segments_len() - st.first_timestamp < 2.0:
            return
        _, probs = self.lang_id(encoder_feature)
        best, _p = max(probs[0].items(), key=lambda kv: kv[1])
        self.create_tokenizer(best)
        st.last_attend_frame = -self.cfg.rewind_threshold
        st.cumulative_time_offset = 0.0
        self.init_tokens()
        self.init_context()
        st.detected_language = best

    # -- the AlignAtt call ----------------------------------------------------------------------
    def infer(self, is_last=False):
        st, cfg = self.state, self.cfg
        if not st.segments:
            return []
        if self.segments_len() < cfg.audio_min_len:
            return []

        encoder_feature, content_mel_len = self._encode(self._concat_segments())
        self._evaluate(encoder_feature)
        self._detect_language_if_needed(encoder_feature)
        self.trim_context()
        tokens = self._current_tokens()
        fire = self.fire_at_boundary(encoder_feature[:, :content_mel_len, :])

        n_before = tokens.shape[1]
        budget = max(50, int(self.segments_len() * 15 * 1.5))     # align_att_base.py:200-201
        device_loop = getattr(self, "device_loop_available", None)
        if device_loop is not None and device_loop():
            # the per-token loop below runs inside the library (wlk_decode_until_stop)
            out = self._decode_until_stop(tokens, content_mel_len, is_last, budget)
            stamps = [f * 0.02 + st.cumulative_time_offset for f in out.step_frames]
            st.last_attend_frame = out.last_attend_frame
            new_ids = list(out.new_tokens)
        else:
            tokens, stamps = self._decode_loop(tokens, encoder_feature, content_mel_len, is_last, budget)
            new_ids = self._tokens_to_list(tokens, n_before)
        times = self._normalize_token_timestamps(stamps, len(new_ids))
        if st.pending_incomplete_tokens:
            new_ids, times = self._prepend_pending_tokens(new_ids, times)
        hypothesis, words, groups = self._split_tokens(new_ids, fire, is_last)
        st.tokens.append(self._make_new_tokens_tensor(hypothesis))
        self._clean_cache()
        if len(stamps) >= 2 and st.first_timestamp is None:
            st.first_timestamp = stamps[0]
        out = self._build_timestamped_words(words, groups, times)
        self._handle_pending_tokens(words, groups, times)
        return out

    def _decode_loop(self, tokens, encoder_feature, content_mel_len, is_last, budget):
        """The reference's per-token loop over the tensor hooks (align_att_base.py:206-286); returns the final token
        matrix and one absolute timestamp per decode step."""
        st, cfg = self.state, self.cfg
        sum_logprobs = self._init_sum_logprobs()
        n_before = tokens.shape[1]
        stamps: List[float] = []
        window: List[Any] = []
        produced = 0
        fresh = True
        done = False
        while not done and tokens.shape[1] < self.max_text_len:
            produced += 1
            if produced > budget:                                 # runaway guard, :208-214
                logger.warning("[Loop Detection] Too many tokens (%d); breaking", produced)
                tokens = tokens[:, :n_before]
                break
            fed = tokens if fresh else tokens[:, -1:]
            logits, cross = self._get_logits_and_cross_attn(fed, encoder_feature)
            self._evaluate(logits)
            window.append(cross)
            window = window[-16:]
            if fresh and self._check_no_speech(logits):
                break
            logits = logits[:, -1, :]
            if fresh:
                logits = self._suppress_blank_tokens(logits)
            fresh = False
            logits = self._apply_token_suppression(logits)
            logits = self._apply_dry_penalty(logits, tokens)
            tokens, done = self._update_tokens(tokens, logits, sum_logprobs)
            self._evaluate(tokens)

            attn = self._process_cross_attention(window, content_mel_len)
            frames, frame = self._get_attended_frames(attn)
            stamps.append(frames[0] * 0.02 + st.cumulative_time_offset)

            if done:                                              # :255-257
                tokens = tokens[:, :-1]
                break
            if not is_last and st.last_attend_frame - frame > cfg.rewind_threshold:   # :260-276
                if tokens.shape[1] > 1 and self._is_special_token(tokens):
                    st.last_attend_frame = frame
                else:
                    st.last_attend_frame = -cfg.rewind_threshold
                    tokens = self._rewind_tokens()
                    break
            else:
                st.last_attend_frame = frame
            if content_mel_len - frame <= (4 if is_last else cfg.frame_threshold):    # :280-286
                tokens = tokens[:, :-1]
                break

        return tokens, stamps

    # -- post-decode helpers --------------------------------------------------------------------
    def _split_tokens(self, ids, fire, is_last):
        """Everything is committed at a boundary; otherwise the last word is held back
        (align_att_base.py:326-337)."""
        words, groups = self.tokenizer.split_to_word_tokens(ids)
        if fire or is_last:
            return ids, words, groups
        kept = [t for g in groups[:-1] for t in g] if len(words) > 1 else []
        return kept, words, groups

    @staticmethod
    def _normalize_token_timestamps(timestamps, expected_len):
        out = [float(t) for t in timestamps[:expected_len]]
        if len(out) < expected_len:
            out += [out[-1] if out else 0.0] * (expected_len - len(out))
        return out

    def _prepend_pending_tokens(self, ids, times):
        st = self.state
        held = list(st.pending_incomplete_tokens)
        held_t = list(st.pending_incomplete_token_timestamps)
        if len(held_t) != len(held):
            fill = held_t[-1] if held_t else (times[0] if times else 0.0)
            held_t = held_t[:len(held)] + [fill] * max(0, len(held) - len(held_t))
        return held + ids, held_t + times

    def _make_asr_token(self, **kw):
        return ASRToken(**kw)

    def _build_timestamped_words(self, words, groups, times):
        """Word start = time of its first token, end = time of the next word's first token
        (or +0.10 s for the last word), at least 0.02 s long (align_att_base.py:386-441)."""
        st = self.state
        out = []
        at = 0
        for word, toks in zip(words, groups):
            n = len(toks)
            if REPLACEMENT in word:
                cleaned = word.replace(REPLACEMENT, "")
                if not cleaned.strip():
                    at += n
                    continue
                word = cleaned
            wt = times[at: at + n]
            if not wt:
                wt = [0.0 if not times else (times[at] if at < len(times) else times[-1])]
            start = wt[0]
            nxt = at + n
            end = times[nxt] if nxt < len(times) else wt[-1] + FALLBACK_WORD_DURATION
            end = max(end, start + MIN_WORD_DURATION)
            at = nxt
            out.append(self._make_asr_token(
                start=round(start, 2), end=round(end, 2), text=word, speaker=st.speaker,
                detected_language=st.detected_language).with_offset(st.global_time_offset))
        return out

    def _handle_pending_tokens(self, words, groups, times):
        """A trailing word that ends inside a UTF-8 sequence is carried to the next call, at most
        twice and at most 10 tokens (align_att_base.py:443-488)."""
        st = self.state

        def drop():
            st.pending_incomplete_tokens = []
            st.pending_incomplete_token_timestamps = []
            st.pending_retries = 0

        if not (words and REPLACEMENT in words[-1]):
            drop()
            return
        st.pending_retries += 1
        if st.pending_retries > 2 or len(groups[-1]) > 10:
            drop()
            return
        st.pending_incomplete_tokens = groups[-1]
        first = sum(len(g) for g in groups[:-1])
        st.pending_incomplete_token_timestamps = self._normalize_token_timestamps(
            times[first: first + len(groups[-1])], len(groups[-1]))

    # -- DRY repetition penalty (align_att_base.py:492-537) ---------------------------------------
    def _apply_dry_penalty(self, logits, current_tokens):
        eot = self.tokenizer.eot
        seq = current_tokens[0].tolist()
        if len(seq) < 5 or seq[-1] >= eot:
            return logits
        last, n = seq[-1], len(seq)
        longest: Dict[int, int] = {}
        for i in range(n - 2, -1, -1):
            if seq[i] != last or seq[i + 1] >= eot:
                continue
            m = 1
            while m < 50:
                j, k = i - m, n - 1 - m
                if j < 0 or k <= i or seq[j] != seq[k] or seq[j] >= eot:
                    break
                m += 1
            if m > longest.get(seq[i + 1], 0):
                longest[seq[i + 1]] = m
        for tok, m in longest.items():
            if m >= 2:
                logits[:, tok] = logits[:, tok] - 1.0 * 2.0 ** (m - 2)
        return logits


# ------------------------------------------------------------------------------------------------
# host half of the token update (whisper/decoding.py:271-376) on top-k read back from the device
# ------------------------------------------------------------------------------------------------
class BeamUpdate:
    """BeamSearchDecoder.update.  The device supplies, per beam row, the beam+1 best
    (log-prob, id) pairs; the ranking of candidate sequences is the reference's dict logic."""

    def __init__(self, beam_size: int, eot: int, patience: float = 1.0):
        self.beam_size, self.eot = beam_size, eot
        self.max_candidates = round(beam_size * patience)
        self.finished: Optional[List[dict]] = None

    def reset(self):
        self.finished = None

    def update(self, tokens: np.ndarray, top_logprobs: np.ndarray, top_ids: np.ndarray,
               sum_logprobs: np.ndarray):
        """-> (new tokens [rows, n+1], completed, source row of every new row)."""
        if tokens.shape[0] % self.beam_size != 0:
            raise ValueError(f"{tokens.shape}[0] % {self.beam_size} != 0")
        n_audio = tokens.shape[0] // self.beam_size
        if self.finished is None:
            self.finished = [{} for _ in range(n_audio)]
        nxt, sources, newly = [], [], []
        for i in range(n_audio):
            scores, src, fin = {}, {}, {}
            for j in range(self.beam_size):
                row = i * self.beam_size + j
                prefix = tokens[row].tolist()
                for lp, tk in zip(top_logprobs[row], top_ids[row]):
                    seq = tuple(prefix + [int(tk)])
                    scores[seq] = float(np.float32(sum_logprobs[row]) + np.float32(lp))
                    src[seq] = row
            saved = 0
            for seq in sorted(scores, key=scores.get, reverse=True):
                if seq[-1] == self.eot:
                    fin[seq] = scores[seq]
                else:
                    sum_logprobs[len(nxt)] = scores[seq]
                    nxt.append(seq)
                    sources.append(src[seq])
                    saved += 1
                    if saved == self.beam_size:
                        break
            newly.append(fin)
        for prev, new in zip(self.finished, newly):
            for seq in sorted(new, key=new.get, reverse=True):
                if len(prev) >= self.max_candidates:
                    break
                prev[seq] = new[seq]
        done = all(len(f) >= self.max_candidates for f in self.finished)
        return np.asarray(nxt, dtype=np.int64), done, sources


class GreedyUpdate:
    """GreedyDecoder.update at temperature 0 (whisper/decoding.py:271-287)."""

    def __init__(self, eot: int):
        self.eot = eot

    def reset(self):
        pass

    def update(self, tokens, top_logprobs, top_ids, sum_logprobs):
        nxt = top_ids[:, 0].astype(np.int64)
        live = tokens[:, -1] != self.eot
        sum_logprobs += np.where(live, top_logprobs[:, 0], 0).astype(np.float32)
        nxt = np.where(live, nxt, self.eot)
        tokens = np.concatenate([tokens, nxt[:, None]], axis=1)
        return tokens, bool((tokens[:, -1] == self.eot).all()), list(range(tokens.shape[0]))


def cif_fire_at_boundary(features: np.ndarray, weight: np.ndarray, bias: float) -> bool:
    """End-of-word test of the optional CIF head (simul_whisper/eow_detection.py:40-77) on the
    content part of the encoder output [T, d]: alpha = sigmoid(Linear(x)); rescale so the sum is
    an integer; fire if the first position after the last full unit lies in the final two frames.

    Host-side and tiny (one [T, d] x [d] product); evaluated with torch's CPU operators in the reference's
    operation order because the decision is discrete and sits on fp32 rounding when the final frames carry
    ~zero weight (the integral then lands within an ulp of an integer; tests/golden/cif_kat.json has such a
    case) - numpy's reduction order flips it."""
    import torch
    t = features.shape[0]
    x = torch.from_numpy(np.ascontiguousarray(features, dtype=np.float32))
    lin_w = torch.from_numpy(np.ascontiguousarray(weight, dtype=np.float32)).reshape(1, -1)
    lin_b = torch.tensor([bias], dtype=torch.float32)
    alphas = torch.sigmoid(torch.nn.functional.linear(x.unsqueeze(0), lin_w, lin_b).squeeze(2))[0]
    total = alphas.sum()
    a = alphas * (torch.round(total).int().float() / total)
    rounds = 0
    while bool((a > 0.999).any()):
        rounds += 1
        if rounds > 10:
            break
        for idx in torch.nonzero(a > 0.999).flatten().tolist():
            if a[idx] >= 0.999:
                mask = a.ne(0).float()
                a = a * 0.5 + (0.5 * a.sum() / mask.sum()) * mask
    integ = torch.cumsum(a[:-1], dim=0)
    if integ.numel() == 0:
        # a 1-frame feature: the reference indexes integrate[-1] of an empty tensor and raises
        # (eow_detection.py:70-71); process_iter turns that into an empty result for the call
        raise IndexError("index -1 is out of bounds for dimension 0 with size 0")
    integ = integ - (integ[-1] // 0.999) * 1.0
    pos = torch.nonzero(integ >= 0).flatten()
    return bool(pos.numel() and int(pos[0]) >= t - 2)

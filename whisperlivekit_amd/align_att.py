"""HipAlignAtt: the tensor hooks of AlignAtt on the MI355X C-ABI library.

This is "boundary B" of the reference (whisperlivekit/simul_whisper/align_att_base.py:541-649): the
abstract methods AlignAttBase.infer() calls.  The reference's PyTorch implementation of the same
hooks is AlignAtt (whisperlivekit/simul_whisper/simul_whisper.py:108-462); the MLX one
(simul_whisper/mlx/simul_whisper.py) is the precedent for a third backend.

Design differences from the PyTorch hooks, none observable by the policy:
  * audio lives on the GPU - only the new chunk is uploaded (insert_audio), eviction is a device copy;
  * logits never come back to the host: the filters the policy applies are recorded on a
    :class:`LazyLogits` proxy and replayed on the device inside ``_update_tokens`` together with
    log-softmax/top-k and the whole AlignAtt read-out, in ONE readback per decode step;
  * the 16-step cross-attention window (align_att_base.py:221-224) is a device ring owned by the
    session; the list the policy keeps holds placeholders.

The class is assembled on whichever policy base is available: the reference's own AlignAttBase when
WhisperLiveKit is installed (so its policy code runs unmodified), otherwise policy.AlignAttPolicy.
"""
from __future__ import annotations

import logging
import math
import os
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import policy as P
from .engine import HipSession, HipWhisperModel

logger = logging.getLogger(__name__)

TOKENS_PER_SECOND = 50


class _Segment:
    """One inserted audio chunk as the policy sees it: something with ``.shape[0]`` samples.
    A host copy is kept so the device buffer can be rebuilt after a policy-side list edit
    (refresh_segment slices ``state.segments`` directly, align_att_base.py:122-126); a run of zeros
    (a short silence, backend.py:86-90) keeps only its length."""
    __slots__ = ("data", "shape", "uid")
    _next = 0
    _lock = __import__("threading").Lock()

    def __init__(self, data: Optional[np.ndarray], n_zeros: int = 0):
        self.data = data
        self.shape = (int(data.shape[0]) if data is not None else int(n_zeros),)
        with _Segment._lock:                 # sessions are created / fed from different threads
            self.uid = _Segment._next
            _Segment._next += 1


class EncoderFeature:
    """Handle for the encoder output that stays on the GPU.  ``[:, :n, :]`` (the policy's slice for
    fire_at_boundary) returns a handle; ``.numpy()`` materialises [1, T, d] on the host."""

    def __init__(self, session: HipSession, n: Optional[int] = None):
        self._s, self._n = session, n
        d = session.model.dims
        self.shape = (1, d.n_audio_ctx if n is None else n, d.n_audio_state)
        self.ndim = 3

    def __getitem__(self, key):
        if isinstance(key, tuple) and len(key) == 3 and isinstance(key[1], slice):
            stop = key[1].stop
            return EncoderFeature(self._s, self.shape[1] if stop is None else min(stop, self.shape[1]))
        raise TypeError("EncoderFeature only supports the [:, :n, :] slice")

    def numpy(self) -> np.ndarray:
        d = self._s.model.dims
        full = self._s.export("enc").reshape(1, d.n_audio_ctx, d.n_audio_state)
        return full[:, : self.shape[1], :]


class _Column:
    """``logits[:, tok]`` inside the reference's DRY penalty (align_att_base.py:535):
    supports ``- x`` and ``+ x`` and remembers which column it is."""
    __slots__ = ("tok", "delta")

    def __init__(self, tok, delta=0.0):
        self.tok, self.delta = tok, delta

    def __sub__(self, x):
        return _Column(self.tok, self.delta - float(x))

    def __add__(self, x):
        return _Column(self.tok, self.delta + float(x))


class LazyLogits:
    """Stand-in for the [rows, n_tok, vocab] logits tensor.  Records in-place edits; the device
    applies them in wlk_select."""

    def __init__(self, n_rows: int, n_tok: int, n_vocab: int):
        self.shape = (n_rows, n_tok, n_vocab)
        self.adjust: Dict[Tuple[int, int], float] = {}   # (row or -1, token) -> additive delta

    def __getitem__(self, key):
        if isinstance(key, tuple) and len(key) == 3:        # logits[:, -1, :]
            return self
        if isinstance(key, tuple) and len(key) == 2:        # logits[:, tok]
            return _Column(key[1])
        raise TypeError("unsupported logits indexing")

    def __setitem__(self, key, value):
        if not (isinstance(key, tuple) and len(key) == 2):
            raise TypeError("unsupported logits assignment")
        toks = key[1]
        if isinstance(value, _Column):
            self.add(-1, [value.tok], value.delta)
        elif isinstance(value, (int, float)) and value == -math.inf:
            self.add(-1, list(toks) if isinstance(toks, (list, tuple, np.ndarray)) else [toks], -math.inf)
        else:
            raise TypeError("only -inf stores and column +/- deltas are supported on device logits")

    def add(self, row: int, tokens: Sequence[int], delta: float) -> None:
        for t in tokens:
            k = (row, int(t))
            self.adjust[k] = self.adjust.get(k, 0.0) + delta

    def float(self):
        return self


class AttendedFrames:
    """Result handle of _process_cross_attention: the frames were computed on the device."""

    def __init__(self, frames: np.ndarray, content_mel_len: int):
        self.frames = frames
        self.content_mel_len = content_mel_len


class HipAlignAttHooks:
    """The 20 hooks.  Mixed into a policy base by :func:`make_alignatt_class`."""

    def __init__(self, cfg, hip_model: HipWhisperModel = None, loaded_model=None, session: HipSession = None,
                 **_ignored) -> None:
        hip_model = hip_model if hip_model is not None else loaded_model
        if not getattr(hip_model, "_wlk_hip_model", False):
            raise TypeError("HipAlignAtt needs a HipWhisperModel (no CPU/PyTorch fallback on this backend)")
        self.device = f"hip:{hip_model.device}"
        self._base_init(cfg, hip_model)
        self.session = session if session is not None else hip_model.new_session(
            beam=cfg.beam_size, max_audio_seconds=max(64.0, 2.0 * cfg.audio_max_len + 4.0))
        self._dev_segments: List[Tuple[int, int]] = []   # (uid, samples) resident on the device, in order
        self._fresh_infer = True
        self._content_mel_len = 0
        self._last_frames: Optional[np.ndarray] = None
        self.counters = {"encode": 0, "decode": 0, "prefill_tokens": 0}
        # optional decision trace (bench.py / tests switch it on by assigning a list): one entry per infer,
        # [content_mel_len, [(token of beam 0, attended frame of beam 0), ...]] - what a golden stream pins
        self.decision_log: Optional[List[list]] = None
        # teacher forcing of single decisions, for parity harnesses only (None in production): {(infer index, decode step):
        # (token or -1, frame or -1)} - when this backend and the reference land on different sides of an fp32 tie, the
        # harness replays the stream with the reference's choice forced at that step, so every later decision is still
        # compared (wlk_loop_params.force_* in the library loop, _update_tokens in the per-token path; beam 1 only)
        self.teacher: Optional[dict] = None
        self._infer_index = -1
        self._step_index = 0
        self.state = P.StreamState()
        self.state.on_clean_cache = self._on_clean_cache
        self._init_state(cfg)

    # === state ===================================================================================
    def _init_state(self, cfg):
        self._init_state_common(cfg)
        st = self.state
        if not cfg.cif_ckpt_path:                                  # eow_detection.py:12-25
            st.always_fire, st.never_fire = (not cfg.never_fire), bool(cfg.never_fire)
        else:
            st.always_fire, st.never_fire = False, bool(cfg.never_fire)
            self._load_cif(cfg.cif_ckpt_path)
        st.align_source = {}
        st.num_align_heads = 0
        for layer, head in self.model.alignment_heads:             # simul_whisper.py:151-159
            st.align_source.setdefault(layer, []).append((st.num_align_heads, head))
            st.num_align_heads += 1
        tok = self.tokenizer
        suppress = [tok.transcribe, tok.translate, tok.sot, tok.sot_prev, tok.sot_lm, tok.no_timestamps]
        suppress += list(tok.all_language_tokens)
        if tok.no_speech is not None:
            suppress.append(tok.no_speech)
        st.suppress_ids = tuple(sorted(set(suppress)))             # simul_whisper.py:161-172
        self.init_tokens()
        self.init_context()
        st.decoder_type = cfg.decoder_type
        if cfg.decoder_type == "greedy":
            self._updater = P.GreedyUpdate(tok.eot)
        else:
            self._updater = P.BeamUpdate(cfg.beam_size, tok.eot)

    def _load_cif(self, path):
        import torch  # checkpoint format of the reference (a Linear(d, 1) state dict)
        ck = torch.load(path, map_location="cpu", weights_only=True)
        self.state.cif_weight = ck["weight"].float().numpy().reshape(-1)
        self.state.cif_bias = float(ck["bias"].float().numpy().reshape(-1)[0])

    def _on_clean_cache(self):
        self._fresh_infer = True
        if getattr(self, "_updater", None) is not None:
            self._updater.reset()

    def init_tokens(self):
        tok = self.tokenizer
        st = self.state
        st.initial_tokens = np.asarray([tok.sot_sequence_including_notimestamps], dtype=np.int64)
        st.initial_token_length = st.initial_tokens.shape[1]
        st.sot_index = list(tok.sot_sequence).index(tok.sot)
        st.tokens = [st.initial_tokens]

    def init_context(self):
        st = self.state
        st.context = P.TextContext("", self.tokenizer, [self.tokenizer.sot_prev])
        if self.cfg.static_init_prompt is not None:
            st.context = P.TextContext(self.cfg.static_init_prompt, self.tokenizer, [self.tokenizer.sot_prev])
        if self.cfg.init_prompt is not None:
            st.context.text += self.cfg.init_prompt

    # === audio (a1) ==============================================================================
    @staticmethod
    def _to_numpy(segment) -> np.ndarray:
        if hasattr(segment, "detach"):
            segment = segment.detach().cpu().numpy()
        if isinstance(segment, np.ndarray) and segment.dtype == np.int16:      # wire-format PCM: widened on the GPU
            return np.ascontiguousarray(segment.reshape(-1))
        return np.ascontiguousarray(np.asarray(segment, dtype=np.float32).reshape(-1))

    def _upload(self, data: Optional[np.ndarray], n_zeros: int = 0) -> None:
        if data is None:
            self.session.append_zeros(n_zeros)
        elif data.dtype == np.int16:
            self.session.append_pcm16(data)
        else:
            self.session.append(data)

    def insert_audio(self, segment=None):
        """simul_whisper.py:219-237.  The chunk goes straight to HBM; eviction shifts the attention
        bookkeeping and moves the evicted chunk's tokens into the text context."""
        st = self.state
        if segment is not None:
            seg = segment if isinstance(segment, _Segment) else _Segment(self._to_numpy(segment))
            self._sync_device_audio()
            st.segments.append(seg)
            self._upload(seg.data, seg.shape[0])
            self._dev_segments.append((seg.uid, seg.shape[0]))
        removed_len = 0
        total = self.segments_len()
        while len(st.segments) > 1 and total > self.cfg.audio_max_len:
            removed_len = st.segments[0].shape[0] / 16000
            total -= removed_len
            st.last_attend_frame -= int(TOKENS_PER_SECOND * removed_len)
            st.cumulative_time_offset += removed_len
            st.segments = st.segments[1:]
            if len(st.tokens) > 1:
                st.context.append_token_ids(st.tokens[1][0, :].tolist())
                st.tokens = [st.initial_tokens] + st.tokens[2:]
        self._sync_device_audio()
        return removed_len

    def insert_silence(self, n_samples: int):
        """``insert_audio(torch.zeros(n))`` of the reference's end_silence (backend.py:86-90) without the upload."""
        return self.insert_audio(_Segment(None, int(n_samples)))

    def _sync_device_audio(self):
        """Make the session's audio buffer equal to ``state.segments`` (which the policy may have
        sliced or emptied behind our back, align_att_base.py:122-126)."""
        want = [(s.uid, s.shape[0]) for s in self.state.segments]
        have = self._dev_segments
        if want == have:
            return
        k = len(have) - len(want)
        if 0 < k <= len(have) and have[k:] == want:         # a prefix was evicted
            self.session.drop_front(sum(n for _, n in have[:k]))
        else:                                               # anything else: rebuild from the host copies
            self.session.clear_audio()
            for seg in self.state.segments:
                self._upload(seg.data, seg.shape[0])
        self._dev_segments = want

    def _concat_segments(self):
        self._sync_device_audio()
        total = sum(s.shape[0] for s in self.state.segments)
        if total != self.session.audio_len:
            raise RuntimeError(f"device audio out of sync: {self.session.audio_len} != {total}")
        return total

    # === encoder (a2, a3, a4) ====================================================================
    def _encode(self, input_segments):
        self._content_mel_len = self.session.encode()
        self.counters["encode"] += 1
        self._fresh_infer = True
        self._infer_index += 1
        self._step_index = 0
        if self.decision_log is not None:
            self.decision_log.append([self._content_mel_len, []])
        return EncoderFeature(self.session), self._content_mel_len

    def fire_at_boundary(self, feature):
        st = self.state
        if st.always_fire:
            return True
        if st.never_fire:
            return False
        if st.cif_weight is None:
            return False
        return P.cif_fire_at_boundary(feature.numpy()[0], st.cif_weight, st.cif_bias)

    def lang_id(self, encoder_features):
        """simul_whisper.py:266-292: one <|sot|> step, language-token softmax."""
        tok = self.tokenizer
        rows = self.session.beam
        self.session.decode(np.full((rows, 1), tok.sot, np.int64), first=True, sot_index=0)
        logits = self.session.export("logits_last").reshape(rows, -1)[:1].astype(np.float32)
        mask = np.ones(logits.shape[-1], bool)
        mask[list(tok.all_language_tokens)] = False
        logits[:, mask] = -np.inf
        lang_tokens = logits.argmax(-1)
        e = np.exp(logits - logits.max(-1, keepdims=True))
        probs = e / e.sum(-1, keepdims=True)
        lang_probs = [{c: float(probs[i, j]) for j, c in zip(tok.all_language_tokens, tok.all_language_codes)}
                      for i in range(logits.shape[0])]
        self._clean_cache()
        return lang_tokens, lang_probs

    # === decoder (a5) ============================================================================
    def _current_tokens(self):
        st = self.state
        toks = list(st.tokens)
        if toks[0].shape[0] == 1:
            toks[0] = np.repeat(toks[0], self.cfg.beam_size, axis=0)
            st.tokens[0] = toks[0]
        if not st.context.is_empty():
            ctx = np.asarray([st.context.as_token_ids()], dtype=np.int64)
            toks = [np.repeat(ctx, self.cfg.beam_size, axis=0)] + toks
        return np.concatenate(toks, axis=1) if len(toks) > 1 else toks[0]

    def _init_sum_logprobs(self):
        return np.zeros(self.cfg.beam_size, dtype=np.float32)

    def _get_logits_and_cross_attn(self, tokens, encoder_feature):
        tokens = np.asarray(tokens, dtype=np.int64)
        self.session.decode(tokens, first=self._fresh_infer, sot_index=self.state.sot_index)
        self.counters["decode"] += 1
        if self._fresh_infer:
            self.counters["prefill_tokens"] += int(tokens.shape[1])
        self._fresh_infer = False
        return LazyLogits(tokens.shape[0], tokens.shape[1], self.model.dims.n_vocab), None

    def _check_no_speech(self, logits):
        if self.tokenizer.no_speech is None:
            return False
        p = self.session.no_speech_prob(self.tokenizer.no_speech)
        self.last_no_speech_prob = float(p[0])
        return bool(p[0] > self.cfg.nonspeech_prob)

    def _suppress_blank_tokens(self, logits):
        logits.add(-1, self.tokenizer.encode(" ") + [self.tokenizer.eot], -math.inf)
        return logits

    def _apply_token_suppression(self, logits):
        logits.add(-1, self.state.suppress_ids, -math.inf)
        return logits

    def _update_tokens(self, current_tokens, logits, sum_logprobs):
        """Device: filters + log-softmax + top-(beam+1) + AlignAtt read-out, one readback.
        Host: the candidate ranking of whisper/decoding.py:317-376."""
        rows, ids, deltas = [], [], []
        for (r, t), dl in logits.adjust.items():
            rows.append(r), ids.append(t), deltas.append(dl)
        k = 1 if self.state.decoder_type == "greedy" else self.cfg.beam_size + 1
        lp, top, frames = self.session.select(rows, ids, deltas, k, self._content_mel_len)
        forced = self.teacher.get((self._infer_index, self._step_index)) if self.teacher else None
        self._step_index += 1
        if forced is not None:
            if self.cfg.beam_size != 1:
                raise RuntimeError("teacher forcing is defined for beam 1 only")
            tok_f, frame_f = forced
            lp, top, frames = np.array(lp), np.array(top), np.array(frames)
            if frame_f >= 0:
                frames[0] = frame_f
            if tok_f >= 0 and top.shape[1] > 1 and int(top[0, 0]) != self.tokenizer.eot and int(top[0, 1]) == tok_f:
                top[0, [0, 1]], lp[0, [0, 1]] = top[0, [1, 0]], lp[0, [1, 0]]
        self._last_frames = frames
        self.last_top = (lp, top)
        tokens, completed, sources = self._updater.update(np.asarray(current_tokens), lp, top, sum_logprobs)
        if self.decision_log is not None:
            self.decision_log[-1][1].append((int(tokens[0, -1]), int(frames[0])))
        if sources != list(range(len(sources))):
            self.session.kv_reorder(sources)
        return tokens, completed

    # === the whole decode loop behind one call ===================================
    def device_loop_available(self) -> bool:
        """Beam 1 with the reference's (hard-wired) beam decoder, on a session that offers the loop; switchable
        per object (``use_device_loop``) and globally (WLK_DEVICE_LOOP=0) so the per-token hook path stays testable."""
        if not getattr(self, "use_device_loop", True) or os.environ.get("WLK_DEVICE_LOOP", "1") == "0":
            return False
        return (self.cfg.beam_size == 1 and self.state.decoder_type == "beam"
                and hasattr(self.session, "decode_until_stop"))

    def _decode_until_stop(self, tokens, content_mel_len, is_last, budget):
        """align_att_base.py:206-286 as ONE library call: returns the loop's outcome (engine.LoopOutcome)."""
        from . import _lib
        tok, cfg, st = self.tokenizer, self.cfg, self.state
        if content_mel_len != self._content_mel_len:
            raise RuntimeError("content_mel_len changed between _encode and the decode loop")
        if content_mel_len <= 0:
            raise RuntimeError("attention read-out over zero content frames")
        p = _lib.LoopParams(
            sot_index=st.sot_index, is_last=1 if is_last else 0, frame_threshold=int(cfg.frame_threshold),
            rewind_threshold=int(cfg.rewind_threshold), last_attend_frame=int(st.last_attend_frame),
            max_text_len=int(self.max_text_len), budget=int(budget), eot=int(tok.eot), dec_pad=P.DEC_PAD,
            no_speech_token=-1 if tok.no_speech is None else int(tok.no_speech),
            no_speech_threshold=float(cfg.nonspeech_prob), content_mel_len=int(content_mel_len))
        if self.teacher:
            p.force([(si, t, f) for (ci, si), (t, f) in sorted(self.teacher.items()) if ci == self._infer_index])
        blank = list(tok.encode(" ")) + [tok.eot]
        out = self.session.decode_until_stop(np.asarray(tokens)[0], p, st.suppress_ids, blank)
        self._fresh_infer = False
        self.counters["decode"] += out.decode_calls
        if out.decode_calls:
            self.counters["prefill_tokens"] += int(np.asarray(tokens).shape[1])
        self.last_no_speech_prob = out.no_speech_prob
        if self.decision_log is not None:
            self.decision_log[-1][1].extend(zip(out.step_tokens, out.step_frames))
        return out

    # === AlignAtt read-out (a8) =====================================================================
    def _process_cross_attention(self, accumulated_cross_attns, content_mel_len):
        if content_mel_len != self._content_mel_len:
            raise RuntimeError("content_mel_len changed between _encode and the read-out")
        return AttendedFrames(self._last_frames, content_mel_len)

    def _get_attended_frames(self, attn):
        if attn.content_mel_len <= 0:
            # torch.argmax over an empty frame axis raises in the reference (simul_whisper.py:436)
            raise RuntimeError("attention read-out over zero content frames")
        frames = [int(f) for f in attn.frames]
        return frames, frames[0]

    # === token plumbing ==============================================================================
    def _is_special_token(self, current_tokens):
        return int(current_tokens[0, -2]) >= P.DEC_PAD

    def _rewind_tokens(self):
        st = self.state
        return np.concatenate(st.tokens, axis=1) if len(st.tokens) > 0 else st.tokens[0]

    def _tokens_to_list(self, current_tokens, start_col):
        return [int(t) for t in np.asarray(current_tokens)[0, start_col:].reshape(-1)]

    def _make_new_tokens_tensor(self, hypothesis):
        row = np.asarray([list(hypothesis)], dtype=np.int64).reshape(1, -1)
        return np.repeat(row, self.cfg.beam_size, axis=0)

    def _evaluate(self, tensor):
        pass

    def debug_print_tokens(self, tokens):  # the reference logs decoded beams at DEBUG level
        if logger.isEnabledFor(logging.DEBUG):
            for i in range(min(self.cfg.beam_size, tokens.shape[0])):
                logger.debug(self.tokenizer.decode_with_timestamps(tokens[i].tolist()))

    def close(self):
        self.session.close()


def _reference_base():
    try:
        from whisperlivekit.simul_whisper.align_att_base import AlignAttBase  # type: ignore
        return AlignAttBase
    except Exception:
        return None


def make_alignatt_class(base=None):
    """HipAlignAtt on the given policy base (default: the reference's AlignAttBase if WhisperLiveKit
    is importable, else this package's AlignAttPolicy)."""
    if base is None:
        base = _reference_base() or P.AlignAttPolicy
    ns: Dict[str, Any] = {"__doc__": HipAlignAttHooks.__doc__}
    if base is not P.AlignAttPolicy:
        # the reference builds ASRToken itself; its _base_init wants model.decoder.blocks
        pass
    cls = type("HipAlignAtt", (HipAlignAttHooks, base), ns)
    if getattr(cls, "__abstractmethods__", None):
        missing = sorted(cls.__abstractmethods__)
        if missing:
            raise TypeError(f"HipAlignAtt does not implement hooks: {missing}")
    return cls


HipAlignAttStandalone = make_alignatt_class(P.AlignAttPolicy)

"""Thin Python objects over the C ABI: one :class:`HipWhisperModel` per GPU (immutable packed
weights, shared by every session on that GPU) and one :class:`HipSession` per audio stream
(device-resident rolling audio, KV caches, alignment window, HIP stream)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .dims import ALIGNMENT_HEADS, MODEL_DIMS, ModelDims, default_alignment_heads
from .melbank import mel_filterbank


def _as_f32(x) -> np.ndarray:
    if hasattr(x, "detach"):  # torch tensor
        x = x.detach().cpu().float().numpy()
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def hann_window_periodic(n: int = 400) -> np.ndarray:
    """torch.hann_window(n) (periodic).  Uses torch when importable so the fp32 values are the
    very ones whisper/audio.py:147 multiplies with."""
    try:
        import torch
        return torch.hann_window(n).numpy().astype(np.float32)
    except Exception:  # pragma: no cover
        k = np.arange(n, dtype=np.float64)
        return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)).astype(np.float32)


def pack_state_dict(dims: ModelDims, sd: Mapping[str, object]) -> Dict[str, np.ndarray]:
    """Reference checkpoint names (whisper/model.py module tree, as produced by load_model,
    whisper/__init__.py:466-596) -> the packed tensors wlk_tensor_lookup() defines.

    * conv weights [out, in, tap] become tap-major [out, tap*in] rows (the conv is a GEMM over a
      time-major activation buffer);
    * query/key/value (key has no bias, model.py:89) are concatenated into one [3d, d] projection,
      cross-attention key/value into one [2d, d] projection."""
    g = lambda k: _as_f32(sd[k])
    out: Dict[str, np.ndarray] = {}
    out["mel.filters"] = np.ascontiguousarray(mel_filterbank(dims.n_mels))
    out["mel.window"] = hann_window_periodic()
    d = dims.n_audio_state
    out["enc.conv1.w"] = np.ascontiguousarray(g("encoder.conv1.weight").transpose(0, 2, 1).reshape(d, -1))
    out["enc.conv1.b"] = g("encoder.conv1.bias")
    out["enc.conv2.w"] = np.ascontiguousarray(g("encoder.conv2.weight").transpose(0, 2, 1).reshape(d, -1))
    out["enc.conv2.b"] = g("encoder.conv2.bias")
    out["enc.pos"] = g("encoder.positional_embedding")

    def block(dst: str, src: str, width: int, cross: bool):
        zeros = np.zeros(width, np.float32)
        a = src + ".attn"
        out[dst + "ln1.w"], out[dst + "ln1.b"] = g(src + ".attn_ln.weight"), g(src + ".attn_ln.bias")
        out[dst + "qkv.w"] = np.concatenate([g(a + ".query.weight"), g(a + ".key.weight"), g(a + ".value.weight")])
        out[dst + "qkv.b"] = np.concatenate([g(a + ".query.bias"), zeros, g(a + ".value.bias")])
        out[dst + "out.w"], out[dst + "out.b"] = g(a + ".out.weight"), g(a + ".out.bias")
        if cross:
            x = src + ".cross_attn"
            out[dst + "lnx.w"], out[dst + "lnx.b"] = g(src + ".cross_attn_ln.weight"), g(src + ".cross_attn_ln.bias")
            out[dst + "xq.w"], out[dst + "xq.b"] = g(x + ".query.weight"), g(x + ".query.bias")
            out[dst + "xkv.w"] = np.concatenate([g(x + ".key.weight"), g(x + ".value.weight")])
            out[dst + "xkv.b"] = np.concatenate([zeros, g(x + ".value.bias")])
            out[dst + "xout.w"], out[dst + "xout.b"] = g(x + ".out.weight"), g(x + ".out.bias")
        out[dst + "ln2.w"], out[dst + "ln2.b"] = g(src + ".mlp_ln.weight"), g(src + ".mlp_ln.bias")
        out[dst + "fc1.w"], out[dst + "fc1.b"] = g(src + ".mlp.0.weight"), g(src + ".mlp.0.bias")
        out[dst + "fc2.w"], out[dst + "fc2.b"] = g(src + ".mlp.2.weight"), g(src + ".mlp.2.bias")

    for i in range(dims.n_audio_layer):
        block(f"enc.{i}.", f"encoder.blocks.{i}", dims.n_audio_state, cross=False)
    out["enc.ln_post.w"], out["enc.ln_post.b"] = g("encoder.ln_post.weight"), g("encoder.ln_post.bias")
    out["dec.tok_emb"] = g("decoder.token_embedding.weight")
    out["dec.pos"] = g("decoder.positional_embedding")
    for i in range(dims.n_text_layer):
        block(f"dec.{i}.", f"decoder.blocks.{i}", dims.n_text_state, cross=True)
    out["dec.ln.w"], out["dec.ln.b"] = g("decoder.ln.weight"), g("decoder.ln.bias")
    return {k: np.ascontiguousarray(v.reshape(-1)) for k, v in out.items()}


def _cdims(dims: ModelDims) -> _lib.Dims:
    return _lib.Dims(*dims.as_tuple())


def arena_floats(dims: ModelDims) -> int:
    n = C.c_uint64()
    _lib.check(_lib.load().wlk_arena_floats(C.byref(_cdims(dims)), C.byref(n)))
    return int(n.value)


def packed_tensor_names(dims: ModelDims) -> List[str]:
    lib = _lib.load()
    names, i = [], 0
    cd = _cdims(dims)
    while True:
        s = C.c_char_p()
        if lib.wlk_tensor_name(C.byref(cd), i, C.byref(s)) != 0:
            break
        names.append(s.value.decode())
        i += 1
    return names


class HipWhisperModel:
    """Packed fp32 weights of one Whisper checkpoint resident on one GPU."""
    _wlk_hip_model = True

    def __init__(self, dims: ModelDims, device: int = 0, arena=None):
        """``arena``: optional torch CUDA float32 tensor of ``arena_floats(dims)`` elements that the
        caller owns (torch.distributed broadcasts it over RCCL); otherwise the library allocates."""
        self.lib = _lib.load()
        self.dims = dims
        self.device = device
        self._arena_keepalive = arena
        self._h = C.c_void_p()
        ptr = None
        if arena is not None:
            if arena.numel() != arena_floats(dims) or str(arena.dtype) != "torch.float32" or not arena.is_cuda:
                raise ValueError("arena must be a CUDA float32 tensor of arena_floats(dims) elements")
            ptr = C.c_void_p(arena.data_ptr())
        _lib.check(self.lib.wlk_model_create(C.byref(_cdims(dims)), device, ptr, C.byref(self._h)))
        self.alignment_heads: List[Tuple[int, int]] = []
        self.finalized = False
        # the reference's AlignAttBase._base_init reads len(model.decoder.blocks) (align_att_base.py:58)
        import types
        self.decoder = types.SimpleNamespace(blocks=[None] * dims.n_text_layer)

    # -- weights --------------------------------------------------------------------------
    def upload_packed(self, packed: Mapping[str, np.ndarray]) -> None:
        expected = set(packed_tensor_names(self.dims))
        missing = expected - set(packed)
        if missing:
            raise KeyError(f"missing packed tensors: {sorted(missing)[:5]} ...")
        for name in expected:
            a = np.ascontiguousarray(packed[name], dtype=np.float32).reshape(-1)
            _lib.check(self.lib.wlk_model_upload(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))

    def load_state_dict(self, sd: Mapping[str, object]) -> None:
        self.upload_packed(pack_state_dict(self.dims, sd))

    def set_alignment_heads(self, pairs: Sequence[Tuple[int, int]]) -> None:
        flat = np.asarray([x for p in pairs for x in p], dtype=np.int32)
        _lib.check(self.lib.wlk_model_set_alignment_heads(self._h, flat.ctypes.data_as(C.c_void_p), len(pairs)))
        self.alignment_heads = [tuple(p) for p in pairs]

    def finalize(self) -> None:
        _lib.check(self.lib.wlk_model_finalize(self._h))
        self.finalized = True

    @classmethod
    def from_state_dict(cls, dims: ModelDims, sd: Mapping[str, object],
                        alignment_heads: Optional[Sequence[Tuple[int, int]]] = None, device: int = 0,
                        arena=None) -> "HipWhisperModel":
        m = cls(dims, device, arena)
        m.load_state_dict(sd)
        m.set_alignment_heads(alignment_heads if alignment_heads is not None else default_alignment_heads(dims))
        m.finalize()
        return m

    @classmethod
    def synthetic(cls, name: str, seed: int = 0, device: int = 0) -> "HipWhisperModel":
        """Seeded random weights of a named architecture (no checkpoint / network needed)."""
        from . import synth
        dims = MODEL_DIMS[name]
        return cls.from_state_dict(dims, synth.synth_state_dict(dims, seed), ALIGNMENT_HEADS[name], device)

    # -- reference-compatible read-only attributes (whisper/model.py:385-395) ----------------
    @property
    def is_multilingual(self) -> bool:
        return self.dims.is_multilingual

    @property
    def num_languages(self) -> int:
        return self.dims.num_languages

    def new_session(self, beam: int = 1, max_audio_seconds: float = 64.0, batched: Optional[bool] = None) -> "HipSession":
        """``batched`` (default: on for beam 1 unless WLK_BATCH_DECODE=0): the session's decode loops share batched
        single-token steps with the other sessions of this GPU (wlk_engine_attach)."""
        s = HipSession(self, beam, int(max_audio_seconds * 16000))
        if batched is None:
            batched = beam == 1 and os.environ.get("WLK_BATCH_DECODE", "1") != "0"
        if batched:
            s.attach_engine()
        return s

    def engine_stats(self) -> Dict[str, int]:
        """Iterations of this GPU's batch engine and the rows (= session steps) advanced in them."""
        v = [C.c_uint64() for _ in range(4)]
        _lib.check(self.lib.wlk_engine_stats(self._h, *[C.byref(x) for x in v]))
        it, rows, bs, br = (int(x.value) for x in v)
        eb, es = C.c_uint64(), C.c_uint64()
        _lib.check(self.lib.wlk_engine_encode_stats(self._h, C.byref(eb), C.byref(es)))
        pb, ps = C.c_uint64(), C.c_uint64()
        _lib.check(self.lib.wlk_engine_prefill_stats(self._h, C.byref(pb), C.byref(ps)))
        return dict(iterations=it, rows=rows, batched_steps=bs, batched_rows=br,
                    mean_rows_per_batched_step=round(br / bs, 3) if bs else None,
                    encode_batches=int(eb.value), encoded_sessions=int(es.value),
                    mean_sessions_per_encode_batch=round(es.value / eb.value, 3) if eb.value else None,
                    prefill_batches=int(pb.value), stacked_prefills=int(ps.value),
                    mean_sessions_per_prefill_batch=round(ps.value / pb.value, 3) if pb.value else None)

    def close(self) -> None:
        # sessions the batch path (transcribe.py) keeps per calling thread hold pointers into this model
        for rows in (self.__dict__.pop("_batch_rows", None) or {}).values():
            for sess in rows.sessions.values():
                sess.close()
        if self._h:
            self.lib.wlk_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class LoopOutcome:
    """What wlk_decode_until_stop / wlk_job_result hand back, as plain Python values."""
    __slots__ = ("new_tokens", "step_tokens", "step_frames", "step_sum_logprobs", "stop_reason", "last_attend_frame",
                 "no_speech_prob", "sum_logprob", "decode_calls")

    def __init__(self, res: "_lib.LoopResult", new, step_tokens, step_frames, step_sums):
        n, k = int(res.n_steps), int(res.n_new_tokens)
        self.new_tokens = [int(x) for x in new[:k]]
        self.step_tokens = [int(x) for x in step_tokens[:n]]
        self.step_frames = [int(x) for x in step_frames[:n]]
        self.step_sum_logprobs = [float(x) for x in step_sums[:n]]
        self.stop_reason = int(res.stop_reason)
        self.last_attend_frame = int(res.last_attend_frame)
        self.no_speech_prob = float(res.no_speech_prob)
        self.sum_logprob = float(res.sum_logprob)
        self.decode_calls = int(res.decode_calls)


class HipSession:
    """Per-stream device state.  One call in flight at a time (the reference's threading
    contract for a session); different sessions may be driven from different threads."""

    def __init__(self, model: HipWhisperModel, beam: int = 1, max_audio_samples: int = 64 * 16000):
        if not model.finalized:
            raise _lib.WlkError("model must be finalized before creating sessions")
        self.model = model
        self.lib = model.lib
        self.beam = beam
        self._h = C.c_void_p()
        _lib.check(self.lib.wlk_session_create(model._h, beam, max_audio_samples, C.byref(self._h)))
        self.max_audio_samples = max_audio_samples

    # -- audio (a1) -------------------------------------------------------------------------
    def append(self, pcm: np.ndarray) -> None:
        a = np.ascontiguousarray(pcm, dtype=np.float32).reshape(-1)
        _lib.check(self.lib.wlk_audio_append(self._h, a.ctypes.data_as(C.c_void_p), a.size))

    def append_pcm16(self, pcm: np.ndarray) -> None:
        """int16 samples as they arrive on the wire; widened to fp32 / 32768 on the device."""
        a = np.ascontiguousarray(pcm, dtype=np.int16).reshape(-1)
        _lib.check(self.lib.wlk_audio_append_pcm16(self._h, a.ctypes.data_as(C.c_void_p), a.size))

    def append_zeros(self, n: int) -> None:
        _lib.check(self.lib.wlk_audio_append_zeros(self._h, int(n)))

    def drop_front(self, n: int) -> None:
        _lib.check(self.lib.wlk_audio_drop_front(self._h, int(n)))

    def clear_audio(self) -> None:
        _lib.check(self.lib.wlk_audio_clear(self._h))

    @property
    def audio_len(self) -> int:
        n = C.c_int()
        _lib.check(self.lib.wlk_audio_len(self._h, C.byref(n)))
        return n.value

    # -- hot path ---------------------------------------------------------------------------
    def encode(self) -> int:
        cml = C.c_int32()
        _lib.check(self.lib.wlk_encode(self._h, C.byref(cml)))
        return cml.value

    def encode_mel(self, mel: np.ndarray) -> None:
        """Encode a [n_mels, 3000] log-mel segment as whisper.transcribe() / find_alignment hand it to the model
        (instead of the session's own audio): encoder + cross-K/V."""
        mel = np.ascontiguousarray(_as_f32(mel))
        if mel.shape != (self.model.dims.n_mels, 3000):
            raise ValueError(f"encode_mel: expected [{self.model.dims.n_mels}, 3000], got {mel.shape}")
        _lib.check(self.lib.wlk_encode_mel(self._h, mel.ctypes.data_as(C.POINTER(C.c_float)), 3000))

    def log_mel(self, audio: np.ndarray, padding: int = 0) -> np.ndarray:
        """whisper/audio.py:log_mel_spectrogram(audio, n_mels, padding) of a whole recording -> [n_mels, frames]."""
        a = np.ascontiguousarray(_as_f32(audio)).reshape(-1)
        n = C.c_int32()
        _lib.check(self.lib.wlk_log_mel(self._h, None, a.size, int(padding), None, 0, C.byref(n)))
        out = np.empty((self.model.dims.n_mels, n.value), np.float32)
        _lib.check(self.lib.wlk_log_mel(self._h, a.ctypes.data_as(C.POINTER(C.c_float)), a.size, int(padding),
                                        out.ctypes.data_as(C.POINTER(C.c_float)), out.size, C.byref(n)))
        return out

    def find_alignment(self, tokens: Sequence[int], n_sot: int, eot: int, num_frames: int, qk_scale: float = 1.0,
                       want_cost: bool = False):
        """Device half of whisper/timing.py:find_alignment on the encoded session.  ``tokens`` = [sot sequence (n_sot
        ids), <|notimestamps|>, text tokens, <|endoftext|>].  -> (dtw step codes [(n_text + 2), num_frames // 2 + 1],
        token probabilities [n_text], cost matrix [n_text + 1, num_frames // 2] or None)."""
        toks = np.ascontiguousarray(np.asarray(tokens, dtype=np.int64))
        n_text = len(toks) - n_sot - 2
        if n_text < 1:
            raise ValueError("find_alignment: no text tokens")
        f = num_frames // 2
        trace = np.empty((n_text + 2, f + 1), dtype=np.int8)
        probs = np.empty(n_text, dtype=np.float32)
        cost = np.empty((n_text + 1, f), dtype=np.float32) if want_cost else None
        _lib.check(self.lib.wlk_find_alignment(
            self._h, toks.ctypes.data_as(C.POINTER(C.c_int64)), len(toks), int(n_sot), int(eot), int(num_frames),
            float(qk_scale), cost.ctypes.data_as(C.POINTER(C.c_float)) if want_cost else None,
            trace.ctypes.data_as(C.POINTER(C.c_int8)), probs.ctypes.data_as(C.POINTER(C.c_float))))
        return trace, probs, cost

    def decode(self, tokens: np.ndarray, first: bool, sot_index: int = 0) -> None:
        t = np.ascontiguousarray(tokens, dtype=np.int64)
        if t.ndim != 2:
            raise ValueError("tokens must be [rows, n_tok]")
        _lib.check(self.lib.wlk_decode(self._h, t.ctypes.data_as(C.c_void_p), t.shape[0], t.shape[1],
                                       1 if first else 0, int(sot_index)))

    def no_speech_prob(self, token: int) -> np.ndarray:
        out = np.empty(self.beam, np.float32)
        _lib.check(self.lib.wlk_no_speech_prob(self._h, int(token), out.ctypes.data_as(C.c_void_p)))
        return out

    def set_rules(self, suppressed: Sequence[int], blank: Sequence[int]) -> None:
        """The token lists of whisper's SuppressTokens / SuppressBlank for wlk_pick_greedy (decoding.py:417-432)."""
        a = np.ascontiguousarray(suppressed, dtype=np.int32).reshape(-1)
        b = np.ascontiguousarray(blank, dtype=np.int32).reshape(-1)
        _lib.check(self.lib.wlk_rules_set(self._h, a.ctypes.data_as(C.c_void_p), a.size, b.ctypes.data_as(C.c_void_p), b.size))

    def pick_greedy(self, *, first_step: bool, without_timestamps: bool, timestamp_begin: int, eot: int, no_timestamps: int,
                    ts_mode: int, ts_bound: int, max_initial: int) -> Tuple[int, float]:
        """Logit rules + argmax + log-probability of the last decode's row on the device (decoding.py:270-287, 417-499)."""
        prm = np.asarray([int(first_step), int(without_timestamps), timestamp_begin, eot, no_timestamps, ts_mode, ts_bound,
                          max_initial], dtype=np.int32)
        tok = np.empty(1, np.int32)
        lp = np.empty(1, np.float32)
        _lib.check(self.lib.wlk_pick_greedy(self._h, prm.ctypes.data_as(C.c_void_p), tok.ctypes.data_as(C.c_void_p),
                                            lp.ctypes.data_as(C.c_void_p)))
        return int(tok[0]), float(lp[0])

    def select(self, adj_rows: Sequence[int], adj_ids: Sequence[int], adj_deltas: Sequence[float], k: int,
               content_mel_len: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        n = len(adj_ids)
        r = np.asarray(adj_rows, dtype=np.int32)
        i = np.asarray(adj_ids, dtype=np.int32)
        dl = np.asarray(adj_deltas, dtype=np.float32)
        lp = np.empty((self.beam, k), np.float32)
        ids = np.empty((self.beam, k), np.int32)
        fr = np.empty(self.beam, np.int32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(self.lib.wlk_select(self._h, vp(r), vp(i), vp(dl), n, k, int(content_mel_len), vp(lp), vp(ids),
                                       vp(fr)))
        return lp, ids, fr

    def decode_until_stop(self, tokens: Sequence[int], params: "_lib.LoopParams", suppress_ids: Sequence[int],
                          blank_ids: Sequence[int]) -> "LoopOutcome":
        """The whole AlignAtt decode loop of one infer (beam 1) in one library call (wlk_decode_until_stop)."""
        t = np.ascontiguousarray(tokens, dtype=np.int64).reshape(-1)
        sup = np.ascontiguousarray(suppress_ids, dtype=np.int32)
        blank = np.ascontiguousarray(blank_ids, dtype=np.int32)
        cap = int(params.max_text_len) + 8
        res = _lib.LoopResult()
        new = np.empty(cap, np.int64)
        st, sf = np.empty(cap, np.int32), np.empty(cap, np.int32)
        ss = np.empty(cap, np.float32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(self.lib.wlk_decode_until_stop(self._h, vp(t), t.size, C.byref(params), vp(sup), sup.size, vp(blank),
                                                  blank.size, C.byref(res), vp(new), vp(st), vp(sf), vp(ss), cap))
        return LoopOutcome(res, new, st, sf, ss)

    def attach_engine(self) -> None:
        _lib.check(self.lib.wlk_engine_attach(self._h))

    def detach_engine(self) -> None:
        _lib.check(self.lib.wlk_engine_detach(self._h))

    def kv_reorder(self, source_rows: Sequence[int]) -> None:
        s = np.asarray(source_rows, dtype=np.int32)
        _lib.check(self.lib.wlk_kv_reorder(self._h, s.ctypes.data_as(C.c_void_p), s.size))

    def sync(self) -> None:
        _lib.check(self.lib.wlk_sync(self._h))

    # -- parity / profiling -------------------------------------------------------------------
    def set_debug(self, on: bool = True) -> None:
        _lib.check(self.lib.wlk_session_set_debug(self._h, 1 if on else 0))

    def export(self, what: str, max_floats: Optional[int] = None) -> np.ndarray:
        d = self.model.dims
        if max_floats is None:
            max_floats = max(d.n_mels * 3000, 1500 * d.n_audio_state, self.beam * d.n_vocab,
                             self.beam * d.n_text_ctx * d.n_text_head * 1500)
        buf = np.empty(max_floats, np.float32)
        n = C.c_uint64()
        _lib.check(self.lib.wlk_export(self._h, what.encode(), buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(n)))
        return buf[: n.value].copy()

    def step_stats(self) -> Dict[str, float]:
        """Wall time of this session's graph-replayed single-token steps so far (wlk_session_step_stats)."""
        n, wall, launch = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _lib.check(self.lib.wlk_session_step_stats(self._h, C.byref(n), C.byref(wall), C.byref(launch)))
        return dict(steps=int(n.value), wall_ns=int(wall.value), launch_ns=int(launch.value))

    def prof_begin(self) -> None:
        _lib.check(self.lib.wlk_prof_begin(self._h))

    def prof_end(self, cap: int = 64) -> Dict[str, Dict[str, float]]:
        """-> {launch tag: {"ms", "launches", "flops", "bytes"}} summed since prof_begin()."""
        names = (C.c_char_p * cap)()
        ms = np.zeros(cap, np.float32)
        cnt = np.zeros(cap, np.int32)
        fl = np.zeros(cap, np.float64)
        by = np.zeros(cap, np.float64)
        n = C.c_int32()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(self.lib.wlk_prof_end(self._h, cap, names, vp(ms), vp(cnt), vp(fl), vp(by), C.byref(n)))
        return {names[i].decode(): dict(ms=float(ms[i]), launches=int(cnt[i]), flops=float(fl[i]), bytes=float(by[i]))
                for i in range(n.value)}

    def close(self) -> None:
        if self._h:
            self.lib.wlk_session_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

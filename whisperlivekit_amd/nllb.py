"""NLLB-200 (M2M-100 architecture) on the HIP library: the translation model of config 5.

The reference reaches this model through the third-party ``nllw`` package (``whisperlivekit/core.py:320-329``
``nllw.load_model``; ``core.py:483-493`` / ``translation.py`` wrap it in ``nllw.OnlineTranslation``).  ``nllw`` is not in the
reference tree; what it executes is the published M2M-100 network of ``transformers``
(``models/m2m_100/modeling_m2m_100.py``) or its CTranslate2 conversion, selected per call by ``forced_bos_token_id`` = the
target language code.  This module is that network behind the C ABI (``wlk_nllb_*``, csrc/nllb.hip) with

* :func:`pack_hf_state_dict` - a ``transformers`` checkpoint's tensors (``model.shared.weight``,
  ``model.encoder.layers.N.self_attn.q_proj.weight`` ...) into the packed arena;
* :class:`HipNllbModel` / :class:`HipNllbSession` - encoder pass, decoder steps with KV cache, beam reorder, top-k;
* :func:`generate` / :func:`beam_search` - ``GenerationMixin.generate`` for the cases the translation backends use: greedy
  or beam search from ``[decoder_start_token_id]`` with the target language forced as the first generated token and
  ``</s>`` ending a hypothesis.

Pinned by ``transformers``' own ``M2M100ForConditionalGeneration`` on seeded weights (``scripts/gen_golden_nllb.py``,
``tests/golden/nllb_kat.npz``).  The streaming policy of ``nllw.OnlineTranslation`` (when to re-translate which prefix) is
host logic of that package and is not restated.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np

from . import _lib


@dataclass(frozen=True)
class NllbConfig:
    """The fields of ``M2M100Config`` the network depends on."""
    vocab_size: int = 256206
    d_model: int = 1024
    encoder_layers: int = 12
    decoder_layers: int = 12
    attention_heads: int = 16
    ffn_dim: int = 4096
    scale_embedding: bool = True
    pad_token_id: int = 1
    eos_token_id: int = 2
    decoder_start_token_id: int = 2
    max_position_embeddings: int = 1024


NLLB_200_DISTILLED_600M = NllbConfig()
NLLB_MICRO = NllbConfig(vocab_size=2003, d_model=128, encoder_layers=2, decoder_layers=2, attention_heads=2, ffn_dim=256,
                        max_position_embeddings=96)


def sinusoid_table(n_rows: int, d: int, padding_idx: Optional[int]) -> np.ndarray:
    """``M2M100SinusoidalPositionalEmbedding.get_embedding`` (tensor2tensor flavour: all sines, then all cosines; the
    padding row zeroed), computed with torch's float32 operators so that the table is the one ``transformers`` builds."""
    import torch
    half = d // 2
    step = math.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half, dtype=torch.int64).float() * -step)
    ang = torch.arange(n_rows, dtype=torch.int64).float().unsqueeze(1) * freq.unsqueeze(0)
    table = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1).view(n_rows, -1)
    if d % 2 == 1:
        table = torch.cat([table, torch.zeros(n_rows, 1)], dim=1)
    if padding_idx is not None:
        table[padding_idx, :] = 0
    return table.numpy().astype(np.float32)


def _f32(x) -> np.ndarray:
    if hasattr(x, "detach"):
        x = x.detach().cpu().float().numpy()
    return np.ascontiguousarray(x, dtype=np.float32)


def pack_hf_state_dict(cfg: NllbConfig, sd: Mapping[str, object], n_positions: int) -> Dict[str, np.ndarray]:
    """``transformers`` parameter names -> packed names (include/wlk_hip.h).  q / k / v projections are stacked row-wise."""
    out: Dict[str, np.ndarray] = {}
    shared = sd.get("model.shared.weight")
    if shared is None:
        shared = sd.get("lm_head.weight", sd.get("model.encoder.embed_tokens.weight"))
    if shared is None:
        raise KeyError("no shared embedding in the state dict (model.shared.weight / lm_head.weight)")
    out["shared.emb"] = _f32(shared)
    out["pos.table"] = sinusoid_table(n_positions, cfg.d_model, cfg.pad_token_id)

    def lin(prefix):
        return _f32(sd[prefix + ".weight"]), _f32(sd[prefix + ".bias"])

    def block(dst: str, src: str, cross: bool):
        for ours, theirs in (("ln1", "self_attn_layer_norm"), ("ln2", "final_layer_norm"), ("fc1", "fc1"), ("fc2", "fc2"),
                             ("out", "self_attn.out_proj")):
            out[dst + ours + ".w"], out[dst + ours + ".b"] = lin(src + theirs)
        q, k, v = (lin(src + "self_attn." + n + "_proj") for n in "qkv")
        out[dst + "qkv.w"] = np.concatenate([q[0], k[0], v[0]], axis=0)
        out[dst + "qkv.b"] = np.concatenate([q[1], k[1], v[1]], axis=0)
        if cross:
            out[dst + "lnx.w"], out[dst + "lnx.b"] = lin(src + "encoder_attn_layer_norm")
            out[dst + "xq.w"], out[dst + "xq.b"] = lin(src + "encoder_attn.q_proj")
            k, v = (lin(src + "encoder_attn." + n + "_proj") for n in "kv")
            out[dst + "xkv.w"] = np.concatenate([k[0], v[0]], axis=0)
            out[dst + "xkv.b"] = np.concatenate([k[1], v[1]], axis=0)
            out[dst + "xout.w"], out[dst + "xout.b"] = lin(src + "encoder_attn.out_proj")

    for i in range(cfg.encoder_layers):
        block(f"enc.{i}.", f"model.encoder.layers.{i}.", False)
    for i in range(cfg.decoder_layers):
        block(f"dec.{i}.", f"model.decoder.layers.{i}.", True)
    out["enc.ln.w"], out["enc.ln.b"] = lin("model.encoder.layer_norm")
    out["dec.ln.w"], out["dec.ln.b"] = lin("model.decoder.layer_norm")
    return out


def synth_state_dict(cfg: NllbConfig, seed: int = 0, eos_gain: float = 2.0) -> Dict[str, np.ndarray]:
    """Seeded random parameters under the ``transformers`` names (there is no checkpoint and no network where this is
    built and measured).  Linear weights ~ N(0, 1/fan_in), biases ~ N(0, 0.02^2), LayerNorm gains ~ 1 + N(0, 0.1^2),
    embedding ~ N(0, 0.04 / d_model) (small next to the layers' contributions: with a tied output projection a large
    embedding makes every random-weight generation repeat its last token); the ``</s>`` row is scaled by ``eos_gain`` so
    that generations end now and then."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, np.ndarray] = {}
    d, f = cfg.d_model, cfg.ffn_dim

    def normal(shape, std):
        return (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

    def linear(name, n_out, n_in):
        sd[name + ".weight"] = normal((n_out, n_in), 1.0 / np.sqrt(n_in))
        sd[name + ".bias"] = normal((n_out,), 0.02)

    def layernorm(name):
        sd[name + ".weight"] = (1.0 + normal((d,), 0.1)).astype(np.float32)
        sd[name + ".bias"] = normal((d,), 0.02)

    emb = normal((cfg.vocab_size, d), 0.2 / np.sqrt(d))
    emb[cfg.eos_token_id] *= np.float32(eos_gain)
    sd["model.shared.weight"] = emb
    for side, n, cross in (("encoder", cfg.encoder_layers, False), ("decoder", cfg.decoder_layers, True)):
        for i in range(n):
            p = f"model.{side}.layers.{i}."
            for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
                linear(p + "self_attn." + proj, d, d)
            layernorm(p + "self_attn_layer_norm")
            if cross:
                for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
                    linear(p + "encoder_attn." + proj, d, d)
                layernorm(p + "encoder_attn_layer_norm")
            linear(p + "fc1", f, d)
            linear(p + "fc2", d, f)
            layernorm(p + "final_layer_norm")
        layernorm(f"model.{side}.layer_norm")
    return sd


class HipNllbModel:
    """Packed weights of one NLLB / M2M-100 network on one GPU."""

    def __init__(self, cfg: NllbConfig, device: int = 0, max_src: int = 256, max_tgt: int = 256):
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = device
        n_pos = max(max_src, max_tgt) + cfg.pad_token_id + 2
        self.cdims = _lib.NllbDims(cfg.vocab_size, cfg.d_model, cfg.attention_heads, cfg.ffn_dim, cfg.encoder_layers,
                                   cfg.decoder_layers, max_src, max_tgt, cfg.pad_token_id, n_pos,
                                   math.sqrt(cfg.d_model) if cfg.scale_embedding else 1.0)
        self.n_positions = n_pos
        self._h = C.c_void_p()
        _lib.check(self.lib.wlk_nllb_create(C.byref(self.cdims), device, C.byref(self._h)))
        self.finalized = False

    def upload_packed(self, packed: Mapping[str, np.ndarray]) -> None:
        for name, arr in packed.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            _lib.check(self.lib.wlk_nllb_upload(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))
        _lib.check(self.lib.wlk_nllb_finalize(self._h))
        self.finalized = True

    @classmethod
    def from_hf_state_dict(cls, cfg: NllbConfig, sd: Mapping[str, object], device: int = 0, max_src: int = 256,
                           max_tgt: int = 256) -> "HipNllbModel":
        m = cls(cfg, device, max_src, max_tgt)
        m.upload_packed(pack_hf_state_dict(cfg, sd, m.n_positions))
        return m

    @classmethod
    def synthetic(cls, cfg: NllbConfig, seed: int = 0, device: int = 0, **kw) -> "HipNllbModel":
        return cls.from_hf_state_dict(cfg, synth_state_dict(cfg, seed), device, **kw)

    def new_session(self, rows: int = 1) -> "HipNllbSession":
        return HipNllbSession(self, rows)

    def close(self) -> None:
        if self._h:
            self.lib.wlk_nllb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class HipNllbSession:
    """Device state of one translation request: encoder output, cross K/V, the self-attention caches of ``rows`` hypotheses."""

    def __init__(self, model: HipNllbModel, rows: int = 1):
        if not model.finalized:
            raise _lib.WlkError("NLLB model must be finalized before creating sessions")
        self.model, self.lib, self.rows = model, model.lib, rows
        self._h = C.c_void_p()
        _lib.check(self.lib.wlk_nllb_session_create(model._h, rows, C.byref(self._h)))

    def encode(self, src_ids: Sequence[int]) -> None:
        a = np.ascontiguousarray(src_ids, dtype=np.int64).reshape(-1)
        _lib.check(self.lib.wlk_nllb_encode(self._h, a.ctypes.data_as(C.c_void_p), a.size))

    def decode(self, tokens, first: bool) -> None:
        t = np.ascontiguousarray(tokens, dtype=np.int64)
        if t.ndim != 2:
            raise ValueError("tokens must be [rows, n_tok]")
        _lib.check(self.lib.wlk_nllb_decode(self._h, t.ctypes.data_as(C.c_void_p), t.shape[0], t.shape[1], 1 if first else 0))

    def step(self, tokens: Sequence[int], k: int = 1) -> Tuple[np.ndarray, np.ndarray]:
        """One token per row after the prompt + the k best continuations of every row, one graph replay (wlk_nllb_step)."""
        t = np.ascontiguousarray(tokens, dtype=np.int64).reshape(-1)
        lp = np.empty((self.rows, k), np.float32)
        ids = np.empty((self.rows, k), np.int32)
        _lib.check(self.lib.wlk_nllb_step(self._h, t.ctypes.data_as(C.c_void_p), t.size, k, lp.ctypes.data_as(C.c_void_p),
                                          ids.ctypes.data_as(C.c_void_p)))
        return lp, ids

    def kv_reorder(self, source_rows: Sequence[int]) -> None:
        a = np.ascontiguousarray(source_rows, dtype=np.int32)
        _lib.check(self.lib.wlk_nllb_kv_reorder(self._h, a.ctypes.data_as(C.c_void_p), a.size))

    def topk(self, k: int) -> Tuple[np.ndarray, np.ndarray]:
        lp = np.empty((self.rows, k), np.float32)
        ids = np.empty((self.rows, k), np.int32)
        _lib.check(self.lib.wlk_nllb_topk(self._h, k, lp.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p)))
        return lp, ids

    def _export(self, what: str, n: int) -> np.ndarray:
        buf = np.empty(n, np.float32)
        got = C.c_uint64()
        _lib.check(self.lib.wlk_nllb_export(self._h, what.encode(), buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(got)))
        return buf[: got.value]

    def logits(self) -> np.ndarray:
        return self._export("logits", self.rows * self.model.cfg.vocab_size).reshape(self.rows, -1)

    def encoder_output(self) -> np.ndarray:
        return self._export("enc", self.model.cdims.max_src * self.model.cfg.d_model).reshape(-1, self.model.cfg.d_model)

    def sync(self) -> None:
        _lib.check(self.lib.wlk_nllb_sync(self._h))

    def close(self) -> None:
        if self._h:
            self.lib.wlk_nllb_session_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def generate(session, src_ids: Sequence[int], forced_bos_token_id: Optional[int] = None, *, max_new_tokens: int = 199,
             forced_eos_token_id: Optional[int] = None) -> List[int]:
    """``model.generate(input_ids, forced_bos_token_id=<target language>, num_beams=1, do_sample=False,
    max_new_tokens=n)`` of ``transformers`` for one sentence: starts from ``[decoder_start_token_id]``, forces the
    target-language token as the first generated token (ForcedBOSTokenLogitsProcessor) and, when ``forced_eos_token_id``
    is given, ``</s>`` as the last allowed one (ForcedEOSTokenLogitsProcessor); stops at ``</s>``.  Returns the ids
    including the start token, as ``generate`` does.  The arg-max runs on the device (``wlk_nllb_topk``): one read-back
    of 8 bytes per token.  Beam search: :func:`beam_search`."""
    cfg = session.model.cfg
    if session.rows != 1:
        raise ValueError("generate: greedy decoding needs a 1-row session")
    eos, start = cfg.eos_token_id, cfg.decoder_start_token_id
    session.encode(src_ids)
    max_length = 1 + max_new_tokens
    out = [start]
    for step in range(max_new_tokens):
        if step == 0:
            session.decode(np.asarray([out], np.int64), first=True)
            best = None
        else:
            best = int(session.step(out[-1:], 1)[1][0, 0])
        cur_len = len(out)
        if forced_bos_token_id is not None and cur_len == 1:
            nxt = int(forced_bos_token_id)
        elif forced_eos_token_id is not None and cur_len == max_length - 1:
            nxt = int(forced_eos_token_id)
        else:
            nxt = best if best is not None else int(session.topk(1)[1][0, 0])
        out.append(nxt)
        if nxt == eos:
            break
    return out


def beam_search(session, src_ids: Sequence[int], forced_bos_token_id: Optional[int] = None, *, num_beams: int,
                max_new_tokens: int = 199, length_penalty: float = 1.0, early_stopping=False,
                forced_eos_token_id: Optional[int] = None) -> List[int]:
    """``model.generate(input_ids, forced_bos_token_id=..., num_beams=n, do_sample=False, max_new_tokens=...)`` of
    ``transformers`` 5.x for one sentence (``GenerationMixin._beam_search``, generation/utils.py): per step the 2 n best
    (beam, token) continuations by accumulated log-probability; those among the n best that end (``</s>`` or the length
    limit) compete, with score / generated_length ** length_penalty, for the n finished slots; the n best others run on;
    stop when no running beam can beat the worst finished one (``early_stopping=False``: judged at the current length;
    ``"never"``: at the maximum length when the penalty favours long outputs; ``True``: as soon as n have finished).
    Returns the best finished sequence including the start token.  The device supplies the 2 n best log-probabilities of
    every beam row (``topk``) and reorders the caches (``kv_reorder``); the bookkeeping here is float32 like the original."""
    cfg = session.model.cfg
    n, K = int(num_beams), 2 * int(num_beams)
    if n != session.rows:
        raise ValueError(f"beam_search: the session has {session.rows} rows, num_beams = {n}")
    if n > 8:
        raise ValueError("beam_search: at most 8 beams (rows of a session)")
    f32 = np.float32
    V, eos, start, pad = cfg.vocab_size, cfg.eos_token_id, cfg.decoder_start_token_id, cfg.pad_token_id
    prompt_len, cur_len = 1, 1
    max_length = prompt_len + max_new_tokens
    NEG = f32(-1.0e9)
    run_seq = np.full((n, max_length), pad, np.int64)
    run_seq[:, 0] = start
    run_sc = np.full(n, NEG, f32)
    run_sc[0] = 0.0                                  # identical beams at the start: only the first one counts
    fin_seq, fin_sc = run_seq.copy(), np.full(n, NEG, f32)
    fin_done, fin_len = np.zeros(n, bool), np.full(n, prompt_len)
    unsatisfied = True
    session.encode(src_ids)
    while True:
        if K <= 8:                                    # the device's top-k takes up to 8 per row
            if cur_len == prompt_len:
                session.decode(run_seq[:, :cur_len], first=True)
                top_lp, top_id = session.topk(K)
            else:
                top_lp, top_id = session.step(run_seq[:, cur_len - 1], K)
        else:                                         # 5 .. 8 beams: the rows' logits come back and are cut here
            session.decode(run_seq[:, :cur_len] if cur_len == prompt_len else run_seq[:, cur_len - 1:cur_len],
                           first=(cur_len == prompt_len))
            lp_rows = session.logits().astype(f32)
            lp_rows = lp_rows - _logsumexp(lp_rows)
            top_id = np.argpartition(-lp_rows, K - 1, axis=-1)[:, :K]
            top_lp = np.take_along_axis(lp_rows, top_id, axis=-1)
        forced = None
        if forced_bos_token_id is not None and cur_len == 1:
            forced = int(forced_bos_token_id)
        elif forced_eos_token_id is not None and cur_len == max_length - 1:
            forced = int(forced_eos_token_id)
        if forced is not None:                       # Forced*TokenLogitsProcessor: 0 for the token, -inf for the rest
            top_lp = np.full((n, K), -np.inf, f32)
            top_lp[:, 0] = 0.0
            top_id = np.tile(np.arange(K, dtype=np.int64), (n, 1))
            top_id[top_id == forced] = K               # the filler ids only have to differ from the forced one
            top_id[:, 0] = forced
        total = (top_lp.astype(f32) + run_sc[:, None]).reshape(-1)
        flat = (np.arange(n)[:, None] * V + top_id.astype(np.int64)).reshape(-1)
        order = np.lexsort((flat, -total))[:K]         # descending score, ties by the flattened (beam, token) index
        cand_sc, cand_b, cand_tok = total[order], flat[order] // V, flat[order] % V
        cand_seq = run_seq[cand_b].copy()
        cand_seq[:, cur_len] = cand_tok
        hits = (cand_tok == eos) | (cur_len + 1 >= max_length)
        # the n best continuations that go on
        going = cand_sc + hits.astype(f32) * NEG
        keep = np.argsort(-going, kind="stable")[:n]
        # the finished slots: candidates among the n best that just ended, against what is already there
        ended = hits & (np.arange(K) < n)
        score = (cand_sc / f32((cur_len + 1 - prompt_len) ** length_penalty)).astype(f32)
        score = score + f32(bool(fin_done.all()) and early_stopping is True) * NEG
        score = score + f32(not unsatisfied) * NEG
        score = score + (~ended).astype(f32) * NEG
        all_sc = np.concatenate([fin_sc, score])
        best = np.argsort(-all_sc, kind="stable")[:n]
        fin_seq = np.concatenate([fin_seq, cand_seq])[best]
        fin_done = np.concatenate([fin_done, ended])[best]
        fin_len = np.concatenate([fin_len, np.full(K, cur_len + 1)])[best]
        fin_sc = all_sc[best]
        run_seq, run_sc = cand_seq[keep], going[keep]
        session.kv_reorder(cand_b[keep])
        cur_len += 1
        horizon = (max_length - prompt_len) if (early_stopping == "never" and length_penalty > 0.0) else (cur_len - prompt_len)
        best_running = run_sc[0] / f32(horizon ** length_penalty)
        worst_finished = np.where(fin_done, fin_sc.min(), NEG)
        unsatisfied = unsatisfied and bool((best_running > worst_finished).any())
        if not (unsatisfied and not (bool(fin_done.all()) and early_stopping is True) and not bool(hits.all())):
            break
    return fin_seq[0, :fin_len[0]].tolist()


def _logsumexp(x: np.ndarray) -> np.ndarray:
    m = x.max(axis=-1, keepdims=True)
    return m + np.log(np.exp(x - m).sum(axis=-1, keepdims=True, dtype=np.float32))

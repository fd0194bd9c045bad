"""The streaming Sortformer diarizer network on the HIP backend (model side).

What the reference obtains from NeMo (``SortformerEncLabelModel``; loaded and configured at
whisperlivekit/diarization/sortformer_backend.py:86-131, driven one chunk at a time at :293-300):

* device part (``wlk_sf_*`` in include/wlk_hip.h): ConvSubsampling(dw_striding x8) -> 17 Conformer blocks with
  relative-position attention -> Linear 512->192 -> 18 post-LN Transformer blocks -> sigmoid speaker head;
* host part (this file, numpy): the streaming state - speaker cache, FIFO, silence profile - and its update rule
  (NeMo ``SortformerModules.streaming_update_async`` / ``_compress_spkcache``); a few hundred scalars per chunk.

``HipSortformerModel`` implements ``diarization.SortformerBackend``; weights arrive as a NeMo-named state dict
(``load_nemo_checkpoint``) or are synthesised for tests/benchmarks.  NeMo is not in the reference tree: the
arithmetic here restates NeMo's published modules.  Pinned since round 5 by transformers' independent ports
(ParakeetFeatureExtractor, ParakeetEncoder: tests/golden/sortformer_hf_kat.npz): log-mel, sub-sampling stem, the 17
Conformer blocks; still unpinned (no NeMo, no checkpoint offline): Transformer-block wiring, sigmoid head, cache update.
There is no CPU fallback: without libwlk_hip.so / a GPU the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

from . import _lib
from .diarization import HipMelSpectrogram, SortformerStreamingParams


@dataclass(frozen=True)
class SortformerDims:
    """diar_streaming_sortformer_4spk-v2 geometry (NeMo model config: encoder / transformer_encoder / sortformer_modules)."""
    n_mels: int = 128
    sub_channels: int = 256
    fc_d_model: int = 512
    fc_layers: int = 17
    fc_heads: int = 8
    fc_ff: int = 2048
    conv_kernel: int = 9
    tf_d_model: int = 192
    tf_layers: int = 18
    tf_heads: int = 8
    tf_inner: int = 768
    n_spk: int = 4
    xscaling: bool = True

    @property
    def freq_out(self) -> int:
        f = self.n_mels
        for _ in range(3):
            f = (f - 1) // 2 + 1
        return f


@dataclass(frozen=True)
class SpkCacheParams:
    """SortformerModules hyper-parameters of the speaker-cache update (NeMo defaults) with the three values the
    reference overrides (sortformer_backend.py:120-126).  ``subsampling_factor`` is the reference's 10, which it
    sets on the module and which NeMo uses to turn the 8/8 frame offsets into 1/1 embedding offsets."""
    spkcache_len: int = 188
    fifo_len: int = 188
    spkcache_update_period: int = 144
    subsampling_factor: int = 10
    spkcache_sil_frames_per_spk: int = 3
    pred_score_threshold: float = 0.25
    scores_boost_latest: float = 0.05
    sil_threshold: float = 0.2
    strong_boost_rate: float = 0.75
    weak_boost_rate: float = 1.5
    min_pos_scores_rate: float = 0.5
    max_index: int = 99999


@dataclass
class SortformerState:
    """StreamingSortformerState for one stream (sortformer_backend.py:212-234): fixed-size zero buffers + lengths."""
    spkcache: np.ndarray
    spkcache_preds: np.ndarray
    fifo: np.ndarray
    fifo_preds: np.ndarray
    mean_sil_emb: np.ndarray
    spkcache_len: int = 0
    fifo_len: int = 0
    n_sil_frames: int = 0


# ---- weights ------------------------------------------------------------------------------------------
def rel_positional_table(length: int, d_model: int) -> np.ndarray:
    """RelPositionalEncoding table for ``length`` frames: positions length-1 ... -(length-1), sin/cos interleaved,
    computed in fp32 like the module's buffer; [2*length-1, d_model]."""
    pos = np.arange(length - 1, -length, -1, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, d_model, 2, dtype=np.float32) * np.float32(-(math.log(10000.0) / d_model))).astype(np.float32)
    pe = np.zeros((2 * length - 1, d_model), np.float32)
    arg = (pos * div[None, :]).astype(np.float32)
    pe[:, 0::2] = np.sin(arg)
    pe[:, 1::2] = np.cos(arg)
    return pe


def synth_sortformer_state_dict(dims: SortformerDims, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded random weights under NeMo's parameter names (there is no checkpoint offline).  Scales keep the
    activations O(1) through 35 blocks so the speaker activities are neither saturated nor constant."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, np.ndarray] = {}

    def lin(name, n_out, n_in, gain=1.0, bias=True):
        sd[name + ".weight"] = (rng.standard_normal((n_out, n_in)) * (gain / math.sqrt(n_in))).astype(np.float32)
        if bias:
            sd[name + ".bias"] = (rng.standard_normal(n_out) * 0.05).astype(np.float32)

    def norm(name, n):
        sd[name + ".weight"] = (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32)
        sd[name + ".bias"] = (0.05 * rng.standard_normal(n)).astype(np.float32)

    C_, d, ff = dims.sub_channels, dims.fc_d_model, dims.fc_ff
    p = "encoder.pre_encode."
    sd[p + "conv.0.weight"] = (rng.standard_normal((C_, 1, 3, 3)) / 3.0).astype(np.float32)
    sd[p + "conv.0.bias"] = (rng.standard_normal(C_) * 0.05).astype(np.float32)
    for dw, pw in ((2, 3), (5, 6)):
        sd[p + f"conv.{dw}.weight"] = (rng.standard_normal((C_, 1, 3, 3)) / 3.0).astype(np.float32)
        sd[p + f"conv.{dw}.bias"] = (rng.standard_normal(C_) * 0.05).astype(np.float32)
        sd[p + f"conv.{pw}.weight"] = (rng.standard_normal((C_, C_, 1, 1)) / math.sqrt(C_)).astype(np.float32)
        sd[p + f"conv.{pw}.bias"] = (rng.standard_normal(C_) * 0.05).astype(np.float32)
    lin(p + "out", d, C_ * dims.freq_out, gain=0.05)
    dk = d // dims.fc_heads
    for i in range(dims.fc_layers):
        q = f"encoder.layers.{i}."
        for n in ("norm_feed_forward1", "norm_self_att", "norm_conv", "norm_feed_forward2", "norm_out"):
            norm(q + n, d)
        for n in ("feed_forward1", "feed_forward2"):
            lin(q + n + ".linear1", ff, d)
            lin(q + n + ".linear2", d, ff, gain=0.5)
        for n in ("linear_q", "linear_k", "linear_v"):
            lin(q + "self_attn." + n, d, d)
        lin(q + "self_attn.linear_out", d, d, gain=0.5)
        lin(q + "self_attn.linear_pos", d, d, bias=False)
        sd[q + "self_attn.pos_bias_u"] = (0.1 * rng.standard_normal((dims.fc_heads, dk))).astype(np.float32)
        sd[q + "self_attn.pos_bias_v"] = (0.1 * rng.standard_normal((dims.fc_heads, dk))).astype(np.float32)
        sd[q + "conv.pointwise_conv1.weight"] = (rng.standard_normal((2 * d, d, 1)) / math.sqrt(d)).astype(np.float32)
        sd[q + "conv.pointwise_conv1.bias"] = (rng.standard_normal(2 * d) * 0.05).astype(np.float32)
        sd[q + "conv.depthwise_conv.weight"] = (rng.standard_normal((d, 1, dims.conv_kernel)) / 3.0).astype(np.float32)
        sd[q + "conv.depthwise_conv.bias"] = (rng.standard_normal(d) * 0.05).astype(np.float32)
        sd[q + "conv.batch_norm.weight"] = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32)
        sd[q + "conv.batch_norm.bias"] = (0.05 * rng.standard_normal(d)).astype(np.float32)
        sd[q + "conv.batch_norm.running_mean"] = (0.05 * rng.standard_normal(d)).astype(np.float32)
        sd[q + "conv.batch_norm.running_var"] = (0.5 + rng.random(d)).astype(np.float32)
        sd[q + "conv.pointwise_conv2.weight"] = (rng.standard_normal((d, d, 1)) * (0.5 / math.sqrt(d))).astype(np.float32)
        sd[q + "conv.pointwise_conv2.bias"] = (rng.standard_normal(d) * 0.05).astype(np.float32)
    dt = dims.tf_d_model
    lin("sortformer_modules.encoder_proj", dt, d)
    for i in range(dims.tf_layers):
        q = f"transformer_encoder.layers.{i}."
        for n in ("query_net", "key_net", "value_net"):
            lin(q + "first_sub_layer." + n, dt, dt, gain=1.5)
        lin(q + "first_sub_layer.out_projection", dt, dt)
        norm(q + "layer_norm_1", dt)
        lin(q + "second_sub_layer.dense_in", dims.tf_inner, dt)
        lin(q + "second_sub_layer.dense_out", dt, dims.tf_inner)
        norm(q + "layer_norm_2", dt)
    lin("sortformer_modules.first_hidden_to_hidden", dt, dt, gain=1.5)
    lin("sortformer_modules.single_hidden_to_spks", dims.n_spk, dt, gain=2.0)
    return sd


def load_nemo_checkpoint(path: str) -> Dict[str, np.ndarray]:
    """State dict of a ``.nemo`` archive (tar with model_weights.ckpt) or a bare torch checkpoint, as numpy fp32."""
    import io
    import tarfile

    import torch

    if path.endswith(".nemo"):
        with tarfile.open(path) as tar:
            member = next(m for m in tar.getmembers() if m.name.endswith("model_weights.ckpt"))
            blob = tar.extractfile(member).read()
        ckpt = torch.load(io.BytesIO(blob), map_location="cpu")
    else:
        ckpt = torch.load(path, map_location="cpu")
    ckpt = ckpt.get("state_dict", ckpt)
    return {k: v.detach().to(torch.float32).numpy() for k, v in ckpt.items() if hasattr(v, "detach") and v.dtype.is_floating_point}


def dims_from_state_dict(sd: Dict[str, np.ndarray], fc_heads: int = 8, tf_heads: int = 8) -> SortformerDims:
    """Geometry implied by the tensor shapes (head counts are not recoverable from shapes, except fc via pos_bias_u)."""
    import re
    fc_layers = 1 + max(int(m.group(1)) for k in sd if (m := re.match(r"encoder\.layers\.(\d+)\.", k)))
    tf_layers = 1 + max(int(m.group(1)) for k in sd if (m := re.match(r"transformer_encoder\.layers\.(\d+)\.", k)))
    c = sd["encoder.pre_encode.conv.0.weight"].shape[0]
    d = sd["encoder.pre_encode.out.weight"].shape[0]
    f3 = sd["encoder.pre_encode.out.weight"].shape[1] // c
    n_mels = {f: m for m in (64, 80, 128) for f in [SortformerDims(n_mels=m).freq_out]}.get(f3, f3 * 8)
    if "encoder.layers.0.self_attn.pos_bias_u" in sd:
        fc_heads = sd["encoder.layers.0.self_attn.pos_bias_u"].shape[0]
    return SortformerDims(
        n_mels=n_mels, sub_channels=c, fc_d_model=d, fc_layers=fc_layers, fc_heads=fc_heads,
        fc_ff=sd["encoder.layers.0.feed_forward1.linear1.weight"].shape[0],
        conv_kernel=sd["encoder.layers.0.conv.depthwise_conv.weight"].shape[-1],
        tf_d_model=sd["sortformer_modules.encoder_proj.weight"].shape[0], tf_layers=tf_layers, tf_heads=tf_heads,
        tf_inner=sd["transformer_encoder.layers.0.second_sub_layer.dense_in.weight"].shape[0],
        n_spk=sd["sortformer_modules.single_hidden_to_spks.weight"].shape[0])


def pack_sortformer_state_dict(dims: SortformerDims, sd: Dict[str, np.ndarray], max_frames: int) -> Dict[str, np.ndarray]:
    """NeMo names -> the packed tensors of ``wlk_sf_tensor_name``.  Re-packings (all value-preserving):
    conv weights tap-major / channels-last, q|k|v concatenated, the Linear after the sub-sampling stem re-ordered
    from NeMo's (channel, freq) feature order to the kernel's (freq, channel), the Macaron 0.5 folded into the
    second feed-forward Linear (exact: a power of two), BatchNorm's 1/sqrt(var + eps) precomputed."""
    f32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    C_, d, F3 = dims.sub_channels, dims.fc_d_model, dims.freq_out
    out: Dict[str, np.ndarray] = {}
    p = "encoder.pre_encode."
    out["pre.conv0.w"] = f32(sd[p + "conv.0.weight"]).reshape(C_, 9)
    out["pre.conv0.b"] = f32(sd[p + "conv.0.bias"])
    for i, (dw, pw) in enumerate(((2, 3), (5, 6)), start=1):
        out[f"pre.dw{i}.w"] = f32(f32(sd[p + f"conv.{dw}.weight"]).reshape(C_, 9).T)
        out[f"pre.dw{i}.b"] = f32(sd[p + f"conv.{dw}.bias"])
        out[f"pre.pw{i}.w"] = f32(sd[p + f"conv.{pw}.weight"]).reshape(C_, C_)
        out[f"pre.pw{i}.b"] = f32(sd[p + f"conv.{pw}.bias"])
    w_out = f32(sd[p + "out.weight"]).reshape(d, C_, F3)
    out["pre.out.w"] = f32(w_out.transpose(0, 2, 1)).reshape(d, F3 * C_)
    out["pre.out.b"] = f32(sd[p + "out.bias"])
    out["pos.table"] = rel_positional_table(max_frames, d)
    half = np.float32(0.5)
    for i in range(dims.fc_layers):
        q, o = f"encoder.layers.{i}.", f"fc.{i}."
        for src, dst in (("norm_feed_forward1", "ln_ff1"), ("norm_self_att", "ln_att"), ("norm_conv", "ln_conv"),
                         ("norm_feed_forward2", "ln_ff2"), ("norm_out", "ln_out")):
            out[o + dst + ".w"], out[o + dst + ".b"] = f32(sd[q + src + ".weight"]), f32(sd[q + src + ".bias"])
        for src, dst in (("feed_forward1", "ff1"), ("feed_forward2", "ff2")):
            out[o + dst + "a.w"], out[o + dst + "a.b"] = f32(sd[q + src + ".linear1.weight"]), f32(sd[q + src + ".linear1.bias"])
            out[o + dst + "b.w"] = f32(sd[q + src + ".linear2.weight"]) * half
            out[o + dst + "b.b"] = f32(sd[q + src + ".linear2.bias"]) * half
        a = q + "self_attn."
        out[o + "qkv.w"] = np.concatenate([f32(sd[a + f"linear_{n}.weight"]) for n in "qkv"], axis=0)
        out[o + "qkv.b"] = np.concatenate([f32(sd[a + f"linear_{n}.bias"]) for n in "qkv"], axis=0)
        out[o + "pos.w"] = f32(sd[a + "linear_pos.weight"])
        out[o + "bias_u"] = f32(sd[a + "pos_bias_u"]).reshape(-1)
        out[o + "bias_v"] = f32(sd[a + "pos_bias_v"]).reshape(-1)
        out[o + "out.w"], out[o + "out.b"] = f32(sd[a + "linear_out.weight"]), f32(sd[a + "linear_out.bias"])
        c = q + "conv."
        out[o + "pw1.w"] = f32(sd[c + "pointwise_conv1.weight"]).reshape(2 * d, d)
        out[o + "pw1.b"] = f32(sd[c + "pointwise_conv1.bias"])
        out[o + "dw.w"] = f32(f32(sd[c + "depthwise_conv.weight"]).reshape(d, dims.conv_kernel).T)
        out[o + "dw.b"] = f32(sd[c + "depthwise_conv.bias"])
        out[o + "bn.mean"] = f32(sd[c + "batch_norm.running_mean"])
        out[o + "bn.invstd"] = (np.float32(1.0) / np.sqrt(f32(sd[c + "batch_norm.running_var"]) + np.float32(1e-5))).astype(np.float32)
        out[o + "bn.w"], out[o + "bn.b"] = f32(sd[c + "batch_norm.weight"]), f32(sd[c + "batch_norm.bias"])
        out[o + "pw2.w"] = f32(sd[c + "pointwise_conv2.weight"]).reshape(d, d)
        out[o + "pw2.b"] = f32(sd[c + "pointwise_conv2.bias"])
    m = "sortformer_modules."
    out["proj.w"], out["proj.b"] = f32(sd[m + "encoder_proj.weight"]), f32(sd[m + "encoder_proj.bias"])
    for i in range(dims.tf_layers):
        q, o = f"transformer_encoder.layers.{i}.", f"tf.{i}."
        a = q + "first_sub_layer."
        out[o + "qkv.w"] = np.concatenate([f32(sd[a + n + ".weight"]) for n in ("query_net", "key_net", "value_net")], axis=0)
        out[o + "qkv.b"] = np.concatenate([f32(sd[a + n + ".bias"]) for n in ("query_net", "key_net", "value_net")], axis=0)
        out[o + "out.w"], out[o + "out.b"] = f32(sd[a + "out_projection.weight"]), f32(sd[a + "out_projection.bias"])
        out[o + "ln1.w"], out[o + "ln1.b"] = f32(sd[q + "layer_norm_1.weight"]), f32(sd[q + "layer_norm_1.bias"])
        s = q + "second_sub_layer."
        out[o + "in.w"], out[o + "in.b"] = f32(sd[s + "dense_in.weight"]), f32(sd[s + "dense_in.bias"])
        out[o + "outd.w"], out[o + "outd.b"] = f32(sd[s + "dense_out.weight"]), f32(sd[s + "dense_out.bias"])
        out[o + "ln2.w"], out[o + "ln2.b"] = f32(sd[q + "layer_norm_2.weight"]), f32(sd[q + "layer_norm_2.bias"])
    out["head.h.w"], out["head.h.b"] = f32(sd[m + "first_hidden_to_hidden.weight"]), f32(sd[m + "first_hidden_to_hidden.bias"])
    out["head.s.w"], out["head.s.b"] = f32(sd[m + "single_hidden_to_spks.weight"]), f32(sd[m + "single_hidden_to_spks.bias"])
    return out


# ---- streaming state update (host, numpy) ------------------------------------------------------------------
def _topk_desc(values: np.ndarray, k: int) -> np.ndarray:
    """Indices of the k largest entries (ties: lowest index first, like a stable descending sort)."""
    return np.argsort(-values, kind="stable")[:k]


def compress_spkcache(sp: SpkCacheParams, emb_seq: np.ndarray, preds: np.ndarray, mean_sil_emb: np.ndarray):
    """SortformerModules._compress_spkcache for one stream: keep the ``spkcache_len`` most informative frames.
    Per-speaker scores = log-odds of the speaker plus the log-probability that nobody else speaks
    (_get_log_pred_scores); frames where the speaker is inactive, and non-positive scores once a speaker has enough
    positive ones, are disabled (_disable_low_scores); the newest frames get a small boost; the best frames per
    speaker are boosted in two tiers so every speaker keeps a share (_boost_topk_scores); a few +inf "silence" slots
    per speaker are appended (they resolve to the mean silence embedding); the global top-k in (speaker, time)
    order selects the frames (_get_topk_indices, _gather_spkcache_and_preds)."""
    n_frames, n_spk = preds.shape
    per_spk = sp.spkcache_len // n_spk - sp.spkcache_sil_frames_per_spk
    strong, weak = math.floor(per_spk * sp.strong_boost_rate), math.floor(per_spk * sp.weak_boost_rate)
    min_pos = math.floor(per_spk * sp.min_pos_scores_rate)
    thr = np.float32(sp.pred_score_threshold)
    log_p = np.log(np.maximum(preds, thr))
    log_1p = np.log(np.maximum(np.float32(1.0) - preds, thr))
    scores = (log_p - log_1p + log_1p.sum(axis=1, keepdims=True) - np.float32(math.log(0.5))).astype(np.float32)
    is_speech = preds > 0.5
    scores = np.where(is_speech, scores, -np.inf).astype(np.float32)
    is_pos = scores > 0
    replace = (~is_pos) & is_speech & (is_pos.sum(axis=0, keepdims=True) >= min_pos)
    scores = np.where(replace, -np.inf, scores).astype(np.float32)
    if sp.scores_boost_latest > 0:
        scores[sp.spkcache_len:, :] += np.float32(sp.scores_boost_latest)
    for n_boost, factor in ((strong, 2.0), (weak, 1.0)):
        for s in range(n_spk):
            idx = _topk_desc(scores[:, s], min(n_boost, n_frames))
            scores[idx, s] -= np.float32(factor * math.log(0.5))
    if sp.spkcache_sil_frames_per_spk > 0:
        scores = np.concatenate([scores, np.full((sp.spkcache_sil_frames_per_spk, n_spk), np.inf, np.float32)], axis=0)
    n_total = scores.shape[0]
    flat = scores.T.reshape(-1)
    idx = _topk_desc(flat, sp.spkcache_len)
    idx = np.where(flat[idx] != -np.inf, idx, sp.max_index)
    idx = np.sort(idx)
    disabled = idx == sp.max_index
    idx = np.remainder(idx, n_total)
    disabled = disabled | (idx >= n_frames)
    idx = np.where(disabled, 0, idx)
    emb = np.where(disabled[:, None], mean_sil_emb[None, :], emb_seq[idx]).astype(np.float32)
    pr = np.where(disabled[:, None], np.float32(0), preds[idx]).astype(np.float32)
    return emb, pr


def streaming_update(sp: SpkCacheParams, st: SortformerState, chunk: np.ndarray, preds: np.ndarray, lc: int, rc: int) -> np.ndarray:
    """SortformerModules.streaming_update_async for one stream.  ``chunk`` [Tc, d] are this step's pre-encode
    embeddings (with lc / rc context rows), ``preds`` the activities of [spkcache | fifo | chunk].  The chunk's own
    rows join the FIFO; when the FIFO overflows its oldest ``spkcache_update_period`` rows move to the speaker
    cache (updating the silence profile), which is compressed back to ``spkcache_len`` rows when it overflows.
    Returns the chunk's activities [Tc - lc - rc, n_spk]."""
    n_spk, d = preds.shape[1], chunk.shape[1]
    max_chunk = chunk.shape[0] - lc - rc
    chunk_len = max(0, min(chunk.shape[0] - lc, max_chunk))
    s_len, f_len = st.spkcache_len, st.fifo_len
    fifo_preds = np.zeros_like(st.fifo_preds)
    fifo_preds[:f_len] = preds[s_len: s_len + f_len]
    chunk_preds = np.zeros((max_chunk, n_spk), np.float32)
    chunk_preds[:chunk_len] = preds[s_len + f_len + lc: s_len + f_len + lc + chunk_len]
    up_fifo = np.zeros((sp.fifo_len + max_chunk, d), np.float32)
    up_fifo_p = np.zeros((sp.fifo_len + max_chunk, n_spk), np.float32)
    pop_max = min(max(sp.spkcache_update_period, max_chunk), max_chunk + sp.fifo_len)
    up_cache = np.zeros((sp.spkcache_len + pop_max, d), np.float32)
    up_cache_p = np.zeros((sp.spkcache_len + pop_max, n_spk), np.float32)
    up_cache[:s_len] = st.spkcache[:s_len]
    up_cache_p[:s_len] = st.spkcache_preds[:s_len]
    up_fifo[:f_len] = st.fifo[:f_len]
    up_fifo_p[:f_len] = fifo_preds[:f_len]
    up_fifo[f_len: f_len + chunk_len] = chunk[lc: lc + chunk_len]
    up_fifo_p[f_len: f_len + chunk_len] = chunk_preds[:chunk_len]
    st.fifo_len = f_len + chunk_len
    if f_len + chunk_len > sp.fifo_len:
        pop = min(max(sp.spkcache_update_period, max_chunk - sp.fifo_len + f_len), f_len + chunk_len)
        st.spkcache_len = s_len + pop
        pop_e, pop_p = up_fifo[:pop], up_fifo_p[:pop]
        is_sil = pop_p.sum(axis=1) < np.float32(sp.sil_threshold)
        n_sil = int(is_sil.sum())
        if n_sil > 0:
            total = st.mean_sil_emb * np.float32(st.n_sil_frames) + (pop_e * is_sil[:, None]).sum(axis=0, dtype=np.float32)
            st.n_sil_frames += n_sil
            st.mean_sil_emb = (total / np.float32(max(st.n_sil_frames, 1))).astype(np.float32)
        up_cache[s_len: s_len + pop] = pop_e
        up_cache_p[s_len: s_len + pop] = pop_p
        st.fifo_len -= pop
        up_fifo[: st.fifo_len] = up_fifo[pop: pop + st.fifo_len].copy()
        up_fifo_p[: st.fifo_len] = up_fifo_p[pop: pop + st.fifo_len].copy()
        up_fifo[st.fifo_len:] = 0
        up_fifo_p[st.fifo_len:] = 0
    st.fifo, st.fifo_preds = up_fifo[: sp.fifo_len].copy(), up_fifo_p[: sp.fifo_len].copy()
    if st.spkcache_len > sp.spkcache_len:
        st.spkcache, st.spkcache_preds = compress_spkcache(sp, up_cache, up_cache_p, st.mean_sil_emb)
        st.spkcache_len = sp.spkcache_len
    else:
        st.spkcache, st.spkcache_preds = up_cache[: sp.spkcache_len].copy(), up_cache_p[: sp.spkcache_len].copy()
    return chunk_preds


# ---- the model handle --------------------------------------------------------------------------------------
class HipSortformerModel:
    """Shared (per GPU) Sortformer network + feature extractor; implements ``diarization.SortformerBackend``.
    Sessions keep their ``SortformerState`` on the host, so any number of streams share one handle; steps of sessions
    that wait for the model at the same time run as one stacked launch chain inside the library (bit-identical per
    session to a step alone)."""

    def __init__(self, dims: SortformerDims, state_dict: Dict[str, np.ndarray], device: int = 0,
                 params: SortformerStreamingParams = SortformerStreamingParams(),
                 cache: SpkCacheParams = SpkCacheParams(), max_feat_frames: int = 256):
        self.lib = _lib.load()
        if _lib.device_count() <= 0:
            raise _lib.WlkError("no HIP device visible: the Sortformer HIP backend has no CPU fallback")
        self.dims, self.params, self.cache, self.device = dims, params, cache, device
        self.n_spk = dims.n_spk
        sub = lambda n: (((n - 1) // 2 + 1 - 1) // 2 + 1 - 1) // 2 + 1
        self.max_frames = cache.spkcache_len + cache.fifo_len + sub(max_feat_frames)
        self.max_feat_frames = max_feat_frames
        self._cd = _lib.SfDims(dims.n_mels, dims.sub_channels, dims.fc_d_model, dims.fc_layers, dims.fc_heads, dims.fc_ff,
                               dims.conv_kernel, dims.tf_d_model, dims.tf_layers, dims.tf_heads, dims.tf_inner,
                               dims.n_spk, self.max_frames, max_feat_frames,
                               math.sqrt(dims.fc_d_model) if dims.xscaling else 1.0)
        self._h = C.c_void_p()
        _lib.check(self.lib.wlk_sf_create(C.byref(self._cd), device, C.byref(self._h)))
        packed = pack_sortformer_state_dict(dims, state_dict, self.max_frames)
        for name in packed_sortformer_names(self._cd):
            if name not in packed:
                raise KeyError(f"state dict does not provide packed tensor {name}")
            a = np.ascontiguousarray(packed[name], dtype=np.float32).reshape(-1)
            _lib.check(self.lib.wlk_sf_upload(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))
        _lib.check(self.lib.wlk_sf_finalize(self._h))
        self._mel: Optional[HipMelSpectrogram] = None

    @classmethod
    def synthetic(cls, dims: SortformerDims = SortformerDims(), seed: int = 0, **kw) -> "HipSortformerModel":
        return cls(dims, synth_sortformer_state_dict(dims, seed), **kw)

    @classmethod
    def from_checkpoint(cls, path: str, **kw) -> "HipSortformerModel":
        sd = load_nemo_checkpoint(path)
        return cls(dims_from_state_dict(sd), sd, **kw)

    # -- SortformerBackend -----------------------------------------------------------------------------------
    def features(self, pcm: np.ndarray) -> np.ndarray:
        if self._mel is None:
            self._mel = HipMelSpectrogram(device=self.device, n_mels=self.dims.n_mels)
        return self._mel(pcm)

    def new_state(self) -> SortformerState:
        d, c = self.dims.fc_d_model, self.cache
        return SortformerState(np.zeros((c.spkcache_len, d), np.float32), np.zeros((c.spkcache_len, self.n_spk), np.float32),
                               np.zeros((c.fifo_len, d), np.float32), np.zeros((c.fifo_len, self.n_spk), np.float32),
                               np.zeros(d, np.float32))

    def step(self, feats: Optional[np.ndarray], ctx_embs: Optional[np.ndarray]):
        """Device part of one step: (chunk embeddings [Tc, d], activities [n_ctx + Tc, n_spk])."""
        d = self.dims.fc_d_model
        n_feat = 0 if feats is None else int(feats.shape[0])
        n_ctx = 0 if ctx_embs is None else int(ctx_embs.shape[0])
        f = np.ascontiguousarray(feats, np.float32) if n_feat else None
        e = np.ascontiguousarray(ctx_embs, np.float32) if n_ctx else None
        cap_c = n_feat // 8 + 2
        chunk = np.empty((cap_c, d), np.float32)
        preds = np.empty((n_ctx + cap_c, self.n_spk), np.float32)
        n_chunk = C.c_int()
        _lib.check(self.lib.wlk_sf_step(
            self._h, f.ctypes.data_as(C.c_void_p) if n_feat else None, n_feat,
            e.ctypes.data_as(C.c_void_p) if n_ctx else None, n_ctx, chunk.ctypes.data_as(C.c_void_p), cap_c,
            C.byref(n_chunk), preds.ctypes.data_as(C.c_void_p), preds.shape[0]))
        return chunk[: n_chunk.value], preds[: n_ctx + n_chunk.value]

    def step_pcm(self, pcm: np.ndarray, prev_feats: Optional[np.ndarray], ctx_embs: Optional[np.ndarray]):
        """``step`` with the front end inside the launch chain (wlk_sf_step_pcm): the chunk's log-mel rows are computed behind
        ``prev_feats`` on the step's own stream.  -> (new feature rows [n, n_mels], chunk embeddings, activities);
        bit-identical to ``features(pcm)`` + ``step(concat(prev_feats, feats), ctx)``."""
        if self._mel is None:
            self._mel = HipMelSpectrogram(device=self.device, n_mels=self.dims.n_mels)
        mel, d = self._mel, self.dims.fc_d_model
        a = np.ascontiguousarray(pcm, dtype=np.float32).reshape(-1)
        n_prev = 0 if prev_feats is None else int(prev_feats.shape[0])
        pf = np.ascontiguousarray(prev_feats, np.float32) if n_prev else None
        n_ctx = 0 if ctx_embs is None else int(ctx_embs.shape[0])
        e = np.ascontiguousarray(ctx_embs, np.float32) if n_ctx else None
        cap_f = a.shape[0] // mel.hop + 2
        feats = np.empty((cap_f, self.dims.n_mels), np.float32)
        cap_c = (n_prev + cap_f) // 8 + 2
        chunk = np.empty((cap_c, d), np.float32)
        preds = np.empty((n_ctx + cap_c, self.n_spk), np.float32)
        n_f, n_chunk = C.c_int(), C.c_int()
        valid = -1 if mel.seq_len_plus_one else a.shape[0] // mel.hop
        _lib.check(self.lib.wlk_sf_step_pcm(
            self._h, mel._h, a.ctypes.data_as(C.c_void_p), a.shape[0], valid, pf.ctypes.data_as(C.c_void_p) if n_prev else None, n_prev,
            feats.ctypes.data_as(C.c_void_p), cap_f, C.byref(n_f), e.ctypes.data_as(C.c_void_p) if n_ctx else None, n_ctx,
            chunk.ctypes.data_as(C.c_void_p), cap_c, C.byref(n_chunk), preds.ctypes.data_as(C.c_void_p), preds.shape[0]))
        return feats[: n_f.value], chunk[: n_chunk.value], preds[: n_ctx + n_chunk.value]

    def forward_streaming_step_pcm(self, pcm: np.ndarray, prev_feats: Optional[np.ndarray], state: SortformerState,
                                   left_offset: int, right_offset: int):
        """``features`` + ``forward_streaming_step`` as ONE device call (one launch chain, one synchronisation): -> (the
        chunk's activities, the chunk's own feature rows for the caller's next call)."""
        ctx = np.concatenate([state.spkcache[: state.spkcache_len], state.fifo[: state.fifo_len]], axis=0)
        feats, chunk, preds = self.step_pcm(pcm, prev_feats, ctx if ctx.shape[0] else None)
        sf = self.cache.subsampling_factor
        return streaming_update(self.cache, state, chunk, preds, round(left_offset / sf), math.ceil(right_offset / sf)), feats

    def forward_streaming_step(self, features: np.ndarray, state: SortformerState, left_offset: int,
                               right_offset: int) -> np.ndarray:
        """SortformerEncLabelModel.forward_streaming_step for one stream: pre-encode the chunk, run the network over
        [spkcache | fifo | chunk], update ``state`` in place, return the chunk's activities."""
        ctx = np.concatenate([state.spkcache[: state.spkcache_len], state.fifo[: state.fifo_len]], axis=0)
        chunk, preds = self.step(features, ctx if ctx.shape[0] else None)
        sf = self.cache.subsampling_factor
        return streaming_update(self.cache, state, chunk, preds, round(left_offset / sf), math.ceil(right_offset / sf))

    def stats(self) -> Dict[str, float]:
        """Batching statistics of the model's lanes: stacked launch chains run, session steps inside them, mean sessions per chain."""
        a, b, r = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _lib.check(self.lib.wlk_sf_stats(self._h, C.byref(a), C.byref(b), C.byref(r)))
        return dict(stacked_steps=a.value, session_steps=b.value, rows=r.value,
                    mean_sessions_per_step=round(b.value / a.value, 3) if a.value else None)

    def export(self, what: str) -> np.ndarray:
        width = {"fc_out": self.dims.fc_d_model, "tf_out": self.dims.tf_d_model}[what]
        buf = np.empty((self.max_frames, width), np.float32)
        n = C.c_uint64()
        _lib.check(self.lib.wlk_sf_export(self._h, what.encode(), buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(n)))
        return buf.reshape(-1)[: n.value].reshape(-1, width).copy()

    def close(self):
        if self._mel is not None:
            self._mel.close()
            self._mel = None
        if self._h:
            self.lib.wlk_sf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def packed_sortformer_names(cdims: "_lib.SfDims"):
    lib = _lib.load()
    names, i = [], 0
    while True:
        s = C.c_char_p()
        if lib.wlk_sf_tensor_name(C.byref(cdims), i, C.byref(s)) != 0:
            break
        names.append(s.value.decode())
        i += 1
    return names

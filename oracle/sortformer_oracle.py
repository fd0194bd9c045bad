"""CPU ORACLE for streaming Sortformer diarization (a12: feature front end, network, speaker-cache update) -
TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it; the product
package never does).

PARITY: PARTLY PINNED.  The arithmetic restated here lives in NeMo (nemo-toolkit[asr] >=3,<4, pyproject.toml:80-85
of the reference; exact pin unknown, uv.lock is not in the tree), which is not installed where this code is built and
measured, and the reference's own tests hold no numeric vectors for it (tests/test_sortformer_real_fixture.py is
statistical and skipped without NeMo).  Since round 5 the front end and the FastConformer are pinned by the independent
ports of those NeMo modules that ship with `transformers` (tests/golden/sortformer_hf_kat.npz, made by
scripts/gen_golden_sortformer_hf.py; tests/test_sortformer_hf_golden.py):
  * ``nemo_log_mel``  == ``ParakeetFeatureExtractor`` (log-mel, valid-frame rule, zero fill)          <= 2e-6
  * ``pre_encode``, ``conformer_layer``, ``conformer_stack`` == ``ParakeetEncoder`` (stem / block / 17 blocks) <= 1e-5
  * ``transformer_layer`` == ``BertEncoder``'s post-LN block (relu, eps 1e-5; an independent implementation of the same
    published block, not a NeMo port)                                                                    <= 1e-5
UNPINNED remainder: that NeMo wires its 18 Transformer blocks and the sigmoid head as restated, and the speaker-cache
update (``streaming_update`` / ``compress_spkcache``; sortformer_backend.py:293-300).

The front end restates ``nemo.collections.asr.parts.preprocessing.features.FilterbankFeatures`` as configured at
whisperlivekit/diarization/sortformer_backend.py:181-187:
  window 25 ms (400 samples, symmetric hann), stride 10 ms, n_fft 512, 128 slaney mel bins 0-8000 Hz,
  pre-emphasis 0.97, centred STFT with zero padding, power 2, log(x + 2^-24), normalize "NA", pad_to 0;
  ``get_seq_len`` = len // hop valid frames, the frames behind them filled with pad_value 0.
"""
import numpy as np
import torch


def nemo_log_mel(pcm: np.ndarray, filters: np.ndarray, n_fft: int = 512, win_length: int = 400, hop: int = 160,
                 preemph: float = 0.97, log_guard: float = 2.0 ** -24, seq_len_plus_one: bool = False) -> np.ndarray:
    """-> [n_frames, n_mels], n_frames = len(pcm) // hop + 1 = what the centred STFT yields and ``get_features`` hands
    back; FilterbankFeatures.get_seq_len counts len(pcm) // hop of them as valid and the forward fills the rest with
    pad_value 0 (``seq_len_plus_one``: the rule of NeMo < 2.0, every frame valid)."""
    x = torch.from_numpy(np.asarray(pcm, np.float32)).unsqueeze(0)
    x = torch.cat((x[:, :1], x[:, 1:] - preemph * x[:, :-1]), dim=1)
    window = torch.hann_window(win_length, periodic=False)
    spec = torch.stft(x, n_fft=n_fft, hop_length=hop, win_length=win_length, center=True, window=window,
                      return_complex=True, pad_mode="constant")
    mag = torch.sqrt(torch.view_as_real(spec).pow(2).sum(-1))
    power = mag.pow(2.0)
    mel = torch.matmul(torch.from_numpy(np.asarray(filters, np.float32)), power)
    out = torch.log(mel + log_guard)
    n_frames = x.shape[1] // hop + 1
    res = out[0, :, :n_frames].transpose(0, 1).contiguous().numpy()
    if not seq_len_plus_one:
        res[x.shape[1] // hop:] = 0.0
    return res


# ================================================================================================
# The Sortformer network and its streaming speaker-cache update (NeMo SortformerEncLabelModel /
# SortformerModules / ConformerEncoder / TransformerEncoder), restated from the published NeMo
# implementation.  PARITY UNPINNED (see the header): no NeMo, no checkpoint, no vectors here.
# Weight names follow the NeMo state dict so that a real checkpoint's tensors can be dropped in.
# ================================================================================================
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch.nn.functional as F


@dataclass(frozen=True)
class SortformerDims:
    n_mels: int = 128
    fc_d_model: int = 512
    fc_layers: int = 17
    fc_heads: int = 8
    conv_kernel: int = 9
    sub_channels: int = 256
    tf_d_model: int = 192
    tf_layers: int = 18
    tf_heads: int = 8
    tf_inner: int = 768
    n_spk: int = 4


def rel_positional_encoding(length: int, d_model: int) -> torch.Tensor:
    """RelPositionalEncoding.forward's pos_emb for an input of ``length`` frames: positions
    length-1 ... -(length-1), interleaved sin/cos; [2*length-1, d_model]."""
    positions = torch.arange(length - 1, -length, -1, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(2 * length - 1, d_model)
    pe[:, 0::2] = torch.sin(positions * div)
    pe[:, 1::2] = torch.cos(positions * div)
    return pe


def _rel_shift(x: torch.Tensor) -> torch.Tensor:
    b, h, qlen, pos_len = x.size()
    x = F.pad(x, pad=(1, 0))
    x = x.view(b, h, -1, qlen)
    return x[:, :, 1:].view(b, h, qlen, pos_len)


def pre_encode(sd, dims: SortformerDims, feats: torch.Tensor) -> torch.Tensor:
    """ConvSubsampling('dw_striding', factor 8): [T, n_mels] -> [T', fc_d_model]."""
    x = feats.unsqueeze(0).unsqueeze(0)
    p = "encoder.pre_encode."
    x = F.relu(F.conv2d(x, sd[p + "conv.0.weight"], sd[p + "conv.0.bias"], stride=2, padding=1))
    for dw, pw in ((2, 3), (5, 6)):
        x = F.conv2d(x, sd[p + f"conv.{dw}.weight"], sd[p + f"conv.{dw}.bias"], stride=2, padding=1,
                     groups=dims.sub_channels)
        x = F.relu(F.conv2d(x, sd[p + f"conv.{pw}.weight"], sd[p + f"conv.{pw}.bias"]))
    b, c, t, f = x.size()
    x = F.linear(x.transpose(1, 2).reshape(b, t, -1), sd[p + "out.weight"], sd[p + "out.bias"])
    return x[0]


def _swish(x):
    return x * torch.sigmoid(x)


def conformer_layer(sd, p: str, dims: SortformerDims, x: torch.Tensor, pos_emb: torch.Tensor) -> torch.Tensor:
    """ConformerLayer.forward for one un-padded sequence x [T, d]."""
    d, h = dims.fc_d_model, dims.fc_heads
    ln = lambda t, n: F.layer_norm(t, (d,), sd[p + n + ".weight"], sd[p + n + ".bias"])
    ff = lambda t, n: F.linear(_swish(F.linear(t, sd[p + n + ".linear1.weight"], sd[p + n + ".linear1.bias"])),
                               sd[p + n + ".linear2.weight"], sd[p + n + ".linear2.bias"])
    res = x + 0.5 * ff(ln(x, "norm_feed_forward1"), "feed_forward1")
    y = ln(res, "norm_self_att")
    a = p + "self_attn."
    T = y.shape[0]
    q = F.linear(y, sd[a + "linear_q.weight"], sd[a + "linear_q.bias"]).view(T, h, -1)
    k = F.linear(y, sd[a + "linear_k.weight"], sd[a + "linear_k.bias"]).view(T, h, -1).permute(1, 0, 2)
    v = F.linear(y, sd[a + "linear_v.weight"], sd[a + "linear_v.bias"]).view(T, h, -1).permute(1, 0, 2)
    pp = F.linear(pos_emb, sd[a + "linear_pos.weight"]).view(-1, h, q.shape[-1]).permute(1, 0, 2)     # [h, 2T-1, dk]
    qu = (q + sd[a + "pos_bias_u"]).permute(1, 0, 2)
    qv = (q + sd[a + "pos_bias_v"]).permute(1, 0, 2)
    bd = _rel_shift(torch.matmul(qv, pp.transpose(-2, -1)).unsqueeze(0))[0]
    ac = torch.matmul(qu, k.transpose(-2, -1))
    scores = (ac + bd[:, :, :T]) / math.sqrt(q.shape[-1])
    ctx = torch.matmul(torch.softmax(scores, dim=-1), v).permute(1, 0, 2).reshape(T, d)
    res = res + F.linear(ctx, sd[a + "linear_out.weight"], sd[a + "linear_out.bias"])
    c = p + "conv."
    y = ln(res, "norm_conv").transpose(0, 1).unsqueeze(0)                                            # [1, d, T]
    y = F.glu(F.conv1d(y, sd[c + "pointwise_conv1.weight"], sd[c + "pointwise_conv1.bias"]), dim=1)
    y = F.conv1d(y, sd[c + "depthwise_conv.weight"], sd[c + "depthwise_conv.bias"], padding=(dims.conv_kernel - 1) // 2,
                 groups=d)
    y = F.batch_norm(y, sd[c + "batch_norm.running_mean"], sd[c + "batch_norm.running_var"],
                     sd[c + "batch_norm.weight"], sd[c + "batch_norm.bias"], training=False, eps=1e-5)
    y = F.conv1d(_swish(y), sd[c + "pointwise_conv2.weight"], sd[c + "pointwise_conv2.bias"])[0].transpose(0, 1)
    res = res + y
    res = res + 0.5 * ff(ln(res, "norm_feed_forward2"), "feed_forward2")
    return ln(res, "norm_out")


def transformer_layer(sd, p: str, dims: SortformerDims, x: torch.Tensor) -> torch.Tensor:
    """Post-LN TransformerEncoderBlock of nemo.collections.asr.modules.transformer."""
    d, h = dims.tf_d_model, dims.tf_heads
    dh = d // h
    T = x.shape[0]
    a = p + "first_sub_layer."
    scale = math.sqrt(math.sqrt(dh))
    q = (F.linear(x, sd[a + "query_net.weight"], sd[a + "query_net.bias"]).view(T, h, dh).permute(1, 0, 2)) / scale
    k = (F.linear(x, sd[a + "key_net.weight"], sd[a + "key_net.bias"]).view(T, h, dh).permute(1, 0, 2)) / scale
    v = F.linear(x, sd[a + "value_net.weight"], sd[a + "value_net.bias"]).view(T, h, dh).permute(1, 0, 2)
    ctx = torch.matmul(torch.softmax(torch.matmul(q, k.transpose(-1, -2)), dim=-1), v).permute(1, 0, 2).reshape(T, d)
    y = F.linear(ctx, sd[a + "out_projection.weight"], sd[a + "out_projection.bias"]) + x
    y = F.layer_norm(y, (d,), sd[p + "layer_norm_1.weight"], sd[p + "layer_norm_1.bias"])
    f = p + "second_sub_layer."
    z = F.linear(F.relu(F.linear(y, sd[f + "dense_in.weight"], sd[f + "dense_in.bias"])),
                 sd[f + "dense_out.weight"], sd[f + "dense_out.bias"]) + y
    return F.layer_norm(z, (d,), sd[p + "layer_norm_2.weight"], sd[p + "layer_norm_2.bias"])


def conformer_stack(sd, dims: SortformerDims, embs: torch.Tensor) -> torch.Tensor:
    """ConformerEncoder.forward_internal(bypass_pre_encode=True): xscaling, relative positions, the Conformer blocks."""
    T = embs.shape[0]
    x = embs * math.sqrt(dims.fc_d_model)                         # xscaling
    pos_emb = rel_positional_encoding(T, dims.fc_d_model)
    for i in range(dims.fc_layers):
        x = conformer_layer(sd, f"encoder.layers.{i}.", dims, x, pos_emb)
    return x


def forward_embeddings(sd, dims: SortformerDims, embs: torch.Tensor) -> torch.Tensor:
    """frontend_encoder(bypass_pre_encode=True) + forward_infer for one un-padded sequence of pre-encode
    embeddings [T, fc_d_model] -> speaker activities [T, n_spk] in [0, 1]."""
    x = conformer_stack(sd, dims, embs)
    m = "sortformer_modules."
    x = F.linear(x, sd[m + "encoder_proj.weight"], sd[m + "encoder_proj.bias"])
    for i in range(dims.tf_layers):
        x = transformer_layer(sd, f"transformer_encoder.layers.{i}.", dims, x)
    x = F.relu(x)
    x = F.relu(F.linear(x, sd[m + "first_hidden_to_hidden.weight"], sd[m + "first_hidden_to_hidden.bias"]))
    return torch.sigmoid(F.linear(x, sd[m + "single_hidden_to_spks.weight"], sd[m + "single_hidden_to_spks.bias"]))


@dataclass
class StreamParams:
    spkcache_len: int = 188
    fifo_len: int = 188
    spkcache_update_period: int = 144
    subsampling_factor: int = 10      # the reference overrides the module's 8 (sortformer_backend.py:121)
    spkcache_sil_frames_per_spk: int = 3
    pred_score_threshold: float = 0.25
    scores_boost_latest: float = 0.05
    sil_threshold: float = 0.2
    strong_boost_rate: float = 0.75
    weak_boost_rate: float = 1.5
    min_pos_scores_rate: float = 0.5
    max_index: int = 99999


@dataclass
class StreamState:
    """StreamingSortformerState as the reference initialises it (sortformer_backend.py:212-234): fixed-size
    zero buffers plus valid lengths (NeMo's asynchronous streaming layout)."""
    spkcache: torch.Tensor
    spkcache_preds: torch.Tensor
    fifo: torch.Tensor
    fifo_preds: torch.Tensor
    mean_sil_emb: torch.Tensor
    spkcache_len: int = 0
    fifo_len: int = 0
    n_sil_frames: int = 0


def new_stream_state(dims: SortformerDims, sp: StreamParams) -> StreamState:
    return StreamState(torch.zeros(sp.spkcache_len, dims.fc_d_model), torch.zeros(sp.spkcache_len, dims.n_spk),
                       torch.zeros(sp.fifo_len, dims.fc_d_model), torch.zeros(sp.fifo_len, dims.n_spk),
                       torch.zeros(dims.fc_d_model))


def compress_spkcache(sp: StreamParams, emb_seq: torch.Tensor, preds: torch.Tensor, mean_sil_emb: torch.Tensor):
    """SortformerModules._compress_spkcache for one stream: keep the spkcache_len most informative frames
    (per-speaker log-odds scores, boosts for the top frames, a few silence slots per speaker)."""
    n_frames, n_spk = preds.shape
    per_spk = sp.spkcache_len // n_spk - sp.spkcache_sil_frames_per_spk
    strong, weak = math.floor(per_spk * sp.strong_boost_rate), math.floor(per_spk * sp.weak_boost_rate)
    min_pos = math.floor(per_spk * sp.min_pos_scores_rate)
    log_p = torch.log(torch.clamp(preds, min=sp.pred_score_threshold))
    log_1p = torch.log(torch.clamp(1.0 - preds, min=sp.pred_score_threshold))
    scores = log_p - log_1p + log_1p.sum(dim=1, keepdim=True) - math.log(0.5)
    is_speech = preds > 0.5
    scores = torch.where(is_speech, scores, torch.tensor(float("-inf")))
    is_pos = scores > 0
    replace = (~is_pos) & is_speech & (is_pos.sum(dim=0, keepdim=True) >= min_pos)
    scores = torch.where(replace, torch.tensor(float("-inf")), scores)
    if sp.scores_boost_latest > 0:
        scores[sp.spkcache_len:, :] += sp.scores_boost_latest
    for n_boost, factor in ((strong, 2.0), (weak, 1.0)):
        _, idx = torch.topk(scores, min(n_boost, n_frames), dim=0, largest=True, sorted=False)
        scores[idx, torch.arange(n_spk).unsqueeze(0).expand_as(idx)] -= factor * math.log(0.5)
    if sp.spkcache_sil_frames_per_spk > 0:
        scores = torch.cat([scores, torch.full((sp.spkcache_sil_frames_per_spk, n_spk), float("inf"))], dim=0)
    n_total = scores.shape[0]
    flat = scores.t().reshape(-1)
    vals, idx = torch.topk(flat, sp.spkcache_len, sorted=False)
    idx = torch.where(vals != float("-inf"), idx, torch.tensor(sp.max_index))
    idx, _ = torch.sort(idx)
    disabled = idx == sp.max_index
    idx = torch.remainder(idx, n_total)
    disabled = disabled | (idx >= n_frames)
    idx = torch.where(disabled, torch.zeros_like(idx), idx)
    emb = torch.where(disabled.unsqueeze(-1), mean_sil_emb.unsqueeze(0).expand(sp.spkcache_len, -1), emb_seq[idx])
    pr = torch.where(disabled.unsqueeze(-1), torch.zeros(()), preds[idx])
    return emb, pr


def streaming_update(sp: StreamParams, st: StreamState, chunk: torch.Tensor, preds: torch.Tensor, lc: int, rc: int):
    """SortformerModules.streaming_update_async for one stream.  ``chunk``: this step's pre-encode embeddings
    [Tc, d]; ``preds``: activities of [spkcache | fifo | chunk]; returns the chunk's own activities."""
    n_spk = preds.shape[1]
    max_chunk = chunk.shape[0] - lc - rc
    chunk_len = max(0, min(chunk.shape[0] - lc, max_chunk))
    s_len, f_len = st.spkcache_len, st.fifo_len
    fifo_preds = torch.zeros_like(st.fifo_preds)
    fifo_preds[:f_len] = preds[s_len: s_len + f_len]
    chunk_preds = torch.zeros(max_chunk, n_spk)
    chunk_preds[:chunk_len] = preds[s_len + f_len + lc: s_len + f_len + lc + chunk_len]
    up_fifo = torch.zeros(sp.fifo_len + max_chunk, chunk.shape[1])
    up_fifo_p = torch.zeros(sp.fifo_len + max_chunk, n_spk)
    pop_max = min(max(sp.spkcache_update_period, max_chunk), max_chunk + sp.fifo_len)
    up_cache = torch.zeros(sp.spkcache_len + pop_max, chunk.shape[1])
    up_cache_p = torch.zeros(sp.spkcache_len + pop_max, n_spk)
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
        is_sil = pop_p.sum(dim=1) < sp.sil_threshold
        if int(is_sil.sum()) > 0:
            total = st.mean_sil_emb * st.n_sil_frames + (pop_e * is_sil.unsqueeze(-1)).sum(dim=0)
            st.n_sil_frames += int(is_sil.sum())
            st.mean_sil_emb = total / max(st.n_sil_frames, 1)
        up_cache[s_len: s_len + pop] = pop_e
        up_cache_p[s_len: s_len + pop] = pop_p       # spkcache_preds starts at zeros (>= 0): "already compressed" branch
        st.fifo_len -= pop
        up_fifo[: st.fifo_len] = up_fifo[pop: pop + st.fifo_len].clone()
        up_fifo_p[: st.fifo_len] = up_fifo_p[pop: pop + st.fifo_len].clone()
        up_fifo[st.fifo_len:] = 0
        up_fifo_p[st.fifo_len:] = 0
    st.fifo, st.fifo_preds = up_fifo[: sp.fifo_len], up_fifo_p[: sp.fifo_len]
    if st.spkcache_len > sp.spkcache_len:
        st.spkcache, st.spkcache_preds = compress_spkcache(sp, up_cache, up_cache_p, st.mean_sil_emb)
        st.spkcache_len = sp.spkcache_len
    else:
        st.spkcache, st.spkcache_preds = up_cache[: sp.spkcache_len], up_cache_p[: sp.spkcache_len]
    return chunk_preds


def forward_streaming_step(sd, dims: SortformerDims, sp: StreamParams, st: StreamState, feats: torch.Tensor,
                           left_offset: int, right_offset: int) -> torch.Tensor:
    """SortformerEncLabelModel.forward_streaming_step for one stream; feats [T, n_mels]."""
    chunk = pre_encode(sd, dims, feats)
    embs = torch.cat([st.spkcache[: st.spkcache_len], st.fifo[: st.fifo_len], chunk], dim=0)
    preds = forward_embeddings(sd, dims, embs)
    return streaming_update(sp, st, chunk, preds, round(left_offset / sp.subsampling_factor),
                            math.ceil(right_offset / sp.subsampling_factor))

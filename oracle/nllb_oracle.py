"""CPU restatement of the NLLB-200 / M2M-100 forward pass - TEST INFRASTRUCTURE (tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() only; the product path is whisperlivekit_amd/csrc/nllb.hip and never imports this).

The reference reaches the model through the third-party `nllw` package (whisperlivekit/core.py:320-329), absent from the
reference tree; the arithmetic restated here is the published network it wraps, `transformers` 5.15.0
models/m2m_100/modeling_m2m_100.py (line numbers below refer to that file).  Pinned by outputs of `transformers`' own
M2M100ForConditionalGeneration on seeded weights: tests/golden/nllb_kat.npz (scripts/gen_golden_nllb.py),
tests/test_nllb.py::test_oracle_matches_transformers.

Parameters are addressed by their `transformers` names (model.encoder.layers.N.self_attn.q_proj.weight ...).
"""
import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F


def sinusoid_table(n_rows: int, d: int, padding_idx: Optional[int]) -> torch.Tensor:
    """M2M100SinusoidalPositionalEmbedding.get_embedding (:99-118)."""
    half = d // 2
    step = math.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half, dtype=torch.int64).float() * -step)
    ang = torch.arange(n_rows, dtype=torch.int64).float().unsqueeze(1) * freq.unsqueeze(0)
    table = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1).view(n_rows, -1)
    if d % 2 == 1:
        table = torch.cat([table, torch.zeros(n_rows, 1)], dim=1)
    if padding_idx is not None:
        table[padding_idx, :] = 0
    return table


def _lin(x, sd, name):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def _ln(x, sd, name):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def _heads(x, n_head):
    return x.view(*x.shape[:-1], n_head, -1).transpose(-3, -2)            # [.., T, d] -> [.., H, T, dh]


def _attend(q, k, v, n_head, mask=None):
    """eager_attention_forward (:186-208): softmax(q k^T * dh^-0.5 + mask) v, heads split on the channel axis."""
    qh, kh, vh = _heads(q, n_head), _heads(k, n_head), _heads(v, n_head)
    w = torch.matmul(qh, kh.transpose(-1, -2)) * (qh.shape[-1] ** -0.5)
    if mask is not None:
        w = w + mask
    w = F.softmax(w, dim=-1)
    o = torch.matmul(w, vh).transpose(-3, -2)
    return o.reshape(*o.shape[:-2], -1)


class NllbOracle:
    def __init__(self, cfg, sd: Dict[str, torch.Tensor]):
        self.cfg = cfg
        self.sd = {k: (v if torch.is_tensor(v) else torch.from_numpy(v)).float() for k, v in sd.items()}
        self.scale = math.sqrt(cfg.d_model) if cfg.scale_embedding else 1.0
        self.pos = sinusoid_table(cfg.max_position_embeddings + 2, cfg.d_model, cfg.pad_token_id)
        self.emb = self.sd["model.shared.weight"]

    def _embed(self, ids: torch.Tensor, past: int) -> torch.Tensor:
        """M2M100ScaledWordEmbedding (:76-77) + positions (:121-145, :166-180) for sequences without padding:
        position id = index + past + padding_idx + 1."""
        n = ids.shape[-1]
        rows = torch.arange(past, past + n) + self.cfg.pad_token_id + 1
        if int(rows[-1]) >= self.pos.shape[0]:
            self.pos = sinusoid_table(int(rows[-1]) + 3, self.cfg.d_model, self.cfg.pad_token_id)
        return F.embedding(ids, self.emb) * self.scale + self.pos[rows]

    @torch.no_grad()
    def encode(self, src_ids: Sequence[int]) -> torch.Tensor:
        """M2M100Encoder.forward (:546-590) over one unpadded sentence -> [S, d]."""
        cfg, sd = self.cfg, self.sd
        x = self._embed(torch.as_tensor(list(src_ids), dtype=torch.int64), 0)
        for i in range(cfg.encoder_layers):
            p = f"model.encoder.layers.{i}."
            h = _ln(x, sd, p + "self_attn_layer_norm")                                     # :364-371
            a = _attend(_lin(h, sd, p + "self_attn.q_proj"), _lin(h, sd, p + "self_attn.k_proj"),
                        _lin(h, sd, p + "self_attn.v_proj"), cfg.attention_heads)
            x = x + _lin(a, sd, p + "self_attn.out_proj")
            h = _ln(x, sd, p + "final_layer_norm")                                         # :373-379
            x = x + _lin(F.relu(_lin(h, sd, p + "fc1")), sd, p + "fc2")
        return _ln(x, sd, "model.encoder.layer_norm")

    def new_cache(self):
        return dict(k=[None] * self.cfg.decoder_layers, v=[None] * self.cfg.decoder_layers,
                    xk=[None] * self.cfg.decoder_layers, xv=[None] * self.cfg.decoder_layers)

    @torch.no_grad()
    def decode(self, tokens: torch.Tensor, enc: torch.Tensor, cache) -> torch.Tensor:
        """M2M100Decoder.forward (:635-713) + lm_head (:880) for tokens [B, n] on top of the cache -> logits [B, n, V]."""
        cfg, sd = self.cfg, self.sd
        past = 0 if cache["k"][0] is None else cache["k"][0].shape[1]
        n = tokens.shape[-1]
        x = self._embed(tokens, past)
        mask = torch.full((n, past + n), float("-inf")).triu_(past + 1)                    # causal over [cache | new]
        for i in range(cfg.decoder_layers):
            p = f"model.decoder.layers.{i}."
            h = _ln(x, sd, p + "self_attn_layer_norm")                                     # :446-456
            k, v = _lin(h, sd, p + "self_attn.k_proj"), _lin(h, sd, p + "self_attn.v_proj")
            if cache["k"][i] is not None:
                k, v = torch.cat([cache["k"][i], k], dim=1), torch.cat([cache["v"][i], v], dim=1)
            cache["k"][i], cache["v"][i] = k, v
            a = _attend(_lin(h, sd, p + "self_attn.q_proj"), k, v, cfg.attention_heads, mask)
            x = x + _lin(a, sd, p + "self_attn.out_proj")
            h = _ln(x, sd, p + "encoder_attn_layer_norm")                                  # :458-471
            if cache["xk"][i] is None:
                cache["xk"][i] = _lin(enc, sd, p + "encoder_attn.k_proj")
                cache["xv"][i] = _lin(enc, sd, p + "encoder_attn.v_proj")
            a = _attend(_lin(h, sd, p + "encoder_attn.q_proj"), cache["xk"][i], cache["xv"][i], cfg.attention_heads)
            x = x + _lin(a, sd, p + "encoder_attn.out_proj")
            h = _ln(x, sd, p + "final_layer_norm")                                         # :473-479
            x = x + _lin(F.relu(_lin(h, sd, p + "fc1")), sd, p + "fc2")
        x = _ln(x, sd, "model.decoder.layer_norm")
        return F.linear(x, self.emb)

    @staticmethod
    def reorder(cache, source_rows: Sequence[int]):
        idx = torch.as_tensor(list(source_rows))
        for i in range(len(cache["k"])):
            if cache["k"][i] is not None:
                cache["k"][i], cache["v"][i] = cache["k"][i][idx], cache["v"][i][idx]


class OracleNllbSession:
    """The methods whisperlivekit_amd.nllb.generate calls on a HipNllbSession, answered by the oracle (CPU tests of the
    host-side generation logic; the gpu-marked twins run the library)."""

    def __init__(self, oracle: NllbOracle, rows: int = 1):
        import types
        self.oracle, self.rows = oracle, rows
        self.model = types.SimpleNamespace(cfg=oracle.cfg)
        self.enc = self.cache = self.last = None

    def encode(self, src_ids):
        self.enc = self.oracle.encode(src_ids)

    def decode(self, tokens, first):
        t = torch.as_tensor(tokens, dtype=torch.int64)
        if first:
            self.cache = self.oracle.new_cache()
        self.last = self.oracle.decode(t, self.enc, self.cache)[:, -1]

    def step(self, tokens, k=1):
        self.decode(torch.as_tensor(tokens, dtype=torch.int64).view(-1, 1), first=False)
        return self.topk(k)

    def kv_reorder(self, source_rows):
        self.oracle.reorder(self.cache, source_rows)

    def logits(self):
        return self.last.numpy().copy()

    def topk(self, k):
        lp, ids = torch.log_softmax(self.last, dim=-1).topk(k, dim=-1)
        return lp.numpy(), ids.numpy().astype("int32")

    def encoder_output(self):
        return self.enc.numpy()

"""CPU ORACLE for the simul_whisper hot path - TEST INFRASTRUCTURE ONLY.

This file restates, in plain fp32 PyTorch-CPU tensor ops, the arithmetic of the reference
WhisperLiveKit path  log-mel -> Whisper encoder -> decoder with cross-attention QK ->
AlignAtt post-processing -> beam/greedy token update -> AlignAtt streaming policy.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package ``whisperlivekit_amd`` never does (tests/test_abi_cpu.py::test_product_never_imports_the_oracle enforces that).

Pinning: the reference ships no numeric golden vectors for this path, so the
oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, produced in the build container by
``scripts/gen_golden.py`` (which imports /root/reference with three harness-side stubs) and
committed under ``tests/golden/``; ``tests/test_oracle_golden.py`` replays them.

Every function cites the reference lines it follows (paths relative to the WhisperLiveKit
repository root).  Weights come in as a ``{name: tensor}`` dict with the reference's checkpoint
names (whisperlivekit/whisper/model.py module tree).
"""
from __future__ import annotations

import math
import string
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

N_FFT, HOP, N_SAMPLES, N_FRAMES = 400, 160, 480000, 3000
DEC_PAD = 50257  # whisperlivekit/simul_whisper/align_att_base.py:9


def to_torch_state_dict(sd) -> Dict[str, torch.Tensor]:
    return {k: (v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v))).float()
            for k, v in sd.items()}


# ---------------------------------------------------------------------------------------
# a2  log-mel                                      whisperlivekit/whisper/audio.py:110-157
# ---------------------------------------------------------------------------------------

def log_mel_spectrogram(audio: torch.Tensor, filters: torch.Tensor, padding: int = 0) -> torch.Tensor:
    """``filters`` is the [n_mels, 201] filterbank (audio.py:91-107)."""
    audio = audio.float()
    if padding > 0:
        audio = F.pad(audio, (0, padding))                                   # audio.py:145-146
    window = torch.hann_window(N_FFT)                                        # audio.py:147
    stft = torch.stft(audio, N_FFT, HOP, window=window, return_complex=True)  # audio.py:148
    power = stft[..., :-1].abs() ** 2                                        # audio.py:149
    mel = filters @ power                                                    # audio.py:152
    log_spec = torch.clamp(mel, min=1e-10).log10()                           # audio.py:154
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)                 # audio.py:155
    return (log_spec + 4.0) / 4.0                                            # audio.py:156


def encoder_input_from_audio(audio: torch.Tensor, filters: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """The mel the streaming path feeds the encoder, and ``content_mel_len``
    (whisperlivekit/simul_whisper/simul_whisper.py:344-350): pad 30 s of zeros, keep the first
    3000 frames, content length = (padded frames - 3000) / 2 encoder positions."""
    mel_padded = log_mel_spectrogram(audio, filters, padding=N_SAMPLES).unsqueeze(0)
    t = mel_padded.shape[2]
    if t > N_FRAMES:
        mel = mel_padded[:, :, :N_FRAMES]                                   # audio.py:70-73
    else:
        mel = F.pad(mel_padded, (0, N_FRAMES - t))
    content_mel_len = int((t - mel.shape[2]) / 2)
    return mel, content_mel_len


# ---------------------------------------------------------------------------------------
# a14/a13/a4/a3  blocks                            whisperlivekit/whisper/model.py:39-254
# ---------------------------------------------------------------------------------------

def _ln(x, sd, prefix):
    return F.layer_norm(x.float(), (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"])  # model.py:39-41


def _lin(x, sd, prefix):
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))     # model.py:44-50


def qkv_attention(q, k, v, n_head, mask=None):
    """model.py:148-173 with SDPA off: scale q and k by d_head**-0.25, fp32 QK, softmax, @V.
    Returns (context, pre-softmax qk)."""
    n_batch, n_ctx, n_state = q.shape
    scale = (n_state // n_head) ** -0.25
    q = q.view(*q.shape[:2], n_head, -1).permute(0, 2, 1, 3)
    k = k.view(*k.shape[:2], n_head, -1).permute(0, 2, 1, 3)
    v = v.view(*v.shape[:2], n_head, -1).permute(0, 2, 1, 3)
    qk = (q * scale) @ (k * scale).transpose(-1, -2)
    if mask is not None:
        qk = qk + mask[:n_ctx, :n_ctx]
    qk = qk.float()
    w = F.softmax(qk, dim=-1)
    out = (w @ v).permute(0, 2, 1, 3).flatten(start_dim=2)
    return out, qk


def _mlp(x, sd, prefix):
    return _lin(F.gelu(_lin(x, sd, prefix + ".mlp.0")), sd, prefix + ".mlp.2")  # model.py:194-197 (erf GELU)


def encoder_forward(sd, dims, mel: torch.Tensor) -> torch.Tensor:
    """AudioEncoder.forward, model.py:238-254.  mel [1, n_mels, 3000] -> [1, 1500, d]."""
    x = F.gelu(F.conv1d(mel, sd["encoder.conv1.weight"], sd["encoder.conv1.bias"], padding=1))
    x = F.gelu(F.conv1d(x, sd["encoder.conv2.weight"], sd["encoder.conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1)
    x = x + sd["encoder.positional_embedding"]
    for i in range(dims.n_audio_layer):
        p = f"encoder.blocks.{i}"
        h = _ln(x, sd, p + ".attn_ln")
        a, _ = qkv_attention(_lin(h, sd, p + ".attn.query"), _lin(h, sd, p + ".attn.key"),
                             _lin(h, sd, p + ".attn.value"), dims.n_audio_head)
        x = x + _lin(a, sd, p + ".attn.out")
        x = x + _mlp(_ln(x, sd, p + ".mlp_ln"), sd, p)
    return _ln(x, sd, "encoder.ln_post")


class DecoderCache:
    """Per-``infer`` KV state: self-attention K/V grown by concatenation, cross K/V computed on
    first use (model.py:117-146).  Cleared after every ``infer`` (decoder_state.py:51-59)."""

    def __init__(self, n_layer: int):
        self.self_k: List[Optional[torch.Tensor]] = [None] * n_layer
        self.self_v: List[Optional[torch.Tensor]] = [None] * n_layer
        self.cross_k: List[Optional[torch.Tensor]] = [None] * n_layer
        self.cross_v: List[Optional[torch.Tensor]] = [None] * n_layer

    def reorder(self, source_indices: Sequence[int]):
        """simul_whisper/beam.py:15-19 - only the self-attention entries follow the beams."""
        if list(source_indices) != list(range(len(source_indices))):
            idx = torch.tensor(list(source_indices))
            for i in range(len(self.self_k)):
                if self.self_k[i] is not None:
                    self.self_k[i] = self.self_k[i][idx]
                    self.self_v[i] = self.self_v[i][idx]


def decoder_forward(sd, dims, tokens: torch.Tensor, xa: torch.Tensor, cache: DecoderCache):
    """TextDecoder.forward with ``return_cross_attn=True`` (model.py:279-332).
    tokens int64 [B, P]; xa [1 or B, 1500, d].  Returns (logits [B,P,V], [L x qk [B,H,P,1500]])."""
    n_ctx = dims.n_text_ctx
    offset = 0 if cache.self_k[0] is None else cache.self_k[0].shape[1]     # model.py:304-310
    x = F.embedding(tokens, sd["decoder.token_embedding.weight"]) \
        + sd["decoder.positional_embedding"][offset: offset + tokens.shape[-1]]
    mask = torch.empty(n_ctx, n_ctx).fill_(-np.inf).triu_(1)                 # model.py:276-277
    cross_qk = []
    for i in range(dims.n_text_layer):
        p = f"decoder.blocks.{i}"
        h = _ln(x, sd, p + ".attn_ln")
        q = _lin(h, sd, p + ".attn.query")
        k = _lin(h, sd, p + ".attn.key")
        v = _lin(h, sd, p + ".attn.value")
        if cache.self_k[i] is None or k.shape[1] > n_ctx:                  # model.py:134-137
            cache.self_k[i], cache.self_v[i] = k, v
        else:
            k = torch.cat([cache.self_k[i], k], dim=1)
            v = torch.cat([cache.self_v[i], v], dim=1)
            cache.self_k[i], cache.self_v[i] = k, v
        a, _ = qkv_attention(q, k, v, dims.n_text_head, mask)
        x = x + _lin(a, sd, p + ".attn.out")
        h = _ln(x, sd, p + ".cross_attn_ln")
        if cache.cross_k[i] is None:                                        # model.py:117-126
            cache.cross_k[i] = _lin(xa, sd, p + ".cross_attn.key")
            cache.cross_v[i] = _lin(xa, sd, p + ".cross_attn.value")
        a, qk = qkv_attention(_lin(h, sd, p + ".cross_attn.query"), cache.cross_k[i], cache.cross_v[i],
                              dims.n_text_head)
        x = x + _lin(a, sd, p + ".cross_attn.out")
        cross_qk.append(qk)
        x = x + _mlp(_ln(x, sd, p + ".mlp_ln"), sd, p)
    x = _ln(x, sd, "decoder.ln")
    logits = (x @ sd["decoder.token_embedding.weight"].transpose(0, 1)).float()  # model.py:326-328
    return logits, cross_qk


# ---------------------------------------------------------------------------------------
# a8  AlignAtt post-processing   simul_whisper/simul_whisper.py:390-437, whisper/timing.py:19-54
# ---------------------------------------------------------------------------------------

def median_filter(x: torch.Tensor, width: int) -> torch.Tensor:
    pad = width // 2
    if x.shape[-1] <= pad:
        return x                                                             # timing.py:22-24
    x = F.pad(x, (pad, pad, 0, 0), mode="reflect")                           # timing.py:35
    return x.unfold(-1, width, 1).sort()[0][..., pad]                        # timing.py:49


def alignatt_attention(accumulated: List[List[torch.Tensor]], align_heads: Sequence[Tuple[int, int]],
                       n_layer: int, content_mel_len: int, beam: int) -> torch.Tensor:
    """``accumulated`` = the kept decode steps (<=16), each a list of L tensors [B,H,q,1500].
    Returns [B, rows, content_mel_len] (simul_whisper.py:390-433)."""
    per_head: List[List[torch.Tensor]] = [[] for _ in align_heads]
    by_layer: Dict[int, List[Tuple[int, int]]] = {}
    for rank, (layer, head) in enumerate(align_heads):
        by_layer.setdefault(layer, []).append((rank, head))
    for step in accumulated:
        for layer in range(n_layer):
            if layer not in by_layer:
                continue
            w = F.softmax(step[layer], dim=-1)
            for rank, head in by_layer[layer]:
                a = w[0, head, :, :].unsqueeze(0) if beam == 1 else w[:, head, :, :]
                per_head[rank].append(a)
    tmp = [torch.cat(m, dim=1) for m in per_head if m]
    if not tmp:
        return torch.zeros(beam, 1, content_mel_len)
    a = torch.stack(tmp, dim=1)
    std, mean = torch.std_mean(a, dim=-2, keepdim=True, unbiased=False)
    a = (a - mean) / (std + 1e-8)
    a = median_filter(a, 7)
    a = a.mean(dim=1)
    return a[:, :, :content_mel_len]


# ---------------------------------------------------------------------------------------
# a6/a7  logit filters and token update
# ---------------------------------------------------------------------------------------

def dry_penalties(seq: Sequence[int], eot: int) -> Dict[int, float]:
    """{token: amount to subtract} per align_att_base.py:492-537."""
    if len(seq) < 5 or seq[-1] >= eot:
        return {}
    last = seq[-1]
    best: Dict[int, int] = {}
    n = len(seq)
    for i in range(n - 2, -1, -1):
        if seq[i] != last or seq[i + 1] >= eot:
            continue
        length = 1
        while length < 50:
            j, k = i - length, n - 1 - length
            if j < 0 or k <= i or seq[j] != seq[k] or seq[j] >= eot:
                break
            length += 1
        nxt = seq[i + 1]
        if length > best.get(nxt, 0):
            best[nxt] = length
    return {tok: 1.0 * 2.0 ** (ln - 2) for tok, ln in best.items() if ln >= 2}


class BeamUpdate:
    """BeamSearchDecoder.update (whisper/decoding.py:317-376); the streaming path uses it even
    for beam_size 1 because the config hard-wires decoder_type="beam" (simul_whisper/backend.py:377)."""

    def __init__(self, beam: int, eot: int):
        self.beam, self.eot = beam, eot
        self.max_candidates = round(beam * 1.0)
        self.finished: Optional[List[dict]] = None

    def reset(self):
        self.finished = None

    def update(self, tokens: torch.Tensor, logits: torch.Tensor, sum_logprobs: torch.Tensor,
               cache: DecoderCache):
        n_audio = tokens.shape[0] // self.beam
        if self.finished is None:
            self.finished = [{} for _ in range(n_audio)]
        logprobs = F.log_softmax(logits.float(), dim=-1)
        next_tokens, sources, newly = [], [], []
        for i in range(n_audio):
            scores, src, fin = {}, {}, {}
            for j in range(self.beam):
                idx = i * self.beam + j
                prefix = tokens[idx].tolist()
                vals, ids = logprobs[idx].topk(self.beam + 1)
                for lp, tk in zip(vals, ids):
                    seq = tuple(prefix + [tk.item()])
                    scores[seq] = (sum_logprobs[idx] + lp).item()
                    src[seq] = idx
            saved = 0
            for seq in sorted(scores, key=scores.get, reverse=True):
                if seq[-1] == self.eot:
                    fin[seq] = scores[seq]
                else:
                    sum_logprobs[len(next_tokens)] = scores[seq]
                    next_tokens.append(seq)
                    sources.append(src[seq])
                    saved += 1
                    if saved == self.beam:
                        break
            newly.append(fin)
        tokens = torch.tensor(next_tokens)
        cache.reorder(sources)
        for prev, new in zip(self.finished, newly):
            for seq in sorted(new, key=new.get, reverse=True):
                if len(prev) >= self.max_candidates:
                    break
                prev[seq] = new[seq]
        completed = all(len(s) >= self.max_candidates for s in self.finished)
        return tokens, completed


# ---------------------------------------------------------------------------------------
# a1/a9/a10/a16  streaming policy
# ---------------------------------------------------------------------------------------

@dataclass
class OracleConfig:
    """AlignAttConfig as the engine fills it (simul_whisper/config.py:5-23, backend.py:369-384,
    defaults from whisperlivekit/config.py:101-113)."""
    frame_threshold: int = 25
    rewind_threshold: int = 200
    audio_max_len: float = 30.0
    audio_min_len: float = 0.0
    beam_size: int = 1
    nonspeech_prob: float = 0.5
    max_context_tokens: Optional[int] = None
    init_prompt: Optional[str] = None
    static_init_prompt: Optional[str] = None
    language: str = "en"
    never_fire: bool = False
    cif_ckpt_path: Optional[str] = None          # a11: optional CIF end-of-word head (state dict of a Linear(d, 1))


def cif_fire_at_boundary(feature: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> bool:
    """a11, simul_whisper/eow_detection.py:40-77 on the content rows [T, d] of the encoder output: sigmoid
    weights rescaled to an integer sum (peaks above 0.999 are flattened, at most 10 rounds), integrated over
    all but the last frame modulo 1; fire when the first frame at/after the last completed unit is one of
    the final two."""
    t = feature.shape[0]
    alphas = torch.sigmoid(feature @ weight.reshape(-1) + bias.reshape(()))
    total = alphas.sum()
    alphas = alphas * (torch.round(total).int().float() / total)
    rounds = 0
    while bool((alphas > 0.999).any()):
        rounds += 1
        if rounds > 10:
            break
        for y in torch.nonzero(alphas > 0.999).flatten().tolist():
            if alphas[y] >= 0.999:
                live = alphas.ne(0).float()
                alphas = alphas * 0.5 + (0.5 * alphas.sum() / live.sum()) * live
    integ = torch.cumsum(alphas[:-1], dim=0)
    integ = integ - (integ[-1] // 0.999) * 1.0
    pos = torch.nonzero(integ >= 0).flatten()
    return bool(pos.numel() and pos[0] >= t - 2)


@dataclass
class Word:
    start: float
    end: float
    text: str
    speaker: int = -1
    detected_language: Optional[str] = None      # align_att_base.py:437


class OracleAlignAtt:
    """One streaming session: the AlignAtt loop of align_att_base.py:174-322 over the oracle
    numerics above, with the PyTorch backend's state handling (simul_whisper.py:219-262)."""

    def __init__(self, sd, dims, align_heads, tokenizer, filters, cfg: OracleConfig = OracleConfig(),
                 tokenizer_factory=None):
        self.sd, self.dims, self.tok, self.cfg = sd, dims, tokenizer, cfg
        self.tokenizer_factory = tokenizer_factory      # language code -> tokenizer (language="auto" only)
        self.detected_language = None if cfg.language == "auto" else cfg.language
        self.cif = None                                 # eow_detection.py:12-37
        if cfg.cif_ckpt_path:
            ck = torch.load(cfg.cif_ckpt_path, map_location="cpu", weights_only=True)
            self.cif = (ck["weight"].float(), ck["bias"].float())
        self.align_heads = list(align_heads)
        self.filters = torch.from_numpy(np.asarray(filters)).float()
        suppress = [tokenizer.transcribe, tokenizer.translate, tokenizer.sot, tokenizer.sot_prev,
                    tokenizer.sot_lm, tokenizer.no_timestamps] + list(tokenizer.all_language_tokens)
        if tokenizer.no_speech is not None:
            suppress.append(tokenizer.no_speech)
        self.suppress = sorted(set(suppress))                               # simul_whisper.py:161-172
        self.max_text_len = dims.n_text_ctx
        self.max_context_tokens = cfg.max_context_tokens or self.max_text_len
        self.global_time_offset = 0.0
        self.speaker = -1
        self.segments: List[torch.Tensor] = []
        self.first_timestamp = None
        self.cumulative_time_offset = 0.0
        self.pending_tokens: List[int] = []
        self.pending_times: List[float] = []
        self.pending_retries = 0
        self.updater = BeamUpdate(cfg.beam_size, tokenizer.eot)
        self.trace: List[dict] = []   # per-call numeric trace for the parity tests
        self.refresh_segment(complete=True)

    # -- state ---------------------------------------------------------------------------
    def _init_tokens(self):
        init = list(self.tok.sot_sequence_including_notimestamps)
        self.initial_tokens = init
        self.sot_index = list(self.tok.sot_sequence).index(self.tok.sot)
        self.tokens: List[List[int]] = [init]                                # simul_whisper.py:192-202

    def _init_context(self):
        self.context_text = ""
        self.context_pending: List[int] = []
        if self.cfg.static_init_prompt is not None:
            self.context_text = self.cfg.static_init_prompt
        if self.cfg.init_prompt is not None:
            self.context_text += self.cfg.init_prompt                        # simul_whisper.py:204-217

    def refresh_segment(self, complete=False):                              # align_att_base.py:115-132
        self._init_tokens()
        self.last_attend_frame = -self.cfg.rewind_threshold
        self.cumulative_time_offset = 0.0
        self._init_context()
        if not complete and len(self.segments) > 2:
            self.segments = self.segments[-2:]
        else:
            self.segments = []
        self.pending_tokens, self.pending_times, self.pending_retries = [], [], 0

    def segments_len(self):
        return sum(s.shape[0] for s in self.segments) / 16000

    def _context_append(self, ids: List[int]):                              # token_buffer.py:66-89
        allt = self.context_pending + ids
        dec = self.tok.decode(allt)
        bad = "�"
        if bad in dec:
            if len(allt) > 1:
                part = self.tok.decode(allt[:-1])
                if bad not in part:
                    self.context_text += part
                    self.context_pending = [allt[-1]]
                else:
                    self.context_pending = allt
            else:
                self.context_pending = allt
        else:
            self.context_text += dec
            self.context_pending = []

    def insert_audio(self, segment: Optional[torch.Tensor] = None):         # simul_whisper.py:219-237
        if segment is not None:
            self.segments.append(torch.as_tensor(segment).float())
        removed = 0
        total = self.segments_len()
        while len(self.segments) > 1 and total > self.cfg.audio_max_len:
            removed = self.segments[0].shape[0] / 16000
            total -= removed
            self.last_attend_frame -= int(50 * removed)
            self.cumulative_time_offset += removed
            self.segments = self.segments[1:]
            if len(self.tokens) > 1:
                self._context_append(list(self.tokens[1]))
                self.tokens = [self.initial_tokens] + self.tokens[2:]
        return removed

    def _context_ids(self):
        return [self.tok.sot_prev] + self.tok.encode(self.context_text)      # token_buffer.py:14-20

    def _trim_context(self):                                                # align_att_base.py:100-113
        c = len(self._context_ids()) - 1
        l = sum(len(t) for t in self.tokens) + c
        after = 0 if self.cfg.static_init_prompt is None else len(self.cfg.static_init_prompt)
        while c > self.max_context_tokens or l > self.max_text_len - 20:
            ids = self.tok.encode(self.context_text[after:])                 # token_buffer.py:46-64
            words, wids = self.tok.split_to_word_tokens(ids)
            if not words:
                break
            self.context_text = self.context_text[:after] + "".join(words[1:])
            t = len(wids[0])
            l -= t
            c -= t
            if t == 0:
                break

    def _current_tokens(self) -> List[List[int]]:                           # simul_whisper.py:239-254
        flat: List[int] = []
        if self.context_text:
            flat += self._context_ids()
        for t in self.tokens:
            flat += list(t)
        return [list(flat) for _ in range(self.cfg.beam_size)]

    def _fire_at_boundary(self, content_feature: torch.Tensor) -> bool:     # simul_whisper.py:256-264
        if self.cif is None:                      # no checkpoint: always fire unless never_fire (eow_detection.py:12-25)
            return not self.cfg.never_fire
        if self.cfg.never_fire:
            return False
        return cif_fire_at_boundary(content_feature, *self.cif)

    def _detect_language_if_needed(self, enc):                              # align_att_base.py:153-170
        """language="auto": once >= 2 s have passed since the first timestamp, one <|sot|> decoder step, soft-max
        over the language tokens (AlignAtt.lang_id, simul_whisper.py:266-292), then restart the token state with
        the detected language's tokenizer."""
        if not (self.cfg.language == "auto" and self.detected_language is None and self.first_timestamp):
            return None
        if self.segments_len() - self.first_timestamp < 2.0:
            return None
        tok = self.tok
        logits, _ = decoder_forward(self.sd, self.dims, torch.tensor([[tok.sot]]), enc, DecoderCache(self.dims.n_text_layer))
        logits = logits[:, 0].clone()
        mask = torch.ones(logits.shape[-1], dtype=torch.bool)
        mask[list(tok.all_language_tokens)] = False
        logits[:, mask] = -np.inf
        probs = logits.softmax(dim=-1)[0]
        table = {c: float(probs[j]) for j, c in zip(tok.all_language_tokens, tok.all_language_codes)}
        top_lan, _ = max(table.items(), key=lambda kv: kv[1])
        self.tok = self.tokenizer_factory(top_lan)
        self.last_attend_frame = -self.cfg.rewind_threshold
        self.cumulative_time_offset = 0.0
        self._init_tokens()
        self._init_context()
        self.detected_language = top_lan
        return sorted(table.items(), key=lambda kv: -kv[1])[:3]

    # -- the AlignAtt call -----------------------------------------------------------------
    @torch.no_grad()
    def infer(self, is_last=False) -> List[Word]:
        cfg, tok = self.cfg, self.tok
        if not self.segments or self.segments_len() < cfg.audio_min_len:
            return []
        audio = torch.cat(self.segments) if len(self.segments) > 1 else self.segments[0]
        mel, content_mel_len = encoder_input_from_audio(audio, self.filters)
        enc = encoder_forward(self.sd, self.dims, mel)
        lang_top = self._detect_language_if_needed(enc)
        tok = self.tok
        self._trim_context()
        cur = torch.tensor(self._current_tokens(), dtype=torch.long)
        fire = self._fire_at_boundary(enc[0, :content_mel_len])
        rec = {"n_samples": int(audio.shape[0]), "content_mel_len": content_mel_len, "mel": mel,
               "enc": enc, "prefill_tokens": cur[0].tolist(), "steps": [], "fire": fire, "lang_top": lang_top}
        self.trace.append(rec)

        sum_logprobs = torch.zeros(cfg.beam_size)
        cache = DecoderCache(self.dims.n_text_layer)
        self.updater.reset()
        completed = False
        len_before = cur.shape[1]
        stamps: List[float] = []
        kept: List[List[torch.Tensor]] = []
        secs = self.segments_len()
        max_tokens = max(50, int(secs * 15 * 1.5))
        produced = 0
        new_segment = True
        while not completed and cur.shape[1] < self.max_text_len:
            produced += 1
            if produced > max_tokens:                                       # align_att_base.py:208-214
                cur = cur[:, :len_before]
                break
            feed = cur if new_segment else cur[:, -1:]
            logits, cross = decoder_forward(self.sd, self.dims, feed, enc, cache)
            kept.append(cross)
            kept = kept[-16:]                                               # align_att_base.py:221-224
            step = {"fed": feed.shape[1]}
            rec["steps"].append(step)
            if new_segment and tok.no_speech is not None:                   # simul_whisper.py:370-377
                p = logits[:, self.sot_index, :].float().softmax(dim=-1)[:, tok.no_speech].tolist()
                step["no_speech_prob"] = p[0]
                if p[0] > cfg.nonspeech_prob:
                    break
            logits = logits[:, -1, :]
            step["raw_logits"] = logits.clone()
            if new_segment:
                logits[:, tok.encode(" ") + [tok.eot]] = -np.inf              # simul_whisper.py:379-381
            new_segment = False
            logits[:, self.suppress] = -np.inf                              # whisper/decoding.py:427-432
            for t, amount in dry_penalties(cur[0].tolist(), tok.eot).items():
                logits[:, t] = logits[:, t] - amount
            cur, completed = self.updater.update(cur, logits, sum_logprobs, cache)
            attn = alignatt_attention(kept, self.align_heads, self.dims.n_text_layer,
                                      content_mel_len, cfg.beam_size)
            frames = torch.argmax(attn[:, -1, :], dim=-1)                    # simul_whisper.py:435-437
            frame = frames[0].item()
            step.update(token=int(cur[0, -1]), completed=bool(completed), frame=frame,
                        attn_last=attn[0, -1].clone(), sum_logprob=float(sum_logprobs[0]))
            stamps.append(frames.tolist()[0] * 0.02 + self.cumulative_time_offset)
            if completed:
                cur = cur[:, :-1]
                break
            if not is_last and self.last_attend_frame - frame > cfg.rewind_threshold:
                if cur.shape[1] > 1 and cur[0, -2].item() >= DEC_PAD:       # align_att_base.py:264-267
                    self.last_attend_frame = frame
                else:
                    self.last_attend_frame = -cfg.rewind_threshold
                    cur = torch.tensor([sum((list(t) for t in self.tokens), [])] * cfg.beam_size)
                    break
            else:
                self.last_attend_frame = frame
            if content_mel_len - frame <= (4 if is_last else cfg.frame_threshold):
                cur = cur[:, :-1]
                break

        new = cur[0, len_before:].flatten().tolist()
        times = [float(t) for t in stamps[:len(new)]]
        if len(times) < len(new):
            times += [times[-1] if times else 0.0] * (len(new) - len(times))
        if self.pending_tokens:                                             # align_att_base.py:339-369
            new = list(self.pending_tokens) + new
            times = list(self.pending_times) + times
        words, groups = tok.split_to_word_tokens(new)
        # _split_tokens align_att_base.py:326-337 (fire decided before decoding, align_att_base.py:193)
        if fire or is_last:
            hypothesis = new
        else:
            hypothesis = [t for g in groups[:-1] for t in g] if len(words) > 1 else []
        self.tokens.append(list(hypothesis))
        rec["hypothesis"] = list(hypothesis)
        if len(stamps) >= 2 and self.first_timestamp is None:
            self.first_timestamp = stamps[0]
        out = self._timestamped_words(words, groups, times)
        self._hold_incomplete(words, groups, times)
        rec["words"] = [(w.start, w.end, w.text) for w in out]
        return out

    def _timestamped_words(self, words, groups, times) -> List[Word]:       # align_att_base.py:386-441
        out: List[Word] = []
        idx = 0
        bad = "�"
        for word, toks in zip(words, groups):
            n = len(toks)
            if bad in word:
                cleaned = word.replace(bad, "")
                if not cleaned.strip():
                    idx += n
                    continue
                word = cleaned
            wt = times[idx: idx + n]
            if not wt:
                wt = [0.0 if not times else (times[idx] if idx < len(times) else times[-1])]
            start = wt[0]
            nxt = idx + n
            end = times[nxt] if nxt < len(times) else wt[-1] + 0.10
            end = max(end, start + 0.02)
            idx += n
            off = self.global_time_offset
            out.append(Word(round(start, 2) + off, round(end, 2) + off, word, self.speaker, self.detected_language))
        return out

    def _hold_incomplete(self, words, groups, times):                       # align_att_base.py:443-488
        bad = "�"
        if words and bad in words[-1]:
            self.pending_retries += 1
            if self.pending_retries > 2:
                self.pending_tokens, self.pending_times, self.pending_retries = [], [], 0
            elif len(groups[-1]) <= 10:
                self.pending_tokens = groups[-1]
                s = sum(len(g) for g in groups[:-1])
                pt = [float(t) for t in times[s: s + len(groups[-1])]]
                if len(pt) < len(groups[-1]):
                    pt += [pt[-1] if pt else 0.0] * (len(groups[-1]) - len(pt))
                self.pending_times = pt
            else:
                self.pending_tokens, self.pending_times, self.pending_retries = [], [], 0
        else:
            self.pending_tokens, self.pending_times, self.pending_retries = [], [], 0


# ---------------------------------------------------------------------------------------
# a16  per-session wrapper and its output guards   simul_whisper/backend.py:38-268
# ---------------------------------------------------------------------------------------

import re as _re

_WORD = _re.compile(r"[^\W_]+(?:'[^\W_]+)*", _re.UNICODE)


def has_repetition_loop(words: List[str], min_words: int = 12) -> bool:
    """backend.py:128-177: a run of 8 equal words, a tail n-gram repeated >=3 times covering
    >=12 words, or one n-gram making up >=55 % of the recent words."""
    if len(words) < min_words:
        return False
    run = 1
    for a, b in zip(words, words[1:]):
        run = run + 1 if a == b else 1
        if run >= 8:
            return True
    top = min(8, len(words) // 2)
    for size in range(2, top + 1):
        reps, cur = 1, len(words)
        while cur - 2 * size >= 0 and words[cur - size:cur] == words[cur - 2 * size:cur - size]:
            reps += 1
            cur -= size
        if reps >= 3 and reps * size >= min_words:
            return True
    for size in range(2, top + 1):
        counts: Dict[tuple, int] = {}
        for i in range(0, len(words) - size + 1):
            g = tuple(words[i:i + size])
            counts[g] = counts.get(g, 0) + 1
        if counts:
            most = max(counts.values())
            if most >= 4 and most * size >= min_words and most * size / len(words) >= 0.55:
                return True
    return False


class OracleOnlineProcessor:
    """SimulStreamingOnlineProcessor restated over :class:`OracleAlignAtt`."""

    def __init__(self, session: OracleAlignAtt):
        self.model = session
        self.end = 0.0
        self.last_committed_end = 0.0
        self.recent_words: List[str] = []

    def insert_audio_chunk(self, audio, end_time):                           # backend.py:95-102
        self.end = end_time
        self.model.insert_audio(torch.from_numpy(np.asarray(audio)).float())

    def start_silence(self):                                                # backend.py:73-75
        return self.process_iter(is_last=True)

    def end_silence(self, duration, offset):                                # backend.py:77-93
        self.end += duration
        if duration < 5:
            gap = int(16000 * duration)
            if gap > 0:
                self.model.insert_audio(torch.zeros(gap))
        else:
            self.model.refresh_segment(complete=True)
            self.model.global_time_offset = duration + offset
            self.last_committed_end = max(self.last_committed_end, self.model.global_time_offset)
            self.recent_words = []

    def new_speaker(self, speaker, start):                                  # backend.py:104-115
        out = self.process_iter(is_last=True)
        self.model.refresh_segment(complete=True)
        self.model.speaker = speaker
        self.model.global_time_offset = start
        self.last_committed_end = max(self.last_committed_end, start)
        self.recent_words = []
        return out

    def _reset(self):                                                       # backend.py:179-184
        self.model.refresh_segment(complete=True)
        self.model.global_time_offset = max(self.last_committed_end, self.end)
        self.recent_words = []

    def _stable(self, words: List[Word]) -> List[Word]:                     # backend.py:186-225
        keep: List[Word] = []
        last_end = self.last_committed_end
        for w in words:
            s, e = float(w.start or 0.0), float(w.end or (w.start or 0.0))
            if e < s or e <= self.last_committed_end + 0.05:
                continue
            if keep and last_end - e > 0.75:
                continue
            keep.append(w)
            last_end = max(last_end, e)
        return keep

    def process_iter(self, is_last=False):                                  # backend.py:227-268
        try:
            words = self.model.infer(is_last=is_last)
        except Exception:                                                   # backend.py:266-268
            return [], self.end
        if not words:
            return [], self.end
        if self.model.cfg.language == "auto" and words[0].detected_language is None:   # backend.py:239-241
            return [], self.end                  # held in the display buffer until the language is known
        stable = self._stable(words)
        if not stable:
            if self.last_committed_end - max(float(w.end or 0.0) for w in words) > 1.0:
                self._reset()
            return [], self.end
        spoken = [x for w in stable for x in _WORD.findall((w.text or "").casefold())]
        if has_repetition_loop(self.recent_words + spoken):
            self._reset()
            return [], self.end
        self.last_committed_end = max(self.last_committed_end, max(float(w.end or 0.0) for w in stable))
        self.recent_words = (self.recent_words + spoken)[-80:]
        return stable, self.end

"""Seeded synthetic weights and audio for parity tests and benchmarks.

There is no Whisper checkpoint (and no network) where this code is built and measured, so
parity and timing use seeded random weights of the exact architecture plus synthetic 16 kHz
audio, as SURVEY.md section 8(d) prescribes.  Everything here is numpy-only and depends on
``numpy.random.Generator(PCG64(seed))`` so the CPU oracle, the golden-fixture generator and
the GPU box regenerate bit-identical inputs.

Weight names follow the reference checkpoint naming that ``load_model`` produces
(whisperlivekit/whisper/__init__.py:466-596; modules in whisperlivekit/whisper/model.py).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .dims import ModelDims, SAMPLE_RATE

EOT_EN = 50256  # <|endoftext|> in both vocabularies


def sinusoid_table(length: int, channels: int, max_timescale: float = 10000.0) -> np.ndarray:
    """Encoder positional table, cat(sin, cos) of log-spaced timescales (whisper/model.py:62-68)."""
    half = channels // 2
    inc = np.float32(np.log(max_timescale) / (half - 1))
    inv = np.exp(-inc * np.arange(half, dtype=np.float32)).astype(np.float32)
    t = np.arange(length, dtype=np.float32)[:, None] * inv[None, :]
    return np.concatenate([np.sin(t), np.cos(t)], axis=1).astype(np.float32)


def synth_state_dict(dims: ModelDims, seed: int = 0, eot_gain: float = 3.0) -> Dict[str, np.ndarray]:
    """Random fp32 parameters with the reference's names and shapes.

    Linear/conv weights ~ N(0, 1/fan_in), biases ~ N(0, 0.02^2), LayerNorm gains ~ 1 + N(0, 0.1^2),
    embeddings ~ N(0, 0.1^2).  The <|endoftext|> embedding row is scaled by ``eot_gain`` so
    that end-of-text wins the arg-max now and then and the decoder's "completed" branch
    (whisper/decoding.py:343-376) is exercised by random-weight runs.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, np.ndarray] = {}

    def normal(shape, std):
        return (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

    def linear(prefix, n_out, n_in, bias=True):
        sd[prefix + ".weight"] = normal((n_out, n_in), 1.0 / np.sqrt(n_in))
        if bias:
            sd[prefix + ".bias"] = normal((n_out,), 0.02)

    def layernorm(prefix, n):
        sd[prefix + ".weight"] = (1.0 + normal((n,), 0.1)).astype(np.float32)
        sd[prefix + ".bias"] = normal((n,), 0.02)

    def attention(prefix, n):
        linear(prefix + ".query", n, n)
        linear(prefix + ".key", n, n, bias=False)
        linear(prefix + ".value", n, n)
        linear(prefix + ".out", n, n)

    def block(prefix, n, cross):
        attention(prefix + ".attn", n)
        layernorm(prefix + ".attn_ln", n)
        if cross:
            attention(prefix + ".cross_attn", n)
            layernorm(prefix + ".cross_attn_ln", n)
        linear(prefix + ".mlp.0", 4 * n, n)
        linear(prefix + ".mlp.2", n, 4 * n)
        layernorm(prefix + ".mlp_ln", n)

    da, dt = dims.n_audio_state, dims.n_text_state
    sd["encoder.conv1.weight"] = normal((da, dims.n_mels, 3), 1.0 / np.sqrt(3 * dims.n_mels))
    sd["encoder.conv1.bias"] = normal((da,), 0.02)
    sd["encoder.conv2.weight"] = normal((da, da, 3), 1.0 / np.sqrt(3 * da))
    sd["encoder.conv2.bias"] = normal((da,), 0.02)
    sd["encoder.positional_embedding"] = sinusoid_table(dims.n_audio_ctx, da)
    for i in range(dims.n_audio_layer):
        block(f"encoder.blocks.{i}", da, cross=False)
    layernorm("encoder.ln_post", da)

    emb = normal((dims.n_vocab, dt), 0.1)
    emb[EOT_EN] *= np.float32(eot_gain)
    sd["decoder.token_embedding.weight"] = emb
    sd["decoder.positional_embedding"] = normal((dims.n_text_ctx, dt), 0.1)
    for i in range(dims.n_text_layer):
        block(f"decoder.blocks.{i}", dt, cross=True)
    layernorm("decoder.ln", dt)
    return sd


# ---------------------------------------------------------------------------------------
# audio
# ---------------------------------------------------------------------------------------

def speech_like(seconds: float, seed: int = 0, sr: int = SAMPLE_RATE) -> np.ndarray:
    """"Speech-like" test signal: three formant-band harmonics of a gliding f0
    with a 4 Hz syllabic envelope, 0.3 s gaps every ~3 s, plus -40 dB white noise; peak 0.5."""
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    n = int(round(seconds * sr))
    t = np.arange(n, dtype=np.float64) / sr
    f0 = 110.0 + 110.0 * (0.5 + 0.5 * np.sin(2 * np.pi * (0.13 + 0.01 * seed) * t + seed))
    phase = 2 * np.pi * np.cumsum(f0) / sr
    sig = np.zeros(n, dtype=np.float64)
    for k, (formant, bw) in enumerate(((700.0, 130.0), (1220.0, 70.0), (2600.0, 160.0))):
        for h in range(1, 40):
            amp = np.exp(-0.5 * ((h * f0 - formant) / (2.5 * bw)) ** 2)
            sig += amp * np.sin(h * phase + 0.7 * k)
    env = 0.55 + 0.45 * np.sin(2 * np.pi * 4.0 * t + 0.3 * seed)
    gap_phase = (t + 0.37 * seed) % 3.0
    env = env * (gap_phase > 0.3)
    sig = sig * env
    peak = np.max(np.abs(sig)) or 1.0
    sig = 0.5 * sig / peak
    sig = sig + 0.005 * rng.standard_normal(n)
    return sig.astype(np.float32)


def white_noise(seconds: float, seed: int = 0, sr: int = SAMPLE_RATE, sigma: float = 0.1) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(2000 + seed))
    return (sigma * rng.standard_normal(int(round(seconds * sr)))).astype(np.float32)


def to_pcm16_roundtrip(x: np.ndarray) -> np.ndarray:
    """float -> int16 PCM -> float32/32768, the conversion AudioProcessor applies to wire audio
    (whisperlivekit/audio_processor.py:416-418)."""
    pcm = np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)
    return (pcm.astype(np.float32) / np.float32(32768.0)).astype(np.float32)

"""Shared plumbing for the parity tests: golden fixtures, seeded inputs, oracle sessions."""
import json
import os
from functools import lru_cache

import numpy as np
import torch

from whisperlivekit_amd import synth
from whisperlivekit_amd.dims import ALIGNMENT_HEADS, MODEL_DIMS
from whisperlivekit_amd.melbank import mel_filterbank
from whisperlivekit_amd.tokenizer import get_tokenizer

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PROBE_IDS = np.arange(7, 51864, 101)


def golden_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def golden_json(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def mel_case_audio(meta):
    n, kind, seed = meta["n"], meta["kind"], meta["seed"]
    if kind == "speech":
        return synth.to_pcm16_roundtrip(synth.speech_like(n / 16000, seed))[:n]
    if kind == "noise":
        return synth.to_pcm16_roundtrip(synth.white_noise(max(n, 16) / 16000, seed))[:n]
    return np.zeros(n, np.float32)


def expand_mel_golden(arr, meta):
    """-> list of (frame_lo, frame_hi, values[n_mels, hi-lo]) windows stored for this case."""
    if meta["windows"] is None:
        return [(0, meta["keep"], arr)]
    return [(w, w + 96, arr[:, i * 96:(i + 1) * 96]) for i, w in enumerate(meta["windows"])]


@lru_cache(maxsize=None)
def synth_sd(name, seed=0):
    return synth.synth_state_dict(MODEL_DIMS[name], seed)


@lru_cache(maxsize=None)
def oracle_sd(name, seed=0):
    from oracle import whisper_oracle as wo
    return wo.to_torch_state_dict(synth_sd(name, seed))


def stream_audio(case_name):
    """The audio each stream_*.json case was generated from (scripts/gen_golden.py:gen_streams)."""
    a12 = lambda: synth.to_pcm16_roundtrip(synth.speech_like(12.0, 0))
    table = {
        "micro_12s": lambda: a12(),
        "micro_34s_evict": lambda: synth.to_pcm16_roundtrip(synth.speech_like(34.0, 1)),
        "micro_beam2": lambda: a12()[:96000],
        "micro_neverfire": lambda: a12()[:96000],
        "micro_nospeech": lambda: a12()[:48000],
        "micro_events": lambda: a12(),
        "micro_noise_ragged": lambda: synth.to_pcm16_roundtrip(synth.white_noise(6.0, 3)),
        "tiny_6s": lambda: synth.to_pcm16_roundtrip(synth.speech_like(6.0, 2)),
        "base_4s": lambda: synth.to_pcm16_roundtrip(synth.speech_like(6.0, 2))[:64000],
        "micro_cif": lambda: a12()[:128000],
        "micro_prompt": lambda: a12(),
        "micro_minlen_beam3": lambda: a12()[:112000],
        "micromulti_auto": lambda: a12()[:128000],
    }
    return table[case_name]()


def resolve_cfg(cfg_over):
    """Stream fixtures name files of tests/golden as ``golden:<file>`` (scripts/gen_golden.py:run_stream)."""
    over = dict(cfg_over or {})
    if str(over.get("cif_ckpt_path") or "").startswith("golden:"):
        over["cif_ckpt_path"] = os.path.join(GOLDEN, over["cif_ckpt_path"][len("golden:"):])
    return over


def asr_kwargs(cfg_over):
    """Golden cfg overrides (AlignAttConfig field names) -> HipSimulStreamingASR keyword names."""
    over = resolve_cfg(cfg_over)
    kw = {}
    if "beam_size" in over:
        kw["beams"] = over.pop("beam_size")
    if "language" in over:
        kw["lan"] = over.pop("language")
    kw.update(over)
    return kw


def make_oracle_session(model_name, cfg_over=None, seed=0):
    from oracle import whisper_oracle as wo
    dims = MODEL_DIMS[model_name]
    cfg = wo.OracleConfig(**resolve_cfg(cfg_over))
    factory = lambda lang: get_tokenizer(dims.is_multilingual, num_languages=dims.num_languages,
                                         language=lang if dims.is_multilingual else None, synthetic=True)
    return wo.OracleAlignAtt(oracle_sd(model_name, seed), dims, ALIGNMENT_HEADS[model_name], factory("en"),
                             mel_filterbank(dims.n_mels), cfg, tokenizer_factory=factory)

"""Shared plumbing for the parity tests: golden fixtures, seeded inputs, oracle sessions."""
import json
import os
import re
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
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path) and os.path.exists(path + ".gz"):
        path += ".gz"
    if path.endswith(".gz"):
        import gzip
        with gzip.open(path, "rt") as fh:
            return json.load(fh)
    with open(path) as fh:
        return json.load(fh)


def golden_exists(name):
    path = os.path.join(GOLDEN, name)
    return os.path.exists(path) or os.path.exists(path + ".gz")


TIE_EPS = 2e-5   # a reference decision whose winning margin is below this is a tie in fp32


def compare_decisions(g, decision_log, emitted=None):
    """Non-asserting form of test_oracle_golden.check_stream_against_golden for bench.py and the concurrency test:
    ``decision_log`` is HipAlignAttHooks.decision_log ([content_mel_len, [(token, frame), ...]] per infer), ``emitted``
    the per-chunk word lists [(start, end, text), ...].  Every decode decision (token id of beam 0, attended frame) is
    compared with the reference's; a mismatch where the reference's own winning margin is below TIE_EPS is reported as
    a tie divergence (this pass stops there, as the stream legitimately takes another path; ``run_resynced`` replays
    the stream with the reference's choice forced at that step and compares the rest), anything else as a mismatch.
    -> dict(decisions, identical, calls, tie_divergence, mismatch, words_identical)."""
    out = dict(decisions=0, identical=0, calls=0, tie_divergence=None, mismatch=None, words_identical=None)
    stop = False
    for ci, ref in enumerate(g["calls"]):
        if ci >= len(decision_log):
            out["mismatch"] = out["mismatch"] or [ci, 0, "missing call"]
            break
        cml, steps = decision_log[ci]
        if cml != ref["content_mel_len"]:
            out["mismatch"] = [ci, 0, "content_mel_len"]
            break
        ref_steps = [rs for rs in ref["steps"] if rs.get("token") is not None]
        for si, rs in enumerate(ref_steps):
            out["decisions"] += 1
            if si >= len(steps):
                out["mismatch"] = [ci, si, "missing step"]
                stop = True
                break
            tok, frame = steps[si]
            if tok != rs["token"]:
                margin = rs["lp_top_vals"][0] - rs["lp_top_vals"][1]
                out["tie_divergence" if margin < TIE_EPS else "mismatch"] = [ci, si, "token", margin]
                stop = True
                break
            if frame != rs["frame"]:
                vals = rs["attn_top_vals"]
                margin = vals[0] - vals[1] if len(vals) > 1 else 1.0
                out["tie_divergence" if margin < TIE_EPS else "mismatch"] = [ci, si, "frame", margin]
                stop = True
                break
            out["identical"] += 1
        if stop:
            break
        if len(steps) != len(ref_steps):
            out["mismatch"] = [ci, len(ref_steps), "extra steps"]
            break
        out["calls"] += 1
    if out["mismatch"] is None and out["tie_divergence"] is None and len(decision_log) != len(g["calls"]):
        out["mismatch"] = [len(g["calls"]), 0, "extra calls"]
    if emitted is not None and out["mismatch"] is None and out["tie_divergence"] is None:
        want = [[(round(s, 2), round(e, 2), x) for s, e, x, _sp in ev["tokens"]] for ev in g["events"] if ev["kind"] == "chunk"]
        got = [[(round(s, 2), round(e, 2), x) for s, e, x in words] for words in emitted]
        out["words_identical"] = bool(want == got)
    return out


def reference_decision(g, ci, si):
    """(token, frame) the reference took at decision ``si`` of call ``ci`` - what a teacher-forced replay feeds back."""
    ref_steps = [rs for rs in g["calls"][ci]["steps"] if rs.get("token") is not None]
    return int(ref_steps[si]["token"]), int(ref_steps[si]["frame"])


MAX_RESYNC = 8


def run_resynced(run, g, compare):
    """Re-synchronise after fp32 ties instead of giving up on the rest of the stream.  ``run(teacher)`` replays the whole
    stream with the decisions in ``teacher`` ({(call, step): (token, frame)}) forced to the reference's side
    (HipAlignAttHooks.teacher) and returns whatever ``compare`` needs; ``compare(result)`` -> (call, step, kind[, margin])
    of the first divergence that is inside the tie margin, or None (it raises / reports real mismatches itself).
    Every tie adds one forced decision and one replay, so all decisions behind it are compared as well.
    -> (last result, [tie, ...]); more than MAX_RESYNC ties is an error (ties are 1-ulp events)."""
    teacher, ties = {}, []
    while True:
        result = run(dict(teacher))
        tie = compare(result)
        if tie is None:
            return result, ties
        ci, si = int(tie[0]), int(tie[1])
        if (ci, si) in teacher or len(ties) >= MAX_RESYNC:
            raise AssertionError(f"re-synchronisation does not converge: {ties + [list(tie)]}")
        teacher[(ci, si)] = reference_decision(g, ci, si)
        ties.append(list(tie))


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
        "large_v3_2s": lambda: synth.to_pcm16_roundtrip(synth.speech_like(2.0, 5)),
        "micro_single_35s": lambda: synth.to_pcm16_roundtrip(synth.speech_like(36.0, 8)),
        "micro_realvocab": lambda: synth.to_pcm16_roundtrip(synth.speech_like(16.0, 6)),
        "micro_realvocab_beam2": lambda: synth.to_pcm16_roundtrip(synth.speech_like(10.0, 7)),
    }
    if case_name.startswith("bench_base_30s_s"):
        return synth.to_pcm16_roundtrip(synth.speech_like(30.0, int(case_name.rsplit("_s", 1)[1])))
    m = re.fullmatch(r"bench_large-v3_(\d+)s_s(\d+)", case_name)
    if m:
        return synth.to_pcm16_roundtrip(synth.speech_like(float(m.group(1)), int(m.group(2))))
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


def real_vocab_dir(tmp_dir):
    """Materialise tests/golden/vocab_gpt2.npz / vocab_multilingual.npz (scripts/gen_golden_vocab.py,
    scripts/gen_golden_tokenizer.py) as ``gpt2.tiktoken`` / ``multilingual.tiktoken`` under ``tmp_dir`` in
    the rank-file format the product loads (WLK_VOCAB_DIR); returns the directory."""
    import base64
    for name in ("gpt2", "multilingual"):
        z = np.load(os.path.join(GOLDEN, f"vocab_{name}.npz"))
        lengths, blob = z["lengths"].astype(np.int64), z["blob"].tobytes()
        ends = np.cumsum(lengths)
        with open(os.path.join(str(tmp_dir), f"{name}.tiktoken"), "wb") as fh:
            for rank, (e, n) in enumerate(zip(ends, lengths)):
                # the multilingual table ends with an EMPTY token spelled "=" in the rank file (rank 50256)
                fh.write((base64.b64encode(blob[e - n:e]) or b"=") + b" " + str(rank).encode() + b"\n")
    return str(tmp_dir)

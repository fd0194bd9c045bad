"""Drop-in check in the build container: the HIP hook class assembled on the REFERENCE's own
AlignAttBase and plugged into the REFERENCE's own SimulStreamingOnlineProcessor (only
_create_alignatt overridden) reproduces the golden streams.  The C ABI is replaced by the CPU fake
session (oracle numerics) because this container has no GPU; what is under test is that the
reference's policy / guard code runs unmodified on top of our hooks.  Skipped without /root/reference."""
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import ref_stubs  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_stubs.reference_available(), reason="reference tree not present")

import helpers as H  # noqa: E402
from fake_session import FakeHipModel  # noqa: E402
from test_oracle_golden import check_stream_against_golden, replay_stream  # noqa: E402
from whisperlivekit_amd.dims import ALIGNMENT_HEADS, MODEL_DIMS  # noqa: E402


def _make(model_name, cfg_over, seed=0, model_factory=None):
    ref_stubs.install(synthetic_vocab=True)
    from whisperlivekit.simul_whisper.config import AlignAttConfig
    from whisperlivekit.timed_objects import ChangeSpeaker
    from whisperlivekit_amd.backend import reference_online_processor_class
    kw = dict(tokenizer_is_multilingual=not model_name.endswith(".en"), segment_length=0.5, frame_threshold=25,
              language="en", audio_max_len=30.0, audio_min_len=0.0, cif_ckpt_path=None, decoder_type="beam",
              beam_size=1, task="transcribe", never_fire=False, init_prompt=None, max_context_tokens=None,
              static_init_prompt=None)
    over = H.resolve_cfg(cfg_over)
    nonspeech = over.pop("nonspeech_prob", None)
    kw.update(over)
    cfg = AlignAttConfig(**kw)
    if nonspeech is not None:
        cfg.nonspeech_prob = nonspeech
    if model_factory is not None:         # tests/test_gpu_reference_dropin.py: a real HipWhisperModel
        fake = model_factory(model_name, seed)
    else:
        fake = FakeHipModel(MODEL_DIMS[model_name], H.oracle_sd(model_name, seed), ALIGNMENT_HEADS[model_name])
    asr = types.SimpleNamespace(cfg=cfg, hip_model=fake, shared_model=fake, use_full_mlx=False, mlx_encoder=None,
                                fw_encoder=None, tokenizer=None)
    proc = reference_online_processor_class()(asr)
    # numeric trace for the comparison (same wrapper idea as tests/test_policy_golden.py)
    from test_policy_golden import RecordingProcessor
    proc.trace = []
    RecordingProcessor._wrap = None
    import numpy as np
    m = proc.model
    enc0, logit0, ns0, upd0, fr0 = m._encode, m._get_logits_and_cross_attn, m._check_no_speech, \
        m._update_tokens, m._get_attended_frames

    def _encode(segs):
        out = enc0(segs)
        proc.trace.append(dict(content_mel_len=out[1], prefill_tokens=None, steps=[]))
        return out

    def _logits(tokens, enc):
        rec = proc.trace[-1]
        if rec["prefill_tokens"] is None:
            rec["prefill_tokens"] = np.asarray(tokens)[0].tolist()
        rec["steps"].append(dict(fed=int(np.asarray(tokens).shape[1])))
        return logit0(tokens, enc)

    def _ns(logits):
        r = ns0(logits)
        proc.trace[-1]["steps"][-1]["no_speech_prob"] = m.last_no_speech_prob
        return r

    def _upd(tokens, logits, slp):
        new, done = upd0(tokens, logits, slp)
        proc.trace[-1]["steps"][-1].update(token=int(new[0, -1]), completed=bool(done), sum_logprob=float(slp[0]))
        return new, done

    def _fr(attn):
        frames, first = fr0(attn)
        proc.trace[-1]["steps"][-1]["frame"] = first
        return frames, first

    m._encode, m._get_logits_and_cross_attn, m._check_no_speech = _encode, _logits, _ns
    m._update_tokens, m._get_attended_frames = _upd, _fr
    orig_new_speaker = proc.new_speaker
    proc.new_speaker = lambda speaker, start: orig_new_speaker(ChangeSpeaker(speaker=speaker, start=start))
    return proc


@pytest.mark.parametrize("case", ["micro_12s", "micro_34s_evict", "micro_beam2", "micro_neverfire", "micro_events",
                                  "micro_cif", "micromulti_auto", "micro_prompt", "micro_minlen_beam3", "micro_single_35s"])
def test_reference_policy_runs_unmodified_on_hip_hooks(case):
    g, proc, got = replay_stream(case, _make)
    from whisperlivekit.simul_whisper.align_att_base import AlignAttBase
    assert isinstance(proc.model, AlignAttBase)
    assert type(proc.model).infer is AlignAttBase.infer          # the policy is the reference's own code
    check_stream_against_golden(g, proc.trace, got)

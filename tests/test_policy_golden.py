"""Host logic of the product (policy.py + align_att.py + backend.py) replayed against the reference's
golden streams on CPU, with tests/fake_session.py standing in for the C ABI (oracle numerics)."""
import numpy as np
import pytest

import helpers as H
from fake_session import FakeHipModel
from test_oracle_golden import STREAMS, replay_stream
from whisperlivekit_amd import policy as P
from whisperlivekit_amd.backend import HipSimulStreamingASR, HipSimulStreamingOnlineProcessor, has_repetition_loop
from whisperlivekit_amd.dims import ALIGNMENT_HEADS, MODEL_DIMS


class RecordingProcessor(HipSimulStreamingOnlineProcessor):
    """Adds the per-call numeric trace the golden comparison wants (tokens, frames, sums)."""

    def __init__(self, asr):
        super().__init__(asr)
        self.trace = []
        m = self.model
        m.use_device_loop = False        # the trace below hangs on the per-token hooks
        enc0, logit0, ns0, upd0, fr0 = m._encode, m._get_logits_and_cross_attn, m._check_no_speech, \
            m._update_tokens, m._get_attended_frames

        def _encode(segs):
            out = enc0(segs)
            self.trace.append(dict(content_mel_len=out[1], prefill_tokens=None, steps=[]))
            return out

        def _logits(tokens, enc):
            rec = self.trace[-1]
            if rec["prefill_tokens"] is None:
                rec["prefill_tokens"] = np.asarray(tokens)[0].tolist()
            rec["steps"].append(dict(fed=int(np.asarray(tokens).shape[1])))
            return logit0(tokens, enc)

        def _ns(logits):
            r = ns0(logits)
            self.trace[-1]["steps"][-1]["no_speech_prob"] = m.last_no_speech_prob
            return r

        def _upd(tokens, logits, slp):
            new, done = upd0(tokens, logits, slp)
            self.trace[-1]["steps"][-1].update(token=int(new[0, -1]), completed=bool(done),
                                               sum_logprob=float(slp[0]))
            return new, done

        def _fr(attn):
            frames, first = fr0(attn)
            self.trace[-1]["steps"][-1]["frame"] = first
            return frames, first

        fire0, lang0 = m.fire_at_boundary, m.lang_id

        def _fire(feature):
            r = bool(fire0(feature))
            self.trace[-1]["fire"] = r
            return r

        def _lang(enc):
            toks, probs = lang0(enc)
            self.trace[-1]["lang_top"] = sorted(probs[0].items(), key=lambda kv: -kv[1])[:3]
            return toks, probs

        m.fire_at_boundary, m.lang_id = _fire, _lang
        m._encode, m._get_logits_and_cross_attn, m._check_no_speech = _encode, _logits, _ns
        m._update_tokens, m._get_attended_frames = _upd, _fr

    def new_speaker(self, speaker, start):           # replay_stream passes (speaker, start)
        return super().new_speaker(P.ChangeSpeaker(speaker=speaker, start=start))


def make_fake_processor(model_name, cfg_over, seed=0):
    dims = MODEL_DIMS[model_name]
    fake = FakeHipModel(dims, H.oracle_sd(model_name, seed), ALIGNMENT_HEADS[model_name])
    asr = HipSimulStreamingASR(model_name, hip_model=fake, **H.asr_kwargs(cfg_over))
    return RecordingProcessor(asr)


@pytest.mark.parametrize("case", STREAMS)
def test_product_host_logic_matches_reference(case):
    from test_oracle_golden import check_stream_against_golden
    g, proc, got = replay_stream(case, make_fake_processor)
    check_stream_against_golden(g, proc.trace, got)
    last = [ev for ev, _, _ in got if ev["kind"] == "chunk"][-1]
    assert proc.model.state.context.text == last["context"]
    assert proc.model.state.last_attend_frame == last["last_attend_frame"]
    assert abs(proc.model.state.cumulative_time_offset - last["cumulative_time_offset"]) < 1e-9


BEAM1_STREAMS = [c for c in STREAMS if "beam" not in c]


def make_loop_processor(model_name, cfg_over, seed=0):
    """The product's processor with the decode loop inside the library (wlk_decode_until_stop's host logic through
    wlk_job_*; oracle numerics from the CPU fake session)."""
    dims = MODEL_DIMS[model_name]
    fake = FakeHipModel(dims, H.oracle_sd(model_name, seed), ALIGNMENT_HEADS[model_name])
    asr = HipSimulStreamingASR(model_name, hip_model=fake, **H.asr_kwargs(cfg_over))

    class P2(HipSimulStreamingOnlineProcessor):
        def new_speaker(self, speaker, start):
            return super().new_speaker(P.ChangeSpeaker(speaker=speaker, start=start))

    proc = P2(asr)
    proc.model.decision_log = []
    assert proc.model.device_loop_available()
    return proc


def check_loop_stream(g, proc, got):
    emitted = [[(t.start, t.end, t.text) for t in toks] for ev, toks, _ in got if ev["kind"] == "chunk"]
    r = H.compare_decisions(g, proc.model.decision_log, emitted)
    assert r["mismatch"] is None and r["tie_divergence"] is None, r
    assert r["identical"] == r["decisions"] and r["words_identical"], r
    for ev, toks, upto in got:                      # silence / speaker events emit words too
        assert [(round(t.start, 2), round(t.end, 2), t.text, t.speaker) for t in toks] == \
               [(round(s, 2), round(e, 2), x, sp) for s, e, x, sp in ev["tokens"]]
        assert abs(upto - ev["upto"]) < 1e-9
    last = [ev for ev, _, _ in got if ev["kind"] == "chunk"][-1]
    assert proc.model.state.context.text == last["context"]
    assert proc.model.state.last_attend_frame == last["last_attend_frame"]
    assert abs(proc.model.state.cumulative_time_offset - last["cumulative_time_offset"]) < 1e-9
    hyp = [t[0].tolist() for t in proc.model.state.tokens[1:]]
    assert (hyp[-1] if hyp else []) == last["hypothesis"]
    return r


@pytest.mark.parametrize("case", BEAM1_STREAMS)
def test_library_decode_loop_matches_reference(case):
    """the per-token loop (budget, no-speech stop, suppression, DRY penalty, beam-1 update, rewind /
    frame-threshold stops) runs inside the library; same decisions, words and end state as the reference."""
    g, proc, got = replay_stream(case, make_loop_processor)
    r = check_loop_stream(g, proc, got)
    assert r["calls"] == len(g["calls"])


def test_repetition_detectors():
    assert not has_repetition_loop(["a"] * 7)
    assert has_repetition_loop(["x"] * 4 + ["a"] * 8)
    assert has_repetition_loop(("the cat sat " * 5).split())
    assert not has_repetition_loop("one two three four five six seven eight nine ten eleven twelve".split())
    words = ("a b " * 4 + "c d e f").split()
    assert not has_repetition_loop(words)      # 4 x 2 words < 12-word minimum
    assert has_repetition_loop(("a b c " * 4 + "d e f g").split())   # 4 x 3 = 12 words, 75 % coverage


def test_pcm16_wire_format_stream_equals_float_stream():
    """s16le chunks straight into the backend (insert_pcm16_chunk) give exactly the stream the
    float path gives after AudioProcessor.convert_pcm_to_float (audio_processor.py:416-418)."""
    from whisperlivekit_amd import synth
    audio = synth.speech_like(6.0, 0)
    pcm = np.clip(np.round(audio * 32768.0), -32768, 32767).astype(np.int16)
    as_float = pcm.astype(np.float32) / 32768.0
    outs = []
    for mode in ("float", "pcm16", "bytes"):
        proc = make_fake_processor("micro.en", {})
        words = []
        for lo in range(0, len(pcm), 8000):
            t_end = min(lo + 8000, len(pcm)) / 16000
            if mode == "float":
                proc.insert_audio_chunk(as_float[lo:lo + 8000], t_end)
            elif mode == "pcm16":
                proc.insert_pcm16_chunk(pcm[lo:lo + 8000], t_end)
            else:
                proc.insert_pcm16_chunk(pcm[lo:lo + 8000].tobytes(), t_end)
            toks, _ = proc.process_iter()
            words += [(t.start, t.end, t.text) for t in toks]
        outs.append((words, [(r["content_mel_len"], [s.get("token") for s in r["steps"]]) for r in proc.trace]))
    assert outs[0] == outs[1] == outs[2] and len(outs[0][1]) == 12
    with pytest.raises(TypeError):
        make_fake_processor("micro.en", {}).insert_pcm16_chunk(as_float[:100], 0.1)

"""Differential test of the library's decode-loop host logic (csrc/loop.hip: DecodeJob through wlk_job_*) against the
Python loop of policy.AlignAttPolicy._decode_loop (itself pinned against the reference's AlignAttBase.infer by the
golden streams) on random scripts: random top-2 candidates and attended frames, repeated n-grams (DRY penalty),
end-of-text wins, attention rewinds, frame-threshold and token-budget stops, special tokens in second-to-last place."""
import ctypes as C
import math
from types import SimpleNamespace

import numpy as np
import pytest

from whisperlivekit_amd import _lib
from whisperlivekit_amd import policy as P
from whisperlivekit_amd.align_att import LazyLogits

EOT = 50256
vp = lambda a: a.ctypes.data_as(C.c_void_p)


class ScriptedPolicy(P.AlignAttPolicy):
    """AlignAttPolicy._decode_loop over hooks that replay a script of (top-2 log-probs, top-2 ids, frame)."""

    def __init__(self, script, cfg, state, max_text_len, suppress, blank, no_speech_prob):
        self.script, self.cfg, self.state, self.max_text_len = script, cfg, state, max_text_len
        self.tokenizer = SimpleNamespace(eot=EOT)
        self.suppress, self.blank, self.nsp = suppress, blank, no_speech_prob
        self.step = 0
        self.adjust_log = []
        self._upd = P.BeamUpdate(1, EOT)

    def _init_sum_logprobs(self):
        return np.zeros(1, np.float32)

    def _get_logits_and_cross_attn(self, fed, enc):
        return LazyLogits(1, fed.shape[1], 51864), None

    def _evaluate(self, x):
        pass

    def _check_no_speech(self, logits):
        return self.nsp > self.cfg.nonspeech_prob

    def _suppress_blank_tokens(self, logits):
        logits.add(-1, self.blank, -math.inf)
        return logits

    def _apply_token_suppression(self, logits):
        logits.add(-1, self.suppress, -math.inf)
        return logits

    def _update_tokens(self, tokens, logits, slp):
        self.adjust_log.append({t: d for (_, t), d in logits.adjust.items()})
        lp, ids, self._frame = self.script[self.step]
        self.step += 1
        new, done, _ = self._upd.update(np.asarray(tokens), np.asarray([lp], np.float32), np.asarray([ids]), slp)
        return new, done

    def _process_cross_attention(self, window, cml):
        return None

    def _get_attended_frames(self, attn):
        return [self._frame], self._frame

    def _is_special_token(self, tokens):
        return int(tokens[0, -2]) >= P.DEC_PAD

    def _rewind_tokens(self):
        return np.asarray([self.state.prompt], dtype=np.int64)


def run_job(tokens, params, suppress, blank, script, nsp):
    lib = _lib.load()
    t = np.asarray(tokens, np.int64)
    sup, bl = np.asarray(suppress, np.int32), np.asarray(blank, np.int32)
    job = C.c_void_p()
    _lib.check(lib.wlk_job_create(C.byref(params), vp(t), t.size, vp(sup), sup.size, vp(bl), bl.size, C.byref(job)))
    adjust_log = []
    step = 0
    while True:
        n_feed = C.c_int32()
        _lib.check(lib.wlk_job_begin_step(job, C.byref(n_feed)))
        if n_feed.value == 0:
            break
        assert n_feed.value == (len(tokens) if step == 0 else 1)
        if step == 0:
            stops = C.c_int32()
            _lib.check(lib.wlk_job_no_speech(job, nsp, C.byref(stops)))
            if stops.value:
                break
        ids_p, dl_p, n = C.POINTER(C.c_int32)(), C.POINTER(C.c_float)(), C.c_int32()
        _lib.check(lib.wlk_job_adjustments(job, C.byref(ids_p), C.byref(dl_p), C.byref(n)))
        adjust_log.append({ids_p[i]: dl_p[i] for i in range(n.value)})
        lp, ids, frame = script[step]
        step += 1
        go = C.c_int32()
        _lib.check(lib.wlk_job_consume(job, vp(np.asarray(lp, np.float32)), vp(np.asarray(ids, np.int32)), frame, C.byref(go)))
        if not go.value:
            break
    cap = 600
    res = _lib.LoopResult()
    new, st, sf, ss = np.empty(cap, np.int64), np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float32)
    _lib.check(lib.wlk_job_result(job, C.byref(res), vp(new), vp(st), vp(sf), vp(ss), cap))
    lib.wlk_job_destroy(job)
    return res, new[:res.n_new_tokens].tolist(), sf[:res.n_steps].tolist(), adjust_log


@pytest.mark.parametrize("seed", range(60))
def test_job_equals_python_loop(seed):
    rng = np.random.default_rng(seed)
    alphabet = rng.integers(300, 340, size=rng.integers(2, 6)).tolist()       # few symbols -> repeated n-grams
    n_ctx = int(rng.integers(0, 30))
    prompt = [50257, 50362] + rng.choice(alphabet, int(rng.integers(0, 20))).tolist()
    ctx = ([50361] + rng.choice(alphabet, n_ctx).tolist()) if n_ctx else []
    tokens = ctx + prompt
    is_last = bool(rng.random() < 0.2)
    cml = int(rng.integers(30, 1500))
    max_text_len = int(rng.choice([448, len(tokens) + 3, len(tokens) + 40]))
    budget = int(rng.choice([50, 3, 200]))
    last_attend = int(rng.choice([-200, 0, cml // 2, cml + 300, 900]))
    script = []
    for i in range(500):
        pool = sorted(set(alphabet + ([EOT, 50300] if rng.random() < 0.15 else []) + [299]))
        a, b = (int(x) for x in rng.choice(pool, 2, replace=False))
        lp0 = -float(rng.random())
        frame = int(np.clip(rng.normal(cml * min(1.0, (i + 1) / 12), 40), 0, cml - 1))
        if rng.random() < 0.05:
            frame = int(rng.integers(0, max(1, cml // 4)))                     # a jump back
        script.append(([lp0, lp0 - float(rng.random())], [int(a), int(b)], frame))
    nsp = float(rng.choice([0.0, 0.2, 0.9]))
    suppress = sorted({50257, 50358, 50359, 50360, 50361, 50362, int(alphabet[-1])} if rng.random() < 0.3 else
                      {50257, 50358, 50359, 50360, 50361, 50362})
    blank = [220, EOT]
    cfg = SimpleNamespace(rewind_threshold=200, frame_threshold=int(rng.choice([25, 4, 10])), nonspeech_prob=0.5)
    state = SimpleNamespace(last_attend_frame=last_attend, cumulative_time_offset=1.25, prompt=prompt)
    pol = ScriptedPolicy(script, cfg, state, max_text_len, suppress, blank, nsp)
    out_tokens, stamps = pol._decode_loop(np.asarray([tokens], np.int64), None, cml, is_last, budget)
    want_new = out_tokens[0, len(tokens):].tolist()

    p = _lib.LoopParams(sot_index=len(ctx), is_last=int(is_last), frame_threshold=cfg.frame_threshold, rewind_threshold=200,
                        last_attend_frame=last_attend, max_text_len=max_text_len, budget=budget, eot=EOT,
                        dec_pad=P.DEC_PAD, no_speech_token=50361, no_speech_threshold=0.5, content_mel_len=cml)
    res, new, frames, adj = run_job(tokens, p, suppress, blank, script, nsp)
    assert new == want_new
    assert [f * 0.02 + 1.25 for f in frames] == stamps
    assert res.last_attend_frame == state.last_attend_frame
    assert len(adj) == len(pol.adjust_log)
    for mine, ref in zip(adj, pol.adjust_log):
        assert set(mine) == set(ref)
        for t in ref:
            assert mine[t] == ref[t] or (math.isinf(mine[t]) and math.isinf(ref[t])), (t, mine[t], ref[t])

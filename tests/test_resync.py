"""Re-synchronisation after fp32 ties (round 5): when this backend and the reference land on different sides of a tie
(two candidates closer than fp32 rounding), the parity harness replays the stream with the reference's choice forced at
that one step (HipAlignAttHooks.teacher -> wlk_loop_params.force_* / the per-token hooks) and keeps comparing, instead of
leaving the rest of the stream unchecked.  CPU: the library's host logic (wlk_job_*) and the Python hooks over the
oracle-backed fake session; an artificial divergence stands in for the GPU's 1-ulp one."""
import copy
import ctypes as C

import numpy as np
import pytest

import helpers as H
from test_oracle_golden import check_stream_against_golden, replay_stream
from test_policy_golden import make_fake_processor, make_loop_processor
from whisperlivekit_amd import _lib


def _params(**kw):
    base = dict(sot_index=0, is_last=0, frame_threshold=25, rewind_threshold=200, last_attend_frame=0, max_text_len=448,
                budget=50, eot=50256, dec_pad=50257, no_speech_token=-1, no_speech_threshold=1.0, content_mel_len=1000)
    base.update(kw)
    return _lib.LoopParams(**base)


def _run_job(p, steps):
    """steps: [(lp[2], ids[2], frame)] -> (step_tokens, step_frames, sum_logprob, stop_reason)"""
    lib = _lib.load()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    toks = np.array([50257, 11, 12], np.int64)
    empty = np.zeros(0, np.int32)
    job = C.c_void_p()
    _lib.check(lib.wlk_job_create(C.byref(p), vp(toks), toks.size, vp(empty), 0, vp(empty), 0, C.byref(job)))
    try:
        for lp, ids, frame in steps:
            n = C.c_int32()
            _lib.check(lib.wlk_job_begin_step(job, C.byref(n)))
            assert n.value > 0
            go = C.c_int32()
            _lib.check(lib.wlk_job_consume(job, vp(np.array(lp, np.float32)), vp(np.array(ids, np.int32)), frame, C.byref(go)))
            if not go.value:
                break
        res = _lib.LoopResult()
        new, st, sf, ss = np.empty(64, np.int64), np.empty(64, np.int32), np.empty(64, np.int32), np.empty(64, np.float32)
        _lib.check(lib.wlk_job_result(job, C.byref(res), vp(new), vp(st), vp(sf), vp(ss), 64))
        return st[:res.n_steps].tolist(), sf[:res.n_steps].tolist(), float(res.sum_logprob), int(res.stop_reason)
    finally:
        lib.wlk_job_destroy(job)


def test_job_takes_the_forced_side_of_a_tie():
    steps = [([-1.0, -1.00001], [100, 200], 300), ([-0.5, -2.0], [101, 201], 310), ([-0.25, -3.0], [102, 202], 320)]
    toks, frames, s, _ = _run_job(_params(), steps)
    assert toks == [100, 101, 102] and frames == [300, 310, 320]
    p = _params()
    p.force([(0, 200, 301), (2, -1, 333)])          # step 0: runner-up token + another frame; step 2: the frame only
    toks, frames, s2, _ = _run_job(p, steps)
    assert toks == [200, 101, 102] and frames == [301, 310, 333]
    assert abs(s2 - (-1.00001 - 0.5 - 0.25)) < 1e-6 and abs(s - (-1.0 - 0.5 - 0.25)) < 1e-6
    # a forced token that is not the runner-up, or a winner that is end-of-text, is left alone
    p = _params()
    p.force([(0, 999, -1)])
    assert _run_job(p, steps)[0] == [100, 101, 102]
    p = _params()
    p.force([(0, 200, -1)])
    toks, _, _, stop = _run_job(p, [([-1.0, -1.00001], [50256, 200], 300)])
    assert toks == [200] and stop == _lib.STOP_COMPLETED
    # a forced frame takes part in the stop rules: 1000 - 980 <= 25 ends the loop at step 1
    p = _params()
    p.force([(1, -1, 980)])
    toks, frames, _, stop = _run_job(p, steps)
    assert toks == [100, 101] and frames == [300, 980] and stop == _lib.STOP_FRAME


def test_force_block_is_validated():
    p = _params()
    with pytest.raises(ValueError):
        p.force([(i, -1, 1) for i in range(5)])
    p.n_force = 9
    lib = _lib.load()
    toks = np.array([50257], np.int64)
    job = C.c_void_p()
    rc = lib.wlk_job_create(C.byref(p), toks.ctypes.data_as(C.c_void_p), 1, None, 0, None, 0, C.byref(job))
    assert rc != 0 and b"n_force" in lib.wlk_last_error()


def _with_teacher_and_flip(make, teacher, flip):
    """Factory: hooks take ``teacher``; the session's read-out is off by one frame at decision ``flip`` = (call, step) -
    the stand-in for a GPU result on the other side of a 1-ulp tie."""
    def mk(model_name, cfg_over, seed=0):
        proc = make(model_name, cfg_over, seed)
        m = proc.model
        m.teacher = dict(teacher) or None
        sess, inner = m.session, m.session.select
        count = {"call": -1, "step": 0}
        enc0 = sess.encode

        def encode():
            count["call"] += 1
            count["step"] = 0
            return enc0()

        def select(*a, **k):
            lp, top, frames = inner(*a, **k)
            if (count["call"], count["step"]) == flip:
                frames = np.array(frames)
                frames[0] += 1
            count["step"] += 1
            return lp, top, frames

        sess.encode, sess.select = encode, select
        return proc
    return mk


def _golden_with_tie(case, flip):
    g = copy.deepcopy(H.golden_json(f"stream_{case}.json"))
    ci, si = flip
    rs = [r for r in g["calls"][ci]["steps"] if r.get("token") is not None][si]
    rs["attn_top_vals"] = [rs["attn_top_vals"][0], rs["attn_top_vals"][0] - 1.2e-7]     # a one-ulp margin
    return g


def _first_multi_step_call(case):
    """(call, step) of a decision whose reference margin is far from a tie (the micro model's median-filtered scores
    hold many EXACT ties, which would make the negative half of the test vacuous)."""
    g = H.golden_json(f"stream_{case}.json")
    for ci, c in enumerate(g["calls"]):
        steps = [r for r in c["steps"] if r.get("token") is not None]
        for si, r in enumerate(steps):
            vals = r["attn_top_vals"]
            if ci >= 3 and 1 <= si < len(steps) - 1 and len(vals) > 1 and vals[0] - vals[1] > 1e-2:
                return ci, si
    raise AssertionError("no suitable decision")


N_EVENTS = 14        # a prefix of the stream keeps the CPU oracle's share of this test short


@pytest.mark.parametrize("path", ["hooks", "library_loop"])
def test_stream_is_compared_to_the_end_after_a_tie(path, monkeypatch):
    case = "micro_12s"
    g_real = H.golden_json(f"stream_{case}.json")
    flip = _first_multi_step_call(case)
    assert flip[0] < N_EVENTS - 2
    g_tie = _golden_with_tie(case, flip)
    served = {"g": g_tie}
    original = H.golden_json
    monkeypatch.setattr(H, "golden_json", lambda name: copy.deepcopy(served["g"]) if name == f"stream_{case}.json" else original(name))
    runs = []

    def run(teacher):
        make = make_fake_processor if path == "hooks" else make_loop_processor
        g, proc, got = replay_stream(case, _with_teacher_and_flip(make, teacher, flip), max_events=N_EVENTS)
        runs.append(proc)
        return g, proc, got

    def compare(res):
        g, proc, got = res
        if path == "hooks":
            return check_stream_against_golden(g, proc.trace, got, allow_ties=True)
        emitted = [[(t.start, t.end, t.text) for t in toks] for ev, toks, _ in got if ev["kind"] == "chunk"]
        r = H.compare_decisions(g, proc.model.decision_log, emitted)
        assert r["mismatch"] is None, r
        return r["tie_divergence"]

    (g, proc, got), ties = H.run_resynced(run, g_tie, compare)
    assert len(runs) == 2 and len(ties) == 1 and tuple(ties[0][:2]) == flip and "frame" in ties[0][2]
    if path == "library_loop":
        emitted = [[(t.start, t.end, t.text) for t in toks] for ev, toks, _ in got if ev["kind"] == "chunk"]
        r = H.compare_decisions(g, proc.model.decision_log, emitted)
        assert r["calls"] == len(g["calls"]) and r["identical"] == r["decisions"] and r["words_identical"], r
    # without the tie margin the same divergence is a failure, not something to resynchronise
    served["g"] = g_real
    with pytest.raises(AssertionError):
        H.run_resynced(run, g_real, compare)

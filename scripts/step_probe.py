#!/usr/bin/env python
"""Wall-clock anatomy of one AlignAtt call on the GPU: encode, prefill, and N single-token decode
steps (decode + select with one readback), each timed host-side after a stream sync.  GPU box only."""
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperlivekit_amd import synth  # noqa: E402
from whisperlivekit_amd.engine import HipWhisperModel  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "base.en"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
prefill = int(sys.argv[4]) if len(sys.argv) > 4 else 60
model = HipWhisperModel.synthetic(name, 0)
sess = model.new_session(beam=1)
sess.append(synth.to_pcm16_roundtrip(synth.speech_like(secs, 0)))
rng = np.random.default_rng(0)
toks = np.concatenate([[50257, 50362], rng.integers(300, 40000, prefill - 2)]).astype(np.int64)[None, :]
supp = list(range(50257, 50363))
enc_t, pre_t, step_t, dec_t, sel_t = [], [], [], [], []
for rep in range(4):
    t0 = time.perf_counter()
    cml = sess.encode()
    sess.sync()
    t1 = time.perf_counter()
    sess.decode(toks, first=True, sot_index=0)
    sess.no_speech_prob(50361)
    lp, ids, fr = sess.select([-1] * len(supp), supp, [-np.inf] * len(supp), 2, cml)
    t2 = time.perf_counter()
    if rep:
        enc_t.append(t1 - t0)
        pre_t.append(t2 - t1)
    cur = ids[:, :1].astype(np.int64)
    for i in range(n_steps):
        a = time.perf_counter()
        sess.decode(cur, first=False)
        b = time.perf_counter()
        lp, ids, fr = sess.select([-1] * len(supp), supp, [-np.inf] * len(supp), 2, cml)
        c = time.perf_counter()
        if rep:
            step_t.append(c - a)
            dec_t.append(b - a)
            sel_t.append(c - b)
        cur = ids[:, :1].astype(np.int64)
us = lambda xs: round(1e6 * statistics.median(xs), 1)
print(dict(model=name, audio_s=secs, prefill_tokens=prefill, encode_us=us(enc_t), prefill_call_us=us(pre_t),
           step_us=us(step_t), step_decode_enqueue_us=us(dec_t), step_select_us=us(sel_t),
           step_p90_us=round(1e6 * float(np.percentile(step_t, 90)), 1)))

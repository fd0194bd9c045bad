#!/usr/bin/env python
"""Batch `transcribe` (greedy, temperature 0, word timestamps) over a 30 s stream on base.en: the per-step logit rules on the
host (the logits row read back every step) against wlk_pick_greedy (rules, arg-max and log-probability on the device)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WLK_SYNTHETIC_VOCAB", "1")
from whisperlivekit_amd import synth  # noqa: E402
from whisperlivekit_amd.engine import HipWhisperModel  # noqa: E402
from whisperlivekit_amd.transcribe import release_sessions, transcribe  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "base.en"
model = HipWhisperModel.synthetic(name, 0, device=0)
audio = synth.speech_like(30.0, seed=0)
kw = dict(language="en", temperature=0.0, logprob_threshold=None, compression_ratio_threshold=None, word_timestamps=True)
transcribe(model, audio[:16000], **kw)
out = {}
for rep in range(2):
    for mode in ("0", "1"):
        os.environ["WLK_TRANSCRIBE_DEVICE_RULES"] = mode
        a = time.perf_counter()
        res = transcribe(model, audio, **kw)
        dt = time.perf_counter() - a
        n_tok = sum(len(s["tokens"]) for s in res["segments"])
        out.setdefault(mode, res)
        print(f"{name} rules on the {'device' if mode == '1' else 'host  '}: {dt:.3f} s = {30.0 / dt:7.1f} audio-s/s, {len(res['segments'])} segments, {n_tok} tokens")
same = [s["tokens"] for s in out["0"]["segments"]] == [s["tokens"] for s in out["1"]["segments"]]
worst = max((abs(a["avg_logprob"] - b["avg_logprob"]) for a, b in zip(out["0"]["segments"], out["1"]["segments"])), default=0.0)
print(f"same tokens: {same}; worst avg_logprob difference {worst:.2e}")
release_sessions(model)
model.close()

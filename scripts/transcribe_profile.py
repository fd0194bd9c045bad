#!/usr/bin/env python
"""cProfile of one greedy batch `transcribe` call (30 s, base.en, word timestamps): where the host time goes."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WLK_SYNTHETIC_VOCAB", "1")
from whisperlivekit_amd import synth  # noqa: E402
from whisperlivekit_amd.engine import HipWhisperModel  # noqa: E402
from whisperlivekit_amd.transcribe import release_sessions, transcribe  # noqa: E402

model = HipWhisperModel.synthetic(sys.argv[1] if len(sys.argv) > 1 else "base.en", 0, device=0)
audio = synth.speech_like(30.0, seed=0)
kw = dict(language="en", temperature=0.0, logprob_threshold=None, compression_ratio_threshold=None, word_timestamps=True)
transcribe(model, audio[:16000], **kw)
transcribe(model, audio, **kw)
pr = cProfile.Profile()
pr.enable()
transcribe(model, audio, **kw)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
release_sessions(model)
model.close()

#!/usr/bin/env python
"""8 concurrent base.en streams on ONE GPU (the metric's second half), alone, for rocprofv3: one warm-up pass and one
timed pass of 8 x 30 s.  Prints audio_s/s and the engine statistics.  GPU box only."""
import os
import sys
import time

os.environ.setdefault("WLK_SYNTHETIC_VOCAB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from whisperlivekit_amd import synth  # noqa: E402
from whisperlivekit_amd.backend import HipSimulStreamingASR, HipSimulStreamingOnlineProcessor  # noqa: E402
from whisperlivekit_amd.dims import ALIGNMENT_HEADS, MODEL_DIMS  # noqa: E402
from whisperlivekit_amd.engine import HipWhisperModel  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dims = MODEL_DIMS["base.en"]
model = HipWhisperModel.from_state_dict(dims, synth.synth_state_dict(dims, 0), ALIGNMENT_HEADS["base.en"])
asr = HipSimulStreamingASR("base.en", hip_model=model)
audios = [bench.make_audio("speech", 30.0, s) for s in range(n)]
for rep in range(2):
    procs = [HipSimulStreamingOnlineProcessor(asr) for _ in range(n)]
    t0 = time.perf_counter()
    bench.run_sessions(procs, audios)
    dt = time.perf_counter() - t0
    print(f"pass {rep}: {n} streams, {30.0 * n / dt:.1f} audio_s/s, wall {dt * 1e3:.1f} ms", flush=True)
    for p in procs:
        p.close()
print(model.engine_stats())

"""N diarizer sessions flat out on ONE Sortformer model (config 4's diarizer half without the ASR sessions): where does a
chunk's time go (features / step / host update), how many sessions ride in a stacked step, what is the aggregate rate.

    python scripts/diar_probe8.py [sessions] [seconds]"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from whisperlivekit_amd import synth  # noqa: E402
from whisperlivekit_amd.diarization import HipSortformerDiarizationOnline  # noqa: E402
from whisperlivekit_amd.sortformer import HipSortformerModel  # noqa: E402

import os
n_sess = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
n_asr = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # base.en ASR sessions running beside the diarizers (config 4)
os.environ.setdefault("WLK_SYNTHETIC_VOCAB", "1")
if "GPU_MAX_HW_QUEUES" not in os.environ:
    os.environ["GPU_MAX_HW_QUEUES"] = "2"                      # bench.py's single-process default
model = HipSortformerModel.synthetic()
audios = [synth.speech_like(seconds, s) for s in range(n_sess)]
acc = dict(features=[], step=[], chunk=[])
lock = threading.Lock()
inner_step, inner_feat, inner_pcm = model.step, model.features, model.step_pcm


def step(feats, ctx):
    a = time.perf_counter()
    out = inner_step(feats, ctx)
    with lock:
        acc["step"].append(1e3 * (time.perf_counter() - a))
    return out


def features(pcm):
    a = time.perf_counter()
    out = inner_feat(pcm)
    with lock:
        acc["features"].append(1e3 * (time.perf_counter() - a))
    return out


def step_pcm(pcm, prev, ctx):
    a = time.perf_counter()
    out = inner_pcm(pcm, prev, ctx)
    with lock:
        acc["step"].append(1e3 * (time.perf_counter() - a))
        acc["features"].append(0.0)
    return out


model.step, model.features, model.step_pcm = step, features, step_pcm
if os.environ.get("PROBE_THREE_CALLS") == "1":      # the pre-round-6 call sequence: extractor call, then the step
    model.forward_streaming_step_pcm = None


def stream(audio):
    online = HipSortformerDiarizationOnline(model)
    for lo in range(0, len(audio) - 15999, 16000):
        online.insert_audio_chunk(audio[lo:lo + 16000])
        a = time.perf_counter()
        online.diarize_sync()
        with lock:
            acc["chunk"].append(1e3 * (time.perf_counter() - a))


asr = None
if n_asr:
    from whisperlivekit_amd.backend import HipSimulStreamingASR, HipSimulStreamingOnlineProcessor
    from whisperlivekit_amd.dims import ALIGNMENT_HEADS, MODEL_DIMS
    from whisperlivekit_amd.engine import HipWhisperModel
    wm = HipWhisperModel.from_state_dict(MODEL_DIMS["base.en"], synth.synth_state_dict(MODEL_DIMS["base.en"], 0), ALIGNMENT_HEADS["base.en"], device=0)
    asr = HipSimulStreamingASR("base.en", hip_model=wm)
    asr_audio = [synth.to_pcm16_roundtrip(synth.speech_like(seconds, s)) for s in range(n_asr)]
asr_done = [0.0]


def asr_stream(audio, t0):
    p = HipSimulStreamingOnlineProcessor(asr)
    t_end = 0.0
    for lo in range(0, len(audio), 8000):
        t_end += 0.5
        p.insert_audio_chunk(audio[lo:lo + 8000], t_end)
        p.process_iter()
    with lock:
        asr_done[0] = max(asr_done[0], time.perf_counter() - t0)
    p.close()


stream(audios[0][:3 * 16000])
if n_asr:
    asr_stream(asr_audio[0][:48000], time.perf_counter())
for rep in range(2):
    for v in acc.values():
        v.clear()
    before = model.stats()
    gate = threading.Barrier(n_sess + n_asr)
    ts = [threading.Thread(target=lambda a=a: (gate.wait(), stream(a))) for a in audios]
    t0 = time.perf_counter()
    if n_asr:
        asr_done[0] = 0.0
        ts += [threading.Thread(target=lambda a=a: (gate.wait(), asr_stream(a, t0))) for a in asr_audio]
    [t.start() for t in ts]
    [t.join() for t in ts]
    wall = time.perf_counter() - t0
    after = model.stats()
    steps, sess = after["stacked_steps"] - before["stacked_steps"], after["session_steps"] - before["session_steps"]
    med = {k: round(float(np.median(v)), 3) for k, v in acc.items()}
    print(f"rep {rep}: {n_sess} sessions x {int(seconds)} chunks in {wall * 1e3:.0f} ms = {n_sess * int(seconds) / wall:.1f} audio_s/s; "
          f"p50 ms {med}; host part p50 {med['chunk'] - med['step'] - med['features']:.3f}; {sess} session steps in {steps} chains "
          f"= {sess / max(steps, 1):.2f} per chain" + (f"; ASR {n_asr} streams done after {asr_done[0] * 1e3:.0f} ms = {n_asr * seconds / asr_done[0]:.1f} audio_s/s" if n_asr else ""),
          flush=True)
model.close()

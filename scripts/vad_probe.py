"""Time the VAD gate: per 0.5 s chunk (8000 samples -> 15-16 windows) on the GPU vs the torch-CPU oracle."""
import sys
import time

import numpy as np

ROOT = __file__.rsplit("/scripts/", 1)[0]
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + "/tests")
import torch  # noqa: E402

from oracle import vad_oracle as vo  # noqa: E402
from whisperlivekit_amd import synth, vad as V  # noqa: E402

w = dict(np.load(ROOT + "/tests/golden/vad_weights_16k.npz"))
audio = synth.to_pcm16_roundtrip(synth.speech_like(30.0, 0))
weights = V.HipSileroVADWeights(w)
model = V.HipSileroVAD(weights)
it = V.HipFixedVADIterator(model)
for rep in range(2):
    it.reset_states()
    ms = []
    for lo in range(0, len(audio), 8000):
        a = time.perf_counter()
        it(audio[lo:lo + 8000])
        ms.append(1e3 * (time.perf_counter() - a))
    print(f"hip rep {rep}: {len(ms)} chunks, p50 {np.median(ms) * 1e3:.1f} us per 0.5 s chunk, max {max(ms) * 1e3:.1f} us")
torch.set_num_threads(1)
oit = vo.OracleVADIterator(vo.OracleSileroVAD(w))
ms = []
for lo in range(0, 10 * 8000, 8000):
    a = time.perf_counter()
    oit(audio[lo:lo + 8000])
    ms.append(1e3 * (time.perf_counter() - a))
print(f"cpu oracle (torch, 1 thread): p50 {np.median(ms):.2f} ms per 0.5 s chunk")

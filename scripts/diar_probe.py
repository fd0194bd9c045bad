"""Drive the streaming Sortformer pass alone (for rocprofv3): N seconds of speech-like audio in 1 s chunks."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from whisperlivekit_amd import synth  # noqa: E402
from whisperlivekit_amd.diarization import HipSortformerDiarizationOnline  # noqa: E402
from whisperlivekit_amd.sortformer import HipSortformerModel  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
model = HipSortformerModel.synthetic()
audio = synth.speech_like(seconds, 0)
for rep in range(2):
    online = HipSortformerDiarizationOnline(model)
    ms = []
    for lo in range(0, len(audio) - 15999, 16000):
        online.insert_audio_chunk(audio[lo:lo + 16000])
        a = time.perf_counter()
        online.diarize_sync()
        ms.append(1e3 * (time.perf_counter() - a))
    print(f"rep {rep}: {len(ms)} chunks, p50 {np.median(ms):.3f} ms, last {ms[-1]:.3f} ms, sum {sum(ms):.1f} ms")
model.close()

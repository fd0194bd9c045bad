"""Host profile of the batch transcribe loop on the GPU box (gpurun): python scripts/transcribe_profile.py [threads]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WLK_SYNTHETIC_VOCAB", "1")
import torch
if len(sys.argv) > 1:
    torch.set_num_threads(int(sys.argv[1]))
from whisperlivekit_amd import synth
from whisperlivekit_amd.engine import HipWhisperModel
from whisperlivekit_amd.transcribe import transcribe
m = HipWhisperModel.synthetic("base.en", 0)
a = synth.speech_like(30.0, seed=1)
kw = dict(language="en", temperature=0.0, logprob_threshold=None, compression_ratio_threshold=None, word_timestamps=True)
transcribe(m, a[:16000], **kw)
t = time.perf_counter(); transcribe(m, a, **kw); print("greedy 30 s:", round(time.perf_counter() - t, 3), "s, torch threads", torch.get_num_threads())
pr = cProfile.Profile(); pr.enable(); transcribe(m, a, **kw); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)

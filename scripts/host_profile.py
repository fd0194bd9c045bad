"""Where does the HOST time of one stream go?  cProfile over bench.run_stream (base.en, 30 s, 0.5 s chunks), plus the
split of each call's wall time into 'inside libwlk_hip.so' (ctypes calls) and 'Python'."""
import os as _os; _os.environ.setdefault("WLK_SYNTHETIC_VOCAB", "1")
import cProfile
import io
import pstats
import sys
import time

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
import bench  # noqa: E402
from whisperlivekit_amd import _lib, synth  # noqa: E402
from whisperlivekit_amd.backend import HipSimulStreamingASR, HipSimulStreamingOnlineProcessor  # noqa: E402
from whisperlivekit_amd.dims import ALIGNMENT_HEADS, MODEL_DIMS  # noqa: E402
from whisperlivekit_amd.engine import HipWhisperModel  # noqa: E402

name = "base.en"
model = HipWhisperModel.from_state_dict(MODEL_DIMS[name], synth.synth_state_dict(MODEL_DIMS[name], 0), ALIGNMENT_HEADS[name])
asr = HipSimulStreamingASR(name, hip_model=model)
audio = bench.make_audio("speech", 30.0, 0)
bench.run_stream(HipSimulStreamingOnlineProcessor(asr), audio)          # warm-up

# time spent inside the C library: wrap every ctypes entry point
lib = _lib.load()
inside = {"t": 0.0, "n": 0, "by": {}}
for sym in _lib.EXPORTED_SYMBOLS:
    fn = getattr(lib, sym)

    def make(fn=fn, sym=sym):
        def wrapped(*a):
            t0 = time.perf_counter()
            try:
                return fn(*a)
            finally:
                dt = time.perf_counter() - t0
                inside["t"] += dt
                inside["n"] += 1
                b = inside["by"].setdefault(sym, [0.0, 0])
                b[0] += dt
                b[1] += 1
        return wrapped
    setattr(lib, sym, make())

proc = HipSimulStreamingOnlineProcessor(asr)
t0 = time.perf_counter()
calls = bench.run_stream(proc, audio)
wall = time.perf_counter() - t0
print(f"stream wall {wall * 1e3:.1f} ms; inside libwlk_hip {inside['t'] * 1e3:.1f} ms over {inside['n']} calls; "
      f"python {(wall - inside['t']) * 1e3:.1f} ms; decode steps {proc.model.counters['decode']}")
for sym, (t, n) in sorted(inside["by"].items(), key=lambda kv: -kv[1][0])[:10]:
    print(f"  {sym:28s} {t * 1e3:8.2f} ms {n:6d} calls {t / n * 1e6:8.1f} us/call")

proc = HipSimulStreamingOnlineProcessor(asr)
pr = cProfile.Profile()
pr.enable()
bench.run_stream(proc, audio)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])

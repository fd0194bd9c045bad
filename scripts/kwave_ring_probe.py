"""k-wave GEMM tiles (16 x 16: force 6, 32 x 32: force 7, 32 x 32 with one LDS slab buffer: force 8) at the row counts of the decoder prefill (~60-80 prompt rows) and of one
Sortformer session (<= 401 frames): microseconds per launch, back to back.  Run once per library (WLK_HIP_LIB) to compare builds."""
import ctypes as C
import sys

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from whisperlivekit_amd import _lib  # noqa: E402

lib = _lib.load()
SHAPES = [("ff_a/fc1", 2048, 512, 16), ("ff_b/fc2", 512, 2048, 2), ("qkv", 1536, 512, 0), ("out/pw2", 512, 512, 2), ("pw1", 1024, 512, 0),
          ("tf_outd", 192, 768, 2), ("proj", 192, 512, 0), ("pre_pw", 256, 256, 8), ("large qkv", 3840, 1280, 0), ("large fc2", 1280, 5120, 2)]


def t(m, n, k, flags, force, reps=40):
    us = C.c_float()
    rc = lib.wlk_diag_linear_time(m, n, k, flags, force, reps, C.byref(us))
    return us.value if rc == 0 else float("nan")


for name, n, k, flags in SHAPES:
    row = []
    for m in (61, 82, 200, 291, 401):
        row.append(f"M {m}: {t(m, n, k, flags, 7):5.1f} / {t(m, n, k, flags, 8):5.1f} / {t(m, n, k, flags, 6):5.1f}")
    print(f"{name:10s} N {n:4d} K {k:4d} (32x32 / 32x32 one LDS buffer / 16x16 us): " + " | ".join(row), flush=True)

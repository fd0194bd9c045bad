"""Does the Infinity Cache (256 MB, memory side) serve a decode step's weight streams when they are already resident?
Back-to-back launches of the single-row GEMV on the SAME weights (resident after the first launch; the per-XCD L2s are
invalidated at every kernel boundary) against the duration the same kernel shows inside a decode step (rocprofv3,
profiles/r06_large_v3_kernel_stats.md), where every launch streams weights that were last touched a whole step ago
(3.7 GB of other traffic in between for large-v3)."""
import ctypes as C
import sys

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from whisperlivekit_amd import _lib  # noqa: E402

lib = _lib.load()
for name, n, k in [("large-v3 fc1", 5120, 1280), ("large-v3 fc2", 1280, 5120), ("large-v3 qkv", 3840, 1280), ("large-v3 out", 1280, 1280),
                   ("large-v3 vocab", 51866, 1280), ("base fc1", 2048, 512), ("base vocab", 51864, 512)]:
    us = C.c_float()
    assert lib.wlk_diag_linear_time(1, n, k, 0, 1, 200, C.byref(us)) == 0, lib.wlk_diag_last_error()
    mb = 4.0 * n * k / 1e6
    print(f"{name:16s} N {n:6d} K {k:5d}: {mb:7.1f} MB in {us.value:6.2f} us = {mb / us.value / 1e3 * 1e3:6.2f} TB/s (same weights every launch)")

"""Which k-pipe tile for the stacked Sortformer projections?  Every tile of the kp family gives the same bits (gemm_f32.hip:
launch_gemm_kp), so the choice is speed only: time each instantiated tile for the network's (N, K) shapes at the row counts
stacked steps produce (2 / 4 / 8 sessions of ~300 frames), next to the 32 x 32 k-wave tiles and the current rule."""
import ctypes as C
import sys

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from whisperlivekit_amd import _lib  # noqa: E402

lib = _lib.load()
TILES = [(3, 4), (3, 3), (3, 2), (3, 1), (2, 4), (2, 2), (2, 1), (4, 2)]
SHAPES = [("ff_a", 2048, 512, 16), ("ff_b", 512, 2048, 2), ("qkv", 1536, 512, 0), ("out/pw2", 512, 512, 2), ("pw1", 1024, 512, 0),
          ("tf_outd", 192, 768, 2), ("proj", 192, 512, 0), ("pre_pw", 256, 256, 8)]


def t(m, n, k, flags, force, reps=30):
    us = C.c_float()
    rc = lib.wlk_diag_linear_time(m, n, k, flags, force, reps, C.byref(us))
    return us.value if rc == 0 else float("nan")


for name, n, k, flags in SHAPES:
    for m in ([600, 1200, 2400, 3200] if name != "pre_pw" else [1600, 6400, 12800]):
        rule = t(m, n, k, flags, 5)
        per = {f"{a}x{b}": t(m, n, k, flags, 500 + 10 * a + b) for a, b in TILES}
        best = min(per, key=per.get)
        gf = 2.0 * m * n * k / 1e9
        print(f"{name:8s} M {m:5d} N {n:4d} K {k:4d}: rule {rule:6.1f} us ({gf / rule * 1e-3:5.1f} TFLOP/s) | best {best} {per[best]:6.1f} us "
              f"({gf / per[best] * 1e-3:5.1f}) | " + " ".join(f"{kk} {v:.1f}" for kk, v in per.items()), flush=True)

print("--- below 512 rows: 32 x 32 k-wave tiles (force 7) against 16 x 16 tiles (force 6), same bits")
for name, n, k, flags in SHAPES + [("tf_qkv(K192: plain)", 576, 192, 4)]:
    if k % 128:
        continue
    row = []
    for m in (50, 100, 200, 291, 401):
        a, b = t(m, n, k, flags, 7), t(m, n, k, flags, 6)
        row.append(f"M {m}: {a:5.1f} / {b:5.1f}")
    print(f"{name:8s} N {n:4d} K {k:4d}: " + " | ".join(row), flush=True)

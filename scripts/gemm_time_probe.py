#!/usr/bin/env python
"""Time the encoder-shaped fp32 GEMMs through wlk_diag_linear_time (GPU box only): microseconds per launch and TFLOP/s
for one stream (M = 1500) and for k stacked streams (M = k * 1500), with the kernel launch_gemm picks by shape (the
one-tile-per-CU k-split kernel for these shapes) and with the 64x64 kernel forced (force = 3), alternating.
Usage: gemm_time_probe.py [reps] [model]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperlivekit_amd import _lib  # noqa: E402

lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
model = sys.argv[2] if len(sys.argv) > 2 else "base"
d, L, mel = {"base": (512, 6, 80), "small": (768, 12, 80), "large-v3": (1280, 32, 128), "tiny": (384, 4, 80)}[model]
SHAPES = [("conv1", 3000, d, 3 * mel, 1, 1), ("conv2", 1500, d, 3 * d, 3, 1), ("qkv", 1500, 3 * d, d, 4, L), ("out", 1500, d, d, 2, L),
          ("fc1", 1500, 4 * d, d, 1, L), ("fc2", 1500, d, 4 * d, 2, L), ("xkv", 1500, 2 * d * L, d, 4, 1)]
for mult in (1, 2, 8):
    tot = {0: 0.0, 3: 0.0, 4: 0.0}
    tot_fl = 0.0
    for tag, M, N, K, flags, w in SHAPES:
        if (N * K * 4 >= 2 ** 31):
            continue
        fl = 2.0 * M * mult * N * K
        res = {}
        for force in (0, 3, 4, 0, 3, 4):
            us = C.c_float()
            rc = lib.wlk_diag_linear_time(M * mult, N, K, flags, force, reps, C.byref(us))
            if rc != 0 and force == 4:       # the one-tile-per-CU kernels do not take this shape (K % 64)
                res[force] = res.get(3, 1e9)
                continue
            assert rc == 0, lib.wlk_diag_last_error()
            res[force] = min(res.get(force, 1e9), us.value)
        tot_fl += w * fl
        for f in tot:
            tot[f] += w * res[f]
        print(f"x{mult} {tag:6s} M={M * mult:6d} N={N:5d} K={K:5d}: by-shape {res[0]:8.2f} us {fl / res[0] / 1e6:6.1f} TF | "
              f"64x64 {res[3]:8.2f} us {fl / res[3] / 1e6:6.1f} TF | widest k-pipe tile {res[4]:8.2f} us {fl / res[4] / 1e6:6.1f} TF")
    print(f"x{mult} {model} encoder GEMMs: by-shape {tot[0]:.0f} us = {tot_fl / tot[0] / 1e6:.1f} TFLOP/s | 64x64 {tot[3]:.0f} us = "
          f"{tot_fl / tot[3] / 1e6:.1f} TFLOP/s | k-pipe everywhere {tot[4]:.0f} us = {tot_fl / tot[4] / 1e6:.1f} TFLOP/s")

#!/usr/bin/env python
"""Time the encoder-shaped fp32 GEMMs through wlk_diag_linear_time (GPU box only): microseconds per launch and TFLOP/s
for one stream (M = 1500) and for k stacked streams (M = k * 1500).  Usage: gemm_time_probe.py [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperlivekit_amd import _lib  # noqa: E402

lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
SHAPES = [("conv1", 3000, 512, 240, 1), ("conv2", 1500, 512, 1536, 3), ("qkv", 1500, 1536, 512, 4), ("out", 1500, 512, 512, 2),
          ("fc1", 1500, 2048, 512, 1), ("fc2", 1500, 512, 2048, 2), ("xkv", 1500, 6144, 512, 4)]
for mult in (1, 2, 4, 8):
    tot_us, tot_fl = 0.0, 0.0
    for tag, M, N, K, flags in SHAPES:
        us = C.c_float()
        rc = lib.wlk_diag_linear_time(M * mult, N, K, flags, 0, reps, C.byref(us))
        assert rc == 0, lib.wlk_diag_last_error()
        fl = 2.0 * M * mult * N * K
        w = {"conv1": 1, "conv2": 1, "xkv": 1}.get(tag, 6)
        tot_us += w * us.value
        tot_fl += w * fl
        print(f"x{mult} {tag:6s} M={M * mult:6d} N={N:5d} K={K:5d}: {us.value:8.2f} us  {fl / us.value / 1e6:7.1f} TFLOP/s")
    print(f"x{mult} encoder GEMMs: {tot_us:.0f} us per {mult} stream(s) = {tot_us / mult:.0f} us/stream, {tot_fl / tot_us / 1e6:.1f} TFLOP/s")

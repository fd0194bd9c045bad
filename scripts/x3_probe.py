#!/usr/bin/env python
"""Average microseconds per launch: X3 wide GEMM (bf16 matrix cores, fp32 accuracy) vs the fp32-MFMA kernels launch_gemm picks,
on the encoder's wide shapes.  GPU box only."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperlivekit_amd import _lib  # noqa: E402

lib = _lib.load()
shapes = [("base fc1", 1500, 2048, 512, 1), ("base qkv", 1500, 1536, 512, 4), ("base cross_kv", 1500, 6144, 512, 4),
          ("small fc1", 1500, 3072, 768, 1), ("large-v3 qkv", 1500, 3840, 1280, 4), ("large-v3 fc1", 1500, 5120, 1280, 1),
          ("8 x base fc1 rows", 12000, 2048, 512, 1)]
if os.environ.get("X3_PROBE_NARROW") == "1":
    # round 6: the shapes the X3 kernel does NOT serve today, measured instead of estimated (round-4 / round-5 reviews): the encoder's
    # narrow projections (N = d: 16 x 4 = 64 workgroups of 96 x 128 on 256 CUs for base.en, 16 x 10 = 160 for large-v3) and the
    # Sortformer's stacked projections (eight sessions, ~2 400 rows)
    shapes = [("base out", 1500, 512, 512, 0), ("base fc2", 1500, 512, 2048, 0), ("base conv2", 1500, 512, 1536, 0),
              ("large-v3 out", 1500, 1280, 1280, 0), ("large-v3 fc2", 1500, 1280, 5120, 0),
              ("8 x base out rows", 12000, 512, 512, 0), ("8 x base fc2 rows", 12000, 512, 2048, 0),
              ("sf x8 ff_a", 2400, 2048, 512, 0), ("sf x8 ff_b", 2400, 512, 2048, 0), ("sf x8 qkv", 2400, 1536, 512, 0),
              ("sf x8 pw1", 2400, 1024, 512, 0), ("sf x4 ff_a", 1200, 2048, 512, 0), ("sf x4 qkv", 1200, 1536, 512, 0)]
for name, m, n, k, flags in shapes:
    us3, us32 = C.c_float(), C.c_float()
    assert lib.wlk_diag_linear_x3_time(m, n, k, flags, 50, C.byref(us3)) == 0, lib.wlk_diag_last_error()
    assert lib.wlk_diag_linear_time(m, n, k, flags, 0, 50, C.byref(us32)) == 0
    gf = 2.0 * m * n * k / 1e9
    # gf GFLOP in us microseconds = gf / us * 1e-3 TFLOP/s... (1e9 / 1e-6 = 1e15): TFLOP/s = gf / us * 1e3
    print(f"{name:20s} M{m} N{n} K{k}: x3 {us3.value:7.2f} us = {gf / us3.value * 1e3:6.1f} TF f32-equivalent ({6 * gf / us3.value * 1e3:7.1f} TF bf16 issued) | "
          f"fp32 mfma {us32.value:7.2f} us = {gf / us32.value * 1e3:6.1f} TF | x{us32.value / us3.value:.2f}")
for name, t, d, h in [] if os.environ.get("X3_PROBE_NARROW") == "1" else [("base attention", 1500, 512, 8), ("small attention", 1500, 768, 12), ("large-v3 attention", 1500, 1280, 20)]:
    us3, us32 = C.c_float(), C.c_float()
    assert lib.wlk_diag_encoder_attention_x3_time(t, d, h, 50, C.byref(us3)) == 0, lib.wlk_diag_last_error()
    assert lib.wlk_diag_encoder_attention_time(t, d, h, 1, 50, C.byref(us32)) == 0
    gf = 4.0 * t * t * 64 * h / 1e9
    print(f"{name:20s} T{t} d{d}: x3 {us3.value:7.2f} us = {gf / us3.value * 1e3:6.1f} TF f32-equivalent | fp32 mfma {us32.value:7.2f} us = "
          f"{gf / us32.value * 1e3:6.1f} TF | x{us32.value / us3.value:.2f}")

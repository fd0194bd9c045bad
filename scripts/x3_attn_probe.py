#!/usr/bin/env python
"""X3 encoder attention timing (base.en / large-v3 shapes); WLK_X3_ATTN_ABL selects an ablation.  GPU box only."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperlivekit_amd import _lib  # noqa: E402

lib = _lib.load()
for name, t, d, h in [("base attention", 1500, 512, 8), ("large-v3 attention", 1500, 1280, 20)]:
    us3 = C.c_float()
    assert lib.wlk_diag_encoder_attention_x3_time(t, d, h, 50, C.byref(us3)) == 0, lib.wlk_diag_last_error()
    print(f"{name:20s} T{t} d{d}: x3 {us3.value:7.2f} us")

#!/usr/bin/env python
"""A few launches of the encoder GEMM shapes for a rocprofv3 --pmc pass (GPU box only)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperlivekit_amd import _lib
lib = _lib.load()
for tag, M, N, K, flags in (("fc2", 1500, 512, 2048, 2), ("out", 1500, 512, 512, 2), ("qkv", 1500, 1536, 512, 4), ("fc1", 1500, 2048, 512, 1)):
    us = C.c_float()
    assert lib.wlk_diag_linear_time(M, N, K, flags, 0, 10, C.byref(us)) == 0
    print(tag, us.value)

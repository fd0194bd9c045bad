#!/usr/bin/env python
"""Run the encoder-shaped fp32 GEMMs (and the flash attention) a few times through the diagnostics
entry points, for rocprofv3 --kernel-trace / --pmc passes.  GPU box only."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperlivekit_amd import _lib  # noqa: E402

lib = _lib.load()
vp = lambda a: a.ctypes.data_as(C.c_void_p)
rng = np.random.default_rng(0)
SHAPES = [("qkv", 1500, 1536, 512), ("out", 1500, 512, 512), ("fc1", 1500, 2048, 512), ("fc2", 1500, 512, 2048),
          ("conv2", 1500, 512, 1536), ("big", 4096, 4096, 1024)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for tag, M, N, K in SHAPES:
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = rng.standard_normal((N, K)).astype(np.float32)
    b = np.zeros(N, np.float32)
    c = np.empty((M, N), np.float32)
    for _ in range(reps):
        rc = lib.wlk_diag_linear(vp(a), K, M * K, vp(w), vp(b), None, N, M, N, K, 0, 1.0, 0, 0, vp(c))
        assert rc == 0, lib.wlk_diag_last_error()
qkv = rng.standard_normal((1500, 1536)).astype(np.float32) * 0.5
out = np.empty((1500, 512), np.float32)
for _ in range(reps):
    assert lib.wlk_diag_encoder_attention(vp(qkv), 1500, 512, 8, vp(out)) == 0
print("probe done")

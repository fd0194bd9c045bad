#!/usr/bin/env python
"""Time the encoder self-attention kernel variants (GPU box only).  WLK_ENC_ATTN=lds|regs picks the kernel."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperlivekit_amd import _lib  # noqa: E402

lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
vp = lambda a: a.ctypes.data_as(C.c_void_p)
for (T, d, H) in ((1500, 512, 8), (1500, 1280, 20)):
    for ks in (1,):
        us = C.c_float()
        rc = lib.wlk_diag_encoder_attention_time(T, d, H, ks, reps, C.byref(us))
        assert rc == 0, lib.wlk_diag_last_error()
        fl = 4.0 * T * T * 64 * H
        print(f"{os.environ.get('WLK_ENC_ATTN', 'regs'):5s} T={T} d={d} H={H} ksplit={os.environ.get('WLK_ENC_KSPLIT', 'auto')}: {us.value:8.2f} us  {fl / us.value / 1e6:6.1f} TFLOP/s")
# numerics: regs kernel vs fp64 reference on a small case
rng = np.random.default_rng(0)
T, d, H = 333, 128, 2
qkv = (rng.standard_normal((T, 3 * d)) * 0.5).astype(np.float32)
out = np.empty((T, d), np.float32)
assert lib.wlk_diag_encoder_attention(vp(qkv), T, d, H, vp(out)) == 0
q, k, v = [qkv[:, i * d:(i + 1) * d].astype(np.float64).reshape(T, H, 64).transpose(1, 0, 2) for i in range(3)]
s = q @ k.transpose(0, 2, 1)
p = np.exp(s - s.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
ref = (p @ v).transpose(1, 0, 2).reshape(T, d)
print("max abs err vs fp64:", float(np.abs(out - ref).max()))
# full-size output (T = 1500, 8 heads) saved for a bitwise comparison between kernel variants
T, d, H = 1500, 512, 8
qkv = (rng.standard_normal((T, 3 * d)) * 0.5).astype(np.float32)
out = np.empty((T, d), np.float32)
assert lib.wlk_diag_encoder_attention(vp(qkv), T, d, H, vp(out)) == 0
tag = os.environ.get("WLK_ENC_ATTN", "default")
np.save(f"/tmp/attn_out_{tag}.npy", out)
if tag != "lds" and os.path.exists("/tmp/attn_out_lds.npy"):
    ref = np.load("/tmp/attn_out_lds.npy")
    print(f"{tag} vs lds: bitwise equal = {bool(np.array_equal(out, ref))}, max abs diff = {float(np.abs(out - ref).max()):.3e}")

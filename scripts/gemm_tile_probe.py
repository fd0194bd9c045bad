#!/usr/bin/env python
"""Tile / slab-depth / ablation sweep of the k-split GEMM (GPU box only).  Each configuration runs in its own process
(WLK_KSPLIT_FORCE="tm,tn,ks,abl" is read once per process).  Usage: gemm_tile_probe.py  (prints one table)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBLEMS = [("fc1", 1500, 2048, 512, 1), ("qkv", 1500, 1536, 512, 4), ("fc2", 1500, 512, 2048, 2), ("out", 1500, 512, 512, 2),
            ("conv2", 1500, 512, 1536, 3), ("fc2x2", 3000, 512, 2048, 2)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from whisperlivekit_amd import _lib
    lib = _lib.load()
    out = []
    for tag, M, N, K, flags in PROBLEMS:
        us = C.c_float()
        best = 1e9
        for _ in range(3):
            assert lib.wlk_diag_linear_time(M, N, K, flags, int(sys.argv[2]), 30, C.byref(us)) == 0, lib.wlk_diag_last_error()
            best = min(best, us.value)
        out.append(f"{best:8.2f} us {2.0 * M * N * K / best / 1e6:6.1f} TF")
    print(" | ".join(out))
    sys.exit(0)

print("config".ljust(16) + " | ".join(t.ljust(19) for t, *_ in PROBLEMS))
configs = [("64x64 kernel", None, 3)]
for ks in (64, 103, 104, 206):  # 64: compiler-scheduled k-split, 2 x 64-deep slabs; 103 / 104: k-pipe, ring of 3 / 4 32-deep slabs;
    for tm, tn in ((3, 4), (3, 3), (3, 2), (3, 1), (2, 4), (2, 2), (4, 2)):   # 206: two slabs per trip, ring of 6
        if ks == 206 and (tm, tn) not in ((3, 1), (2, 2)):
            continue
        configs.append((f"{tm}x{tn} ks{ks}", f"{tm},{tn},{ks},0", 0))
for name, force, mode in configs:
    env = dict(os.environ)
    if force:
        env["WLK_KSPLIT_FORCE"] = force
    r = subprocess.run([sys.executable, __file__, "child", str(mode)], env=env, capture_output=True, text=True)
    print(name.ljust(16) + (r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "FAILED " + r.stderr[-200:]))

"""Builds libwlk_hip.so in-tree with hipcc for gfx950 (explicit `hipcc -shared -fPIC`; no JIT cache,
so the built library travels with the source tree)."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libwlk_hip.so")
SOURCES = ["api.hip", "gemm_f32.hip", "gemm_x3.hip", "layernorm.hip", "mel.hip", "attention.hip", "attention_x3.hip", "decoder.hip", "select.hip", "diag.hip", "melspec.hip",
           "sortformer.hip", "sortformer_api.hip", "vad.hip", "loop.hip", "engine.hip", "dtw.hip", "word_align.hip", "nllb.hip"]
# -amdgpu-kernarg-preload-count: leading scalar kernel arguments arrive in SGPRs at wave start (gfx950 firmware preloads
# them; the compiler keeps a fallback prologue) instead of behind an s_load round trip - the decode-step kernels are
# chains of dependent latencies, and this is the first link of every one of them
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def source_digest() -> str:
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + ["../../include/wlk_hip.h"]:
        path = os.path.join(CSRC, name)
        if os.path.isfile(path):
            h.update(name.encode())
            with open(path, "rb") as fh:
                h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source for gfx950 and link libwlk_hip.so; returns the library path."""
    stamp = LIB_PATH + ".digest"
    digest = source_digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read() == digest:
        return LIB_PATH
    hipcc = _hipcc()
    objs = []
    build_dir = os.path.join(CSRC, "build")
    os.makedirs(build_dir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(build_dir, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out, file=sys.stderr)
        if p.returncode != 0:
            failed = True
            print(f"hipcc failed on {src}:\n{out}", file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc compilation failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as fh:
        fh.write(digest)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

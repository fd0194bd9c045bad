"""Which kernels load behind exec-mask branches?  hipcc joins such a branch with `s_waitcnt vmcnt(0)`, i.e. a predicated load
(`ok ? p[i] : 0`, `if (ptr) x = ptr[i]`) costs a full memory round trip with nothing else in flight - the pattern behind the
first Sortformer attention kernel's 20 us.  Compiles every .hip file of the library to ISA and counts, per kernel,
its loads, those issued within eight instructions behind a `s_cbranch_exec*` / `s_and_saveexec`, and its `s_waitcnt vmcnt(0)`.

usage: python scripts/isa_scan_loads.py [min_predicated=3]   (needs hipcc; no GPU)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from whisperlivekit_amd import build as B  # noqa: E402


def scan(path):
    txt = open(path).read().split("\n")
    cur, stats = None, {}
    for i, l in enumerate(txt):
        m = re.match(r"^(_ZN3wlk\S+):", l)
        if m:
            cur = m.group(1)
            stats[cur] = [0, 0, 0]
            continue
        if cur is None:
            continue
        if "s_endpgm" in l:
            cur = None
            continue
        if re.search(r"\b(global_load|buffer_load|flat_load)", l):
            stats[cur][0] += 1
            if any("s_cbranch_exec" in x or "s_and_saveexec" in x for x in txt[max(0, i - 8):i]):
                stats[cur][1] += 1
        if "s_waitcnt vmcnt(0)" in l:
            stats[cur][2] += 1
    return stats


def main():
    least = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    hipcc = B._hipcc()
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for src in B.SOURCES:
            out = os.path.join(tmp, src.replace(".hip", ".s"))
            flags = [f for f in B.FLAGS if f != "-fPIC"]
            procs.append((src, out, subprocess.Popen([hipcc, *flags, "--cuda-device-only", "-S", os.path.join(B.CSRC, src), "-o", out],
                                                     stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
        for src, out, p in procs:
            p.wait()
            if not os.path.exists(out):
                print(f"{src}: no ISA (hipcc failed)")
                continue
            for k, (n, pred, w0) in scan(out).items():
                if pred >= least:
                    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:110] or k[:110]
                    print(f"{src:18s} loads {n:4d}  behind exec branches {pred:4d}  vmcnt(0) {w0:3d}  {name}")


if __name__ == "__main__":
    main()

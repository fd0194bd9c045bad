"""Timeline of one process_iter call out of a rocprofv3 kernel-trace database: kernel durations and idle gaps.

usage: python scripts/trace_gaps.py <results.db> [call_index] [--all]
"""
import re
import sqlite3
import sys


def load(path):
    c = sqlite3.connect(path)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if x.startswith("rocpd_kernel_dispatch")][0]
    ks = [x for x in t if x.startswith("rocpd_info_kernel_symbol")][0]
    return c.execute(
        f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s "
        "on d.kernel_id=s.id order by d.start").fetchall()


def main():
    rows = load(sys.argv[1])
    which = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 100
    names = [r[0] for r in rows]
    idx = [i for i, n in enumerate(names) if "mel_frame" in n]
    a, b = idx[which], idx[which + 1]
    prev = None
    busy = gaps = 0.0
    big = []
    for r in rows[a:b]:
        gap = (r[1] - prev) / 1e3 if prev else 0.0
        dur = (r[2] - r[1]) / 1e3
        n = re.sub(r"^_ZN3wlk\d+", "", r[0])[:44]
        if "--all" in sys.argv or gap > 1.0:
            print(f"gap {gap:7.2f}  dur {dur:7.2f}  wg={r[3] // max(r[4], 1):5d} {n}")
        busy += dur
        gaps += max(gap, 0.0)
        prev = r[2]
    span = (rows[b - 1][2] - rows[a][1]) / 1e3
    nxt = (rows[b][1] - rows[b - 1][2]) / 1e3
    print(f"call {which}: {b - a} kernels, span {span:.1f} us, busy {busy:.1f} us, gaps inside {gaps:.1f} us, "
          f"idle before the next call {nxt:.1f} us")
    # whole-run: total of idle gaps by what follows them
    tot = {}
    prev = None
    for r in rows[idx[10]:idx[-1]]:
        if prev is not None:
            g = (r[1] - prev) / 1e3
            if g > 1.0:
                n = re.sub(r"^_ZN3wlk\d+", "", r[0])[:30]
                tot.setdefault(n, [0, 0.0])
                tot[n][0] += 1
                tot[n][1] += g
        prev = r[2]
    print("idle gaps > 1 us by the kernel that follows them (calls 10..end):")
    for n, (cnt, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"  {n:32s} {cnt:6d} x {us / cnt:7.1f} us = {us / 1e3:8.2f} ms")
    print(f"  window {(rows[idx[-1]][1] - rows[idx[10]][1]) / 1e6:.1f} ms, {len(idx) - 11} calls")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""GPU busy time from a rocprofv3 --kernel-trace sqlite result: union of kernel intervals vs the span, plus the
per-kernel sums, over the LAST `window_ms` of the trace (the timed pass)."""
import sqlite3
import sys

db, window_ms = sys.argv[1], float(sys.argv[2])
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = next(t for t in tabs if t.startswith("kernels") or t == "kernels")
rows = c.execute(f"select name, start, end from {kt} order by start").fetchall()
t_end = max(r[2] for r in rows)
t_lo = t_end - window_ms * 1e6
rows = [r for r in rows if r[1] >= t_lo]
span = (t_end - rows[0][1]) / 1e6
busy, cur_s, cur_e = 0, None, None
for _, s, e in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = {}
for n, s, e in rows:
    k = n.split("(")[0][-60:]
    a = tot.setdefault(k, [0, 0])
    a[0] += e - s
    a[1] += 1
print(f"window {span:.1f} ms: GPU busy (union of kernel intervals) {busy / 1e6:.1f} ms = {100 * busy / 1e6 / span:.1f} %, "
      f"sum of kernel durations {sum(v[0] for v in tot.values()) / 1e6:.1f} ms, {len(rows)} kernels")
for k, (d, n) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {d / 1e6:8.1f} ms {n:7d} x {d / n / 1e3:8.2f} us  {k}")

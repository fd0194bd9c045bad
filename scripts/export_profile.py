#!/usr/bin/env python
"""Turn a rocprofv3 `--kernel-trace --stats` sqlite result (gpurun_out/...) into the small text
summaries that are committed under profiles/.

    python scripts/export_profile.py gpurun_out/r1/bench_kt_results.db profiles/r01_bench_kernel_stats.md "title"
"""
import sqlite3
import sys


def provenance():
    """Where and when: the commit the measured tree was built from (GRAFT_GIT_HEAD is exported by the job script: the GPU
    box has no .git) and the time of the export."""
    import os
    import time
    return dict(git_head=os.environ.get("GRAFT_GIT_HEAD", "unknown"), exported=time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()))


def main(db, out, title):
    c = sqlite3.connect(db)
    rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    total = sum(r[2] for r in rows)
    span = c.execute("select min(start), max(end) from kernels").fetchone()
    lines = [f"# {title}", "",
             f"source: `rocprofv3 --kernel-trace --stats` ({db}); durations in microseconds", "",
             f"total kernel time {total / 1e3:.1f} ms; first-to-last-kernel span {((span[1] - span[0]) / 1e6):.1f} ms "
             "(the span includes model upload, session creation and Python start-up gaps)", "",
             "| kernel | calls | total us | avg us | % |", "|---|---:|---:|---:|---:|"]
    for name, calls, tot, avg, pct in rows:
        lines.append(f"| `{name[:90]}` | {calls} | {tot:.0f} | {avg:.2f} | {pct:.1f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    import json
    js = {name: dict(calls=int(calls), total_us=float(tot), avg_us=float(avg), pct=float(pct)) for name, calls, tot, avg, pct in rows}
    js["_provenance"] = provenance()
    json.dump(js, open(out.rsplit(".", 1)[0] + ".json", "w"), indent=0)   # bench.py reads avg_us of the dominant kernel
    print(out, "written;", len(rows), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 kernel stats")

#!/usr/bin/env python
"""Aggregate rocprofv3 `--pmc ... --kernel-trace --output-format csv` passes per kernel.

    python scripts/export_pmc.py OUT.md OUT.json PASS_DIR [PASS_DIR ...]

Each PASS_DIR holds one pass's *_counter_collection.csv (one row per dispatch and counter).  Per kernel name:
launches, mean duration, and per-launch means of every counter.  Corrections applied as
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE is reported in KiB and
counts wide coalesced reads at half their bytes -> HBM read bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE * 1024 is
used as is (uncalibrated).  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * duration * 2.4 GHz)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("wlk::", "")
    return name.split("(")[0][:70]


def main(out_md, out_json, dirs):
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))       # kernel -> counter -> [sum, n]
    dur = defaultdict(lambda: [0.0, 0])
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    agg[k][row["Counter_Name"]][0] += float(row["Counter_Value"])
                    agg[k][row["Counter_Name"]][1] += 1
                    did = row.get("Dispatch_Id")
                    if did not in seen and "Start_Timestamp" in row:
                        seen.add(did)
                        dur[k][0] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3
                        dur[k][1] += 1
    rows = {}
    for k, counters in agg.items():
        n = max(dur[k][1], 1)
        r = {"launches_seen": dur[k][1] // max(len(dirs), 1), "avg_us_under_pmc": round(dur[k][0] / n, 2)}
        for c, (s, cnt) in counters.items():
            r[c] = s / max(cnt, 1)
        if "FETCH_SIZE" in r:
            r["hbm_read_bytes_per_launch"] = r["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in r:
            r["hbm_write_bytes_per_launch"] = r["WRITE_SIZE"] * 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES" in r and r["avg_us_under_pmc"] > 0:
            r["mfma_util"] = r["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * r["avg_us_under_pmc"] * 1e-6 * 2.4e9)
        rows[k] = r
    order = sorted(rows, key=lambda k: -rows[k]["avg_us_under_pmc"] * rows[k]["launches_seen"])
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from export_profile import provenance
    js = {k: rows[k] for k in order}
    js["_provenance"] = provenance()
    json.dump(js, open(out_json, "w"), indent=1)
    lines = ["# rocprofv3 PMC passes over `python bench.py` (per-kernel means per launch)", "",
             "FETCH_SIZE/WRITE_SIZE are KiB; HBM read = FETCH_SIZE x 1024 x 2 (gfx950 half-count correction, MI355X_MICROARCH.md), "
             "write = WRITE_SIZE x 1024 (uncalibrated). Durations are under counter collection (slower than the timed runs).", "",
             "| kernel | launches | avg us | MFMA util | HBM read / launch | HBM write / launch |", "|---|---:|---:|---:|---:|---:|"]
    for k in order[:24]:
        r = rows[k]
        fmt = lambda v: "-" if v is None else (f"{v / 1e6:.2f} MB" if v >= 1e5 else f"{v / 1e3:.1f} KB")
        lines.append(f"| `{k}` | {r['launches_seen']} | {r['avg_us_under_pmc']} | "
                     f"{'-' if 'mfma_util' not in r else format(r['mfma_util'], '.1%')} | "
                     f"{fmt(r.get('hbm_read_bytes_per_launch'))} | {fmt(r.get('hbm_write_bytes_per_launch'))} |")
    open(out_md, "w").write("\n".join(lines) + "\n")
    print(out_md, "written;", len(rows), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])

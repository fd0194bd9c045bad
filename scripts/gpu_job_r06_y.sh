#!/bin/bash
# round 6, job y: where the Sortformer kernels' wave cycles go - SQ_WAIT_ANY (parked: s_waitcnt / barrier), SQ_WAIT_INST_ANY (issue
# stall: MFMA dependency / pipe), SQ_ACTIVE_INST_ANY per kernel, one session
set -u
O=gpurun_out/r06y; mkdir -p $O; R=$PWD
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/sfy
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/sfy -o p -- python $R/scripts/diar_probe.py 30 > $R/$O/pmc.log 2>&1
cd $R
python - <<'PY' | tee gpurun_out/r06y/sf_wave_cycles.txt
import csv, glob, collections, re
f = glob.glob('/tmp/sfy/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in f:
    for r in csv.DictReader(open(path)):
        k = re.sub(r'\(.*', '', r['Kernel_Name'])[:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
print(f"{'kernel':60s} {'launches':>8s} {'wave-cycles/launch':>18s} {'parked':>7s} {'issue-stall':>11s} {'active':>7s}")
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]['SQ_WAVE_CYCLES']):
    w = c['SQ_WAVE_CYCLES'] or 1
    print(f"{k:60s} {n[k]:8d} {w / max(n[k], 1):18.0f} {c['SQ_WAIT_ANY'] / w:7.2f} {c['SQ_WAIT_INST_ANY'] / w:11.2f} {c['SQ_ACTIVE_INST_ANY'] / w:7.2f}")
PY

#!/bin/bash
# round 6, job z: where the base.en stream's wave cycles go, per kernel (same four SQ counters as job y)
set -u
O=gpurun_out/r06z; mkdir -p $O; R=$PWD
export WLK_SYNTHETIC_VOCAB=1 TMPDIR=/tmp; cd /tmp; rm -rf /tmp/bz
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams --no-large-v3 --full-out /tmp/prof_full.json"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/bz -o p -- $B > $R/$O/pmc.log 2>&1
cd $R
python - <<'PY' | tee gpurun_out/r06z/bench_wave_cycles.txt
import csv, glob, collections, re
f = glob.glob('/tmp/bz/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in f:
    for r in csv.DictReader(open(path)):
        k = re.sub(r'\(.*', '', r['Kernel_Name'])[:64]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
print(f"{'kernel':64s} {'launches':>8s} {'wave-cycles/launch':>18s} {'parked':>7s} {'issue-stall':>11s} {'active':>7s}")
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]['SQ_WAVE_CYCLES'])[:40]:
    w = c['SQ_WAVE_CYCLES'] or 1
    print(f"{k:64s} {n[k]:8d} {w / max(n[k], 1):18.0f} {c['SQ_WAIT_ANY'] / w:7.2f} {c['SQ_WAIT_INST_ANY'] / w:11.2f} {c['SQ_ACTIVE_INST_ANY'] / w:7.2f}")
PY

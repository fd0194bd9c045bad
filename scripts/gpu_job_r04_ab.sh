#!/bin/bash
# Round 4 A/B on one box: the in-tree library against the round-3 library kept beside it (libwlk_hip_r3.so, WLK_HIP_LIB),
# same Python, alternating runs: step anatomy (scripts/step_probe.py) and short bench lines.  usage: gpu_job_r04_ab.sh [tag] [pytest args]
set -u
TAG="${1:-a}"
PYT="${2:-tests}"
OUT=gpurun_out/r04${TAG}
mkdir -p $OUT
export WLK_SYNTHETIC_VOCAB=1
R3=$PWD/whisperlivekit_amd/libwlk_hip_r3.so
timeout 1200 python -m pytest $PYT -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
: > $OUT/step_probe.log
for i in 1 2; do
  echo -n "new " >> $OUT/step_probe.log; timeout 200 python scripts/step_probe.py base.en 15 40 60 2>/dev/null | tail -1 >> $OUT/step_probe.log
  echo -n "r3  " >> $OUT/step_probe.log; WLK_HIP_LIB=$R3 timeout 200 python scripts/step_probe.py base.en 15 40 60 2>/dev/null | tail -1 >> $OUT/step_probe.log
done
cat $OUT/step_probe.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams"
: > $OUT/ab.log
for i in 1 2; do
  echo "new" >> $OUT/ab.log; WLK_STEP_TIMING=1 timeout 300 $B 2>$OUT/bench_new_$i.err | tail -1 >> $OUT/ab.log
  echo "r3" >> $OUT/ab.log; WLK_STEP_TIMING=1 WLK_HIP_LIB=$R3 timeout 300 $B 2>$OUT/bench_r3_$i.err | tail -1 >> $OUT/ab.log
done
grep -h "step" $OUT/bench_new_1.err | tail -3
grep -h "step" $OUT/bench_r3_1.err | tail -3
python - "$OUT" <<'PY'
import json, sys
lab=None
for l in open(sys.argv[1] + '/ab.log'):
    l=l.strip()
    if l in('new','r3'): lab=l; continue
    try:
        j=json.loads(l); pc=j.get('parity_checked') or {}
        print(lab, j['value'], j.get('ms_per_step'), pc.get('decisions'), pc.get('identical'), pc.get('tie_divergences'), pc.get('mismatches'))
    except Exception as e: print(lab,'ERR',l[:300])
PY
# per-kernel averages of the new library (rocprofv3 kernel trace of one bench pass) and the 8-stream leg, new vs r3
R=$PWD
export TMPDIR=/tmp; cd /tmp
B1="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof/stats -o st -- $B1 > $R/$OUT/prof_stats.log 2>&1
cd $R
python scripts/export_profile.py $(find $OUT/prof/stats -name "*.db" | head -1) $OUT/bench_kernel_stats.md "python bench.py --steps 1 --warmup 1 --no-eight-streams (base.en, 1 stream): rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
rm -rf $OUT/prof
head -45 $OUT/bench_kernel_stats.md
B8="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-diarization"
timeout 300 $B8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new eight', (d.get('eight_streams') or {}).get('audio_s_per_s'), 'value', d['value'])"
WLK_HIP_LIB=$R3 timeout 300 $B8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r3  eight', (d.get('eight_streams') or {}).get('audio_s_per_s'), 'value', d['value'])"

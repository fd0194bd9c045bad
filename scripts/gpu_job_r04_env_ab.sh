#!/bin/bash
# usage: gpu_job_r04_env_ab.sh <tag> "<ENV=.. for the B arm>" [pytest target]: full GPU suite on the default build, then
# alternating short bench runs default / B arm (same library), step timing printed
set -u
TAG="$1"; BENV="$2"; PYT="${3:-tests}"
OUT=gpurun_out/r04${TAG}; mkdir -p $OUT
export WLK_SYNTHETIC_VOCAB=1
timeout 1200 python -m pytest $PYT -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams"
: > $OUT/ab.log
for i in 1 2 3; do
  echo "A" >> $OUT/ab.log; WLK_STEP_TIMING=1 timeout 300 $B 2>$OUT/a_$i.err | tail -1 >> $OUT/ab.log
  echo "B" >> $OUT/ab.log; env $BENV WLK_STEP_TIMING=1 timeout 300 $B 2>$OUT/b_$i.err | tail -1 >> $OUT/ab.log
done
echo "A (default):"; grep -h "one-replay" $OUT/a_1.err $OUT/a_2.err | tail -2
echo "B ($BENV):"; grep -h "one-replay" $OUT/b_1.err $OUT/b_2.err | tail -2
python - "$OUT" <<'PY'
import json, sys
lab=None
for l in open(sys.argv[1] + '/ab.log'):
    l=l.strip()
    if l in('A','B'): lab=l; continue
    try:
        j=json.loads(l); pc=j.get('parity_checked') or {}
        print(lab, j['value'], j.get('ms_per_step'), pc.get('decisions'), pc.get('identical'), pc.get('tie_divergences'), pc.get('mismatches'))
    except Exception as e: print(lab,'ERR',l[:300])
PY

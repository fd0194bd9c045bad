#!/bin/bash
# alternating single-stream bench runs: default environment vs the environment given as $1 (3 pairs)
set -u
NEW_ENV="$1"
mkdir -p gpurun_out
export WLK_SYNTHETIC_VOCAB=1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams"
: > gpurun_out/ab_env.log
for i in 1 2 3; do
  echo "env" >> gpurun_out/ab_env.log; env $NEW_ENV timeout 300 $B 2>/dev/null | tail -1 >> gpurun_out/ab_env.log
  echo "base" >> gpurun_out/ab_env.log; timeout 300 $B 2>/dev/null | tail -1 >> gpurun_out/ab_env.log
done
python - <<'PY'
import json
lab=None
for l in open('gpurun_out/ab_env.log'):
    l=l.strip()
    if l in('env','base'): lab=l; continue
    try:
        j=json.loads(l); pc=j.get('parity_checked') or {}
        print(lab, j['value'], pc.get('decisions'), pc.get('identical'))
    except Exception as e: print(lab,'ERR',l[:300])
PY

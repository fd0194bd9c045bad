#!/bin/bash
# usage: gpu_job_ab_generic.sh "<ENV=1 ...>"  : full GPU suite, then alternating bench runs with / without the given env
set -u
OLD_ENV="$1"
mkdir -p gpurun_out
export WLK_SYNTHETIC_VOCAB=1
timeout 1000 python -m pytest tests -m gpu -x -q > gpurun_out/ab_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/ab_pytest.log
tail -4 gpurun_out/ab_pytest.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams"
: > gpurun_out/ab.log
for i in 1 2 3; do
  echo "new" >> gpurun_out/ab.log; timeout 300 $B 2>/dev/null | tail -1 >> gpurun_out/ab.log
  echo "old" >> gpurun_out/ab.log; env $OLD_ENV timeout 300 $B 2>/dev/null | tail -1 >> gpurun_out/ab.log
done
python - <<'PY'
import json
lab=None
for l in open('gpurun_out/ab.log'):
    l=l.strip()
    if l in('new','old'): lab=l; continue
    try:
        j=json.loads(l); pc=j.get('parity_checked') or {}
        print(lab, j['value'], pc.get('decisions'), pc.get('identical'), pc.get('tie_divergences'), pc.get('mismatches'))
    except Exception as e: print(lab,'ERR',l[:300])
PY

#!/bin/bash
# A/B on one box: single-token step as one graph replay (no copy nodes) + merged first-step read-back vs round-2 chain.
set -u
mkdir -p gpurun_out
export WLK_SYNTHETIC_VOCAB=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_serving.py -m gpu -x -q > gpurun_out/ab_step_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/ab_step_pytest.log
tail -5 gpurun_out/ab_step_pytest.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams"
: > gpurun_out/ab_step.log
for i in 1 2 3; do
  echo "new" >> gpurun_out/ab_step.log; timeout 300 $B 2>/dev/null | tail -1 >> gpurun_out/ab_step.log
  echo "old" >> gpurun_out/ab_step.log; WLK_FUSED_STEP=0 WLK_NO_FIRST_MERGE=1 timeout 300 $B 2>/dev/null | tail -1 >> gpurun_out/ab_step.log
done
echo "stepOnly" >> gpurun_out/ab_step.log; WLK_NO_FIRST_MERGE=1 timeout 300 $B 2>/dev/null | tail -1 >> gpurun_out/ab_step.log
python - <<'PY'
import json
lab=None
for l in open('gpurun_out/ab_step.log'):
    l=l.strip()
    if l in('new','old','stepOnly'): lab=l; continue
    try:
        j=json.loads(l); pc=j.get('parity_checked') or {}
        print(lab, j['value'], pc.get('decisions'), pc.get('identical'), pc.get('tie_divergences'), pc.get('mismatches'))
    except Exception as e: print(lab,'ERR',l[:300])
PY

#!/bin/bash
# A/B on one box: fused select tail + merged prefill projection vs the separate launches, alternating runs.
set -u
mkdir -p gpurun_out
export WLK_SYNTHETIC_VOCAB=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_serving.py -m gpu -x -q > gpurun_out/ab_tail_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/ab_tail_pytest.log
tail -3 gpurun_out/ab_tail_pytest.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams"
: > gpurun_out/ab_tail.log
for i in 1 2 3; do
  echo "new" >> gpurun_out/ab_tail.log; timeout 300 $B 2>/dev/null | tail -1 >> gpurun_out/ab_tail.log
  echo "old" >> gpurun_out/ab_tail.log; WLK_SELECT_FUSED=0 WLK_NO_PREFILL_MERGE=1 timeout 300 $B 2>/dev/null | tail -1 >> gpurun_out/ab_tail.log
done
python - <<'PY'
import json
lab=None
for l in open('gpurun_out/ab_tail.log'):
    l=l.strip()
    if l in('new','old'): lab=l; continue
    try: j=json.loads(l); print(lab, j['value'], j.get('parity_checked'))
    except Exception as e: print(lab,'ERR',l[:200])
PY

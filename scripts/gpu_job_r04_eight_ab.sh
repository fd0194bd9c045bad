#!/bin/bash
# usage: gpu_job_r04_eight_ab.sh "<ENV for the B arm>": alternating 8-stream legs (bench.py), default vs B arm
set -u
export WLK_SYNTHETIC_VOCAB=1
B8="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-diarization"
for i in 1 2 3; do
  timeout 300 $B8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('A eight', (d.get('eight_streams') or {}).get('audio_s_per_s'), 'value', d['value'], d['parity_ok'])"
  env $1 timeout 300 $B8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B eight', (d.get('eight_streams') or {}).get('audio_s_per_s'), 'value', d['value'], d['parity_ok'])"
done

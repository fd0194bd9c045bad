#!/bin/bash
# round 6, job w: the multi-row / vocabulary GEMV with a branch-free main loop (two chunks per batch where K <= 512) - parity tests,
# then new library / previous library alternating: base.en stream, large-v3 stream, 8 streams
set -u
O=gpurun_out/r06w; mkdir -p $O
export WLK_SYNTHETIC_VOCAB=1
PREV=$PWD/whisperlivekit_amd/libwlk_hip_prev.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_serving.py tests/test_nllb.py -m gpu -x -q 2>&1 | tail -3
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams --no-large-v3"
line() { python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], 'step us', j['roofline']['step']['us'], j['parity_checked'].get('identical'), '/', j['parity_checked'].get('decisions'))"; }
for i in 1 2 3; do
  echo "new  $(timeout 300 $B 2>/dev/null | tail -1 | line)"
  echo "prev $(WLK_HIP_LIB=$PREV timeout 300 $B 2>/dev/null | tail -1 | line)"
done | tee $O/ab_stream.txt
L="python bench.py --model large-v3 --seconds 30 --seed 8 --steps 1 --warmup 1 --no-cpu-baseline --no-eight-streams --no-diarization --no-large-v3"
for i in 1 2; do
  echo "new  $(timeout 400 $L 2>/dev/null | tail -1 | line)"
  echo "prev $(WLK_HIP_LIB=$PREV timeout 400 $L 2>/dev/null | tail -1 | line)"
done | tee $O/ab_large.txt
for i in 1 2; do
  echo "new  $(timeout 300 python scripts/eight_stream_probe.py 8 2>&1 | grep '^pass 1')"
  echo "prev $(WLK_HIP_LIB=$PREV timeout 300 python scripts/eight_stream_probe.py 8 2>&1 | grep '^pass 1')"
done | tee $O/ab_eight.txt

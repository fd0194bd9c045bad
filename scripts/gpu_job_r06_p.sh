#!/bin/bash
# round 6, job p: four register slabs in flight in the 16 x 16 k-wave GEMM - bit-identity tests, per-shape probe against the
# previous library (whisperlivekit_amd/libwlk_hip_prev.so, git-ignored), alternating stream / diarizer runs either way
set -u
O=gpurun_out/r06p; mkdir -p $O
export WLK_SYNTHETIC_VOCAB=1
PREV=$PWD/whisperlivekit_amd/libwlk_hip_prev.so
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sortformer.py tests/test_gpu_serving.py -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for i in 1 2; do
  echo "== new"; timeout 200 python scripts/kwave_ring_probe.py 2>&1 | grep -v amdgpu.ids
  echo "== prev"; WLK_HIP_LIB=$PREV timeout 200 python scripts/kwave_ring_probe.py 2>&1 | grep -v amdgpu.ids
done | tee $O/kwave_ring_probe.txt
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams --no-large-v3"
for i in 1 2 3; do
  echo "new  $(timeout 300 $B 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['p50_call_ms'], j['parity_checked'].get('identical'), '/', j['parity_checked'].get('decisions'))")"
  echo "prev $(WLK_HIP_LIB=$PREV timeout 300 $B 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['p50_call_ms'], j['parity_checked'].get('identical'), '/', j['parity_checked'].get('decisions'))")"
done | tee $O/ab_stream.txt
for i in 1 2 3; do
  echo "new  $(timeout 200 python scripts/diar_probe.py 30 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' ')"
  echo "prev $(WLK_HIP_LIB=$PREV timeout 200 python scripts/diar_probe.py 30 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' ')"
done | tee $O/ab_diar.txt

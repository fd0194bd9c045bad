#!/bin/bash
# round 6: HIP_FORCE_DEV_KERNARG=1 (kernel arguments in device memory) on the Sortformer's eager chain, the ASR stream and 8 streams
set -u
O=gpurun_out/r06kk; mkdir -p $O
export WLK_SYNTHETIC_VOCAB=1
for i in 1 2 3; do
  echo "devkernarg=1 $(HIP_FORCE_DEV_KERNARG=1 timeout 200 python scripts/diar_probe.py 30 2>&1 | grep -v amdgpu.ids | tail -1)"
  echo "devkernarg=0 $(timeout 200 python scripts/diar_probe.py 30 2>&1 | grep -v amdgpu.ids | tail -1)"
done | tee $O/ab_diar.txt
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams --no-large-v3"
line() { python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], 'step us', j['roofline']['step']['us'], 'p50 call', j['p50_call_ms'])"; }
for i in 1 2 3; do
  echo "devkernarg=1 $(HIP_FORCE_DEV_KERNARG=1 timeout 300 $B 2>/dev/null | tail -1 | line)"
  echo "devkernarg=0 $(timeout 300 $B 2>/dev/null | tail -1 | line)"
done | tee $O/ab_stream.txt
for i in 1 2; do
  echo "devkernarg=1 $(HIP_FORCE_DEV_KERNARG=1 timeout 300 python scripts/eight_stream_probe.py 8 2>&1 | grep '^pass 1')"
  echo "devkernarg=0 $(timeout 300 python scripts/eight_stream_probe.py 8 2>&1 | grep '^pass 1')"
done | tee $O/ab_eight.txt

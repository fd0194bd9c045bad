#!/bin/bash
# round 6, job v: Sortformer attention with branch-free loads and operand rings - suite, new / previous library alternating
set -u
O=gpurun_out/r06v; mkdir -p $O
PREV=$PWD/whisperlivekit_amd/libwlk_hip_prev.so
timeout 1200 python -m pytest tests/test_gpu_sortformer.py tests/test_nllb.py tests/test_translation.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
  echo "new  $(timeout 200 python scripts/diar_probe.py 30 2>&1 | grep -v amdgpu.ids | tail -1)"
  echo "prev $(WLK_HIP_LIB=$PREV timeout 200 python scripts/diar_probe.py 30 2>&1 | grep -v amdgpu.ids | tail -1)"
done | tee $O/ab_diar.txt
for i in 1 2; do
  echo "new  $(timeout 300 python scripts/diar_probe8.py 8 30 2>&1 | grep 'rep 1' | cut -c1-220)"
  echo "prev $(WLK_HIP_LIB=$PREV timeout 300 python scripts/diar_probe8.py 8 30 2>&1 | grep 'rep 1' | cut -c1-220)"
done | tee $O/ab_diar8.txt

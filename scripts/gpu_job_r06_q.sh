#!/bin/bash
# round 6, job q: 32 x 32 k-wave tiles with ONE LDS slab buffer (four workgroups per CU) - bit-identity, Sortformer suite, diarizer A/B against the previous library
set -u
O=gpurun_out/r06q; mkdir -p $O
PREV=$PWD/whisperlivekit_amd/libwlk_hip_prev.so
timeout 900 python -m pytest tests/test_gpu_sortformer.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
  echo "new  $(timeout 200 python scripts/diar_probe.py 30 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' ')"
  echo "prev $(WLK_HIP_LIB=$PREV timeout 200 python scripts/diar_probe.py 30 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' ')"
done | tee $O/ab_diar.txt
for i in 1 2; do
  echo "new  $(timeout 300 python scripts/diar_probe8.py 8 30 2>&1 | grep 'rep 1' | cut -c1-200)"
  echo "prev $(WLK_HIP_LIB=$PREV timeout 300 python scripts/diar_probe8.py 8 30 2>&1 | grep 'rep 1' | cut -c1-200)"
done | tee $O/ab_diar8.txt

#!/bin/bash
# round 6, job r: K = 192 projections (the Sortformer's Transformer blocks) on the k-wave tiles - suite, then WLK_KP_SHORT_K=0 / 1
# alternating; the 32 x 32 k-wave kernel with four register slabs in flight (force 9) in the per-shape probe
set -u
O=gpurun_out/r06r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sortformer.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python scripts/kwave_ring_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/kwave_probe.txt
for i in 1 2 3; do
  for v in 1 0; do echo "short_k=$v $(WLK_KP_SHORT_K=$v timeout 200 python scripts/diar_probe.py 30 2>&1 | grep -v amdgpu.ids | tail -1)"; done
done | tee $O/ab_diar.txt
for i in 1 2; do
  for v in 1 0; do echo "short_k=$v $(WLK_KP_SHORT_K=$v timeout 300 python scripts/diar_probe8.py 8 30 2>&1 | grep 'rep 1' | cut -c1-200)"; done
done | tee $O/ab_diar8.txt

#!/bin/bash
# round 6, job t: the Transformer blocks' LayerNorms inside the K = 192 projections - suite, then WLK_SF_TF_LN_FUSE=0 / 1 alternating
set -u
O=gpurun_out/r06t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sortformer.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -5
for i in 1 2 3; do
  for v in 1 0; do echo "tf_ln_fuse=$v $(WLK_SF_TF_LN_FUSE=$v timeout 200 python scripts/diar_probe.py 30 2>&1 | grep -v amdgpu.ids | tail -1)"; done
done | tee $O/ab_diar.txt
for i in 1 2; do
  for v in 1 0; do echo "tf_ln_fuse=$v $(WLK_SF_TF_LN_FUSE=$v timeout 300 python scripts/diar_probe8.py 8 30 2>&1 | grep 'rep 1' | cut -c1-200)"; done
done | tee $O/ab_diar8.txt

#!/bin/bash
# round 6, job o: the early z-score in the engine's batched steps - serving tests, then the 8-stream leg either way (alternating)
set -u
O=gpurun_out/r06o; mkdir -p $O
export WLK_SYNTHETIC_VOCAB=1
timeout 1200 python -m pytest tests/test_gpu_serving.py tests/test_gpu_pipeline.py -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
bash scripts/gpu_job_ab8.sh "WLK_EARLY_Z=1" "WLK_EARLY_Z=0" "WLK_EARLY_Z=1" "WLK_EARLY_Z=0" "WLK_EARLY_Z=1" "WLK_EARLY_Z=0" 2>&1 | tee $O/ab8.txt

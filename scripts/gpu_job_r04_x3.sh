#!/bin/bash
# X3 path, first contact: parity tests of the path, GEMM timing probe, then the whole suite and the A/B against round 3
set -u
OUT=gpurun_out/r04${1:-c}
mkdir -p $OUT
export WLK_SYNTHETIC_VOCAB=1
timeout 600 python -m pytest tests/test_gpu_x3.py -m gpu -x -q -s > $OUT/pytest_x3.log 2>&1; echo "x3 pytest exit $?" >> $OUT/pytest_x3.log
grep -h "x3 max\|passed\|failed\|Error\|error\|exit" $OUT/pytest_x3.log | head -30
timeout 300 python scripts/x3_probe.py > $OUT/x3_probe.txt 2>&1; cat $OUT/x3_probe.txt
bash scripts/gpu_job_r04_ab.sh ${1:-c} "${2:-tests}"

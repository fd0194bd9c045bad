#!/bin/bash
set -u
OUT=gpurun_out/r04${1:-j}; mkdir -p $OUT
export WLK_SYNTHETIC_VOCAB=1
timeout 300 python -m pytest tests/test_gpu_x3.py -m gpu -x -q -s -k attention 2>&1 | grep -h "x3 max\|passed\|failed\|rror" | head
for v in 0 1 2 3; do echo "== WLK_X3_ATTN_ABL=$v" | tee -a $OUT/x3_attn_abl.txt; WLK_X3_ATTN_ABL=$v timeout 100 python scripts/x3_attn_probe.py 2>/dev/null | tee -a $OUT/x3_attn_abl.txt; done

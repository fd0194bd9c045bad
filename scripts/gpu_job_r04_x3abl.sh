#!/bin/bash
# where does the X3 wide kernel's time go: loaders only / compute only / MFMAs only / tile-order variants
set -u
OUT=gpurun_out/r04${1:-e}; mkdir -p $OUT
for v in "WLK_X3_ABL=0" "WLK_X3_ABL=1" "WLK_X3_ABL=2" "WLK_X3_ABL=3" "WLK_X3_MAP=1" "WLK_X3_MAP=2"; do
  echo "== $v" | tee -a $OUT/x3_abl.txt
  env $v timeout 200 python scripts/x3_probe.py 2>/dev/null | tee -a $OUT/x3_abl.txt
done

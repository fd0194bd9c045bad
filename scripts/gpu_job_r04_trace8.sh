#!/bin/bash
# the 8-stream kernel trace of the evidence run on its own (rocprofv3 crashed on it once): gpurun_out/r04/trace8_busy.txt
O=gpurun_out/r04; mkdir -p $O; R=$PWD
export WLK_SYNTHETIC_VOCAB=1
export TMPDIR=/tmp; cd /tmp
for try in 1 2; do
  rm -rf $R/$O/trace8
  timeout 300 rocprofv3 --kernel-trace -d $R/$O/trace8 -o t -- python $R/scripts/eight_stream_probe.py 8 > $R/$O/trace8.log 2>&1 && break
done
cd $R
( grep "^pass\|^{" $O/trace8.log; python scripts/trace_busy.py $(find $O/trace8 -name "*.db" | head -1) 900 ) > $O/trace8_busy.txt; rm -rf $O/trace8
head -16 $O/trace8_busy.txt

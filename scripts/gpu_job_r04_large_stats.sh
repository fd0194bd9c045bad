#!/bin/bash
# rocprofv3 per-kernel averages of the large-v3 line (config 3): gpurun_out/r04/large_v3_kernel_stats.{md,json}
O=gpurun_out/r04; mkdir -p $O; R=$PWD
export WLK_SYNTHETIC_VOCAB=1
export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --model large-v3 --seconds 10 --steps 1 --warmup 1 --no-cpu-baseline --no-diarization"
for try in 1 2; do
  rm -rf $R/$O/prof_l
  timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_l -o st -- $B > $R/$O/prof_large.log 2>&1 && break
done
cd $R
python scripts/export_profile.py $(find $O/prof_l -name "*.db" | head -1) $O/large_v3_kernel_stats.md "python bench.py --model large-v3 --seconds 10 --steps 1 --warmup 1 (large-v3, 1 stream, 10 s): rocprofv3 --kernel-trace --stats"
rm -rf $O/prof_l
head -40 $O/large_v3_kernel_stats.md | cut -c1-160

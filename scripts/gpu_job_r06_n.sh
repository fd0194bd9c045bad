#!/bin/bash
# round 6, job n: kernel timeline of one call with the early z-score on / off
set -u
O=gpurun_out/r06n; mkdir -p $O; R=$PWD
export WLK_SYNTHETIC_VOCAB=1 TMPDIR=/tmp
cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams --no-large-v3 --full-out /tmp/prof_full.json"
for mode in 1 0; do
  rm -rf /tmp/tr$mode
  WLK_EARLY_Z=$mode timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr$mode -o st -- $B > $R/$O/trace$mode.log 2>&1
  DB=$(find /tmp/tr$mode -name "*.db" | head -1)
  python $R/scripts/trace_gaps.py $DB 100 --all > $R/$O/call100_early$mode.txt 2>&1
  python $R/scripts/export_profile.py $DB $R/$O/stats_early$mode.md "early z = $mode" > /dev/null
done
cd $R
tail -62 $O/call100_early1.txt
echo ======
tail -14 $O/call100_early0.txt

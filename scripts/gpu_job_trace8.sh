mkdir -p gpurun_out; R=$PWD; export TMPDIR=/tmp; cd /tmp
for cfg in "batch:" "nobatch:WLK_BATCH_ENCODE=0 WLK_BATCH_DECODE=0"; do
  n=${cfg%%:*}; e=${cfg#*:}
  env $e timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/trace8_$n -o t -- python $R/scripts/eight_stream_probe.py 8 > $R/gpurun_out/trace8_$n.log 2>&1
  grep "^pass\|^{" $R/gpurun_out/trace8_$n.log
  DB=$(find $R/gpurun_out/trace8_$n -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/scripts/trace_busy.py $DB 900
done
rm -rf $R/gpurun_out/trace8_*/

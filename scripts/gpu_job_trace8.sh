# rocprofv3 kernel trace of the 8-stream workload on one GPU: how busy is the GPU, and with what (profiles/r02_trace8_*.txt)
mkdir -p gpurun_out/r02; R=$PWD; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/trace8_batch -o t -- python $R/scripts/eight_stream_probe.py 8 > $R/gpurun_out/trace8_batch.log 2>&1
( grep "^pass\|^{" $R/gpurun_out/trace8_batch.log; python $R/scripts/trace_busy.py $(find $R/gpurun_out/trace8_batch -name "*.db" | head -1) 900 ) > $R/gpurun_out/r02/trace8_busy.txt
cat $R/gpurun_out/r02/trace8_busy.txt
rm -rf $R/gpurun_out/trace8_batch/

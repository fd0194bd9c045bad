mkdir -p gpurun_out; R=$PWD
timeout 300 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for S in 1 2 3; do WLK_ENC_KSPLIT=$S timeout 120 python bench.py --no-cpu-baseline --no-diarization > gpurun_out/bench_eks$S.json 2> gpurun_out/bench_eks$S.log; done
python - <<PY
import json
for n in (1,2,3):
    d=json.load(open(f"gpurun_out/bench_eks{n}.json"))
    print("enc_ksplit",n, d["value"], d["p50_call_ms"], d["launch_tags"]["enc_attention"])
PY
export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-diarization"
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pmc/stats -o st -- $B > $R/gpurun_out/pmc_stats.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc/mfma -o p -- $B > $R/gpurun_out/pmc_mfma.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc/fetch -o p -- $B > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc/write -o p -- $B > $R/gpurun_out/pmc_write.log 2>&1
cd $R; du -sh gpurun_out/pmc/*; find gpurun_out/pmc -name "*.csv" | head; rm -f gpurun_out/pmc/*/*kernel_trace.csv gpurun_out/pmc/*/*agent_info.csv

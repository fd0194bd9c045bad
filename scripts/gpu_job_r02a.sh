# Round-2 evidence run A (gpurun): full GPU suite, the drop-in over the staged reference, default bench with the
# reference's own CPU path timed beside it, rocprofv3 kernel stats of the bench command.
mkdir -p gpurun_out; R=$PWD
S=$(date +%s); timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -25 > gpurun_out/pytest_gpu.log; echo "pytest gpu $(( $(date +%s) - S )) s"; tail -5 gpurun_out/pytest_gpu.log
if [ -d "$R/.ref_stage/whisperlivekit" ]; then
  export WLK_REFERENCE_ROOT=$R/.ref_stage
  S=$(date +%s); timeout 900 python -m pytest tests/test_gpu_reference_dropin.py -q -m gpu 2>&1 | tail -12 > gpurun_out/dropin_gpu.log; echo "dropin $(( $(date +%s) - S )) s"; tail -3 gpurun_out/dropin_gpu.log
fi
S=$(date +%s); timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.log; echo "default bench rc=$? $(( $(date +%s) - S )) s"
unset WLK_REFERENCE_ROOT
export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/stats -o st -- $B > $R/gpurun_out/prof_stats.log 2>&1
cd $R; ls gpurun_out/prof/stats/* | head; rm -f gpurun_out/prof/stats/*/*kernel_trace.csv gpurun_out/prof/stats/*/*agent_info.csv
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_default.json"))
    print("value", d["value"], "rtf", d["rtf"], "p50 lat", d["p50_committed_token_latency_ms"], "roof", d["roofline"]["frac"])
    print("parity", d["parity_checked"])
    print("eight", d["eight_streams"])
    print("cpu", d["cpu_baseline"])
except Exception as e: print("failed", e)
PY

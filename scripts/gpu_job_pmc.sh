# Evidence run on the GPU box (gpurun): parity suite, default bench, 8-stream bench, rocprofv3 kernel stats and the
# three PMC passes over the same bench command, large-v3 line.  Outputs under gpurun_out/ (copied to profiles/ by hand).
mkdir -p gpurun_out; R=$PWD
timeout 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
S=$(date +%s); timeout 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.log; echo "default bench rc=$? $(( $(date +%s) - S )) s"
timeout 200 python bench.py --streams-per-gpu 8 --steps 2 --warmup 1 --no-cpu-baseline --no-diarization > gpurun_out/bench_8streams.json 2> gpurun_out/bench_8streams.log
timeout 300 python bench.py --streams-per-gpu 32 --steps 1 --warmup 1 --no-cpu-baseline --no-diarization > gpurun_out/bench_32streams.json 2> gpurun_out/bench_32streams.log
export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-diarization"
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pmc/stats -o st -- $B > $R/gpurun_out/pmc_stats.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc/mfma -o p -- $B > $R/gpurun_out/pmc_mfma.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc/fetch -o p -- $B > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc/write -o p -- $B > $R/gpurun_out/pmc_write.log 2>&1
cd $R; rm -f gpurun_out/pmc/*/*kernel_trace.csv gpurun_out/pmc/*/*agent_info.csv
timeout 420 python bench.py --model large-v3 --seconds 10 --steps 1 --warmup 1 --no-cpu-baseline --no-diarization > gpurun_out/bench_large_v3.json 2> gpurun_out/bench_large_v3.log; echo "large-v3 rc=$?"
python - <<PY
import json
for n in ("default","8streams","32streams","large_v3"):
    try:
        d=json.load(open(f"gpurun_out/bench_{n}.json")); print(n, d["value"], d["rtf"], d["p50_call_ms"], d["p50_committed_token_latency_ms"], d["roofline"]["frac"], (d.get("diarization") or {}).get("p50_chunk_ms"), (d.get("vad") or {}).get("p50_chunk_ms"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e: print(n, "failed", e)
PY

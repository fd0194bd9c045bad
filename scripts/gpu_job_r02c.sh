# Round-2 run C: batched encodes (grid.y = sessions) + batched decode steps: parity first, then A/B numbers.
mkdir -p gpurun_out; R=$PWD
S=$(date +%s); timeout 900 python -m pytest tests/test_gpu_serving.py tests/test_gpu_parity.py -q -m gpu -x -k "eight_threads or library_decode_loop or bench_checks or stream_matches or encoder_decoder or mel" 2>&1 | tail -25 > gpurun_out/pytest_c.log; echo "pytest $(( $(date +%s) - S )) s"; tail -8 gpurun_out/pytest_c.log
for cfg in "c:" "c_noencbatch:WLK_BATCH_ENCODE=0" "c_nobatch:WLK_BATCH_ENCODE=0 WLK_BATCH_DECODE=0"; do
  n=${cfg%%:*}; e=${cfg#*:}
  env $e timeout 600 python bench.py --no-cpu-baseline --no-diarization > gpurun_out/bench_$n.json 2> gpurun_out/bench_$n.log
done
python - <<PY
import json
for n in ("c","c_noencbatch","c_nobatch"):
    try:
        d=json.load(open(f"gpurun_out/bench_{n}.json"))
        e=d["eight_streams"]
        print(n, "value", d["value"], "p50 call", d["p50_call_ms"], "| eight", e["audio_s_per_s"], "p50/p95 lat", e["p50_committed_token_latency_ms"], e["p95_committed_token_latency_ms"], "p50/p95 call", e["p50_call_ms"], e["p95_call_ms"], e.get("batch_engine_rank0"))
        pc=d["parity_checked"]; print("   parity", pc["decisions"], pc["identical"], pc["tie_divergences"], pc["mismatches"], pc["words_identical_sessions"], "/", pc["sessions"])
    except Exception as ex: print(n, "failed", ex)
PY

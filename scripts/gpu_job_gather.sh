for g in 0 500 1500 4000; do
  WLK_ENCODE_GATHER_US=$g timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-diarization > gpurun_out/bench_g$g.json 2> gpurun_out/bench_g$g.log
done
python - <<PY
import json
for g in (0,500,1500,4000):
    try:
        d=json.load(open(f"gpurun_out/bench_g{g}.json")); e=d["eight_streams"]; st=e["batch_engine_rank0"]
        print(g, "eight", e["audio_s_per_s"], "p50/p95 lat", e["p50_committed_token_latency_ms"], e["p95_committed_token_latency_ms"], "call", e["p50_call_ms"], e["p95_call_ms"], "enc batch", st["mean_sessions_per_encode_batch"], "dec rows", st["mean_rows_per_batched_step"], "parity", d["parity_checked"]["identical"], d["parity_checked"]["decisions"])
    except Exception as ex: print(g, "failed", ex)
PY

mkdir -p gpurun_out; R=$PWD
S=$(date +%s); timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 > gpurun_out/pytest_d.log; echo "pytest $(( $(date +%s) - S )) s"; tail -5 gpurun_out/pytest_d.log
for cfg in "d:" "d_ldsattn:WLK_ENC_ATTN=lds"; do
  n=${cfg%%:*}; e=${cfg#*:}
  env $e timeout 600 python bench.py --no-cpu-baseline --no-diarization > gpurun_out/bench_$n.json 2> gpurun_out/bench_$n.log
done
python - <<PY
import json
for n in ("d","d_ldsattn"):
    try:
        d=json.load(open(f"gpurun_out/bench_{n}.json"))
        e=d["eight_streams"]
        print(n, "value", d["value"], "p50 call", d["p50_call_ms"], "| eight", e["audio_s_per_s"], "p50/p95 lat", e["p50_committed_token_latency_ms"], e["p95_committed_token_latency_ms"], "p50/p95 call", e["p50_call_ms"], e["p95_call_ms"])
        print("   enc_attention", d["launch_tags"]["enc_attention"], "roof", d["roofline"]["frac"])
        pc=d["parity_checked"]; print("   parity", pc["decisions"], pc["identical"], pc["tie_divergences"], pc["mismatches"], pc["words_identical_sessions"], "/", pc["sessions"])
    except Exception as ex: print(n, "failed", ex)
PY

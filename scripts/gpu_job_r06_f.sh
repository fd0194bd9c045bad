O=gpurun_out/r06f; mkdir -p $O
S=$(date +%s); timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_sortformer.py -q -m gpu -x 2>&1 | tail -25 > $O/pytest.log; echo "pytest $(( $(date +%s) - S )) s: $(tail -1 $O/pytest.log)"; grep -E "FAILED|Error|assert|^E " $O/pytest.log | head -30
S=$(date +%s); timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_line.json 2> $O/bench.log; echo "bench rc=$? $(( $(date +%s) - S )) s; line bytes $(tail -1 $O/bench_line.json | wc -c)"
cp bench_full.json $O/
grep -E "pipeline|config-4|8-stream|diarization leg|config-3 leg done|summary" $O/bench.log | cut -c1-700
python - <<PY
import json
d=json.load(open("$O/bench_full.json"))
print(json.dumps(d.get("pipeline"))[:1500])
print(json.dumps(d.get("asr_plus_diarization_8_sessions"))[:900])
print(json.dumps(d.get("diarization"))[:900])
l=d.get("large_v3") or {}
print({k:l.get(k) for k in ("audio_s_per_s","rtf","p50_committed_token_latency_ms","p95_committed_token_latency_ms","committed_tokens","decisions","identical","parity_ok","error")}, json.dumps(l.get("cpu_baseline"))[:600])
PY

timeout 600 python -m pytest tests/test_gpu_serving.py -q -m gpu -k "eight_threads or stacked" 2>&1 | grep -v amdgpu.ids | tail -3
echo "8 streams (lane closed below 9 sessions): $(python scripts/eight_stream_probe.py 8 2>&1 | grep '^pass 1')"
mkdir -p gpurun_out/r03
timeout 600 python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.log; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/r03/bench_default.json")); e=d.get("eight_streams") or {}
print("value", d["value"], "eight", e.get("audio_s_per_s"), "roof", d["roofline"]["frac"], d["roofline"].get("frac_at_rocprof_duration"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "parity", d["parity_ok"])
print(json.dumps(d.get("batch_transcribe"))); print(json.dumps(d.get("translation")))
PY

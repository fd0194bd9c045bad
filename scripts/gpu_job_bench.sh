O=gpurun_out/r03d; mkdir -p $O
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "bench rc=$?"
tail -3 $O/bench_default.log | cut -c1-1500
python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
for k in ("asr_plus_diarization_8_sessions","word_alignment","dtw","cpu_baseline"):
    print(k, json.dumps(d.get(k))[:900])
print("roofline", json.dumps(d["roofline"])[:1200])
PY

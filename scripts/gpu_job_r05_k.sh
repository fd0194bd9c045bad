# Round 5, eleventh GPU call: the GPU tests added / repaired since the evidence run, then the default bench line once more
# (its roofline now reads the committed r05 profiles)
O=gpurun_out/r05k; mkdir -p $O
S=$(date +%s); timeout 900 python -m pytest tests/test_checkpoint_formats.py tests/test_checkpoint_ingest.py tests/test_nllb.py tests/test_translation.py tests/test_gpu_reference_dropin.py -q -m gpu 2>&1 | tail -25 > $O/pytest.log; echo "pytest $(( $(date +%s) - S )) s: $(tail -1 $O/pytest.log)"
S=$(date +%s); timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "default bench rc=$? $(( $(date +%s) - S )) s"
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1]); e=d.get("eight_streams") or {}; l=d.get("large_v3") or {}; r=d["roofline"]
print("value", d["value"], "rtf", d["rtf"], "| eight", e.get("audio_s_per_s"), "| lv3", l.get("audio_s_per_s"), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
print({k: r.get(k) for k in ("frac","frac_of_method_ceiling","avg_launch_us","traffic","frac_at_rocprof_duration","rocprof_avg_launch_us","mfma_util_pmc")}); print(r.get("rocprof_source")); print(r.get("traffic_source"))
lr=l.get("roofline") or {}; print("lv3", {k: lr.get(k) for k in ("frac","frac_of_method_ceiling","traffic","frac_at_rocprof_duration","rocprof_avg_launch_us")})
PY

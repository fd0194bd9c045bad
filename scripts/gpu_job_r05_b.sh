# Round 5, second GPU call: the GPU tests added since the first call (config-5 session glue, checkpoint ingest, NLLB 600M
# against transformers), then the DEFAULT bench line exactly as the driver runs it (now with the config-3 leg inside) with
# its wall time.
O=gpurun_out/r05b; mkdir -p $O
S=$(date +%s); timeout 900 python -m pytest tests/test_translation.py tests/test_checkpoint_ingest.py tests/test_nllb.py -q -m gpu -x 2>&1 | tail -15 > $O/pytest_new.log; echo "pytest new $(( $(date +%s) - S )) s: $(tail -1 $O/pytest_new.log)"
S=$(date +%s); timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "default bench rc=$? $(( $(date +%s) - S )) s"
tail -5 $O/bench_default.log
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
e=d.get("eight_streams") or {}; l=d.get("large_v3") or {}
print("value", d["value"], "ms", d["ms_per_step"], "roof", d["roofline"]["frac"], d["roofline"].get("frac_of_method_ceiling"), "| eight", e.get("audio_s_per_s"), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
print("encode", d["roofline"].get("encode")); print("step", d["roofline"].get("step"))
print("large_v3", {k: l.get(k) for k in ("audio_s_per_s","decisions","identical","parity_ok","leg_wall_s","error","cpu_baseline")})
print("lv3 roof", {k: (l.get("roofline") or {}).get(k) for k in ("frac","traffic","encode","step")})
p=d["parity_checked"]; print("parity", d["parity_ok"], p and {k: p.get(k) for k in ("sessions","decisions","identical","tie_divergences","unchecked_calls","mismatches","words_identical_sessions")})
PY

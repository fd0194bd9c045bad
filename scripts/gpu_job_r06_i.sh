O=gpurun_out/r06i; mkdir -p $O
S=$(date +%s); timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_x3.py tests/test_gpu_sortformer.py -q -m gpu -k "medium or large or x3 or kp_family" 2>&1 | tail -8 > $O/pytest.log; echo "pytest $(( $(date +%s) - S )) s: $(tail -1 $O/pytest.log)"; grep -E "FAILED|^E " $O/pytest.log | head
python scripts/kp_tile_probe.py 2>&1 | grep -A 12 "below 512" > $O/kp16_probe.txt; cat $O/kp16_probe.txt | cut -c1-250
B="python bench.py --model large-v3 --seconds 30 --seed 8 --steps 1 --warmup 1 --no-cpu-baseline --no-diarization --no-large-v3"
for rep in 1 2; do
  $B --full-out $O/lv3_fused_$rep.json > /dev/null 2> $O/lv3_fused_$rep.log; echo "fused rc=$?"
  WLK_X3_NARROW=0 $B --full-out $O/lv3_fp32narrow_$rep.json > /dev/null 2> $O/lv3_fp32narrow_$rep.log; echo "fp32 narrow rc=$?"
done
python - <<PY
import json
for n in ("fused_1","fp32narrow_1","fused_2","fp32narrow_2"):
    d=json.load(open("$O/lv3_%s.json"%n)); r=d["roofline"]; pc=d["parity_checked"]
    print(n, "value", d["value"], "p50 call", d["p50_call_ms"], "encode us", r["encode"]["us"], "parity", d["parity_ok"], pc["decisions"], pc["identical"], len(pc["tie_divergences"]),
          {k.split(" (")[0]: (v["avg_launch_us"], v["launches"]) for k, v in r["mfma_families"].items()})
PY

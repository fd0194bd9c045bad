#!/bin/bash
# A/B of the Infinity-Cache prefetcher beside the decode step (decoder.hip: mall_prefetch_step_kernel), same box, alternating
mkdir -p gpurun_out/r06k
MODEL=${MODEL:-large-v3}
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 400 python bench.py --model $MODEL --seconds 30 --seed ${SEED:-8} --steps 1 --warmup 1 --no-cpu-baseline --no-eight-streams --no-diarization --no-large-v3 --full-out gpurun_out/r06k/full_$name.json 2>gpurun_out/r06k/err_$name.log | tail -1 > gpurun_out/r06k/line_$name.json
  python - <<PY
import json
try:
    l=json.load(open("gpurun_out/r06k/line_$name.json"))
    r=l.get("roofline",{})
    p=l.get("parity_checked",{})
    print("$name", "audio_s/s", l["value"], "p50_call_ms", l.get("p50_call_ms"), "step_us", r.get("step",{}).get("us"), "step_frac", r.get("step",{}).get("frac_of_hbm"), "encode_us", r.get("encode",{}).get("us"), "decisions", p.get("identical"), "/", p.get("decisions"))
except Exception as e:
    print("$name FAILED", e)
PY
}
for i in 1 2; do
  run pf0_$i WLK_MALL_PREFETCH=0
  run pf1_lead0_$i WLK_MALL_PREFETCH=1 WLK_MALL_LEAD=0
  run pf1_lead1_$i WLK_MALL_PREFETCH=1 WLK_MALL_LEAD=1
done 2>&1 | tee gpurun_out/r06k/mall_ab_$MODEL.txt

#!/bin/bash
# 64-row tiles of the two-wave X3 kernel where 96-row tiles leave CUs idle (N = 1280 projections of large-v3): tests, per-shape probe,
# large-v3 stream A/B (WLK_X3_BM=96 = every launch on 96-row tiles), same box, alternating
mkdir -p gpurun_out/r06l
timeout 400 python -m pytest tests/test_gpu_x3.py -x -q -m gpu 2>&1 | tail -3
for bm in 96 0; do echo "== WLK_X3_BM=$bm (0 = the launch rule)"; WLK_X3_BM=$bm X3_PROBE_NARROW=1 timeout 200 python scripts/x3_probe.py 2>&1 | grep -v amdgpu.ids | grep "large-v3\|sf x" | cut -c1-110; done | tee gpurun_out/r06l/x3_bm64_probe.txt
run() {
  local name=$1; shift
  env "$@" timeout 400 python bench.py --model ${MODEL:-large-v3} --seconds 30 --seed 8 --steps 1 --warmup 1 --no-cpu-baseline --no-eight-streams --no-diarization --no-large-v3 --full-out gpurun_out/r06l/full_$name.json 2>gpurun_out/r06l/err_$name.log | tail -1 > gpurun_out/r06l/line_$name.json
  python - <<PY
import json
try:
    l=json.load(open("gpurun_out/r06l/line_$name.json")); r=l.get("roofline",{}); p=l.get("parity_checked",{})
    print("$name", "audio_s/s", l["value"], "p50_call_ms", l.get("p50_call_ms"), "encode_us", r.get("encode",{}).get("us"), "x3 avg us", r.get("avg_launch_us"), "frac", r.get("frac"), "decisions", p.get("identical"), "/", p.get("decisions"))
except Exception as e:
    print("$name FAILED", e)
PY
}
for i in 1 2; do
  run bm96_$i WLK_X3_BM=96
  run rule_$i WLK_X3_BM=0
done 2>&1 | tee gpurun_out/r06l/bm64_ab_large-v3.txt

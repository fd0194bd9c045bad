#!/bin/bash
# Round 4: large-v3 decode-step anatomy under environment variants of the same library.
# usage: gpu_job_r04_large.sh <tag> "<ENV..>" ["<ENV..>" ...]   (an empty string = the defaults)
set -u
TAG="$1"; shift
OUT=gpurun_out/r04${TAG}; mkdir -p $OUT
export WLK_SYNTHETIC_VOCAB=1
B="python bench.py --model large-v3 --seconds 10 --steps 1 --warmup 1 --no-cpu-baseline --no-diarization"
i=0
for V in "$@"; do
  i=$((i+1))
  for rep in 1 2; do
    env $V WLK_STEP_TIMING=1 timeout 300 $B > $OUT/v${i}_$rep.json 2> $OUT/v${i}_$rep.err; rc=$?
    python - "$OUT/v${i}_$rep.json" "$V" $rc <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("variant [%s] rc=%s: no line (%s)" % (sys.argv[2], sys.argv[3], e)); sys.exit(0)
pc = d.get("parity_checked") or {}
tags = d.get("launch_tags") or {}
print("variant [%s] rc=%s value %.3f ms_per_step %.1f p50_call %.2f parity %s/%s mism %s" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d.get("p50_call_ms") or 0, pc.get("identical"), pc.get("decisions"), pc.get("mismatches")))
keys = [k for k in tags if k.startswith("dec_") or k.startswith("sel_")]
keys.sort(key=lambda k: -tags[k]["ms"])
print("   " + "  ".join("%s %.1fus x%d" % (k, 1e3 * tags[k]["ms"] / max(tags[k]["launches"], 1), tags[k]["launches"]) for k in keys[:14]))
PY
  done
  grep -h "one-replay" $OUT/v${i}_1.err | tail -1
done

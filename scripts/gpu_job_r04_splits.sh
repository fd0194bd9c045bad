#!/bin/bash
# Round 4: key ranges of the prefill cross-attention (WLK_FLASH_SPLITS) - full GPU suite once, then short bench lines per value
set -u
TAG="$1"; shift
OUT=gpurun_out/r04${TAG}; mkdir -p $OUT
export WLK_SYNTHETIC_VOCAB=1
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams"
for rep in 1 2; do
for V in "$@"; do
  env WLK_FLASH_SPLITS=$V timeout 300 $B 2>$OUT/s${V}_$rep.err | tail -1 > $OUT/s${V}_$rep.json
  python - "$OUT/s${V}_$rep.json" $V <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read()); pc = j.get('parity_checked') or {}; t = j.get('launch_tags') or {}
    x = t.get('dec_cross_attention_prefill') or {}
    print('splits', sys.argv[2], 'value', j['value'], 'ms', j['ms_per_step'], 'decisions', pc.get('decisions'), 'identical', pc.get('identical'), 'ties', pc.get('tie_divergences'), 'mism', pc.get('mismatches'), 'cross prefill us', round(1e3 * x.get('ms', 0) / max(x.get('launches', 1), 1), 2))
except Exception as e:
    print('splits', sys.argv[2], 'ERR', e)
PY
done
done
python scripts/step_probe.py base.en 15 40 60 2>/dev/null | tail -1

O=gpurun_out/r03c; mkdir -p $O
S=$(date +%s); timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > $O/pytest_gpu.log; echo "pytest gpu $(( $(date +%s) - S )) s: $(tail -1 $O/pytest_gpu.log)"; grep -i "fail\|error" $O/pytest_gpu.log | head
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export WLK_NO_XQ_FOLD=1; else unset WLK_NO_XQ_FOLD; fi
  WLK_STEP_TIMING=1 timeout 300 python bench.py --no-cpu-baseline --no-diarization --steps 5 2> $O/bench_fold$v.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['parity_checked']; e=d['eight_streams']
print('nofold=$v', 'value', d['value'], 'eight', e['audio_s_per_s'], 'parity', d['parity_ok'], p['identical'], p['decisions'], p['tie_divergences'], p['mismatches'])"
  grep "one-replay steps" $O/bench_fold$v.log | tail -2
done

# Round 5, thirteenth GPU call: Sortformer workspace pool - tests, config 4 with 1 / 4 / 8 workspaces
O=gpurun_out/r05m; mkdir -p $O
S=$(date +%s); timeout 900 python -m pytest tests/test_gpu_sortformer.py tests/test_gpu_serving.py -q -m gpu -x 2>&1 | tail -12 > $O/pytest.log; echo "pytest $(( $(date +%s) - S )) s: $(tail -1 $O/pytest.log)"
BS="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-large-v3 --no-parity"
: > $O/ab.txt
for i in 1 2; do for v in 1 4 8; do
  export WLK_SF_WORKSPACES=$v
  echo -n "workspaces=$v " >> $O/ab.txt
  timeout 300 $BS 2>$O/bench_${v}_$i.err | tail -1 > $O/bench_${v}_$i.json
  python -c "import json; d=json.load(open('$O/bench_${v}_$i.json')); c=d.get('asr_plus_diarization_8_sessions') or {}; e=d.get('eight_streams') or {}; print(d['value'], 'audio-s/s, eight', e.get('audio_s_per_s'), '| cfg4 asr', c.get('asr_audio_s_per_s'), 'asr p50/p95 call', c.get('asr_p50_call_ms'), c.get('asr_p95_call_ms'), 'diar p50/p95 chunk ms', c.get('diar_p50_chunk_ms'), c.get('diar_p95_chunk_ms'), 'diar audio-s/s', c.get('diar_audio_s_per_s'), 'wall', c.get('wall_s'))" >> $O/ab.txt 2>&1
done; done
unset WLK_SF_WORKSPACES
cat $O/ab.txt

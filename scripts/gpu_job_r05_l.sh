# Round 5, twelfth GPU call: config 4 on one GPU - the diarizer's stream at high priority (WLK_SF_PRIORITY=hi) against normal
O=gpurun_out/r05l; mkdir -p $O
BS="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-large-v3 --no-parity"
: > $O/ab.txt
for i in 1 2; do for v in normal hi; do
  if [ $v = hi ]; then export WLK_SF_PRIORITY=hi; else unset WLK_SF_PRIORITY; fi
  echo -n "sf_priority=$v " >> $O/ab.txt
  timeout 300 $BS 2>$O/bench_${v}_$i.err | tail -1 > $O/bench_${v}_$i.json
  python -c "import json; d=json.load(open('$O/bench_${v}_$i.json')); c=d.get('asr_plus_diarization_8_sessions') or {}; e=d.get('eight_streams') or {}; print(d['value'], 'audio-s/s, eight', e.get('audio_s_per_s'), '| cfg4 asr', c.get('asr_audio_s_per_s'), 'asr p50/p95 call', c.get('asr_p50_call_ms'), c.get('asr_p95_call_ms'), 'diar p50/p95 chunk ms', c.get('diar_p50_chunk_ms'), c.get('diar_p95_chunk_ms'), 'diar alone', (d.get('diarization') or {}).get('ms_per_chunk'))" >> $O/ab.txt 2>&1
done; done
unset WLK_SF_PRIORITY
cat $O/ab.txt

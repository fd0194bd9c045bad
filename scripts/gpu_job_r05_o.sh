# Round 5: the 8-stream kernel trace at the final tree (GPU busy fraction, per-kernel sums of the timed pass), and two more
# default bench lines (box kinds)
O=gpurun_out/r05o; mkdir -p $O; R=$PWD
export WLK_SYNTHETIC_VOCAB=1
export TMPDIR=/tmp; cd /tmp
for try in 1 2; do
  rm -rf $R/$O/trace8
  timeout 300 rocprofv3 --kernel-trace -d $R/$O/trace8 -o t -- python $R/scripts/eight_stream_probe.py 8 > $R/$O/trace8.log 2>&1 && break
done
cd $R
( grep "^pass\|^{" $O/trace8.log; python scripts/trace_busy.py $(find $O/trace8 -name "*.db" | head -1) 800 ) > $O/trace8_busy.txt; rm -rf $O/trace8
head -18 $O/trace8_busy.txt | cut -c1-200
unset WLK_SYNTHETIC_VOCAB
for i in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.log
  python -c "import json; d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); e=d.get('eight_streams') or {}; l=d.get('large_v3') or {}; c=d.get('asr_plus_diarization_8_sessions') or {}; print('bench $i:', d['value'], 'audio-s/s, eight', e.get('audio_s_per_s'), 'lv3', l.get('audio_s_per_s'), 'cfg4 asr', c.get('asr_audio_s_per_s'), 'diar', c.get('diar_audio_s_per_s'), 'diar p50/p95', c.get('diar_p50_chunk_ms'), c.get('diar_p95_chunk_ms'), 'wall', c.get('wall_s'), 'parity', d['parity_ok'])"
done

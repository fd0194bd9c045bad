# Round 5, third GPU call: the persistent X3 wide kernel - its tests, the probe on the encoder shapes with one workgroup per
# tile (WLK_X3_PERSIST=0 = the round-4 launch) against the persistent walk, a short bench pair, and the NLLB divergence probe.
O=gpurun_out/r05c; mkdir -p $O
S=$(date +%s); timeout 600 python -m pytest tests/test_gpu_x3.py -q -m gpu -x 2>&1 | tail -15 > $O/pytest_x3.log; echo "pytest x3 $(( $(date +%s) - S )) s: $(tail -1 $O/pytest_x3.log)"
for v in 0 1; do echo "== WLK_X3_PERSIST=$v"; WLK_X3_PERSIST=$v timeout 200 python scripts/x3_probe.py 2>&1 | grep -v attention; done > $O/x3_persist_probe.txt
for v in 1 2 3; do echo "== WLK_X3_ABL=$v (persistent)"; WLK_X3_ABL=$v timeout 200 python scripts/x3_probe.py 2>&1 | grep -v attention; done >> $O/x3_persist_probe.txt
cat $O/x3_persist_probe.txt | cut -c1-110
timeout 200 python scripts/probes/nllb_divergence_probe.py > $O/nllb_divergence.txt 2>&1; grep -v "^   step" $O/nllb_divergence.txt | head; grep -B2 -A2 "hip [0-9]* oracle" $O/nllb_divergence.txt | awk '{ if ($NF != $(NF-2)) print }' | head -8
BS="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diarization --no-large-v3"
: > $O/ab.txt
for i in 1 2; do for v in 1 0; do
  echo -n "persist=$v " >> $O/ab.txt
  WLK_X3_PERSIST=$v timeout 300 $BS 2>/dev/null | tail -1 > $O/bench_p${v}_$i.json
  python -c "import json; d=json.load(open('$O/bench_p${v}_$i.json')); e=d.get('eight_streams') or {}; print(d['value'], 'audio-s/s', d['ms_per_step'], 'ms/stream, eight', e.get('audio_s_per_s'), 'parity_ok', d['parity_ok'], 'x3 us', d['roofline'].get('avg_launch_us'), 'encode us', (d['roofline'].get('encode') or {}).get('us'))" >> $O/ab.txt 2>&1
done; done
cat $O/ab.txt
for v in 1 0; do WLK_X3_PERSIST=$v timeout 400 python bench.py --model large-v3 --seconds 10 --steps 1 --warmup 1 --no-cpu-baseline --no-diarization --no-large-v3 --no-eight-streams 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('large-v3 persist=$v', d['value'], 'audio-s/s parity', d['parity_ok'], 'x3 us', d['roofline'].get('avg_launch_us'), 'encode us', (d['roofline'].get('encode') or {}).get('us'))"; done | tee $O/ab_large.txt

# Round 5, fifth GPU call: the wide X3 kernel with the weight fragments read straight into registers (W3F layout, A-only ring
# of five slots, three loader waves) - tests, probe with ablations, bench pair against the previous commit's library.
O=gpurun_out/r05e; mkdir -p $O
S=$(date +%s); timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -15 > $O/pytest.log; echo "pytest $(( $(date +%s) - S )) s: $(tail -1 $O/pytest.log)"
( for v in "A=0" "WLK_X3_ABL=1" "WLK_X3_ABL=2" "WLK_X3_ABL=3" "WLK_X3_ABL=4"; do echo "== $v"; env $v timeout 200 python scripts/x3_probe.py 2>&1 | grep -v "attention\|amdgpu.ids"; done ) > $O/x3_probe.txt
cut -c1-100 $O/x3_probe.txt
BS="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diarization --no-large-v3"
: > $O/ab.txt
for i in 1 2; do for lib in tree r5a; do
  if [ $lib = r5a ]; then export WLK_HIP_LIB=$PWD/whisperlivekit_amd/libwlk_hip_r5a.so; else unset WLK_HIP_LIB; fi
  echo -n "$lib " >> $O/ab.txt
  timeout 300 $BS 2>$O/bench_${lib}_$i.err | tail -1 > $O/bench_${lib}_$i.json
  python -c "import json; d=json.load(open('$O/bench_${lib}_$i.json')); e=d.get('eight_streams') or {}; print(d['value'], 'audio-s/s', d['ms_per_step'], 'ms/stream, eight', e.get('audio_s_per_s'), 'parity_ok', d['parity_ok'], 'x3 us', d['roofline'].get('avg_launch_us'), 'encode us', (d['roofline'].get('encode') or {}).get('us'))" >> $O/ab.txt 2>&1
done; done
unset WLK_HIP_LIB
cat $O/ab.txt
timeout 400 python bench.py --model large-v3 --seconds 10 --steps 1 --warmup 1 --no-cpu-baseline --no-diarization --no-large-v3 --no-eight-streams 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('large-v3', d['value'], 'audio-s/s parity', d['parity_ok'], 'x3 us', d['roofline'].get('avg_launch_us'), 'encode us', (d['roofline'].get('encode') or {}).get('us'))" | tee $O/ab_large.txt

# Round 5, tenth GPU call: why do the reference drop-in tests fail in the full suite? (alone, then behind the new test files);
# the shared erf epilogue of the wide X3 kernel: tests, probe, bench pair
O=gpurun_out/r05j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_reference_dropin.py -q -m gpu -x 2>&1 | tail -40 > $O/dropin_alone.log; echo "dropin alone: $(tail -1 $O/dropin_alone.log)"
timeout 600 python -m pytest tests/test_checkpoint_ingest.py tests/test_translation.py tests/test_gpu_reference_dropin.py -q -m gpu 2>&1 | tail -60 > $O/dropin_after.log; echo "dropin after new tests: $(tail -1 $O/dropin_after.log)"
S=$(date +%s); timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -15 > $O/pytest.log; echo "pytest x3+parity $(( $(date +%s) - S )) s: $(tail -1 $O/pytest.log)"
( for v in "A=0" "WLK_X3_ABL=3"; do echo "== $v"; env $v timeout 200 python scripts/x3_probe.py 2>&1 | grep -v "attention\|amdgpu.ids"; done ) > $O/x3_probe.txt
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

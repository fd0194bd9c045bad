# Round 5, first GPU call: the full GPU suite on the tree (new: Sortformer vs transformers' Parakeet ports, tie
# re-synchronisation), then short same-box bench lines: tree vs the round-4 library kept beside it (libwlk_hip_r4.so).
O=gpurun_out/r05a; mkdir -p $O
S=$(date +%s); timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > $O/pytest_gpu.log; echo "pytest gpu $(( $(date +%s) - S )) s: $(tail -1 $O/pytest_gpu.log)"
cp gpurun_out/parity_report.json $O/ 2>/dev/null
BS="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diarization"
: > $O/same_box_ab.txt
for i in 1 2; do
  for lib in tree r4; do
    if [ $lib = r4 ]; then export WLK_HIP_LIB=$PWD/whisperlivekit_amd/libwlk_hip_r4.so; else unset WLK_HIP_LIB; fi
    echo -n "$lib " >> $O/same_box_ab.txt
    timeout 300 $BS 2>$O/bench_${lib}_$i.err | tail -1 > $O/bench_${lib}_$i.json
    python -c "import json,sys; d=json.load(open('$O/bench_${lib}_$i.json')); e=d.get('eight_streams') or {}; print(d['value'], 'audio-s/s', d['ms_per_step'], 'ms/stream, eight', e.get('audio_s_per_s'), 'parity_ok', d['parity_ok'], (d.get('parity_checked') or {}).get('tie_divergences'))" >> $O/same_box_ab.txt 2>&1
  done
done
unset WLK_HIP_LIB
cat $O/same_box_ab.txt

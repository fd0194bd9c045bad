# Round 5, ninth GPU call: 8 streams on one GPU - the engine's gather / prefill-lane knobs, two passes each
O=gpurun_out/r05i; mkdir -p $O
: > $O/eight_sweep.txt
for v in "A=0" "WLK_ENCODE_GATHER_US=300" "WLK_ENCODE_GATHER_US=1000" "WLK_ENCODE_GATHER_US=2500" "WLK_PREFILL_MIN_SESSIONS=2" "WLK_PREFILL_MIN_SESSIONS=4 WLK_PREFILL_GATHER_US=200" "WLK_ENCODE_GATHER_US=1000 WLK_PREFILL_MIN_SESSIONS=2" "A=1"; do
  echo "== $v" >> $O/eight_sweep.txt
  env $v timeout 200 python scripts/eight_stream_probe.py 8 2>&1 | grep -v "amdgpu.ids" >> $O/eight_sweep.txt
done
cut -c1-260 $O/eight_sweep.txt

# A/B of environment switches on the 8-stream leg (each: one warm-up + one timed pass of 8 x 30 s)
mkdir -p gpurun_out
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 300 python scripts/eight_stream_probe.py 8 2>&1 | grep "^pass 1\|^{" | sed "s/^/[$e] /"
done

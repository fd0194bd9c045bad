# What does one prefill chain cost under load?  8 streams (and 1) with every prefill enqueued once / twice / three times
# (WLK_PROBE_PREFILL_REPEAT: same results, the chain just runs again); alternating so that box drift cancels.
for rep in 1 2 1 2 3 1; do
  echo "== 8 streams, prefill x $rep: $(WLK_PROBE_PREFILL_REPEAT=$rep python scripts/eight_stream_probe.py 8 2>&1 | grep '^pass 1')"
done
for rep in 1 2 1 2; do
  echo "== 1 stream, prefill x $rep: $(WLK_PROBE_PREFILL_REPEAT=$rep python scripts/eight_stream_probe.py 1 2>&1 | grep '^pass 1')"
done

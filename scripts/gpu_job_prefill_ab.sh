for v in 1 0 1 0 1; do
  echo "== 8 streams, WLK_BATCH_PREFILL=$v: $(WLK_BATCH_PREFILL=$v python scripts/eight_stream_probe.py 8 2>&1 | grep '^pass 1\|prefill' | tr '\n' ' ' | cut -c1-400)"
done

for n in 16 16 32 32; do for v in 1 0; do
  echo "== $n streams, WLK_BATCH_PREFILL=$v: $(WLK_BATCH_PREFILL=$v python scripts/eight_stream_probe.py $n 2>&1 | grep '^pass 1\|prefill' | sed 's/.iterations.*prefill_batches/ prefill_batches/' | tr '\n' ' ' | cut -c1-300)"
done; done

for q in 2 4 8 1 2 4; do
  echo "== 8 streams, GPU_MAX_HW_QUEUES=$q: $(GPU_MAX_HW_QUEUES=$q python scripts/eight_stream_probe.py 8 2>&1 | grep '^pass 1')"
done

O=gpurun_out/r06e; mkdir -p $O
S=$(date +%s); timeout 900 python -m pytest tests/test_gpu_sortformer.py -q -m gpu -x 2>&1 | tail -6 > $O/pytest_sf.log; echo "pytest $(( $(date +%s) - S )) s: $(tail -1 $O/pytest_sf.log)"; grep -E "FAILED|Error|assert" $O/pytest_sf.log | head
echo "--- solo"; python scripts/diar_probe8.py 1 30 2>&1 | grep "rep 1"
echo "--- 8 diar alone"; python scripts/diar_probe8.py 8 30 2>&1 | grep "rep 1"
echo "--- 8 diar alone, 1 lane"; WLK_SF_WORKSPACES=1 python scripts/diar_probe8.py 8 30 2>&1 | grep "rep 1"
echo "--- cfg4 default"; python scripts/diar_probe8.py 8 30 8 2>&1 | grep rep
echo "--- cfg4 1 lane"; WLK_SF_WORKSPACES=1 python scripts/diar_probe8.py 8 30 8 2>&1 | grep rep
echo "--- cfg4 three calls"; PROBE_THREE_CALLS=1 python scripts/diar_probe8.py 8 30 8 2>&1 | grep rep
echo "--- cfg4 4 hw queues"; GPU_MAX_HW_QUEUES=4 python scripts/diar_probe8.py 8 30 8 2>&1 | grep rep
echo "--- cfg4 4 hw queues 1 lane"; GPU_MAX_HW_QUEUES=4 WLK_SF_WORKSPACES=1 python scripts/diar_probe8.py 8 30 8 2>&1 | grep rep

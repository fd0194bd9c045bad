O=gpurun_out/r06c; mkdir -p $O
echo "--- cfg4: 8 diar + 8 asr, default (2 lanes, batch 8, 2 hw queues)"; python scripts/diar_probe8.py 8 30 8 2>&1 | grep rep
echo "--- 1 lane"; WLK_SF_WORKSPACES=1 python scripts/diar_probe8.py 8 30 8 2>&1 | grep rep
echo "--- batch 1, 4 lanes (round 5)"; WLK_SF_BATCH=1 WLK_SF_WORKSPACES=4 python scripts/diar_probe8.py 8 30 8 2>&1 | grep rep
echo "--- 4 hw queues"; GPU_MAX_HW_QUEUES=4 python scripts/diar_probe8.py 8 30 8 2>&1 | grep rep
echo "--- 4 hw queues, 1 lane"; GPU_MAX_HW_QUEUES=4 WLK_SF_WORKSPACES=1 python scripts/diar_probe8.py 8 30 8 2>&1 | grep rep
echo "--- asr only"; python scripts/diar_probe8.py 0 30 8 2>&1 | grep rep

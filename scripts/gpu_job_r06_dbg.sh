python -m pytest tests/test_gpu_pipeline.py tests/test_checkpoint_ingest.py -q -m gpu 2>&1 | grep -E "^E|assert|Error|passed|failed" | head -40

O=gpurun_out/r06j; mkdir -p $O
S=$(date +%s); timeout 900 python -m pytest tests/test_gpu_sortformer.py tests/test_gpu_pipeline.py -q -m gpu 2>&1 | tail -6 > $O/pytest.log; echo "pytest $(( $(date +%s) - S )) s: $(tail -1 $O/pytest.log)"; grep -E "FAILED|^E " $O/pytest.log | head
python scripts/diar_probe.py 30 2>&1 | grep rep
python scripts/diar_probe8.py 8 30 2>&1 | grep "rep 1" | cut -c1-300
python scripts/diar_probe8.py 8 30 8 2>&1 | grep "rep 1" | cut -c1-420

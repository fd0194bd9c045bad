O=gpurun_out/r06g; mkdir -p $O
S=$(date +%s); timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $O/pytest_gpu.log; echo "pytest gpu $(( $(date +%s) - S )) s: $(tail -1 $O/pytest_gpu.log)"
grep -E "FAILED|ERROR" $O/pytest_gpu.log | head -30

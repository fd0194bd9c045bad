# Round-3 start-of-round check: GPU suite + default bench (reference CPU leg validated on the box's host cores)
O=gpurun_out/r03a; mkdir -p $O
S=$(date +%s); timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > $O/pytest_gpu.log; echo "pytest gpu $(( $(date +%s) - S )) s: $(tail -1 $O/pytest_gpu.log)"
S=$(date +%s); timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "default bench rc=$? $(( $(date +%s) - S )) s"
tail -3 $O/bench_default.log | cut -c1-1500

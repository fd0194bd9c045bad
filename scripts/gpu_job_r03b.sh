O=gpurun_out/r03b; mkdir -p $O
S=$(date +%s); timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > $O/pytest_gpu.log; echo "pytest gpu $(( $(date +%s) - S )) s: $(tail -1 $O/pytest_gpu.log)"; grep -i "fail\|error" $O/pytest_gpu.log | head
S=$(date +%s); timeout 600 python bench.py --no-cpu-baseline --no-diarization > $O/bench.json 2> $O/bench.log; echo "bench rc=$? $(( $(date +%s) - S )) s"
tail -2 $O/bench.log | cut -c1-1200
WLK_GEMM=classic timeout 600 python bench.py --no-cpu-baseline --no-diarization > $O/bench_classic.json 2> $O/bench_classic.log; tail -1 $O/bench_classic.log | cut -c1-400

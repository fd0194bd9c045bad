mkdir -p gpurun_out/gemm
timeout 300 python scripts/gemm_time_probe.py 50 base 2>&1 | grep -v amdgpu | grep "x1 \|x2 " | tee gpurun_out/gemm/probe_base2.txt
timeout 300 python scripts/gemm_time_probe.py 20 large-v3 2>&1 | grep -v amdgpu | grep "x1 " | tee gpurun_out/gemm/probe_large2.txt
timeout 300 python scripts/gemm_time_probe.py 20 small 2>&1 | grep -v amdgpu | grep "x1 " | tee gpurun_out/gemm/probe_small2.txt

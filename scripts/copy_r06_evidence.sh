#!/bin/bash
# gpurun_out/r06/* (scripts/gpu_job_r06_evidence.sh) -> profiles/r06_*: the files the bench line and the texts cite
set -e
O=gpurun_out/r06
for f in bench_driver_cmd.json bench_driver_cmd.log bench_full.json bench_kernel_stats.json bench_kernel_stats.md pmc_bench.json pmc_bench.md \
         large_v3_kernel_stats.json large_v3_kernel_stats.md large_v3_pmc.json large_v3_pmc.md diar_kernel_stats.json diar_kernel_stats.md \
         diar8_kernel_stats.json diar8_kernel_stats.md diar_pmc.json diar_pmc.md diar_lanes.txt kp_tile_probe.txt x3_narrow_probe.txt x3_probe.txt \
         trace8_busy.txt parity_report.json pytest_gpu.log dropin_gpu_report.txt; do
  [ -f $O/$f ] && cp $O/$f profiles/r06_$f
done
ls -la profiles/r06_bench_driver_cmd.json

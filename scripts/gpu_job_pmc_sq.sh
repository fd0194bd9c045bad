# SQ stall-breakdown counters over the bench command (one pass, 8 SQ slots); aggregated ON the GPU box because the
# raw per-dispatch CSV (~100 MB) exceeds what gpurun copies back.
mkdir -p gpurun_out; R=$PWD; export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-diarization"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc2/sq -o p -- $B > $R/gpurun_out/pmc2_sq.log 2>&1
cd $R; python scripts/export_pmc.py gpurun_out/pmc_sq.md gpurun_out/pmc_sq.json /tmp/pmc2/sq; head -20 gpurun_out/pmc_sq.md

#!/bin/bash
# PMC passes (MFMA busy, FETCH_SIZE, WRITE_SIZE - each its own run, kernel trace only) over the large-v3 line
O=gpurun_out/r04; mkdir -p $O; R=$PWD
export WLK_SYNTHETIC_VOCAB=1
export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --model large-v3 --seconds 10 --steps 1 --warmup 1 --no-cpu-baseline --no-diarization"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/$O/prof_l/mfma -o p -- $B > $R/$O/pmc_l_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_l/fetch -o p -- $B > $R/$O/pmc_l_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_l/write -o p -- $B > $R/$O/pmc_l_write.log 2>&1
cd $R
python scripts/export_pmc.py $O/large_v3_pmc.md $O/large_v3_pmc.json $O/prof_l/mfma $O/prof_l/fetch $O/prof_l/write
rm -rf $O/prof_l
head -30 $O/large_v3_pmc.md | cut -c1-170

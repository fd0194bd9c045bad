#!/bin/bash
# kernel trace of one single-stream bench run (timeline analysis: gaps between kernels)
set -u
mkdir -p gpurun_out/trace1
export WLK_SYNTHETIC_VOCAB=1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/trace1 -o t1 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams > $GRAFT_REPO_ROOT/gpurun_out/trace1/bench.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/trace1/bench.log | cut -c1-300
find $GRAFT_REPO_ROOT/gpurun_out/trace1 -name "*.db" | head

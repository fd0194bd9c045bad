#!/bin/bash
# round 6, job s: kernel stats of one Sortformer session after the k-wave changes
set -u
O=gpurun_out/r06s; mkdir -p $O; R=$PWD
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/sfs
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/sfs -o st -- python $R/scripts/diar_probe.py 30 > $R/$O/diar_probe.log 2>&1
cd $R
python scripts/export_profile.py $(find /tmp/sfs -name "*.db" | head -1) $O/diar_kernel_stats.md "python scripts/diar_probe.py 30 (streaming Sortformer, ONE session, 2 x 30 chunks of 1 s): rocprofv3 --kernel-trace --stats" > /dev/null
head -30 $O/diar_kernel_stats.md

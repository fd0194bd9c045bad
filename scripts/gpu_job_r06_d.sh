O=gpurun_out/r06d; mkdir -p $O; R=$PWD
export TMPDIR=/tmp; cd /tmp
WLK_SF_WORKSPACES=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof/diar8 -o st -- python $R/scripts/diar_probe8.py 8 30 > $R/$O/diar8_probe.log 2>&1
cd $R
grep rep $O/diar8_probe.log
python scripts/export_profile.py $(find $O/prof/diar8 -name "*.db" | head -1) $O/diar8_kernel_stats.md "WLK_SF_WORKSPACES=1 python scripts/diar_probe8.py 8 30 (8 diarizer sessions flat out on one model, stacked steps, ~4 sessions per chain): rocprofv3 --kernel-trace --stats" > /dev/null
rm -rf $O/prof
head -34 $O/diar8_kernel_stats.md | tail -28 | cut -c1-170

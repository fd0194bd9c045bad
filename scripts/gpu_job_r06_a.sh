# Round-6 job A: the driver's bench command on the new (compact) line, Sortformer kernel profile (baseline before the
# round's work), large-v3 seed scan for a word-committing 30 s stream.
O=gpurun_out/r06a; mkdir -p $O; R=$PWD
S=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.log; echo "bench rc=$? $(( $(date +%s) - S )) s; line bytes $(tail -1 $O/bench_driver_cmd.json | wc -c)"
cp bench_full.json $O/bench_full.json 2>/dev/null
tail -3 $O/bench_driver_cmd.log | cut -c1-600
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof/diar -o st -- python $R/scripts/diar_probe.py 30 > $R/$O/diar_probe.log 2>&1
cd $R
python scripts/export_profile.py $(find $O/prof/diar -name "*.db" | head -1) $O/diar_kernel_stats.md "python scripts/diar_probe.py 30 (streaming Sortformer alone, 2 x 30 chunks of 1 s): rocprofv3 --kernel-trace --stats"
rm -rf $O/prof
cat $O/diar_probe.log | tail -3
head -30 $O/diar_kernel_stats.md | cut -c1-160
S=$(date +%s); timeout 600 python scripts/lv3_seed_scan.py 30 12 > $O/lv3_seed_scan.txt 2>&1; echo "scan rc=$? $(( $(date +%s) - S )) s"; tail -14 $O/lv3_seed_scan.txt

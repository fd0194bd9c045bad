# Round-6 evidence run (gpurun): smoke, full GPU suite, the DEFAULT bench line as the driver runs it, rocprofv3 kernel stats +
# the three PMC passes over the base.en command and over the large-v3 command (30 s, seed 8), kernel stats of the Sortformer
# alone (one session) and of eight stacked sessions, one PMC pass over the solo Sortformer, the 8-stream kernel trace.
# Outputs under gpurun_out/r06 (copied to profiles/ by hand).  GRAFT_GIT_HEAD is exported by the caller.
O=gpurun_out/r06; mkdir -p $O; R=$PWD
S=$(date +%s); timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$? $(( $(date +%s) - S )) s: $(tail -1 $O/smoke.log)"
S=$(date +%s); timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 > $O/pytest_gpu.log; echo "pytest gpu $(( $(date +%s) - S )) s: $(tail -1 $O/pytest_gpu.log)"
cp gpurun_out/parity_report.json gpurun_out/dropin_gpu_report.txt $O/ 2>/dev/null
S=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.log; echo "driver-command bench rc=$? $(( $(date +%s) - S )) s; line bytes $(tail -1 $O/bench_driver_cmd.json | wc -c)"
cp bench_full.json $O/bench_full.json
export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams --no-large-v3 --full-out /tmp/prof_full.json"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof/stats -o st -- $B > $R/$O/prof_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/$O/prof/mfma -o p -- $B > $R/$O/pmc_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof/fetch -o p -- $B > $R/$O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof/write -o p -- $B > $R/$O/pmc_write.log 2>&1
cd $R
python scripts/export_profile.py $(find $O/prof/stats -name "*.db" | head -1) $O/bench_kernel_stats.md "python bench.py --steps 1 --warmup 1 --no-eight-streams --no-large-v3 (base.en, 1 stream): rocprofv3 --kernel-trace --stats" > /dev/null
python scripts/export_pmc.py $O/pmc_bench.md $O/pmc_bench.json $O/prof/mfma $O/prof/fetch $O/prof/write > /dev/null
rm -rf $O/prof
cd /tmp
B="python $R/bench.py --model large-v3 --seconds 30 --seed 8 --steps 1 --warmup 1 --no-cpu-baseline --no-diarization --no-large-v3 --full-out /tmp/prof_full_l.json"
timeout 500 rocprofv3 --kernel-trace --stats -d $R/$O/prof_l/stats -o st -- $B > $R/$O/prof_large.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/$O/prof_l/mfma -o p -- $B > $R/$O/pmc_l_mfma.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_l/fetch -o p -- $B > $R/$O/pmc_l_fetch.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_l/write -o p -- $B > $R/$O/pmc_l_write.log 2>&1
cd $R
python scripts/export_profile.py $(find $O/prof_l/stats -name "*.db" | head -1) $O/large_v3_kernel_stats.md "python bench.py --model large-v3 --seconds 30 --seed 8 --steps 1 --warmup 1 (large-v3, 1 stream, 30 s): rocprofv3 --kernel-trace --stats" > /dev/null
python scripts/export_pmc.py $O/large_v3_pmc.md $O/large_v3_pmc.json $O/prof_l/mfma $O/prof_l/fetch $O/prof_l/write > /dev/null
rm -rf $O/prof_l
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_d/solo -o st -- python $R/scripts/diar_probe.py 30 > $R/$O/diar_probe.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_d/eight -o st -- python $R/scripts/diar_probe8.py 8 30 > $R/$O/diar_probe8.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/$O/prof_d/mfma -o p -- python $R/scripts/diar_probe.py 30 > $R/$O/diar_pmc_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_d/fetch -o p -- python $R/scripts/diar_probe.py 30 > $R/$O/diar_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_d/write -o p -- python $R/scripts/diar_probe.py 30 > $R/$O/diar_pmc_write.log 2>&1
cd $R
python scripts/export_profile.py $(find $O/prof_d/solo -name "*.db" | head -1) $O/diar_kernel_stats.md "python scripts/diar_probe.py 30 (streaming Sortformer, ONE session, 2 x 30 chunks of 1 s): rocprofv3 --kernel-trace --stats" > /dev/null
python scripts/export_profile.py $(find $O/prof_d/eight -name "*.db" | head -1) $O/diar8_kernel_stats.md "python scripts/diar_probe8.py 8 30 (EIGHT diarizer sessions flat out on one model, stacked steps): rocprofv3 --kernel-trace --stats" > /dev/null
python scripts/export_pmc.py $O/diar_pmc.md $O/diar_pmc.json $O/prof_d/mfma $O/prof_d/fetch $O/prof_d/write > /dev/null
grep rep $O/diar_probe.log $O/diar_probe8.log | cut -c1-400
rm -rf $O/prof_d
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/$O/trace8 -o t -- python $R/scripts/eight_stream_probe.py 8 > $R/$O/trace8.log 2>&1
cd $R
( grep "^pass\|^{" $O/trace8.log; python scripts/trace_busy.py $(find $O/trace8 -name "*.db" | head -1) 900 ) > $O/trace8_busy.txt
rm -rf $O/trace8
( echo "# scripts/diar_probe8.py <diarizer sessions> 30 <ASR streams>: lanes (WLK_SF_WORKSPACES), stacking (WLK_SF_BATCH), front end inside the step or as three calls"
  for v in "A=0" "WLK_SF_WORKSPACES=2" "WLK_SF_BATCH=1 WLK_SF_WORKSPACES=4" "PROBE_THREE_CALLS=1"; do
    echo "== $v: 8 diarizer sessions alone"; env $v python scripts/diar_probe8.py 8 30 2>&1 | grep "rep 1"
    echo "== $v: 8 diarizer sessions beside 8 base.en ASR streams (config 4)"; env $v python scripts/diar_probe8.py 8 30 8 2>&1 | grep "rep 1"
  done ) > $O/diar_lanes.txt; cut -c1-420 $O/diar_lanes.txt
python scripts/kp_tile_probe.py > $O/kp_tile_probe.txt 2>&1
X3_PROBE_NARROW=1 python scripts/x3_probe.py 2>&1 | grep -v amdgpu.ids > $O/x3_narrow_probe.txt
python scripts/x3_probe.py 2>&1 | grep -v amdgpu.ids > $O/x3_probe.txt
python - <<PY
import json
d=json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1]); e=d.get("eight_streams") or {}; l=d.get("large_v3") or {}
print("value", d["value"], "rtf", d["rtf"], "p50 call", d["p50_call_ms"], "roof", d["roofline"]["frac"], d["roofline"].get("frac_at_rocprof_duration"), "| eight", e.get("audio_s_per_s"), "| cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("kind"))
print("cfg4", d.get("asr_plus_diarization_8_sessions")); print("diar", d.get("diarization")); print("pipeline", d.get("pipeline"))
print("large_v3", {k: l.get(k) for k in ("audio_s_per_s","decisions","identical","parity_ok","p50_committed_token_latency_ms","committed_tokens","leg_wall_s","error")}, l.get("roofline"), l.get("cpu_baseline"))
print("parity", d["parity_ok"], d["parity_checked"])
PY
head -16 $O/bench_kernel_stats.md | cut -c1-150; head -24 $O/diar_kernel_stats.md | tail -14 | cut -c1-150; tail -12 $O/trace8_busy.txt | cut -c1-200

# Round-2 evidence run (gpurun): full GPU suite, drop-in over the staged reference (when present), default bench with the
# reference's CPU path beside it, 2-run A/B of the batch engine on the 8-stream workload, rocprofv3 kernel stats and the
# three PMC passes over the bench command, large-v3 line.  Outputs under gpurun_out/r02 (copied to profiles/ by hand).
O=gpurun_out/r02; mkdir -p $O; R=$PWD
S=$(date +%s); timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/pytest_gpu.log; echo "pytest gpu $(( $(date +%s) - S )) s: $(tail -1 $O/pytest_gpu.log)"
if [ -d "$R/.ref_stage/whisperlivekit" ]; then
  export WLK_REFERENCE_ROOT=$R/.ref_stage
  rm -f gpurun_out/dropin_gpu_report.txt
  timeout 600 python -m pytest tests/test_gpu_reference_dropin.py -q -m gpu 2>&1 | tail -3 > $O/dropin_gpu.log; tail -1 $O/dropin_gpu.log
  cp gpurun_out/dropin_gpu_report.txt $O/ 2>/dev/null
fi
S=$(date +%s); timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "default bench rc=$? $(( $(date +%s) - S )) s"
unset WLK_REFERENCE_ROOT
bash scripts/gpu_job_ab8.sh "WLK_X=1" "WLK_BATCH_ENCODE=0 WLK_BATCH_DECODE=0" "WLK_X=2" "WLK_BATCH_ENCODE=0 WLK_BATCH_DECODE=0" 2>&1 | grep "^\[" | cut -c1-330 > $O/ab_batch_engine_8streams.txt; cat $O/ab_batch_engine_8streams.txt | grep pass
export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof/stats -o st -- $B > $R/$O/prof_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/$O/prof/mfma -o p -- $B > $R/$O/pmc_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof/fetch -o p -- $B > $R/$O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof/write -o p -- $B > $R/$O/pmc_write.log 2>&1
cd $R
python scripts/export_profile.py $(find $O/prof/stats -name "*.db" | head -1) $O/bench_kernel_stats.md "python bench.py --steps 1 --warmup 1 --no-eight-streams (base.en, 1 stream): rocprofv3 --kernel-trace --stats" 
python scripts/export_pmc.py $O/pmc_bench.md $O/pmc_bench.json $O/prof/mfma $O/prof/fetch $O/prof/write
rm -rf $O/prof
timeout 500 python bench.py --model large-v3 --seconds 10 --steps 1 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams > $O/bench_large_v3.json 2> $O/bench_large_v3.log; echo "large-v3 rc=$?"
python - <<PY
import json
for n in ("default","large_v3"):
    try:
        d=json.load(open(f"$O/bench_{n}.json")); e=d.get("eight_streams") or {}
        print(n, "value", d["value"], "rtf", d["rtf"], "p50 call", d["p50_call_ms"], "p50 lat", d["p50_committed_token_latency_ms"], "roof", d["roofline"]["frac"], d["roofline"].get("frac_at_rocprof_duration"), "| eight", e.get("audio_s_per_s"), e.get("p50_committed_token_latency_ms"), e.get("p95_committed_token_latency_ms"), "| cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("kind"))
        print("   parity", d["parity_checked"] and {k: d["parity_checked"][k] for k in ("sessions","decisions","identical","tie_divergences","mismatches","words_identical_sessions")})
    except Exception as e: print(n, "failed", e)
PY

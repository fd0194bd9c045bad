# Round-5 evidence run (gpurun): full GPU suite, the DEFAULT bench line as the driver runs it (config-3 leg inside), rocprofv3
# kernel stats + the three PMC passes over the base.en command, kernel stats + FETCH / WRITE passes over the large-v3 command.
# Outputs under gpurun_out/r05 (copied to profiles/ by hand).  GRAFT_GIT_HEAD is exported by the caller.
O=gpurun_out/r05; mkdir -p $O; R=$PWD
S=$(date +%s); timeout 1300 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/pytest_gpu.log; echo "pytest gpu $(( $(date +%s) - S )) s: $(tail -1 $O/pytest_gpu.log)"
cp gpurun_out/parity_report.json $O/ 2>/dev/null
S=$(date +%s); timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "default bench rc=$? $(( $(date +%s) - S )) s"
export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams --no-large-v3"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof/stats -o st -- $B > $R/$O/prof_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/$O/prof/mfma -o p -- $B > $R/$O/pmc_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof/fetch -o p -- $B > $R/$O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof/write -o p -- $B > $R/$O/pmc_write.log 2>&1
cd $R
python scripts/export_profile.py $(find $O/prof/stats -name "*.db" | head -1) $O/bench_kernel_stats.md "python bench.py --steps 1 --warmup 1 --no-eight-streams --no-large-v3 (base.en, 1 stream): rocprofv3 --kernel-trace --stats"
python scripts/export_pmc.py $O/pmc_bench.md $O/pmc_bench.json $O/prof/mfma $O/prof/fetch $O/prof/write
rm -rf $O/prof
export WLK_SYNTHETIC_VOCAB=1
cd /tmp
B="python $R/bench.py --model large-v3 --seconds 10 --steps 1 --warmup 1 --no-cpu-baseline --no-diarization --no-large-v3"
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_l/stats -o st -- $B > $R/$O/prof_large.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/$O/prof_l/mfma -o p -- $B > $R/$O/pmc_l_mfma.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_l/fetch -o p -- $B > $R/$O/pmc_l_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_l/write -o p -- $B > $R/$O/pmc_l_write.log 2>&1
cd $R
python scripts/export_profile.py $(find $O/prof_l/stats -name "*.db" | head -1) $O/large_v3_kernel_stats.md "python bench.py --model large-v3 --seconds 10 --steps 1 --warmup 1 (large-v3, 1 stream, 10 s): rocprofv3 --kernel-trace --stats"
python scripts/export_pmc.py $O/large_v3_pmc.md $O/large_v3_pmc.json $O/prof_l/mfma $O/prof_l/fetch $O/prof_l/write
rm -rf $O/prof_l
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1]); e=d.get("eight_streams") or {}; l=d.get("large_v3") or {}
print("value", d["value"], "rtf", d["rtf"], "p50 call", d["p50_call_ms"], "roof", d["roofline"]["frac"], d["roofline"].get("frac_at_rocprof_duration"), "| eight", e.get("audio_s_per_s"), "| cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("kind"))
print("large_v3", {k: l.get(k) for k in ("audio_s_per_s","decisions","identical","parity_ok","leg_wall_s","error")}, (l.get("cpu_baseline") or {}).get("kind"), (l.get("cpu_baseline") or {}).get("value"))
p=d["parity_checked"]; print("parity", d["parity_ok"], p and {k: p.get(k) for k in ("sessions","decisions","identical","tie_divergences","unchecked_calls","mismatches","words_identical_sessions")})
PY
head -14 $O/bench_kernel_stats.md | cut -c1-150

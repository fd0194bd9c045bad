# Round-4 evidence run (gpurun): full GPU suite, default bench (reference CPU path validated on the box's host cores, config-4
# and word-alignment legs), rocprofv3 kernel stats + the three PMC passes over the bench command, large-v3 line (parity
# checked against its own golden now), 8-stream kernel trace.  Outputs under gpurun_out/r04 (copied to profiles/ by hand).
# GRAFT_GIT_HEAD is exported by the caller: the box has no .git, the exports stamp it into the JSON files.
O=gpurun_out/r04; mkdir -p $O; R=$PWD
S=$(date +%s); timeout 1300 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/pytest_gpu.log; echo "pytest gpu $(( $(date +%s) - S )) s: $(tail -1 $O/pytest_gpu.log)"
cp gpurun_out/parity_report.json $O/ 2>/dev/null
S=$(date +%s); timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "default bench rc=$? $(( $(date +%s) - S )) s"
# same-box calibration: boxes of this pool differ by up to 10 % (host, fabric, memory) - the round-3 library (kept beside the
# tree, git-ignored) and the tree's library alternate on THIS box, short bench lines
if [ -f whisperlivekit_amd/libwlk_hip_r3.so ]; then
  BS="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams"
  : > $O/same_box_ab.txt
  for i in 1 2; do
    echo -n "tree   " >> $O/same_box_ab.txt; timeout 300 $BS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'audio-s/s', d['ms_per_step'], 'ms per 30 s stream, parity_ok', d['parity_ok'])" >> $O/same_box_ab.txt
    echo -n "round3 " >> $O/same_box_ab.txt; WLK_HIP_LIB=$PWD/whisperlivekit_amd/libwlk_hip_r3.so timeout 300 $BS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'audio-s/s', d['ms_per_step'], 'ms per 30 s stream, parity_ok', d['parity_ok'])" >> $O/same_box_ab.txt
  done
  cat $O/same_box_ab.txt
fi
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
timeout 500 python bench.py --model large-v3 --seconds 10 --steps 1 --warmup 1 --no-cpu-baseline --no-diarization > $O/bench_large_v3.json 2> $O/bench_large_v3.log; echo "large-v3 rc=$?"
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/$O/trace8 -o t -- python $R/scripts/eight_stream_probe.py 8 > $R/$O/trace8.log 2>&1
cd $R
( grep "^pass\|^{" $O/trace8.log; python scripts/trace_busy.py $(find $O/trace8 -name "*.db" | head -1) 900 ) > $O/trace8_busy.txt; rm -rf $O/trace8
python - <<PY
import json
for n in ("default","large_v3"):
    try:
        d=json.load(open(f"$O/bench_{n}.json")); e=d.get("eight_streams") or {}
        print(n, "value", d["value"], "rtf", d["rtf"], "p50 call", d["p50_call_ms"], "p50 lat", d["p50_committed_token_latency_ms"], "roof", d["roofline"]["frac"], d["roofline"].get("frac_at_rocprof_duration"), "| eight", e.get("audio_s_per_s"), "| cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("kind"), (d.get("cpu_baseline") or {}).get("validated"))
        p=d["parity_checked"]; print("   parity", d["parity_ok"], p and {k: p[k] for k in ("sessions","decisions","identical","tie_divergences","mismatches","words_identical_sessions")})
    except Exception as e: print(n, "failed", e)
PY
cat $O/trace8_busy.txt | head -12

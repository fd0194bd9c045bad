# Round-6 job B: the Sortformer rework (MFMA attention, kp GEMM family, stacked steps) - parity tests, solo timing A/B,
# kernel profile, config-4 bench with and without stacking.
O=gpurun_out/r06b; mkdir -p $O; R=$PWD
S=$(date +%s); timeout 1200 python -m pytest tests/test_gpu_sortformer.py tests/test_nllb.py tests/test_translation.py -q -m gpu -x 2>&1 | tail -15 > $O/pytest_sf.log; echo "pytest $(( $(date +%s) - S )) s: $(tail -1 $O/pytest_sf.log)"; grep -E "FAILED|Error|assert" $O/pytest_sf.log | head -20
python scripts/diar_probe.py 30 2>&1 | grep rep | sed 's/^/mfma attention: /'
WLK_SF_ATTN=valu python scripts/diar_probe.py 30 2>&1 | grep rep | sed 's/^/valu attention: /'
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof/diar -o st -- python $R/scripts/diar_probe.py 30 > $R/$O/diar_probe.log 2>&1
cd $R
python scripts/export_profile.py $(find $O/prof/diar -name "*.db" | head -1) $O/diar_kernel_stats.md "python scripts/diar_probe.py 30 (streaming Sortformer alone, 2 x 30 chunks of 1 s): rocprofv3 --kernel-trace --stats" > /dev/null
rm -rf $O/prof
head -26 $O/diar_kernel_stats.md | tail -20 | cut -c1-150
B="python bench.py --steps 3 --warmup 1 --no-large-v3 --no-cpu-baseline"
for rep in 1 2; do
  $B --full-out $O/bench_stack8_$rep.json > /dev/null 2> $O/bench_stack8_$rep.log; echo "stack8 rc=$?"
  WLK_SF_BATCH=1 WLK_SF_WORKSPACES=4 $B --full-out $O/bench_stack1_$rep.json > /dev/null 2> $O/bench_stack1_$rep.log; echo "stack1 rc=$?"
done
python - <<PY
import json
for n in ("stack8_1","stack1_1","stack8_2","stack1_2"):
    try:
        d=json.load(open("$O/bench_%s.json"%n))
    except Exception as e:
        print(n, "no record", e); continue
    c=d.get("asr_plus_diarization_8_sessions") or {}; di=d.get("diarization") or {}
    print(n, "value", d["value"], "eight", (d.get("eight_streams") or {}).get("audio_s_per_s"), "| cfg4 asr", c.get("asr_audio_s_per_s"), "diar", c.get("diar_audio_s_per_s"), "p50", c.get("diar_p50_chunk_ms"), "p95", c.get("diar_p95_chunk_ms"), "alone", c.get("diar_alone_p50_chunk_ms"), "sess/step", c.get("diar_mean_sessions_per_step"), "| diar leg p50", di.get("p50_chunk_ms"), "err", di.get("max_abs_err_vs_oracle_last_chunk"), (di.get("roofline") or {}).get("frac"), "parity", d.get("parity_ok"), c.get("error"), di.get("error"))
PY

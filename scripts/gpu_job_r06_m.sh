#!/bin/bash
# round 6, job m: the early z-score (side workgroups of the vocabulary projection) - parity tests, alternating bench runs, and the
# kernel timeline of one call either way
set -u
O=gpurun_out/r06m; mkdir -p $O; R=$PWD
export WLK_SYNTHETIC_VOCAB=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "early_zscore or library_decode_loop or smoke" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
bash scripts/gpu_job_ab_env.sh "WLK_EARLY_Z=0" > $O/ab.txt 2>&1
python - <<'PY'
import json
lab=None
for l in open('gpurun_out/ab_env.log'):
    l=l.strip()
    if l in('env','base'): lab=l; continue
    try:
        j=json.loads(l); pc=j.get('parity_checked') or {}
        print('early z off' if lab=='env' else 'early z on ', j['value'], 'step us', j['roofline']['step']['us'], pc.get('decisions'), pc.get('identical'))
    except Exception as e: print(lab,'ERR',l[:200])
PY
cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-diarization --no-eight-streams --no-large-v3 --full-out /tmp/prof_full.json"
for mode in 1 0; do
  rm -rf /tmp/tr$mode
  WLK_EARLY_Z=$mode timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr$mode -o st -- $B > $R/$O/trace$mode.log 2>&1
  DB=$(find /tmp/tr$mode -name "*.db" | head -1)
  python $R/scripts/trace_gaps.py $DB 100 --all > $R/$O/call100_early$mode.txt 2>&1
  python $R/scripts/export_profile.py $DB $R/$O/stats_early$mode.md "early z = $mode" > /dev/null
done
cd $R
grep "gemv_f32_kernel<1, 4>\|select_stage" $O/stats_early*.md

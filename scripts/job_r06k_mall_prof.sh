#!/bin/bash
# kernel durations of the large-v3 decode step with and without the Infinity-Cache prefetcher beside it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06k
for pf in 0 1; do
  WLK_MALL_PREFETCH=$pf WLK_MALL_LEAD=${LEAD:-0} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06k/prof_pf$pf -o st -- python bench.py --model large-v3 --seconds 10 --seed 8 --steps 1 --warmup 0 --no-cpu-baseline --no-eight-streams --no-diarization --no-large-v3 --no-parity --full-out gpurun_out/r06k/full_prof.json > gpurun_out/r06k/prof_pf$pf.log 2>&1
  f=$(find gpurun_out/r06k/prof_pf$pf -name "*kernel_stats.csv" | head -1)
  echo "== prefetch $pf ($f)"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>7s} avg_us {float(r['AverageNs'])/1e3:8.2f} pct {r['Percentage']}")
PY
done 2>&1 | tee gpurun_out/r06k/mall_prof.txt
find gpurun_out/r06k -name "*.db" -delete; find gpurun_out/r06k -name "*kernel_trace.csv" -delete

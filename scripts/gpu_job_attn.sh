mkdir -p gpurun_out/attn
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "attention" 2>&1 | tail -2
for v in "lds 0" "pw 1" "pw 0" "pw 3"; do set -- $v; WLK_ENC_ATTN=$1 WLK_ENC_KSPLIT=$2 timeout 200 python scripts/attn_time_probe.py 30 2>&1 | grep -v amdgpu.ids | grep "T=\|vs lds" ; done | tee gpurun_out/attn/probe2.txt
for v in "pw 0" "pw 1" "lds 0" "pw 0" "pw 1"; do set -- $v; WLK_ENC_ATTN=$1 WLK_ENC_KSPLIT=$2 timeout 300 python bench.py --no-cpu-baseline --no-diarization --no-eight-streams --steps 5 2> gpurun_out/attn/bench_$1_$2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['launch_tags']['enc_attention']; p=d['parity_checked']
print('$1 ksplit=$2', 'value', d['value'], 'enc_attention', round(1e3*t['ms']/t['launches'],2), 'us', t['tflops'], 'TF', 'parity', d['parity_ok'], p['identical'], p['decisions'], p['tie_divergences'], p['mismatches'])"; done | tee gpurun_out/attn/bench_ab2.txt

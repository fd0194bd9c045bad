# Round 5, seventh GPU call: weight loads between the MFMAs (one per six) against six in front (WLK_X3_ABL=6), column-major walk
O=gpurun_out/r05g; mkdir -p $O
( for v in "A=0" "WLK_X3_ABL=6" "WLK_X3_ABL=4" "WLK_X3_COLMAJOR=1" "WLK_X3_ABL=3"; do echo "== $v"; env $v timeout 200 python scripts/x3_probe.py 2>&1 | grep -v "attention\|amdgpu.ids"; done ) > $O/x3_probe.txt
cut -c1-100 $O/x3_probe.txt
S=$(date +%s); WLK_X3_COLMAJOR=1 timeout 600 python -m pytest tests/test_gpu_x3.py -q -m gpu -x 2>&1 | tail -5 > $O/pytest.log; echo "pytest x3 (colmajor) $(( $(date +%s) - S )) s: $(tail -1 $O/pytest.log)"

# Round 5, sixth GPU call: where do the weight loads of the direct-weight X3 kernel lose their time?  Probe variants:
# column-major walk inside a band (WLK_X3_COLMAJOR), L2 prefetch of the weight stream by the loader waves (WLK_X3_WPF),
# weight loads from one cached address (WLK_X3_ABL=5).
O=gpurun_out/r05f; mkdir -p $O
( for v in "A=0" "WLK_X3_COLMAJOR=1" "WLK_X3_WPF=1" "WLK_X3_COLMAJOR=1 WLK_X3_WPF=1" "WLK_X3_ABL=5" "WLK_X3_ABL=5 WLK_X3_COLMAJOR=1" "WLK_X3_ABL=4 WLK_X3_COLMAJOR=1"; do echo "== $v"; env $v timeout 200 python scripts/x3_probe.py 2>&1 | grep -v "attention\|amdgpu.ids"; done ) > $O/x3_probe.txt
cut -c1-100 $O/x3_probe.txt
S=$(date +%s); WLK_X3_COLMAJOR=1 WLK_X3_WPF=1 timeout 600 python -m pytest tests/test_gpu_x3.py -q -m gpu -x 2>&1 | tail -5 > $O/pytest.log; echo "pytest (colmajor + wpf) $(( $(date +%s) - S )) s: $(tail -1 $O/pytest.log)"

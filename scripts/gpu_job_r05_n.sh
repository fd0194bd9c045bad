# Round 5: issue priority of the X3 kernel's compute (ABL 7) / loader (ABL 8) waves - correct results, timing only
O=gpurun_out/r05n; mkdir -p $O
( for v in "A=0" "WLK_X3_ABL=7" "WLK_X3_ABL=8" "A=1" "WLK_X3_ABL=7"; do echo "== $v"; env $v timeout 200 python scripts/x3_probe.py 2>&1 | grep -v "attention\|amdgpu.ids"; done ) > $O/x3_probe.txt
cut -c1-100 $O/x3_probe.txt

# Round 5, final check at HEAD: __graft_entry__.smoke(), the full GPU suite
O=gpurun_out/r05final; mkdir -p $O
S=$(date +%s); timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$? $(( $(date +%s) - S )) s: $(tail -1 $O/smoke.log)"
S=$(date +%s); timeout 1300 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/pytest_gpu.log; echo "pytest gpu $(( $(date +%s) - S )) s: $(tail -1 $O/pytest_gpu.log)"

export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/pmcg; mkdir -p $O; cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/p1 -o p -- python $R/scripts/gemm_pmc_probe.py > $O/p1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_MISC --output-format csv -d $O/p2 -o p -- python $R/scripts/gemm_pmc_probe.py > $O/p2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmcg/p1","gpurun_out/pmcg/p2"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            k=row["Kernel_Name"].replace("void wlk::","").split("(")[0][:40]
            agg[(k, row.get("Grid_Size","?"))][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,c in agg.items():
        print(k, {n: round(sum(v)/len(v)) for n,v in c.items()})
PY
tail -3 $O/p1.log

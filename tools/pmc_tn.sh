cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 100 rocprofv3 --pmc $pass -d $R/gpurun_out/pmc_tn_$tag -o c --output-format csv -- python3 $R/tools/prof_tn.py > $R/gpurun_out/pmc_tn_$tag.log 2>&1 || { tail -3 $R/gpurun_out/pmc_tn_$tag.log; exit 1; }
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("$R/gpurun_out/pmc_tn_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = "tn" if "gemm_tn" in r["Kernel_Name"] else "nt" if "pipe64" in r["Kernel_Name"] else None
        if k: agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in agg:
    for c, v in agg[k].items(): print(k, c, "%.4g" % v)
PY
done

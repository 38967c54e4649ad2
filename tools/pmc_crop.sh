cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 100 python tools/bench_crop.py 2>&1 | grep -v amdgpu.ids
cd /tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 100 rocprofv3 --pmc $pass -d $R/gpurun_out/pmc_crop_$tag -o c --output-format csv -- python3 $R/tools/prof_crop.py 700 450 > $R/gpurun_out/pmc_crop_$tag.log 2>&1 || { tail -3 $R/gpurun_out/pmc_crop_$tag.log; exit 1; }
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float)
for fn in glob.glob("$R/gpurun_out/pmc_crop_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "crop_" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items(): print(k, "%.4g" % v)
PY
done

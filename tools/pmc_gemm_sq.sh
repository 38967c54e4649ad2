# SQ counters and the clock of the tower GEMM shapes (one launch series per shape, tools/prof_gemm.py at M rows):
# matrix-pipe busy cycles against the cycles the chip ran, i.e. whether a shape is bound by the clock the chip holds
# under load or by instruction issue.  Run on the GPU box:  bash tools/pmc_gemm_sq.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 100 rocprofv3 --pmc $pass -d $R/gpurun_out/pmc_gemm_$tag -o c --output-format csv -- python3 $R/tools/prof_gemm.py > $R/gpurun_out/pmc_gemm_$tag.log 2>&1 || { tail -3 $R/gpurun_out/pmc_gemm_$tag.log; exit 1; }
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in glob.glob("$R/gpurun_out/pmc_gemm_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "gemm" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
dur = collections.defaultdict(dict)
for fn in glob.glob("$R/gpurun_out/pmc_gemm_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "gemm" not in r["Kernel_Name"]: continue
        dur[(r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"])][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
dur = {k: list(v.values()) for k, v in dur.items()}
for k, v in agg.items():
    d = sorted(dur.get(k, [0]))[len(dur.get(k, [0])) // 2]
    print(k, "median ns", d, {c: "%.4g" % (x / n[(k, c)]) for c, x in v.items()})
PY
done

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
timeout -s KILL 100 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d $R/gpurun_out/pmc_tn_tcc_$tag -o c --output-format csv -- python3 $R/tools/prof_tn.py > $R/gpurun_out/pmc_tn_tcc_$tag.log 2>&1 || { tail -3 $R/gpurun_out/pmc_tn_tcc_$tag.log; exit 1; }
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for fn in glob.glob("$R/gpurun_out/pmc_tn_tcc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = "tn" if "gemm_tn" in r["Kernel_Name"] else "nt" if "pipe64" in r["Kernel_Name"] else None
        if k: agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in agg:
    for c, v in agg[k].items(): print("$tag", k, c, "%.4g per launch" % (v / 3))
PY

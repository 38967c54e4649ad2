cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export M=262144 ONLY=qkv,fc1,fc2
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $R/gpurun_out/pmc_l2 -o l2 --output-format csv -- python3 $R/tools/bench_gemm.py > $R/gpurun_out/pmc_l2.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmc_l2/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    if h + m > 0: print(k, "hit rate %.3f" % (h / (h + m)), "req %.3g" % v.get("TCC_REQ_sum", 0), "hit %.3g miss %.3g" % (h, m))
PY

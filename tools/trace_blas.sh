R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_blas -o blas --output-format csv -- python3 tools/calib_blas.py > gpurun_out/prof_blas.log 2>&1 || exit 1
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_blas/blas_kernel_stats.csv")))
for r in rows[:14]:
    print(r["Name"][:400], r["Calls"], r["AverageNs"])
PY
python3 - <<'PY'
import csv
seen = {}
for r in csv.DictReader(open("gpurun_out/prof_blas/blas_kernel_trace.csv")):
    n = r["Kernel_Name"]
    if "Cijk" in n and n not in seen:
        seen[n] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size_X"), r.get("Grid_Size_X"), r.get("Scratch_Size"))
for k, v in seen.items():
    print(v, k[:300])
PY

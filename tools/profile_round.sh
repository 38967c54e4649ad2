# Kernel-trace stats and the two PMC traffic passes over the bench command; run on the GPU box from the repo root.
# BENCH_STREAMS=0 in the environment gives the one-stream variant (kernel durations exclusive of overlap).
TAG=${1:-r01_g}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o $TAG --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1 || exit 1
if [ "$2" = "pmc" ]; then
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_rd_$TAG -o rd --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_rd_$TAG.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_wr_$TAG -o wr --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_wr_$TAG.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $(find gpurun_out/pmc_rd_$TAG -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_wr_$TAG -name "*counter_collection.csv" | head -1) gpurun_out/pmc_traffic_$TAG.json
fi

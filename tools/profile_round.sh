# Kernel-trace stats and the two PMC traffic passes over the bench command; run on the GPU box from the repo root.
#   bash tools/profile_round.sh <tag> [pmc]
# Writes gpurun_out/prof_<tag>/ (default: towers on side streams) and gpurun_out/prof_<tag>_one/ (BENCH_STREAMS=0:
# everything on one stream, kernel durations exclusive of overlap), and with `pmc` gpurun_out/pmc_traffic_<tag>.json.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
BENCH="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-c4 --no-train --no-forward-api"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o $TAG --output-format csv -- python3 $BENCH > gpurun_out/prof_$TAG.log 2>&1 || exit 1
export BENCH_STREAMS=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG}_one -o $TAG --output-format csv -- python3 $BENCH > gpurun_out/prof_${TAG}_one.log 2>&1 || exit 1
if [ "$2" = "pmc" ]; then
BENCH1="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c4 --no-train --no-forward-api"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_rd_$TAG -o rd --output-format csv -- python3 $BENCH1 > gpurun_out/pmc_rd_$TAG.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_wr_$TAG -o wr --output-format csv -- python3 $BENCH1 > gpurun_out/pmc_wr_$TAG.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $(find gpurun_out/pmc_rd_$TAG -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_wr_$TAG -name "*counter_collection.csv" | head -1) gpurun_out/pmc_traffic_$TAG.json
fi

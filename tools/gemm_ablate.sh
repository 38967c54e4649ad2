for b in 1 2 3 4; do BENCH_PAIR_SPLITS=$b timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/bench_s$b.json 2> gpurun_out/bench_s$b.err || { tail -5 gpurun_out/bench_s$b.err; exit 1; }; python -c "
import json;d=json.loads(open('gpurun_out/bench_s$b.json').read().strip().splitlines()[-1]);print($b, d['ms_per_step'], d['value'])"; done

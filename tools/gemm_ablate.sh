for b in 0 1; do BENCH_STREAMS=$b timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/bench_s$b.json 2> gpurun_out/bench_s$b.err || { tail -5 gpurun_out/bench_s$b.err; exit 1; }; python -c "
import json;d=json.loads(open('gpurun_out/bench_s$b.json').read().strip().splitlines()[-1]);print($b, d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['roofline']['achieved'])"; done
timeout -k 10 300 python -u -m pytest tests/test_gpu_forward.py tests/test_gpu_kernels.py -x -q --timeout 180 --timeout-method thread -m gpu 2>&1 | tail -2

# A/B of the headline bench between two settings of one environment variable, interleaved in one GPU call:
#   bash tools/ab_env.sh SGCLIP_LN_FOLD 0 1
VAR=$1; shift
for i in 1 2 3; do
for val in "$@"; do
env $VAR=$val timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-c4 --no-train --no-forward-api 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$VAR=$val', round(d['ms_per_step'],2), round(d['value']), round(d['roofline']['achieved'],1), d['kernel_ms_per_step'])"
done; done

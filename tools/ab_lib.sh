# A/B of the headline bench between kernel libraries laser_amd/libsgclip_<name>.so (default: hip alt; tools/build_alt.sh
# builds the others), interleaved in one GPU call; run on the GPU box from the repo root:  bash tools/ab_lib.sh [names...]
NAMES=${@:-hip alt}
for i in 1 2 3; do
for name in $NAMES; do
lib=laser_amd/libsgclip_$name.so
SGCLIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-c4 --no-train --no-forward-api 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$lib', round(d['ms_per_step'],2), round(d['value']), round(d['roofline']['achieved'],1), d['kernel_ms_per_step'])"
done; done

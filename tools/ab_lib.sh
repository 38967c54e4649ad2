for i in 1 2 3; do
for lib in laser_amd/libsgclip_hip.so laser_amd/libsgclip_alt.so; do
SGCLIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$lib', d['ms_per_step'], d['value'], d['roofline']['achieved'])"
done; done

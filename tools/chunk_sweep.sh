for c in 16384 32768 65536 131072 262144; do
  SGCLIP_VIT_CHUNK_ROWS=$c timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/chunk_$c.json 2> gpurun_out/chunk_$c.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/chunk_$c.json").read().strip().splitlines()[-1])
print($c, round(d["ms_per_step"],1), d["kernel_ms_per_step"], round(d["roofline"]["achieved"]))
PY
done

"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over bench.py.

usage: python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
Bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB, and on gfx950
FETCH_SIZE counts half of the wide streaming reads."""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        tot[k] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    return {k: (tot[k] / len(cnt[k]), len(cnt[k])) for k in tot}


def main():
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python3 bench.py --steps 1 --warmup 1 "
                   "--no-cpu-baseline`; KB per launch averaged per kernel; corrected bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                   "(gfx950 FETCH_SIZE counts half of wide streaming reads); that command makes eight passes over the step "
                   "(warm-up, timed step, five single-clip latency steps, one-stream kernel pass)", "passes": 8, "kernels": {}}
    for k in f:
        if not k.startswith("sgclip::"):
            continue
        fk, n = f[k]
        wk = w.get(k, (0.0, 0))[0]
        out["kernels"][k] = {"launches": n, "fetch_kb": fk, "write_kb": wk,
                             "hbm_bytes_per_launch_corrected": (2 * fk + wk) * 1024}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print("wrote", sys.argv[3], len(out["kernels"]), "kernels")


if __name__ == "__main__":
    main()

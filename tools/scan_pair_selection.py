"""Seed scan of the training-mode pair selection against the CPU oracle (GPU box): per seed the oracle's smallest
third / fourth margin, whether the selected pair lists equal the oracle's with the bf16 ranking alone and with the
ranking guard, and what the guard did.  Usage: python tools/scan_pair_selection.py [first_seed] [count] > table"""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from selection_scan import make_pair, scan_seed  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 120
torch.set_num_threads(16)
model, oracle = make_pair("vit-b-32", 22)
rows = []
print("# ViT-B/32, weight seed 22, clip = tiny_batch(seed, 2 videos x 3 frames x 5 objects, 320x240, train=True)")
print("# seed  oracle_margin  bf16_ranking  guarded  objects_rescored  undecided")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for seed in range(first, first + count):
        r = scan_seed(model, oracle, seed)
        rows.append(r)
        print(f"{seed:5d}  {r['margin']:.3e}  {'same' if r['same_plain'] else 'FLIP'}  {'same' if r['same_guarded'] else 'FLIP'}  "
              f"{r['rescored']:3d}  {r['undecided']}", flush=True)
n = len(rows)
print(f"# flip rate, bf16 ranking alone: {sum(not r['same_plain'] for r in rows)}/{n}")
print(f"# flip rate, with the ranking guard: {sum(not r['same_guarded'] for r in rows)}/{n} "
      f"(seeds the guard reports undecided: {sum(bool(r['undecided']) for r in rows)})")

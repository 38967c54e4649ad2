"""Experiment: one image-tower call over N crops against two half-size calls on two HIP streams (tails and
kernel boundaries of one stream filled by the other)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from laser_amd import ops
from laser_amd.config import config_by_name
from laser_amd.weights import random_state_dict

cfg = config_by_name("vit-b-32")
dev = torch.device("cuda:0")
sd = random_state_dict(cfg, 1)
t1 = ops.Tower(cfg, "vision", sd, dev)
n = int(os.environ.get("N", 7680))
P, kpad = cfg.n_patches, 3072
patches = (torch.randn(n * P, kpad, device=dev) * 0.5).to(torch.bfloat16)
out = torch.empty(n, cfg.proj, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h = n // 2

def single():
    t1.encode_images(patches, n, out)

def dual():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        t1.encode_images(patches[: h * P], h, out[:h])
    with torch.cuda.stream(s2):
        t1.encode_images(patches[h * P:], n - h, out[h:], ws_slot=1)
    cur.wait_stream(s1); cur.wait_stream(s2)

for name, f in (("single", single), ("dual", dual), ("single", single), ("dual", dual)):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    print(name, round((time.perf_counter() - t0) / 5 * 1e3, 2), "ms", flush=True)

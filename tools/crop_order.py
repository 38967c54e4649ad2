"""Crop kernel: natural crop order against largest-first order (mixed box sizes as in the headline clip's pair crops)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_amd import _lib, ops
from laser_amd.predicate_model import CROP_DTYPE

dev = torch.device("cuda:0")
H, W, NF, NM, N = 720, 1280, 8, 64, 7680
rng = np.random.default_rng(0)
frames = torch.from_numpy(ops.pad_bytes(rng.integers(0, 256, size=(NF, H, W, 3), dtype=np.uint8))).to(dev)
masks = torch.from_numpy(ops.pad_bytes((rng.random((NM, H, W)) < 0.5).astype(np.uint8))).to(dev)
out = torch.zeros((N * 49, 3072), dtype=torch.bfloat16, device=dev)
rows = []
for i in range(N):
    # union of two log-uniform boxes, as the pair crops of the synthetic clip
    bs = []
    for _ in range(2):
        w = int(np.exp(rng.uniform(np.log(24), np.log(640)))); h = int(np.exp(rng.uniform(np.log(24), np.log(640))))
        x0 = int(rng.integers(0, W - w + 1)); y0 = int(rng.integers(0, H - h + 1))
        bs.append((x0, y0, x0 + w, y0 + h))
    x1, y1, x2, y2 = min(bs[0][0], bs[1][0]), min(bs[0][1], bs[1][1]), max(bs[0][2], bs[1][2]), max(bs[0][3], bs[1][3])
    rows.append(((i % NF) * H * W * 3, (i % NM) * H * W, ((i + 7) % NM) * H * W, W, H, x1, y1, x2, y2, _lib.CROP_PAIR, 0))
for name in ("natural", "largest first", "natural", "largest first"):
    r = rows if name == "natural" else sorted(rows, key=lambda t: -(t[7] - t[5]) * (t[8] - t[6]))
    tab = np.array(r, dtype=CROP_DTYPE)
    tdev = torch.from_numpy(tab.view(np.uint8).reshape(-1).copy()).to(dev)
    kw = dict(layout=_lib.LAYOUT_PATCHES, patch=32, out_dtype=_lib.DTYPE_BF16, out=out, entries_dev=tdev, entries_dev_ready=True)
    for _ in range(2):
        ops.crop_preprocess(frames, masks, tab, 224, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.crop_preprocess(frames, masks, tab, 224, **kw)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:14s}: {e0.elapsed_time(e1) / 5:7.3f} ms", flush=True)

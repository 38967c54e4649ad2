"""Crop kernel microbenchmark: time vs source box size (pair mode, S=224, patch layout)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_amd import _lib, ops
from laser_amd.predicate_model import CROP_DTYPE

dev = torch.device("cuda:0")
H, W, NF, NM, N = 720, 1280, 4, 32, 4096
rng = np.random.default_rng(0)
frames = torch.from_numpy(ops.pad_bytes(rng.integers(0, 256, size=(NF, H, W, 3), dtype=np.uint8))).to(dev)
masks = torch.from_numpy(ops.pad_bytes((rng.random((NM, H, W)) < 0.5).astype(np.uint8))).to(dev)
out = torch.zeros((N * 49, 3072), dtype=torch.bfloat16, device=dev)
for (w, h) in [(64, 64), (224, 224), (448, 448), (700, 450), (900, 600), (1280, 720)]:
    rows = []
    for i in range(N):
        x1 = int(rng.integers(0, W - w + 1)); y1 = int(rng.integers(0, H - h + 1))
        rows.append(((i % NF) * H * W * 3, (i % NM) * H * W, ((i + 7) % NM) * H * W, W, H, x1, y1, x1 + w, y1 + h, _lib.CROP_PAIR, 0))
    tab = np.array(rows, dtype=CROP_DTYPE)
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
    ms = e0.elapsed_time(e1) / 5
    by = N * (5.0 * w * h + 301056)
    print(f"{w:5d}x{h:<5d}: {ms:7.3f} ms  {ms*1e3/N:6.2f} us/crop  {by/ms/1e6:7.1f} GB/s algorithmic", flush=True)

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_amd import _lib, ops
from laser_amd.predicate_model import CROP_DTYPE
dev = torch.device("cuda:0")
H, W, NF, NM, N = 720, 1280, 4, 32, 2048
w, h = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(0)
frames = torch.from_numpy(ops.pad_bytes(rng.integers(0, 256, size=(NF, H, W, 3), dtype=np.uint8))).to(dev)
masks = torch.from_numpy(ops.pad_bytes((rng.random((NM, H, W)) < 0.5).astype(np.uint8))).to(dev)
out = torch.zeros((N * 49, 3072), dtype=torch.bfloat16, device=dev)
rows = []
for i in range(N):
    x1 = int(rng.integers(0, W - w + 1)); y1 = int(rng.integers(0, H - h + 1))
    rows.append(((i % NF) * H * W * 3, (i % NM) * H * W, ((i + 7) % NM) * H * W, W, H, x1, y1, x1 + w, y1 + h, _lib.CROP_PAIR, 0))
tab = np.array(rows, dtype=CROP_DTYPE)
for _ in range(2):
    ops.crop_preprocess(frames, masks, tab, 224, layout=_lib.LAYOUT_PATCHES, patch=32, out_dtype=_lib.DTYPE_BF16, out=out)
torch.cuda.synchronize()

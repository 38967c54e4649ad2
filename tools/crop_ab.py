"""The matrix-core crop kernel against the vector kernel on seeded boxes (bits must agree): prints where they differ.
    python tools/crop_ab.py [S]        (runs itself twice, SGCLIP_CROP_MFMA=0 then 1)"""
import os, subprocess, sys
import numpy as np

S = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "child" else 64
if "child" in sys.argv:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from laser_amd import _lib, ops
    S = int(sys.argv[sys.argv.index("child") + 1])
    rng = np.random.default_rng(5)
    H, W = 333, 517
    frames = rng.integers(0, 256, size=(2, H, W, 3), dtype=np.uint8)
    pat = os.environ.get("CROP_AB_PATTERN", "")
    if pat == "const":
        frames[:] = 100
    elif pat == "hgrad":
        frames[:] = (np.arange(W) % 256).astype(np.uint8)[None, None, :, None]
    elif pat == "vgrad":
        frames[:] = (np.arange(H) % 256).astype(np.uint8)[None, :, None, None]
    masks = rng.random((3, H, W)) < 0.5
    ft = torch.from_numpy(ops.pad_bytes(frames)).cuda()
    mt = torch.from_numpy(ops.pad_bytes(masks.astype(np.uint8))).cuda()
    boxes = [(0, 0, W, H), (250, 100, 314, 164), (3, 3, 227, 227), (7, 300, 500, 333), (100, 50, 101, 51), (13, 17, 517, 333)]
    rows = []
    for i, (x1, y1, x2, y2) in enumerate(boxes):
        mode = _lib.CROP_PLAIN if pat else (_lib.CROP_PLAIN, _lib.CROP_OBJECT, _lib.CROP_PAIR)[i % 3]
        rows.append((0, 0, H * W, W, H, x1, y1, x2, y2, mode))
    arr = (_lib.CropEntry * len(rows))()
    for e, r in zip(arr, rows):
        (e.frame_off, e.mask_a_off, e.mask_b_off, e.frame_w, e.frame_h, e.x1, e.y1, e.x2, e.y2, e.mode) = r
    out = ops.crop_preprocess(ft, mt, arr, S).cpu().numpy()
    np.save(sys.argv[-1], out)
    sys.exit(0)
outs = []
for v in ("0", "1"):
    fn = f"/tmp/crop_ab_{v}.npy"
    subprocess.run([sys.executable, __file__, "child", str(S), fn], env=dict(os.environ, SGCLIP_CROP_MFMA=v), check=True)
    outs.append(np.load(fn))
a, b = outs
for i in range(a.shape[0]):
    d = a[i] != b[i]
    print(f"crop {i}: {int(d.sum())} of {d.size} differ; per channel {d.sum(axis=(1, 2)).tolist()}")
    if d.any():
        rows_bad = np.where(d.any(axis=(0, 2)))[0]
        cols_bad = np.where(d.any(axis=(0, 1)))[0]
        print("   rows", rows_bad[:20].tolist(), "... cols", cols_bad[:40].tolist())
        ys, xs = np.where(d[0])
        for y, x in list(zip(ys, xs))[:6]:
            print(f"   ch0 ({y},{x}): vector {a[i, 0, y, x]:.5f} mfma {b[i, 0, y, x]:.5f}")

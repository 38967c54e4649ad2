"""Bit-exactness sweep of the crop kernels against the oracle on 1280 x 720 frames: boxes drawn log-uniformly in size
(the bench's distribution), plus thin, tiny, frame-wide and frame-high ones, all three crop modes, S = 224 and 336.
Usage: python tools/fuzz_crop.py [n_boxes] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from laser_amd import _lib, ops
from oracle import crop_ref, resample_ref


def draw_boxes(rng, n, W, H):
    boxes = []
    for k in range(n):
        kind = k % 8
        if kind == 0:      # thin and wide
            w, h = int(rng.integers(200, W + 1)), int(rng.integers(1, 12))
        elif kind == 1:    # thin and tall
            w, h = int(rng.integers(1, 12)), int(rng.integers(100, H + 1))
        elif kind == 2:    # tiny
            w, h = int(rng.integers(1, 24)), int(rng.integers(1, 24))
        elif kind == 3:    # large
            w, h = int(rng.integers(W // 2, W + 1)), int(rng.integers(H // 2, H + 1))
        else:              # log-uniform 24 .. 640
            w = int(np.exp(rng.uniform(np.log(24), np.log(640))))
            h = int(np.exp(rng.uniform(np.log(24), np.log(640))))
        w, h = min(w, W), min(h, H)
        x1, y1 = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
        boxes.append((x1, y1, x1 + w, y1 + h))
    return boxes


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    H, W = 720, 1280
    frames = (rng.integers(0, 256, size=(2, H, W, 3)) // 2 + np.linspace(0, 127, W)[None, None, :, None]).astype(np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    masks = np.stack([((xx - 400) / 350.0) ** 2 + ((yy - 300) / 250.0) ** 2 <= 1.0,
                      ((xx - 800) / 420.0) ** 2 + ((yy - 420) / 280.0) ** 2 <= 1.0, rng.random((H, W)) < 0.5])
    frames_t = torch.from_numpy(ops.pad_bytes(frames)).to(dev)
    masks_t = torch.from_numpy(ops.pad_bytes(masks.astype(np.uint8))).to(dev)
    boxes = draw_boxes(rng, n, W, H)
    rows = []
    for i, (x1, y1, x2, y2) in enumerate(boxes):
        f, a, b = i % 2, i % 3, (i + 1) % 3
        mode = (_lib.CROP_OBJECT, _lib.CROP_PAIR, _lib.CROP_PAIR, _lib.CROP_PLAIN)[i % 4]
        rows.append((f * H * W * 3, a * H * W, b * H * W, W, H, x1, y1, x2, y2, mode))
    arr = (_lib.CropEntry * len(rows))()
    for e, r in zip(arr, rows):
        (e.frame_off, e.mask_a_off, e.mask_b_off, e.frame_w, e.frame_h, e.x1, e.y1, e.x2, e.y2, e.mode) = r
    blended = {}
    bad = 0
    for S in (224, 336):
        out = ops.crop_preprocess(frames_t, masks_t, arr, S).cpu().numpy()
        for i, r in enumerate(rows):
            foff, aoff, boff, _, _, x1, y1, x2, y2, mode = r
            key = (foff, aoff, boff, mode)
            if key not in blended:
                fr = frames[foff // (H * W * 3)]
                ma, mb = masks[aoff // (H * W)][..., None], masks[boff // (H * W)][..., None]
                blended[key] = (crop_ref.extract_single_object(fr, ma, 0.8) if mode == _lib.CROP_OBJECT else
                                crop_ref.extract_object_subject(fr, ma, mb, 0.5, 0.8) if mode == _lib.CROP_PAIR else fr)
            want = resample_ref.preprocess(blended[key][y1:y2, x1:x2], S)
            if not np.array_equal(out[i], want):
                bad += 1
                print(f"MISMATCH S={S} crop {i} box {boxes[i]} mode {mode}: {(out[i] != want).sum()} values differ", flush=True)
        print(f"S={S}: {len(rows)} crops checked, {bad} mismatches so far", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

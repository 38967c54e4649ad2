"""TEST INFRASTRUCTURE ONLY - CPU oracle, never imported by the product path.

numpy restatement of the reference's mask/crop helpers
(reference: laser/models/model_utils.py:29-56, 87-117, 119-140).

`cv2.addWeighted` (third-party `opencv-python==4.10.0.84`, not installed here)
is restated from its published 8-bit semantics: float32 weights,
`dst = saturate_cast<uchar>(src1*alpha + src2*beta + gamma)` with round to
nearest even.  With the reference's white_alpha = 0.8 the blended value
0.2*p + 204 has a fractional part in {0, .2, .4, .6, .8}, never a tie, so the
rounding mode cannot matter; known answers: p=100 -> 224, p=0 -> 204,
p=255 -> 255 outside the mask, tint 100 -> 177, 254 -> 254.
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import numpy as np


def add_weighted_u8(a: np.ndarray, alpha: float, b: np.ndarray, beta: float, gamma: float = 0.0) -> np.ndarray:
    t = a.astype(np.float32) * np.float32(alpha) + b.astype(np.float32) * np.float32(beta) + np.float32(gamma)
    return np.clip(np.rint(t), 0, 255).astype(np.uint8)


def extract_single_object(img: np.ndarray, mask: np.ndarray, alpha: float = 0.8) -> np.ndarray:
    """model_utils.py:29-56: keep pixels inside the mask, blend the rest towards white."""
    mask = mask.astype(bool)
    if mask.ndim == 2:
        mask = mask[..., None]
    outside = np.logical_not(mask)
    white = np.full(img.shape, 255, dtype=np.uint8)
    masked_white = np.where(outside, white, img)
    return add_weighted_u8(img.astype(np.uint8), 1 - alpha, masked_white.astype(np.uint8), alpha, 0)


def union_box(bbox_ls: Sequence, data_id=None) -> Tuple[int, int, int, int]:
    """model_utils.py:93-115: per-box min/max swap, then union; asserts non-empty."""
    xs1, ys1, xs2, ys2 = [], [], [], []
    for bbox in bbox_ls:
        if isinstance(bbox, dict):
            bx1, by1, bx2, by2 = bbox["x1"], bbox["y1"], bbox["x2"], bbox["y2"]
        elif isinstance(bbox, (list, tuple, np.ndarray)):
            bx1, by1, bx2, by2 = map(int, bbox[:4])
        else:
            raise ValueError(f"Unsupported bbox format: {type(bbox)}")
        bx1, bx2 = min(bx1, bx2), max(bx1, bx2)
        by1, by2 = min(by1, by2), max(by1, by2)
        xs1.append(bx1); xs2.append(bx2); ys1.append(by1); ys2.append(by2)
    x1, x2, y1, y2 = min(xs1), max(xs2), min(ys1), max(ys2)
    assert x1 < x2, f"image bbox issue: {data_id}"
    assert y1 < y2, f"image bbox issue: {data_id}"
    return x1, y1, x2, y2


def crop_image_contain_bboxes(img: np.ndarray, bbox_ls: Sequence, data_id=None) -> np.ndarray:
    """model_utils.py:87-117: end-exclusive slice of the union box (Python
    slicing clamps to the frame and wraps negative starts)."""
    x1, y1, x2, y2 = union_box(bbox_ls, data_id)
    return img[y1:y2, x1:x2]


def extract_object_subject(img: np.ndarray, red_mask: np.ndarray, blue_mask: np.ndarray,
                           alpha: float = 0.5, white_alpha: float = 0.8) -> np.ndarray:
    """model_utils.py:119-140.  Channels are addressed by index (0, 1, 2) under
    the names (b, g, r) whatever the true colour order is: channel 2 is tinted
    where `red_mask`, channel 0 where `blue_mask`; outside both masks every
    channel is blended towards white."""
    red_mask = red_mask.astype(bool)
    blue_mask = blue_mask.astype(bool)
    non_masked = ~(red_mask | blue_mask)
    c0, c1, c2 = img[:, :, 0], img[:, :, 1], img[:, :, 2]
    c2 = np.where(red_mask[:, :, 0], np.clip(c2 + (255 - c2) * alpha, 0, 255), c2).astype(np.uint8)
    c0 = np.where(blue_mask[:, :, 0], np.clip(c0 + (255 - c0) * alpha, 0, 255), c0).astype(np.uint8)
    out = np.stack([c0, c1, c2], axis=-1)
    white = np.full_like(out, 255, dtype=np.uint8)
    return np.where(non_masked, add_weighted_u8(out, 1 - white_alpha, white, white_alpha, 0), out)


def blend_luts(alpha: float = 0.5, white_alpha: float = 0.8) -> Dict[str, np.ndarray]:
    """Per-level tables equivalent to the three blends above:
    `white[p]` outside the mask(s), `keep[p]` inside the object mask (the
    self-blend `p*(1-a) + p*a`), `tint[p]` for a tinted channel."""
    p = np.arange(256, dtype=np.uint8)
    w255 = np.full(256, 255, dtype=np.uint8)
    return {
        "white": add_weighted_u8(p, 1 - white_alpha, w255, white_alpha),
        "keep": add_weighted_u8(p, 1 - white_alpha, p, white_alpha),
        "tint": np.clip(p + (255 - p) * alpha, 0, 255).astype(np.uint8),
    }

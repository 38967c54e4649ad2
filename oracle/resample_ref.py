"""TEST INFRASTRUCTURE ONLY - CPU oracle, never imported by the product path.

numpy restatement of the image preprocessing the reference delegates to
HuggingFace `CLIPImageProcessor` (reference call sites:
laser/models/llava_clip_model_v3.py:244 and :443), i.e.

  1. shortest-edge resize to S with PIL BICUBIC   (transformers/image_transforms.py:295-310,367)
  2. center crop S x S                            (transformers/image_transforms.py:493-503)
  3. u8 -> float64 * (1/255) -> float32           (transformers/image_transforms.py:118-122)
  4. (x - mean) / std in float32, per channel     (transformers/image_transforms.py:436-439)
  5. HWC -> CHW

Step 1 follows Pillow's published two-pass 8-bit resampler (third-party
dependency `Pillow`, pinned 10.4.0 by the reference's
environments/laser_train_env.yml:114, 12.2.0 installed here; algorithm in
Pillow `src/libImaging/Resample.c`: `bicubic_filter` with a = -0.5,
`precompute_coeffs`, `normalize_coeffs_8bpc` with PRECISION_BITS = 22, horizontal
pass to u8 followed by vertical pass to u8).  Pinned in
tests/test_oracle_resample.py against Pillow itself, bit for bit.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
BICUBIC_SUPPORT = 2.0
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def bicubic_filter(x: np.ndarray) -> np.ndarray:
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


def precompute_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray, int]:
    """(xmin[out], count[out], kk[out, ksize] int32 fixed point, ksize) for a
    full-extent box (in0 = 0, in1 = in_size)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = BICUBIC_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    ss = 1.0 / filterscale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # C (int) cast truncates
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.float64)[None, :]
    w = bicubic_filter((x + xmin[:, None] - center[:, None] + 0.5) * ss)
    valid = x < xmax[:, None]
    w = np.where(valid, w, 0.0)
    # sequential left-to-right double accumulation, as the C loop does
    ww = np.zeros(out_size, dtype=np.float64)
    for j in range(ksize):
        ww = ww + w[:, j]
    nz = ww != 0.0
    k = np.where(nz[:, None], w / np.where(nz, ww, 1.0)[:, None], w)
    k = np.where(valid, k, 0.0)
    fixed = np.where(k < 0, (-0.5 + k * (1 << PRECISION_BITS)), (0.5 + k * (1 << PRECISION_BITS)))
    kk = np.trunc(fixed).astype(np.int32)
    return xmin.astype(np.int32), xmax.astype(np.int32), kk, ksize


def _clip8(acc: np.ndarray) -> np.ndarray:
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_output_size(h: int, w: int, S: int) -> Tuple[int, int]:
    """(new_h, new_w): shortest edge -> S, long edge -> int(S*long/short)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = S, int(S * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def resize_center_crop_u8(img: np.ndarray, S: int) -> np.ndarray:
    """u8[h,w,3] -> u8[S,S,3]: PIL-bicubic shortest-edge resize + center crop.
    Only the output window that survives the crop is computed; each output
    pixel is independent, so this equals cropping the full resize."""
    h, w, _ = img.shape
    new_h, new_w = resize_output_size(h, w, S)
    top, left = (new_h - S) // 2, (new_w - S) // 2
    src = img
    # horizontal pass (skipped by Pillow when the width is unchanged)
    if new_w != w:
        xmin, cnt, kk, ksize = precompute_coeffs(w, new_w)
        xmin, cnt, kk = xmin[left:left + S], cnt[left:left + S], kk[left:left + S]
        idx = np.minimum(xmin[:, None] + np.arange(ksize)[None, :], w - 1)       # taps beyond cnt carry k = 0
        g = src[:, idx, :].astype(np.int64)                                        # [h, S, ksize, 3]
        acc = (g * kk[None, :, :, None].astype(np.int64)).sum(axis=2) + (1 << (PRECISION_BITS - 1))
        src = _clip8(acc)
    else:
        src = src[:, left:left + S, :]
    if new_h != h:
        ymin, cnt, kk, ksize = precompute_coeffs(h, new_h)
        ymin, cnt, kk = ymin[top:top + S], cnt[top:top + S], kk[top:top + S]
        idx = np.minimum(ymin[:, None] + np.arange(ksize)[None, :], h - 1)
        g = src[idx, :, :].astype(np.int64)                                        # [S, ksize, S, 3]
        acc = (g * kk[:, :, None, None].astype(np.int64)).sum(axis=1) + (1 << (PRECISION_BITS - 1))
        src = _clip8(acc)
    else:
        src = src[top:top + S, :, :]
    return np.ascontiguousarray(src)


def normalize_lut(mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """f32[3,256]: value of a u8 level after rescale + normalize, computed with
    the reference pipeline's exact dtype sequence."""
    levels = np.arange(256, dtype=np.uint8)
    x = (levels.astype(np.float64) * (1 / 255)).astype(np.float32)
    m = np.array(mean, dtype=np.float32)
    s = np.array(std, dtype=np.float32)
    return np.stack([(x - m[c]) / s[c] for c in range(3)]).astype(np.float32)


def resize_center_crop_pillow(img: np.ndarray, S: int) -> np.ndarray:
    """The same u8[S,S,3] through Pillow's own C resampler, as the reference's processor runs it
    (transformers/image_transforms.py:367); bit-equal to `resize_center_crop_u8` (tests/test_oracle.py)."""
    from PIL import Image
    h, w, _ = img.shape
    new_h, new_w = resize_output_size(h, w, S)
    top, left = (new_h - S) // 2, (new_w - S) // 2
    full = np.asarray(Image.fromarray(np.ascontiguousarray(img)).resize((new_w, new_h), resample=Image.BICUBIC))
    return np.ascontiguousarray(full[top:top + S, left:left + S])


def preprocess(img: np.ndarray, S: int, mean=CLIP_MEAN, std=CLIP_STD, pillow: bool = False) -> np.ndarray:
    """u8[h,w,3] crop -> f32[3,S,S] pixel_values.  pillow: resize with Pillow itself instead of the restatement."""
    u8 = resize_center_crop_pillow(img, S) if pillow else resize_center_crop_u8(img, S)
    lut = normalize_lut(mean, std)
    out = np.empty((3, S, S), dtype=np.float32)
    for c in range(3):
        out[c] = lut[c][u8[:, :, c]]
    return out

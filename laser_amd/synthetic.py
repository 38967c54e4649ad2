"""Seeded synthetic clips shaped like the inputs of `PredicateModel.forward`.

Frames are uniform noise on a low-frequency gradient (so resampling is
exercised); every object is a filled ellipse inscribed in a random box, its
bbox is derived from the mask with the inclusive-max convention of the
reference loader (reference: laser/loading.py:84-104); object ids persist
across frames; pairs are all ordered (a, b), a != b, per frame in
lexicographic order (reference: laser/training/llava_dataset.py:218-222).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np


@dataclass
class SyntheticClipBatch:
    batched_video_ids: List[str]
    batched_videos: List[np.ndarray]            # flat list of u8[H,W,3]
    batched_masks: List[np.ndarray]             # bool[H,W,1]
    batched_bboxes: List[Dict[str, int]]
    batched_names: List[List[str]]
    batched_object_ids: List[Tuple[int, int, int]]
    batched_unary_kws: List[List[str]]
    batched_binary_kws: List[List[str]]
    batched_obj_pairs: List[Tuple[int, int, Tuple[int, int]]]
    batched_video_splits: List[int]
    batched_binary_predicates: List[object] = field(default_factory=lambda: [None])

    def forward_kwargs(self):
        return dict(
            batched_video_ids=self.batched_video_ids, batched_videos=self.batched_videos,
            batched_masks=self.batched_masks, batched_bboxes=self.batched_bboxes,
            batched_names=self.batched_names, batched_object_ids=self.batched_object_ids,
            batched_unary_kws=self.batched_unary_kws, batched_binary_kws=self.batched_binary_kws,
            batched_obj_pairs=self.batched_obj_pairs, batched_video_splits=list(self.batched_video_splits),
            batched_binary_predicates=self.batched_binary_predicates,
        )


def vocab(prefix: str, k: int) -> List[str]:
    return [f"{prefix} word{i} item{(i * 7919) % 1013}" for i in range(k)]


def make_frame(rng: np.random.Generator, H: int, W: int) -> np.ndarray:
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    phase = rng.uniform(0, 2 * np.pi, size=3).astype(np.float32)
    freq = rng.uniform(1.0, 3.0, size=3).astype(np.float32)
    base = np.stack([
        96 + 80 * np.sin(freq[c] * 2 * np.pi * xx / W + phase[c]) * np.cos(freq[c] * 2 * np.pi * yy / H)
        for c in range(3)], axis=-1)
    noise = rng.integers(-48, 49, size=(H, W, 3))
    return np.clip(base + noise, 0, 255).astype(np.uint8)


def ellipse_mask(H: int, W: int, x0: int, y0: int, w: int, h: int) -> np.ndarray:
    """bool[H,W,1] filled ellipse inscribed in the box x0..x0+w-1, y0..y0+h-1."""
    m = np.zeros((H, W, 1), dtype=bool)
    ys = np.arange(y0, y0 + h)[:, None]
    xs = np.arange(x0, x0 + w)[None, :]
    cy, cx = y0 + (h - 1) / 2.0, x0 + (w - 1) / 2.0
    ry, rx = max(h / 2.0, 0.5), max(w / 2.0, 0.5)
    inside = ((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2 <= 1.0
    m[y0:y0 + h, x0:x0 + w, 0] = inside
    return m


def mask_to_bbox(mask: np.ndarray) -> Dict[str, int]:
    m = mask[..., 0] if mask.ndim == 3 else mask
    xs = np.where(m.any(axis=0))[0]
    ys = np.where(m.any(axis=1))[0]
    return {"x1": int(xs[0]), "y1": int(ys[0]), "x2": int(xs[-1]), "y2": int(ys[-1])}


def make_clip_batch(seed: int, n_videos: int = 1, frames: int = 32, objects: int = 16,
                    H: int = 720, W: int = 1280, k_cate: int = 35, k_unary: int = 200, k_binary: int = 50,
                    min_box: int = 24, max_box: int = 640, all_pairs: bool = True,
                    distinct_frames: int | None = None) -> SyntheticClipBatch:
    """One batch of `n_videos` clips of `frames` frames with `objects` objects
    each.  `distinct_frames` (< frames) reuses frame pixel data round-robin to
    bound generator time and host memory; masks and boxes stay per-frame."""
    rng = np.random.default_rng(seed)
    max_box = min(max_box, min(H, W))
    vids, masks, bboxes, oids, pairs, splits = [], [], [], [], [], []
    total = 0
    pool: List[np.ndarray] = []
    for v in range(n_videos):
        for f in range(frames):
            if distinct_frames is not None and len(pool) >= distinct_frames:
                vids.append(pool[(v * frames + f) % distinct_frames])
            else:
                fr = make_frame(rng, H, W)
                pool.append(fr)
                vids.append(fr)
            for o in range(objects):
                w = int(np.exp(rng.uniform(np.log(min_box), np.log(max_box))))
                h = int(np.exp(rng.uniform(np.log(min_box), np.log(max_box))))
                w, h = max(w, 4), max(h, 4)
                x0 = int(rng.integers(0, W - w + 1))
                y0 = int(rng.integers(0, H - h + 1))
                m = ellipse_mask(H, W, x0, y0, w, h)
                masks.append(m)
                bboxes.append(mask_to_bbox(m))
                oids.append((v, f, o))
            if all_pairs:
                for a in range(objects):
                    for b in range(objects):
                        if a != b:
                            pairs.append((v, f, (a, b)))
        total += frames
        splits.append(total)
    return SyntheticClipBatch(
        batched_video_ids=[f"synthetic_{seed}_{v}" for v in range(n_videos)],
        batched_videos=vids, batched_masks=masks, batched_bboxes=bboxes,
        batched_names=[vocab("cate", k_cate) for _ in range(n_videos)],
        batched_object_ids=oids,
        batched_unary_kws=[vocab("unary", k_unary) for _ in range(n_videos)],
        batched_binary_kws=[vocab("binary", k_binary) for _ in range(n_videos)],
        batched_obj_pairs=pairs, batched_video_splits=splits,
        batched_binary_predicates=[None for _ in range(n_videos)],
    )

"""TEST INFRASTRUCTURE - CPU restatement of the reference's mask -> box derivation.

Only tests/ may import this module; the product (laser_amd/) never does.

`bitmasks2bboxes` restates laser/loading.py:84-104 (inclusive x2 / y2; empty masks are dropped,
which shortens the list); it is pinned against outputs of the reference function itself
(tests/golden/consumers.json).  `mask_size` is the pixel count of
laser/training/llava_dataset.py:135.
"""
from __future__ import annotations

import numpy as np


def bitmasks2bboxes(bitmasks):
    if len(bitmasks) == 0:
        return []
    arr = np.stack(bitmasks)
    boxes = []
    x_any = np.any(arr, axis=1)
    y_any = np.any(arr, axis=2)
    for idx in range(arr.shape[0]):
        x = np.where(x_any[idx, :])[0]
        y = np.where(y_any[idx, :])[0]
        if len(x) > 0 and len(y) > 0:
            boxes.append({"x1": int(x[0]), "x2": int(x[-1]), "y1": int(y[0]), "y2": int(y[-1])})
    return boxes


def mask_size(mask) -> int:
    return int(np.count_nonzero(mask))

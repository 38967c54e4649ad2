"""Bit-packed segmentation masks: storage format, upload, and boxes on the GPU.

The reference keeps every mask as a bool array ([1,H,W] in the `*_mask.pkl` files,
laser/preprocess/preprocess_masks_gdsam2.py:205-211; [H,W,1] after
laser/training/llava_dataset.py:130), derives the boxes on the host
(laser/loading.py:84-104 `bitmasks2bboxes`) and hands forward() one byte per pixel.
Stored one bit per pixel the masks are 8x smaller on disk and over PCIe; one HIP
kernel (`sgclip_mask_unpack`) expands them into the layout the crop kernel reads and
returns each mask's tight box and pixel count from the same pass.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib, ops


class PackedMasks:
    """Host-side container: one flat uint8 buffer of row-padded, MSB-first bit rows."""

    def __init__(self, data: np.ndarray, offsets: List[int], shapes: List[tuple], ndims: List[int]):
        self.data, self.offsets, self.shapes, self.ndims = data, offsets, shapes, ndims

    def __len__(self):
        return len(self.offsets)

    @property
    def nbytes(self) -> int:
        return int(self.data.nbytes)

    def subset(self, indices: Sequence[int]) -> "PackedMasks":
        """The masks `indices`, in that order, as a new container."""
        chunks, offsets, off = [], [], 0
        for i in indices:
            H, W = self.shapes[i]
            n = H * ((W + 7) // 8)
            chunks.append(self.data[self.offsets[i]: self.offsets[i] + n])
            offsets.append(off)
            off += n
        data = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)
        return PackedMasks(data, offsets, [self.shapes[i] for i in indices], [self.ndims[i] for i in indices])

    def unpack_host(self, i: int) -> np.ndarray:
        """Mask i as a bool array of its original rank ([H,W], [H,W,1] or [1,H,W])."""
        H, W = self.shapes[i]
        rb = (W + 7) // 8
        bits = np.unpackbits(self.data[self.offsets[i]: self.offsets[i] + H * rb].reshape(H, rb), axis=1)[:, :W].astype(bool)
        return bits[:, :, None] if self.ndims[i] == 3 else bits


def pack_masks(masks: Sequence[np.ndarray]) -> PackedMasks:
    """Pack masks of shape [H,W], [H,W,1] or [1,H,W] (anything `.astype(bool)` accepts)."""
    chunks, offsets, shapes, ndims, off = [], [], [], [], 0
    for m in masks:
        m = np.asarray(m)
        nd = m.ndim
        if nd == 3:
            if m.shape[2] == 1:
                m = m[:, :, 0]
            elif m.shape[0] == 1:
                m = m[0]
            else:
                raise ValueError(f"masks must be [H,W], [H,W,1] or [1,H,W], got {m.shape}")
        elif nd != 2:
            raise ValueError(f"masks must be [H,W], [H,W,1] or [1,H,W], got {m.shape}")
        p = np.packbits(m.astype(bool), axis=-1)
        offsets.append(off); shapes.append(m.shape); ndims.append(nd); chunks.append(p.reshape(-1))
        off += p.size
    data = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)
    return PackedMasks(data, offsets, shapes, ndims)


class UnpackedMasks:
    """Result of `unpack_to_device`: byte-per-pixel masks in HBM + per-mask boxes."""

    def __init__(self, masks: torch.Tensor, offsets: List[int], ndims: List[int], boxes: torch.Tensor,
                 shapes: Optional[List[tuple]] = None):
        self.masks, self.offsets, self.ndims, self._boxes_dev = masks, offsets, ndims, boxes
        self.shapes = list(shapes) if shapes is not None else None    # (H, W) per mask
        self._boxes: Optional[np.ndarray] = None

    @property
    def boxes(self) -> np.ndarray:
        """int32 [n, 5] = x1, y1, x2, y2 (inclusive, as bitmasks2bboxes), pixel count."""
        if self._boxes is None:
            self._boxes = self._boxes_dev.cpu().numpy()
        return self._boxes

    def bbox_dicts(self) -> List[Optional[Dict[str, int]]]:
        """Boxes in the reference's dict form; None for an empty mask (the reference
        drops such masks before this point, laser/training/llava_dataset.py:127)."""
        out = []
        for x1, y1, x2, y2, cnt in self.boxes.tolist():
            out.append({"x1": x1, "x2": x2, "y1": y1, "y2": y2} if cnt > 0 else None)
        return out


def unpack_to_device(packed: PackedMasks, device) -> UnpackedMasks:
    lib = _lib.load()
    n = len(packed)
    offsets, off = [], 0
    table = (_lib.MaskEntry * max(n, 1))()
    for i, (po, (H, W)) in enumerate(zip(packed.offsets, packed.shapes)):
        off = (off + 7) & ~7  # 8-byte aligned rows let the kernel store 8 pixels at a time when W % 8 == 0
        table[i].packed_off, table[i].out_off, table[i].H, table[i].W = po, off, H, W
        offsets.append(off)
        off += H * W
    out = torch.zeros(off + ops.PAD, dtype=torch.uint8, device=device)
    boxes = torch.empty((n, 5), dtype=torch.int32, device=device)
    if n:
        pk = torch.from_numpy(packed.data).to(device, non_blocking=True)
        table_dev = torch.empty(n * C.sizeof(_lib.MaskEntry), dtype=torch.uint8, device=device)
        _lib.check(lib.sgclip_mask_unpack(pk.data_ptr(), pk.numel(), C.addressof(table), table_dev.data_ptr(), n,
                                          out.data_ptr(), off, boxes.data_ptr(), ops._stream(out)))
        out._sgclip_keepalive = (pk, table_dev, table)
    return UnpackedMasks(out, offsets, list(packed.ndims), boxes, [tuple(s) for s in packed.shapes])

"""Upload of the next clip's frames and masks while the current clip is being scored.

The reference hands `forward()` host arrays and pays the host-to-device copy of the finished
`pixel_values` inside the call (laser/models/llava_clip_model_v3.py:245, 444); its caller loops over
clips one after the other (scripts/eval_sgclip.py:339-352).  Here the raw frames and masks go to the
GPU instead (88 MB + 472 MB for a 32-frame 720p clip with 16 masks per frame), and `InputStager`
takes that transfer off the scoring path: a worker thread packs clip n + 1 into pinned host memory
(several threads share the byte copies; numpy releases the interpreter lock in them) and queues the
copies on a HIP stream of its own into one of `depth` device buffer sets, while the caller's thread
builds the crop tables of clip n and the GPU scores it.  `forward(device_inputs=pending)` makes its
stream wait for the copies' event; a buffer set is reused only after the forward() that read it has
finished on the GPU.
"""
from __future__ import annotations

import threading
from concurrent.futures import Future, ThreadPoolExecutor
from typing import List, Optional

import numpy as np
import torch

from . import ops
from .masks import PackedMasks, unpack_to_device


class _Slot:
    """One set of staging buffers: pinned host + device, for frames and for masks."""

    def __init__(self):
        self.pinned = {}      # name -> pinned host tensor
        self.device = {}      # name -> device tensor
        self.consumed: Optional[torch.cuda.Event] = None   # recorded by the forward() that read the device buffers
        self.released = threading.Event()                  # cleared while a staged clip waits for its forward()
        self.released.set()

    def release(self, stream=None):
        """Called by the forward() that was given this slot's clip, once all its kernels are queued on `stream`."""
        if stream is not None:
            ev = torch.cuda.Event()
            ev.record(stream)
            self.consumed = ev
        self.released.set()

    def buffers(self, name: str, nbytes: int, device):
        host = self.pinned.get(name)
        if host is None or host.numel() < nbytes:
            host = self.pinned[name] = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8).pin_memory()
        dev = self.device.get(name)
        if dev is None or dev.numel() < nbytes:
            dev = self.device[name] = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        return host, dev


class PendingInputs:
    """Handle of a clip being staged; `result()` is the DeviceInputs (waits for the host side of the staging only:
    the device copies are ordered by an event that forward() makes its stream wait for)."""

    def __init__(self, future: Future):
        self._future = future

    def result(self):
        return self._future.result()

    def discard(self):
        """Give the staged clip up without scoring it: its buffer set becomes reusable.  (A clip that is staged and then
        neither passed to forward() nor discarded keeps its buffer set, and the submission that needs the set next
        waits for it.)"""
        di = self._future.result()
        if di.slot is not None:
            di.slot.release()
            di.slot = None


def _frame_layout(batched_videos):
    frame_off, frame_hw, arrs, off = [], [], [], 0
    for f in batched_videos:
        f = np.asarray(f)
        if f.dtype != np.uint8 or f.ndim != 3 or f.shape[2] != 3:
            raise ValueError(f"frames must be uint8 [H,W,3], got {f.dtype} {f.shape}")
        frame_off.append(off); frame_hw.append((f.shape[0], f.shape[1])); arrs.append(f)
        off += f.size
    return frame_off, frame_hw, arrs, off


def _mask_layout(batched_masks):
    mask_off, mask_ndim, mask_hw, arrs, off = [], [], [], [], 0
    for m in batched_masks:
        m = np.asarray(m)
        mask_ndim.append(m.ndim)
        if m.ndim == 3:
            if m.shape[2] != 1:
                raise ValueError(f"masks must be [H,W,1] or [H,W], got {m.shape}")
            m = m[:, :, 0]
        elif m.ndim != 2:
            raise ValueError(f"masks must be [H,W,1] or [H,W], got {m.shape}")
        mask_hw.append((m.shape[0], m.shape[1]))
        mask_off.append(off); arrs.append(m)
        off += m.size
    return mask_off, mask_ndim, mask_hw, arrs, off


def _copy_frames(hv, arrs, offs):
    for f, o in zip(arrs, offs):
        np.copyto(hv[o:o + f.size].reshape(f.shape), f)


def _copy_masks(hv, arrs, offs):
    for m, o in zip(arrs, offs):
        dst = hv[o:o + m.size].reshape(m.shape)
        if m.dtype == np.bool_:
            np.copyto(dst, m.view(np.uint8))
        else:
            np.not_equal(m, 0, out=dst.view(np.bool_))      # .astype(bool) of the reference (model_utils.py:34)


class InputStager:
    def __init__(self, model, depth: int = 2, pack_threads: int = 4):
        self.model = model
        self.device = model.device
        with torch.cuda.device(self.device):
            self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots: List[_Slot] = [_Slot() for _ in range(max(2, depth))]
        self._next = 0
        self._worker = ThreadPoolExecutor(1, thread_name_prefix="sgclip-stage")       # clips are staged in submission order
        self._packers = ThreadPoolExecutor(max(1, pack_threads), thread_name_prefix="sgclip-pack")
        self._pack_threads = max(1, pack_threads)
        self.reuse_timeout_s = 600.0

    def submit(self, batched_videos, batched_masks) -> PendingInputs:
        slot = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        return PendingInputs(self._worker.submit(self._stage, slot, batched_videos, batched_masks))

    def close(self):
        self._worker.shutdown(wait=True)
        self._packers.shutdown(wait=True)

    # -- worker thread -----------------------------------------------------------------------
    def _parallel(self, fn, hv, arrs, offs):
        n = self._pack_threads
        if len(arrs) < 2 * n:
            fn(hv, arrs, offs)
            return
        step = (len(arrs) + n - 1) // n
        jobs = [self._packers.submit(fn, hv, arrs[i:i + step], offs[i:i + step]) for i in range(0, len(arrs), step)]
        for j in jobs:
            j.result()

    def _stage(self, slot: _Slot, batched_videos, batched_masks):
        # the clip staged here before has been handed to its forward() (or discarded)
        if not slot.released.wait(timeout=self.reuse_timeout_s):
            raise RuntimeError(f"stage_inputs_async: the clip staged {len(self.slots)} submissions ago was neither passed to "
                               f"forward(device_inputs=...) nor discarded within {self.reuse_timeout_s:.0f} s; its buffers "
                               f"cannot be reused (call PendingInputs.discard() for clips that are dropped)")
        if slot.consumed is not None:
            slot.consumed.synchronize()      # ... and that forward() has finished on the GPU
            slot.consumed = None
        slot.released.clear()
        try:
            return self._fill(slot, batched_videos, batched_masks)
        except BaseException:
            slot.release()                   # nothing was handed out: the slot is free again
            raise

    def _fill(self, slot: _Slot, batched_videos, batched_masks):
        from .predicate_model import DeviceInputs
        with torch.cuda.device(self.device), torch.cuda.stream(self.copy_stream):
            frame_off, frame_hw, arrs, nb = _frame_layout(batched_videos)
            host, dev = slot.buffers("frames", nb + ops.PAD, self.device)
            hv = host.numpy()
            self._parallel(_copy_frames, hv, arrs, frame_off)
            hv[nb:nb + ops.PAD] = 0
            frames = dev[:nb + ops.PAD]
            frames.copy_(host[:nb + ops.PAD], non_blocking=True)
            if isinstance(batched_masks, PackedMasks):
                un = unpack_to_device(batched_masks, self.device)
                di = DeviceInputs(frames, frame_off, frame_hw, un.masks if len(batched_masks) else None, un.offsets,
                                  un.ndims, unpacked=un, mask_hw=un.shapes)
                di.transient = [un.masks, un._boxes_dev]   # allocated under the copy stream for this clip only
            else:
                mask_off, mask_ndim, mask_hw, marrs, mb = _mask_layout(batched_masks)
                masks = None
                if marrs:
                    host, dev = slot.buffers("masks", mb + ops.PAD, self.device)
                    hv = host.numpy()
                    self._parallel(_copy_masks, hv, marrs, mask_off)
                    hv[mb:mb + ops.PAD] = 0
                    masks = dev[:mb + ops.PAD]
                    masks.copy_(host[:mb + ops.PAD], non_blocking=True)
                di = DeviceInputs(frames, frame_off, frame_hw, masks, mask_off, mask_ndim, mask_hw=mask_hw)
            di.ready_event = torch.cuda.Event()
            di.ready_event.record(self.copy_stream)
            di.slot = slot
        return di

"""SGCLIP scoring module for MI355X: same constructor arguments, `forward()`
signature, return layout, asserts and exception classes as the reference's
`PredicateModel` (reference: laser/models/llava_clip_model_v3.py:9-487).

What runs where
  host (Python, integer work):  box resolution, crop / pair index tables, pair
      selection, key tables of the returned mappings        (:188-217, :342-392)
  device (libsgclip_hip.so):     mask blend + crop + resize + normalize, the
      three CLIP image towers, the text towers, similarity, softmax / sigmoid,
      per-object category means                              (:137-186, :233-339, :397-474)
There is no CPU fallback: without the HIP library or a GPU, construction fails.
"""
from __future__ import annotations

import functools
import os
import warnings
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from torch import nn

from . import _lib, ops, training
from .config import ClipConfig, config_by_name
from .masks import PackedMasks, unpack_to_device
from .prob_dict import ProbDict
from .weights import TOWER_PREFIXES, random_state_dict

CROP_DTYPE = np.dtype([("frame_off", "<i8"), ("mask_a_off", "<i8"), ("mask_b_off", "<i8"), ("frame_w", "<i4"),
                       ("frame_h", "<i4"), ("x1", "<i4"), ("y1", "<i4"), ("x2", "<i4"), ("y2", "<i4"), ("mode", "<i4"),
                       ("order", "<i4")])
assert CROP_DTYPE.itemsize == 56


def _no_grad_unless_training(fn):
    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        # every kernel of the call is launched on, and every stream / event belongs to, the model's own GPU,
        # whatever the caller's current device is (the reference's `.to(device)` modules allow the same)
        with torch.cuda.device(self.device):
            if self._grad_mode():
                return fn(self, *args, **kwargs)
            with torch.no_grad():
                return fn(self, *args, **kwargs)
    return wrapper


def _on_own_device(fn):
    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        with torch.cuda.device(self.device):
            return fn(self, *args, **kwargs)
    return wrapper


class _ParamTree(nn.Module):
    """Nested parameter containers that reproduce HF's dotted state-dict names."""

    def __init__(self, sd: Dict[str, torch.Tensor], device):
        super().__init__()
        for name, t in sd.items():
            node = self
            parts = name.split(".")
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _ParamTree({}, device))
                node = node._modules[p]
            node.register_parameter(parts[-1], nn.Parameter(t.to(device), requires_grad=False))


def resolve_union_box(bbox_ls: Sequence, data_id=None):
    """Union box of `bbox_ls` exactly as the reference computes it
    (laser/models/model_utils.py:93-115), including its two asserts."""
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
    return int(x1), int(y1), int(x2), int(y2)


def slice_box(x1, y1, x2, y2, W, H, data_id=None):
    """The pixels `img[y1:y2, x1:x2]` selects (model_utils.py:117): Python slice
    clamping / negative wrap-around included."""
    xa, xb, _ = slice(x1, x2).indices(W)
    ya, yb, _ = slice(y1, y2).indices(H)
    if xb <= xa or yb <= ya:
        raise ValueError(f"crop box [{y1}:{y2}, {x1}:{x2}] selects no pixel of a {H}x{W} frame: {data_id}")
    return xa, ya, xb, yb


class DeviceInputs:
    """Frames and masks of one forward() call resident in HBM."""

    def __init__(self, frames: torch.Tensor, frame_off: List[int], frame_hw: List[tuple], masks: Optional[torch.Tensor],
                 mask_off: List[int], mask_ndim: List[int], unpacked=None, mask_hw: Optional[List[tuple]] = None):
        self.frames, self.frame_off, self.frame_hw = frames, frame_off, frame_hw
        self.masks, self.mask_off, self.mask_ndim = masks, mask_off, mask_ndim
        self.mask_hw = mask_hw    # (H, W) per mask; forward() checks each against its frame
        self.unpacked = unpacked  # laser_amd.masks.UnpackedMasks when the masks arrived bit-packed
        # set by laser_amd.staging.InputStager: the event behind the uploads, the buffer set to hand back, and tensors
        # that were allocated under the copy stream
        self.ready_event = None
        self.slot = None
        self.transient = []

    def bbox_dicts(self):
        """Boxes of bit-packed masks, computed on the GPU, in the reference's dict form
        (laser/loading.py:84-104); usable as forward()'s batched_bboxes."""
        if self.unpacked is None:
            raise ValueError("boxes are only derived for masks staged from PackedMasks")
        return self.unpacked.bbox_dicts()


class PredicateModel(nn.Module):
    def __init__(self, hidden_dim=None, num_top_pairs=None, device="cuda", model_name="openai/clip-vit-base-patch32",
                 tokenizer=None, config: Optional[ClipConfig] = None, init_seed: Optional[int] = None):
        """hidden_dim / num_top_pairs are accepted and unused, as in the reference (:11).
        Weights: pretrained checkpoints cannot be fetched here; load them with
        `load_state_dict` (keys `clip_{cate,unary,binary}_model.*`).  `init_seed`
        fills the three towers with seeded random weights instead."""
        super().__init__()
        _lib.load()  # fail loudly when the HIP extension is missing
        if not torch.cuda.is_available():
            raise RuntimeError("laser_amd.PredicateModel needs an MI355X: no GPU is visible and there is no CPU path")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"laser_amd.PredicateModel runs on a GPU device, got {device!r}")
        # any CLIP geometry the reference's AutoModel.from_pretrained(model_name) would build (:15-23): a hub name whose
        # geometry is known offline, a local HF snapshot directory / config.json, an HF config dict, or a ClipConfig
        self.cfg = config_by_name(config if config is not None else model_name)
        self.num_top_pairs = num_top_pairs
        self.use_sparse = False
        if tokenizer is None:
            tokenizer = self._default_tokenizer(model_name)
        self.clip_tokenizer = tokenizer
        if getattr(self.clip_tokenizer, "pad_token", None) is None:
            self.clip_tokenizer.pad_token = self.clip_tokenizer.unk_token if self.clip_tokenizer.unk_token else self.clip_tokenizer.eos_token
        seed = 0 if init_seed is None else init_seed
        for k, prefix in enumerate(TOWER_PREFIXES):
            self.add_module(prefix, _ParamTree(random_state_dict(self.cfg, seed * 1000 + k), self.device))
        self._towers = None          # packed device blobs, rebuilt when weights change
        self._train_towers = None    # training-form towers over the fp32 parameters (built on first use)
        self._text_cache: Dict = {}
        self.cache_text_features = True
        # Training-mode pair selection (:347-386) is a discrete function of a floating-point ranking.  With the guard on,
        # objects whose mean category probability lies within the bf16 towers' error of the boundary between the
        # `topk_cate`-th and the next candidate of a name the predicates use are scored again in fp32
        # (laser_amd/rescore.py) before ranking.  rank_guard_cos_err: bound on the error of a cosine similarity through
        # the bf16 towers; rank_exact_cos_err: the same for the fp32 path (measured against the CPU oracle: per-object
        # means agree to 1e-8 .. 4e-7 absolute at logit scale 100, i.e. cosines to about 1e-8).  What the guard did in the last forward() is in `ranking_report`;
        # boundaries that stay closer than the fp32 bound are listed there with decided=False and warned about.
        self.rank_guard = True
        self.rank_guard_cos_err = 2.5e-3
        self.rank_exact_cos_err = 1e-7
        self.ranking_report: List[dict] = []
        self._exact_towers: Dict = {}
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed_weights())

    # -- weights ---------------------------------------------------------------
    def _default_tokenizer(self, model_name):
        """The reference's AutoTokenizer.from_pretrained(model_name) (:15), from local files only.  Without the
        vocabulary files it raises, as the reference does: a stand-in tokenizer under real weights would score
        confidently against the wrong token ids.  Tests and benchmarks pass `tokenizer=` explicitly."""
        try:
            from transformers import AutoTokenizer
            return AutoTokenizer.from_pretrained(model_name, local_files_only=True)
        except Exception as e:
            raise OSError(f"no CLIP tokenizer files for {model_name!r} on this machine (looked in the local HuggingFace "
                          f"cache / path only, HF_HOME={os.environ.get('HF_HOME', '~/.cache/huggingface')!r}; "
                          f"{type(e).__name__}: {e}).  Pass tokenizer=<a CLIPTokenizer> (or, for synthetic weights, "
                          f"laser_amd.tokenizer.HashTokenizer) to PredicateModel.") from e

    def invalidate_packed_weights(self):
        self._towers = None
        self._text_cache = {}

    def enable_training(self, on: bool = True):
        """Make the CLIP parameters trainable (the reference's are by default, llava_clip_model_v3.py:19-21;
        here they start frozen so that scoring never records activations).  With trainable parameters and
        autograd enabled, forward() runs the training form of the towers and its outputs carry gradients;
        under torch.no_grad() it stays the inference path, reading the current parameter values."""
        self.requires_grad_(on)
        return self

    @_on_own_device
    def flatten_parameters(self) -> "training.FlatParameters":
        """Move every parameter (and its gradient) into one fp32 buffer each, see `training.FlatParameters`:
        weight gradients are accumulated in place, `training.Adam(flat)` is one launch, and
        `training.allreduce_gradients(flat)` needs no packing.  Returns the (cached) FlatParameters."""
        if getattr(self, "_flat", None) is None:
            tt = self._trainers()
            named = dict(self.named_parameters())
            order: List[str] = []
            for prefix in TOWER_PREFIXES:
                for tower in tt[prefix][:2]:
                    order += [f"{prefix}.{n}" for n in tower.names]
                order.append(f"{prefix}.logit_scale")
            order += [n for n in named if n not in set(order)]
            self._flat = training.FlatParameters([(n, named[n]) for n in order])
            for prefix in TOWER_PREFIXES:
                for tower in tt[prefix][:2]:
                    tower.flat, tower.flat_prefix = self._flat, prefix + "."
            self.invalidate_packed_weights()
        return self._flat

    def gradient_reducer(self, group=None, max_bucket_bytes: int = 256 << 20, wire_dtype=None) -> "training.GradientReducer":
        """`training.GradientReducer` over this model's flat gradient buffer with one bucket per tower (vision;
        text + logit scale) in the buffer's layout order, plus one for any parameter outside the towers: each bucket's
        all-reduce starts when that tower's backward is done.  Replaces the reference's DDP wrapper
        (scripts/train_sgclip.py:203-213)."""
        flat = self.flatten_parameters()
        tt = self._trainers()
        buckets, off = [], 0
        for prefix in TOWER_PREFIXES:
            vision, text = tt[prefix][:2]
            v_end = flat.index[f"{prefix}.{text.names[0]}"][0]
            ls = flat.index[f"{prefix}.logit_scale"]
            t_end = ls[0] + (ls[1] + flat.ALIGN - 1) // flat.ALIGN * flat.ALIGN
            buckets += [(f"{prefix}.vision", off, v_end), (f"{prefix}.text", v_end, t_end)]
            vision.bucket_key, text.bucket_key = f"{prefix}.vision", f"{prefix}.text"
            off = t_end
        if off < flat.total:
            buckets.append(("rest", off, flat.total))
        return training.GradientReducer(flat, buckets, group, max_bucket_bytes, wire_dtype,
                                        towers=[t for prefix in TOWER_PREFIXES for t in tt[prefix][:2]])

    def set_recompute_chunk(self, images: Optional[int], keep_activations: bool = True, max_kept_bytes: Optional[int] = None):
        """Bound the activation memory of the train step: the image towers run `images` crops at a time and recompute
        a chunk's forward during backward (None: one pass, keep everything).  With keep_activations, chunks keep
        their activations while the card has memory to spare and, if given, while the kept chunks of ALL towers together
        stay within `max_kept_bytes`; only the rest are recomputed.  keep_activations=False is the bounded-memory mode:
        peak memory is one chunk's activations plus the parameters, independent of the batch and of allocator state."""
        budget = training.KeptBudget(max_kept_bytes)
        for prefix in TOWER_PREFIXES:
            tower = self._trainers()[prefix][0]
            tower.recompute_chunk = images
            tower.keep_activations = keep_activations
            tower.kept_budget = budget

    def _grad_mode(self) -> bool:
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def _trainers(self):
        if self._train_towers is None:
            tt = {}
            for prefix in TOWER_PREFIXES:
                params = dict(self._modules[prefix].named_parameters())
                tt[prefix] = (training.TrainTower(self.cfg, "vision", params), training.TrainTower(self.cfg, "text", params),
                              params["logit_scale"])
            self._train_towers = tt
        return self._train_towers

    def _drop_stale_packed_weights(self):
        """Called once per forward(): an optimizer step or an in-place edit of a parameter since the
        blobs were packed bumps its version counter."""
        plist = getattr(self, "_param_tuple", None)
        if plist is None:          # the Parameter objects never change identity; walking the module tree costs milliseconds
            plist = self._param_tuple = tuple(self.parameters())
        versions = tuple(p._version for p in plist)
        if versions != getattr(self, "_packed_versions", None):
            self.invalidate_packed_weights()
            self._packed_versions = versions

    @_on_own_device
    def _packed(self):
        if self._towers is None:
            towers = {}
            for prefix in TOWER_PREFIXES:
                sd = {k: v.detach() for k, v in self._modules[prefix].state_dict().items()}
                towers[prefix] = (ops.Tower(self.cfg, "vision", sd, self.device), ops.Tower(self.cfg, "text", sd, self.device),
                                  float(sd["logit_scale"].exp()))
            self._towers = towers
        return self._towers

    # -- inputs ----------------------------------------------------------------
    @_on_own_device
    def stage_inputs(self, batched_videos, batched_masks) -> DeviceInputs:
        """Upload frames (u8[H,W,3]) and masks (anything `.astype(bool)` accepts, [H,W,1]
        or [H,W]; or a `laser_amd.masks.PackedMasks`, uploaded as bits and expanded on the
        GPU) into two flat device buffers."""
        # Frames and masks go into pinned host buffers that persist across calls (one pass over the bytes, no
        # concatenated temporary) and from there to the device with one asynchronous copy each.
        frame_off, frame_hw, off = [], [], 0
        arrs = []
        for f in batched_videos:
            f = np.asarray(f)
            if f.dtype != np.uint8 or f.ndim != 3 or f.shape[2] != 3:
                raise ValueError(f"frames must be uint8 [H,W,3], got {f.dtype} {f.shape}")
            frame_off.append(off); frame_hw.append((f.shape[0], f.shape[1])); arrs.append(f)
            off += f.size
        host = self._pinned("frames", off + ops.PAD)
        hv = host.numpy()
        for f, o in zip(arrs, frame_off):
            np.copyto(hv[o:o + f.size].reshape(f.shape), f)
        hv[off:off + ops.PAD] = 0
        frames = torch.empty(off + ops.PAD, dtype=torch.uint8, device=self.device)
        frames.copy_(host[:off + ops.PAD], non_blocking=True)
        self._pinned_done("frames")
        if isinstance(batched_masks, PackedMasks):
            un = unpack_to_device(batched_masks, self.device)
            return DeviceInputs(frames, frame_off, frame_hw, un.masks if len(batched_masks) else None, un.offsets,
                                un.ndims, unpacked=un, mask_hw=un.shapes)
        mask_off, mask_ndim, mask_hw, off = [], [], [], 0
        arrs = []
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
        masks = None
        if arrs:
            host = self._pinned("masks", off + ops.PAD)
            hv = host.numpy()
            for m, o in zip(arrs, mask_off):
                dst = hv[o:o + m.size].reshape(m.shape)
                if m.dtype == np.bool_:
                    np.copyto(dst, m.view(np.uint8))
                else:
                    np.not_equal(m, 0, out=dst.view(np.bool_))      # .astype(bool) of the reference (model_utils.py:34)
            hv[off:off + ops.PAD] = 0
            masks = torch.empty(off + ops.PAD, dtype=torch.uint8, device=self.device)
            masks.copy_(host[:off + ops.PAD], non_blocking=True)
            self._pinned_done("masks")
        return DeviceInputs(frames, frame_off, frame_hw, masks, mask_off, mask_ndim, mask_hw=mask_hw)

    def stage_inputs_async(self, batched_videos, batched_masks, depth: int = 2):
        """`stage_inputs` off the caller's thread and off the scoring stream: returns at once with a handle that
        `forward(device_inputs=...)` accepts.  A worker thread packs the arrays into pinned memory and queues the
        copies on a HIP stream of its own, so the upload of clip n + 1 runs under the scoring of clip n
        (laser_amd/staging.py).  Submit at most `depth - 1` clips ahead of the forward() calls; pass each handle to
        exactly one forward()."""
        if getattr(self, "_stager", None) is None:
            from .staging import InputStager
            self._stager = InputStager(self, depth=depth)
        return self._stager.submit(batched_videos, batched_masks)

    def _pinned(self, name: str, nbytes: int) -> torch.Tensor:
        """A pinned host staging buffer of at least `nbytes`, free to be overwritten (the copy that last read it
        has finished)."""
        bufs = self.__dict__.setdefault("_pinned_bufs", {})
        buf, ev = bufs.get(name, (None, None))
        if ev is not None:
            ev.synchronize()
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8).pin_memory()
        bufs[name] = (buf, None)
        return buf

    def _pinned_done(self, name: str):
        """Marks the point in the current stream after which the staging buffer `name` may be reused."""
        buf, _ = self._pinned_bufs[name]
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._pinned_bufs[name] = (buf, ev)

    def _text_features(self, prefix: str, kws: Sequence[str]) -> torch.Tensor:
        if self._grad_mode():
            tokens = self.clip_tokenizer(list(kws), return_tensors="pt", max_length=75, truncation=True, padding="max_length")
            am = tokens.get("attention_mask") if hasattr(tokens, "get") else tokens["attention_mask"]
            return training.encode_text(self._trainers()[prefix][1], tokens["input_ids"], am)
        key = (prefix, tuple(kws))
        if self.cache_text_features and not self.training and key in self._text_cache:
            return self._text_cache[key]
        tokens = self.clip_tokenizer(list(kws), return_tensors="pt", max_length=75, truncation=True, padding="max_length")
        ids = tokens["input_ids"]
        am = tokens.get("attention_mask") if hasattr(tokens, "get") else tokens["attention_mask"]
        feats = self._packed()[prefix][1].encode_text(ids, am)
        if self.cache_text_features and not self.training:
            self._text_cache[key] = feats
        return feats

    def _crops(self, dev_in: DeviceInputs, table: np.ndarray, alpha, white_alpha) -> torch.Tensor:
        cfg = self.cfg
        return ops.crop_preprocess(dev_in.frames, dev_in.masks, table, cfg.image, layout=_lib.LAYOUT_PATCHES,
                                   patch=cfg.patch, out_dtype=_lib.DTYPE_BF16, alpha=alpha, white_alpha=white_alpha)

    def _side_streams(self):
        """Three side HIP streams for the inference path: the image towers are independent until the similarity
        step, so the unary tower and the two halves of the pair tower are queued beside the category tower; the
        tails and kernel boundaries of one launch sequence are then filled by the others.
        SGCLIP_CONCURRENT_TOWERS=0 keeps everything on the caller's stream."""
        if not getattr(self, "concurrent_towers", True):
            return []
        if getattr(self, "_streams", None) is None:
            on = os.environ.get("SGCLIP_CONCURRENT_TOWERS", "1") != "0"
            self._streams = [torch.cuda.Stream(device=self.device) for _ in range(4)] if on else []
        return self._streams

    def _encode_images(self, prefix: str, patches: torch.Tensor, n: int) -> torch.Tensor:
        if self._grad_mode():
            return training.encode_images(self._trainers()[prefix][0], patches, n)
        return self._packed()[prefix][0].encode_images(patches, n)

    def _similarity(self, prefix: str, text: torch.Tensor, img: torch.Tensor, mode: int) -> torch.Tensor:
        if self._grad_mode():
            return training.similarity(text, img, self._trainers()[prefix][2], mode)
        return ops.similarity(text, img, self._packed()[prefix][2], mode)

    # -- ranking guard -----------------------------------------------------------
    def _exact(self, prefix: str):
        """(vision, text) fp32 towers over the model's own fp32 parameters, built on first use."""
        if prefix not in self._exact_towers:
            from .rescore import ExactTower
            params = dict(self._modules[prefix].named_parameters())
            self._exact_towers[prefix] = (ExactTower(self.cfg, "vision", params), ExactTower(self.cfg, "text", params), params)
        return self._exact_towers[prefix]

    def _rank_guard(self, vid, host, c_rows, names, oids, seg_host, table, di, preds, topk, alpha, white_alpha, softmax,
                    pool):
        """Returns `host` (per-object mean category scores [rows of c_rows, objects], float64 on the host) with the
        columns of every object the ranking's outcome depends on replaced by fp32 values.

        The selected pairs depend on the SET of the best `topk` objects of each name the predicates mention
        (:350-357, 368-386).  With each value known up to its error bound, the set is certain once the lowest
        possible value inside it exceeds the highest possible value outside; while it is not, the bf16-scored
        objects that straddle the boundary are scored again in fp32 (all their frames, the whole vocabulary, mean
        over frames as in :303-311) and the test is repeated with their error bound reduced to the fp32 one."""
        from . import rescore
        used = set()
        for pred in preds or []:
            used.add(pred[1]); used.add(pred[2])
        rows = [(r, nm) for r, nm in c_rows if nm in used]
        if not rows:
            return host
        G_ = len(oids)
        vis, txt, params = self._exact("clip_cate_model")
        scale = float(params["logit_scale"].detach().exp())
        exact = np.zeros(G_, dtype=bool)
        seg = np.asarray(seg_host, dtype=np.int64)
        text_exact = None

        def bounds(v, is_exact):
            cos_err = np.where(is_exact, self.rank_exact_cos_err, self.rank_guard_cos_err)
            # a logit moves by scale * cos_err; a softmax probability by at most that fraction of itself (twice: the
            # normaliser moves too)
            return np.minimum(1.0, 2.0 * scale * cos_err) * np.abs(v) if softmax else scale * cos_err * np.ones_like(v)

        def undecided():
            todo, open_rows = set(), []
            for r, nm in rows:
                v = host[r]
                order = np.argsort(-v, kind="stable")
                inside, outside = order[:topk], order[topk:]
                eps = bounds(v, exact)
                lo_in, hi_out = (v - eps)[inside].min(), (v + eps)[outside].max()
                if lo_in > hi_out:
                    continue
                strad = [g for g in inside if v[g] - eps[g] <= hi_out] + [g for g in outside if v[g] + eps[g] >= lo_in]
                open_rows.append((nm, float(v[inside].min() - v[outside].max()), len(strad)))
                todo.update(int(g) for g in strad if not exact[g])
            return todo, open_rows

        n_rescored = 0
        while True:
            todo, open_rows = undecided()
            if not todo:
                break
            groups = sorted(todo)
            cols = np.nonzero(np.isin(seg, groups))[0]
            sub = np.ascontiguousarray(table[cols])
            patches = ops.crop_preprocess(di.frames, di.masks, sub, self.cfg.image, layout=_lib.LAYOUT_PATCHES,
                                          patch=self.cfg.patch, out_dtype=_lib.DTYPE_F32, alpha=alpha, white_alpha=white_alpha)
            feats = vis.image_features(patches, len(cols))
            if text_exact is None:
                tokens = self.clip_tokenizer(list(names), return_tensors="pt", max_length=75, truncation=True,
                                             padding="max_length")
                am = tokens.get("attention_mask") if hasattr(tokens, "get") else tokens["attention_mask"]
                text_exact = txt.text_features(tokens["input_ids"], am)
            dense = rescore.probabilities_exact(text_exact, feats, scale, softmax)
            local = {g: j for j, g in enumerate(groups)}
            seg_local = torch.tensor([local[int(seg[c])] for c in cols], dtype=torch.int32, device=self.device)
            mean = pool(ops.segment_mean(dense, seg_local, len(groups))).detach().cpu().numpy().astype(np.float64)
            host = host.copy()
            host[:, groups] = mean
            exact[groups] = True
            n_rescored += len(groups)
        for nm, margin, n in open_rows:
            warnings.warn(f"pair selection of video {vid}: the candidates of {nm!r} at the top-{topk} boundary differ by "
                          f"{margin:.3e}, below what fp32 arithmetic resolves; the selected pairs may differ from another "
                          f"fp32 implementation's")
        self.ranking_report.append({"video": vid, "objects": G_, "rescored": n_rescored,
                                    "undecided": [(nm, margin) for nm, margin, _ in open_rows]})
        return host

    # -- forward ---------------------------------------------------------------
    @_no_grad_unless_training
    def forward(self, batched_video_ids, batched_videos, batched_masks, batched_bboxes, batched_names,
                batched_object_ids, batched_unary_kws, batched_binary_kws, batched_obj_pairs, batched_video_splits,
                batched_binary_predicates, unary_segment_size=None, binary_segment_size=None, alpha=0.5,
                white_alpha=0.8, topk_cate=3, dummy_str="$$$", multi_class=False, output_logit=False,
                output_embeddings=False, device_inputs=None, allow_dummy_pair: bool = True):
        """Reference signature (llava_clip_model_v3.py:86-107) plus two optional extras: `device_inputs`
        (frames / masks already staged in HBM: a DeviceInputs from `stage_inputs`, or the handle
        `stage_inputs_async` returned) and `allow_dummy_pair` (False: a call without any pair
        scores no whole-frame stand-in, for shards of a larger call that does have pairs).
        Gradients flow to the CLIP parameters when they are trainable (`enable_training`) and autograd
        is enabled; otherwise the call runs under torch.no_grad()."""
        if device_inputs is not None and hasattr(device_inputs, "result"):
            device_inputs = device_inputs.result()
        slot = getattr(device_inputs, "slot", None)
        main_stream = torch.cuda.current_stream(self.device)
        if device_inputs is not None and device_inputs.ready_event is not None:
            main_stream.wait_event(device_inputs.ready_event)
            for t in device_inputs.transient:
                t.record_stream(main_stream)
        try:
            return self._forward(batched_video_ids, batched_videos, batched_masks, batched_bboxes, batched_names,
                                 batched_object_ids, batched_unary_kws, batched_binary_kws, batched_obj_pairs,
                                 batched_video_splits, batched_binary_predicates, alpha, white_alpha, topk_cate, dummy_str,
                                 multi_class, output_logit, output_embeddings, device_inputs, allow_dummy_pair)
        finally:
            if slot is not None:
                device_inputs.slot = None
                slot.release(main_stream)     # every side stream has been joined into main_stream by now

    def _forward(self, batched_video_ids, batched_videos, batched_masks, batched_bboxes, batched_names,
                 batched_object_ids, batched_unary_kws, batched_binary_kws, batched_obj_pairs, batched_video_splits,
                 batched_binary_predicates, alpha, white_alpha, topk_cate, dummy_str, multi_class, output_logit,
                 output_embeddings, device_inputs, allow_dummy_pair):
        dev = self.device
        B = len(batched_video_ids)
        grad_mode = self._grad_mode()
        if not grad_mode:
            self._drop_stale_packed_weights()

        def fill_empty(batched_kw):
            return [list(kw) if len(kw) else [dummy_str] for kw in batched_kw]

        batched_names = fill_empty(batched_names)
        batched_unary_kws = fill_empty(batched_unary_kws)
        batched_binary_kws = fill_empty(batched_binary_kws)
        dummy_terms: List[torch.Tensor] = []
        self.ranking_report = []

        # Step 1: text features of the three vocabularies (:137-186).  They are needed at the similarity step only: on
        # the inference path the text towers (small launches) are queued on a side stream beside the image towers.
        side = [] if grad_mode else self._side_streams()
        main_stream = torch.cuda.current_stream(self.device)

        def all_text_features():
            return ([self._text_features("clip_cate_model", n) for n in batched_names],
                    [self._text_features("clip_unary_model", n) for n in batched_unary_kws],
                    [self._text_features("clip_binary_model", n) for n in batched_binary_kws])
        text_stream = side[3] if len(side) > 3 else None
        if text_stream is not None:
            # no wait for the caller's stream: the text towers read token ids they upload themselves and the packed
            # weights, which were uploaded with blocking copies when they were packed
            if not ops.PINNED_TABLES:
                text_stream.wait_stream(main_stream)
            with torch.cuda.stream(text_stream):
                name_feats, unary_feats, binary_feats = all_text_features()
            for f in name_feats + unary_feats + binary_feats:
                f.record_stream(main_stream)
        else:
            name_feats, unary_feats, binary_feats = all_text_features()

        # Step 2: object crop table (:188-217)
        splits = [0] + list(batched_video_splits)
        assert len(batched_object_ids) > 0, f"No object bbox: {batched_video_ids}"
        if device_inputs is None:
            device_inputs = self.stage_inputs(batched_videos, batched_masks)
        di = device_inputs
        inputs_ready = torch.cuda.Event()
        inputs_ready.record(main_stream)     # frames and masks are in HBM: side streams start from here
        if batched_bboxes is None:                       # bit-packed masks: boxes come from the unpack kernel
            batched_bboxes = di.bbox_dicts()
        obj_index = {}                                   # (vid, fid, oid) -> position in batched_object_ids
        ids_lookup: Dict[int, list] = {v: [] for v in range(B)}
        per_vid_rows: Dict[int, list] = {v: [] for v in range(B)}
        n_listed = min(len(batched_object_ids), len(batched_bboxes), len(di.mask_off))
        obj_box = np.zeros((n_listed, 4), dtype=np.int64)  # every object's own box, corners ordered (:100-108)
        for pos, ((vid, fid, oid), bbox) in enumerate(zip(batched_object_ids, batched_bboxes)):
            if pos >= len(di.mask_off):
                break                                    # zip() in the reference stops at the shortest list
            fidx = splits[vid] + fid
            H, W = di.frame_hw[fidx]
            if di.mask_hw is not None and tuple(di.mask_hw[pos]) != (H, W):
                # the reference's np.where / addWeighted raise on such a pair of arrays (model_utils.py:48-54)
                raise ValueError(f"mask {pos} of object {(vid, fid, oid)} is {tuple(di.mask_hw[pos])}, its frame is "
                                 f"{(H, W)}: {batched_video_ids}")
            x1, y1, x2, y2 = resolve_union_box([bbox], batched_video_ids)
            obj_box[pos] = (x1, y1, x2, y2)
            xa, ya, xb, yb = slice_box(x1, y1, x2, y2, W, H, batched_video_ids)
            per_vid_rows[vid].append((di.frame_off[fidx], di.mask_off[pos], -1, W, H, xa, ya, xb, yb, _lib.CROP_OBJECT, 0))
            obj_index[(vid, fid, oid)] = pos
            ids_lookup[vid].append((fid, oid))
        obj_rows = [r for v in range(B) for r in per_vid_rows[v]]
        obj_start = np.cumsum([0] + [len(per_vid_rows[v]) for v in range(B)])
        obj_table = np.array(obj_rows, dtype=CROP_DTYPE)
        n_obj = len(obj_rows)

        # per-object groups of the category mean (:296-311), uploaded before any kernel is queued so
        # that the transfer never waits for the towers
        groups: Dict[int, tuple] = {}
        for vid in range(B):
            group_of, oids = {}, []
            for (_, oid) in ids_lookup[vid]:
                if oid not in group_of:
                    group_of[oid] = len(oids)
                    oids.append(oid)
            seg_host = [group_of[oid] for (_, oid) in ids_lookup[vid]]
            # from pinned memory and non-blocking: a blocking copy would make the host wait for everything queued on
            # this stream, the previous call's kernels included, and end its run ahead of the GPU
            seg_t = torch.tensor(seg_host, dtype=torch.int32)
            seg_t = seg_t.pin_memory().to(dev, non_blocking=True) if ops.PINNED_TABLES else seg_t.to(dev)
            groups[vid] = (oids, seg_t, seg_host)

        need_ranking = batched_binary_predicates[0] is not None
        pair_lookup: Dict[int, dict] = {v: {} for v in range(B)}
        for (vid, fid, (a, b)) in batched_obj_pairs:
            pair_lookup[vid].setdefault((a, b), []).append(fid)

        def queue_pairs(obj_per_cate_all):
            """Step 4 (:342-449): pair selection, the pair crop table, and the pair crops through the binary tower.
            Returns (features, selected pairs per video, first column of each video, join): `join()` makes the caller's
            stream wait for the side streams the tower was queued on."""
            if not need_ranking:
                selected = list(batched_obj_pairs)
            else:
                sel = set()
                for bp_vid, preds in enumerate(batched_binary_predicates):
                    cands = {nm: [oid for _, oid in lst[:topk_cate]] for nm, lst in obj_per_cate_all[bp_vid].items()}
                    for (rel_name, from_name, to_name) in preds:
                        if from_name in cands and to_name in cands:
                            for a in cands[from_name]:
                                for b in cands[to_name]:
                                    if (a, b) in pair_lookup[bp_vid]:
                                        for fid in pair_lookup[bp_vid][(a, b)]:
                                            sel.add((bp_vid, fid, (a, b)))
                selected = list(sel)
            new_select: Dict[int, list] = {v: [] for v in range(B)}
            for item in selected:
                new_select[item[0]].append(item)
            ordered = [item for v in range(B) for item in new_select[v]]       # crops grouped by video, as the towers see them
            pair_table = np.zeros(len(ordered), dtype=CROP_DTYPE)
            if ordered:
                # the whole table at once: union of the two objects' boxes (:93-115), then the pixels that
                # img[y1:y2, x1:x2] selects (:117)
                pa = np.fromiter((obj_index[(vid, fid, pr[0])] for (vid, fid, pr) in ordered), dtype=np.int64, count=len(ordered))
                pb = np.fromiter((obj_index[(vid, fid, pr[1])] for (vid, fid, pr) in ordered), dtype=np.int64, count=len(ordered))
                ndim = np.asarray(di.mask_ndim)
                if (ndim[pa] != 3).any() or (ndim[pb] != 3).any():
                    raise IndexError("too many indices for array: pair masks must be 3-dimensional [H,W,1]")
                fidx = np.fromiter((splits[vid] + fid for (vid, fid, _) in ordered), dtype=np.int64, count=len(ordered))
                hw = np.asarray(di.frame_hw, dtype=np.int64)
                Hs, Ws = hw[fidx, 0], hw[fidx, 1]
                x1 = np.minimum(obj_box[pa, 0], obj_box[pb, 0]); y1 = np.minimum(obj_box[pa, 1], obj_box[pb, 1])
                x2 = np.maximum(obj_box[pa, 2], obj_box[pb, 2]); y2 = np.maximum(obj_box[pa, 3], obj_box[pb, 3])

                def clamp(v, n):                              # slice.indices() for a positive step
                    return np.where(v < 0, np.maximum(v + n, 0), np.minimum(v, n))
                xa, xb, ya, yb = clamp(x1, Ws), clamp(x2, Ws), clamp(y1, Hs), clamp(y2, Hs)
                bad = np.nonzero((xb <= xa) | (yb <= ya))[0]
                if len(bad):
                    i = int(bad[0])
                    raise ValueError(f"crop box [{y1[i]}:{y2[i]}, {x1[i]}:{x2[i]}] selects no pixel of a {Hs[i]}x{Ws[i]} "
                                     f"frame: {batched_video_ids}")
                moff = np.asarray(di.mask_off, dtype=np.int64)
                pair_table["frame_off"] = np.asarray(di.frame_off, dtype=np.int64)[fidx]
                pair_table["mask_a_off"], pair_table["mask_b_off"] = moff[pa], moff[pb]
                pair_table["frame_w"], pair_table["frame_h"] = Ws, Hs
                pair_table["x1"], pair_table["y1"], pair_table["x2"], pair_table["y2"] = xa, ya, xb, yb
                pair_table["mode"] = _lib.CROP_PAIR
            elif allow_dummy_pair:                          # default image when no pair exists (:416-420)
                selected.append((0, -1, (-1, -1)))
                new_select[0] = [(0, -1, (-1, -1))]
                H, W = di.frame_hw[0]
                pair_table = np.array([(di.frame_off[0], -1, -1, W, H, 0, 0, W, H, _lib.CROP_PLAIN, 0)], dtype=CROP_DTYPE)
            pair_start = np.cumsum([0] + [len(new_select[v]) for v in range(B)])
            n_pair = len(pair_table)
            pair_feat, join = None, (lambda: None)
            if n_pair:
                if side and n_pair >= 1024:
                    # queued from the point where the inputs were staged, not behind the object towers
                    s_p0, s_p1 = side[1], side[2]
                    P = self.cfg.n_patches
                    half = n_pair // 2
                    tower = self._packed()["clip_binary_model"][0]
                    s_p0.wait_event(inputs_ready)
                    with torch.cuda.stream(s_p0):
                        ppatches = self._crops(di, pair_table, alpha, white_alpha)
                        pair_feat = torch.empty((n_pair, self.cfg.proj), dtype=torch.float32, device=ppatches.device)
                        s_p1.wait_stream(s_p0)
                        tower.encode_images(ppatches[: half * P], half, out=pair_feat[:half], ws_slot=0)
                    with torch.cuda.stream(s_p1):
                        tower.encode_images(ppatches[half * P:], n_pair - half, out=pair_feat[half:], ws_slot=1)
                    ppatches.record_stream(s_p1)
                    pair_feat.record_stream(s_p1)
                    pair_feat.record_stream(main_stream)

                    def join(s_p0=s_p0, s_p1=s_p1):
                        main_stream.wait_stream(s_p0)
                        main_stream.wait_stream(s_p1)
                else:
                    ppatches = self._crops(di, pair_table, alpha, white_alpha)
                    pair_feat = self._encode_images("clip_binary_model", ppatches, n_pair)
            return pair_feat, new_select, pair_start, join

        # Without predicates the selected pairs are the given ones (:368-370) and do not wait for the category scores, so
        # the pair tower could be queued before the object towers (SGCLIP_PAIRS_FIRST=1); measured slower by 2 % (the
        # pair table is then built on the host before the first launch), so the reference's order stays the default.
        pairs_first = not need_ranking and os.environ.get("SGCLIP_PAIRS_FIRST", "0") != "0"
        pairs_queued = queue_pairs(None) if pairs_first else None

        # Step 3: object crops through the category and unary towers (:233-339)
        cate_feat = unary_feat = None
        if n_obj:
            patches = self._crops(di, obj_table, alpha, white_alpha)
            if side:
                s_un = side[0]
                s_un.wait_stream(main_stream)
                with torch.cuda.stream(s_un):
                    unary_feat = self._encode_images("clip_unary_model", patches, n_obj)
                patches.record_stream(s_un)
                cate_feat = self._encode_images("clip_cate_model", patches, n_obj)
                main_stream.wait_stream(s_un)
                unary_feat.record_stream(main_stream)
            else:
                cate_feat = self._encode_images("clip_cate_model", patches, n_obj)
                unary_feat = self._encode_images("clip_unary_model", patches, n_obj)
        sim_mode = _lib.SIM_LOGIT if output_logit else _lib.SIM_SOFTMAX
        if text_stream is not None:
            main_stream.wait_stream(text_stream)

        cate_out: Dict[int, object] = {v: {} for v in range(B)}
        unary_out: Dict[int, object] = {v: {} for v in range(B)}
        cate_feats_out: Dict[int, object] = {v: {} for v in range(B)}
        unary_feats_out: Dict[int, object] = {v: {} for v in range(B)}
        obj_per_cate_all: Dict[int, dict] = {v: {} for v in range(B)}
        # the reference zips this loop with batched_binary_predicates (:233-238)
        for vid in range(min(B, len(batched_binary_predicates))):
            s, e = int(obj_start[vid]), int(obj_start[vid + 1])
            object_ids = ids_lookup[vid]
            if e > s:
                cf, uf = cate_feat[s:e], unary_feat[s:e]
            else:
                cf = uf = torch.tensor([])
            cate_feats_out[vid], unary_feats_out[vid] = cf, uf
            names, ukws = batched_names[vid], batched_unary_kws[vid]
            if e > s:
                cate_dense = self._similarity("clip_cate_model", name_feats[vid], cf, sim_mode)
                unary_dense = self._similarity("clip_unary_model", unary_feats[vid], uf, sim_mode)
                for r, nm in enumerate(names):
                    if nm == dummy_str:
                        dummy_terms.append(cate_dense[r].sum())
                for r, nm in enumerate(ukws):
                    if nm == dummy_str:
                        dummy_terms.append(unary_dense[r].sum())
                # per-object mean over frames (:296-311); groups in order of first appearance
                oids, seg, seg_host = groups[vid]
                if grad_mode:
                    cate_mean = training.segment_mean(cate_dense, seg, len(oids), seg_host)
                else:
                    cate_mean = ops.segment_mean(cate_dense, seg, len(oids))
                # a name listed twice pools both rows' probabilities into one mean in the reference (:296-311)
                uniq: Dict[str, list] = {}
                for r, nm in enumerate(names):
                    if nm != dummy_str:
                        uniq.setdefault(nm, []).append(r)
                pooled_names = any(len(rs) > 1 for rs in uniq.values())

                def pool(mean, uniq=uniq, pooled_names=pooled_names):
                    return torch.stack([mean[rs].mean(dim=0) for rs in uniq.values()]) if pooled_names else mean
                cate_mean = pool(cate_mean)
                if pooled_names:
                    c_rows = list(enumerate(uniq.keys()))
                else:
                    c_rows = [(r, nm) for r, nm in enumerate(names) if nm != dummy_str]
                cate_out[vid] = ProbDict(cate_mean, c_rows, list(enumerate(oids)), lambda oid, nm: (oid, nm), "col_major")
                u_rows = [(r, nm) for r, nm in enumerate(ukws) if nm != dummy_str]
                unary_out[vid] = ProbDict(unary_dense, u_rows, list(enumerate(object_ids)),
                                          lambda fo, nm: (fo[0], fo[1], nm), "row_major")
                if need_ranking:
                    host = cate_mean.detach().cpu().numpy().astype(np.float64)
                    if self.rank_guard and len(oids) > topk_cate:
                        host = self._rank_guard(vid, host, c_rows, names, oids, seg_host, obj_table[s:e], di,
                                                batched_binary_predicates[vid], topk_cate, alpha, white_alpha,
                                                not output_logit, pool)
                    # per category: (prob, oid) sorted descending, ties by descending oid (:313-314)
                    obj_per_cate_all[vid] = {nm: sorted([(float(host[r, g]), oid) for g, oid in enumerate(oids)], reverse=True)
                                             for r, nm in c_rows}

        # Step 4: with predicates, the pairs among the best candidates of the names they mention (:342-392)
        pair_feat, new_select, pair_start, join_pairs = pairs_queued if pairs_queued is not None else queue_pairs(obj_per_cate_all)
        join_pairs()

        binary_out: List[object] = []
        pair_feats_out: Dict[int, object] = {v: torch.tensor([]) for v in range(B)}
        if output_logit:
            bmode = _lib.SIM_LOGIT
        else:
            bmode = _lib.SIM_SIGMOID if multi_class else _lib.SIM_SOFTMAX
        for vid in range(B):
            s, e = int(pair_start[vid]), int(pair_start[vid + 1])
            if e == s:
                binary_out.append({})
                continue
            pf = pair_feat[s:e]
            pair_feats_out[vid] = pf
            bkws = batched_binary_kws[vid]
            dense = self._similarity("clip_binary_model", binary_feats[vid], pf, bmode)
            cols = []
            for c, (_, fid, pr) in enumerate(new_select[vid]):
                if fid == -1:
                    for r, nm in enumerate(bkws):
                        if nm != dummy_str:
                            dummy_terms.append(dense[r, c])
                else:
                    cols.append((c, (fid, pr)))
            for r, nm in enumerate(bkws):
                if nm == dummy_str:
                    dummy_terms.append(dense[r].sum())
            b_rows = [(r, nm) for r, nm in enumerate(bkws) if nm != dummy_str]
            binary_out.append(ProbDict(dense, b_rows, cols, lambda fp, nm: (fp[0], fp[1], nm), "row_major"))

        dummy_prob = torch.stack(dummy_terms).sum() if dummy_terms else torch.zeros((), dtype=torch.float32, device=dev)   # a fill, not a blocking copy
        if output_embeddings:
            emb = {"cate_obj_clip_features": cate_feats_out, "cate_object_ids": ids_lookup,
                   "unary_obj_clip_features": unary_feats_out, "unary_object_ids": ids_lookup,
                   "binary_obj_pair_features": pair_feats_out, "binary_object_pairs": new_select}
            return cate_out, unary_out, binary_out, dummy_prob, emb
        return cate_out, unary_out, binary_out, dummy_prob

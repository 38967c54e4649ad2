"""Checkpoint ingest: the reference's checkpoint shapes -> one state dict with the keys
`clip_{cate,unary,binary}_model.<HF CLIPModel key>` that `PredicateModel.load_state_dict` takes.

The reference accepts five shapes (scripts/eval_sgclip.py:257-265,
scripts/train_sgclip.py:190-204): a state dict, a pickled `PredicateModel`, a pickled
`DistributedDataParallel` wrapper, and pickled V1 / V2 predecessors whose towers map onto the
three current ones as `load_from_v1` / `load_from_v2` do
(laser/models/llava_clip_model_v3.py:25-39).  Files are read with loaders that execute nothing
from them (`torch.load(weights_only=True)`, safetensors): a pickled module object therefore has
to go through the reference's own scripts/convert_model.py first, whose output (a plain state
dict, :52-73) loads here directly.  State dicts of every generation are handled by key layout:

  v3   clip_cate_model.* clip_unary_model.* clip_binary_model.*      taken as is
  v2   clip_unary_model.* clip_binary_model.*    cate <- v2 unary, binary <- v2 binary, unary <- base
  v1   clip_model.*                               cate, binary <- clip_model,         unary <- base
  hf   vision_model.* text_model.* ...            one CLIPModel: all three towers start from it
                                                  (what the constructor's from_pretrained x3 does, :15-19)

`base` is a single-CLIPModel state dict (the pretrained weights the reference leaves in the unary
tower for v1 / v2).  A `module.` prefix (state dict of a DDP wrapper) is stripped.
"""
from __future__ import annotations

import os
from typing import Dict, Mapping, Optional

import torch

from .weights import TOWER_PREFIXES


def _strip_module(sd: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        while k.startswith("module."):
            k = k[len("module."):]
        out[k] = v
    return out


def _sub(sd: Mapping[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    p = prefix + "."
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


def layout_of(sd: Mapping[str, torch.Tensor]) -> str:
    keys = list(sd.keys())
    has = lambda p: any(k.startswith(p + ".") for k in keys)
    if has("clip_cate_model"):
        return "v3"
    if has("clip_unary_model") and has("clip_binary_model"):
        return "v2"
    if has("clip_model"):
        return "v1"
    if has("vision_model") and has("text_model"):
        return "hf"
    raise ValueError("unrecognised checkpoint layout: expected clip_{cate,unary,binary}_model.*, clip_model.* "
                     "or a CLIPModel state dict (vision_model.*, text_model.*)")


def convert_state_dict(sd: Mapping[str, torch.Tensor], base: Optional[Mapping[str, torch.Tensor]] = None
                       ) -> Dict[str, torch.Tensor]:
    sd = _strip_module(sd)
    kind = layout_of(sd)
    if kind == "v3":
        towers = {p: _sub(sd, p) for p in TOWER_PREFIXES}
        missing = [p for p, t in towers.items() if not t]
        if missing:
            raise ValueError(f"checkpoint lacks the towers {missing}")
    elif kind == "hf":
        one = dict(sd)
        towers = {p: one for p in TOWER_PREFIXES}
    else:
        if base is None:
            raise ValueError(f"a {kind} checkpoint leaves the unary tower at the pretrained weights: pass them as `base`")
        base = _strip_module(base)
        if layout_of(base) != "hf":
            raise ValueError("`base` must be a single CLIPModel state dict")
        if kind == "v2":
            towers = {"clip_cate_model": _sub(sd, "clip_unary_model"), "clip_unary_model": dict(base),
                      "clip_binary_model": _sub(sd, "clip_binary_model")}
        else:
            v1 = _sub(sd, "clip_model")
            towers = {"clip_cate_model": v1, "clip_unary_model": dict(base), "clip_binary_model": v1}
    out = {}
    for p in TOWER_PREFIXES:
        for k, v in towers[p].items():
            if k.endswith("position_ids"):
                continue  # non-persistent buffers of older transformers versions
            out[f"{p}.{k}"] = v
    return out


def read_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """Tensors of a checkpoint file, read without executing anything from it."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:  # a pickled nn.Module (PredicateModel / DDP / V1 / V2 object)
        raise ValueError(f"{os.path.basename(path)} is not a plain state dict (pickled module objects are not "
                         f"unpickled here): convert it with the reference's scripts/convert_model.py first") from e
    if not isinstance(obj, dict) or not all(torch.is_tensor(v) for v in obj.values()):
        raise ValueError(f"{os.path.basename(path)} does not hold a state dict")
    return obj


def load_checkpoint(model, path: str, base_path: Optional[str] = None):
    """Read `path` (and the pretrained `base_path` for v1 / v2 files) and load it into `model`."""
    sd = convert_state_dict(read_state_dict(path), read_state_dict(base_path) if base_path else None)
    return model.load_state_dict(sd)

"""Checkpoint ingest: every state-dict layout of the reference maps onto the three towers as its
own loaders do (laser/models/llava_clip_model_v3.py:25-39, scripts/eval_sgclip.py:257-265)."""
import os

import pytest
import torch

from laser_amd import checkpoint
from laser_amd.config import TINY
from laser_amd.weights import TOWER_PREFIXES, predicate_model_state_dict, random_state_dict


def _eq(a, b):
    return a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)


def test_layouts():
    v3 = predicate_model_state_dict(TINY, 5)
    assert checkpoint.layout_of(v3) == "v3" and _eq(checkpoint.convert_state_dict(v3), v3)
    ddp = {"module." + k: v for k, v in v3.items()}
    assert _eq(checkpoint.convert_state_dict(ddp), v3)
    hf, base = random_state_dict(TINY, 7), random_state_dict(TINY, 8)
    out = checkpoint.convert_state_dict(hf)
    for p in TOWER_PREFIXES:
        assert all(torch.equal(out[f"{p}.{k}"], v) for k, v in hf.items())
    v1 = {"clip_model." + k: v for k, v in hf.items()}
    out = checkpoint.convert_state_dict(v1, base)
    assert all(torch.equal(out[f"clip_cate_model.{k}"], v) and torch.equal(out[f"clip_binary_model.{k}"], v)
               and torch.equal(out[f"clip_unary_model.{k}"], base[k]) for k, v in hf.items())
    other = random_state_dict(TINY, 9)
    v2 = {**{"clip_unary_model." + k: v for k, v in hf.items()}, **{"clip_binary_model." + k: v for k, v in other.items()}}
    out = checkpoint.convert_state_dict(v2, base)
    assert all(torch.equal(out[f"clip_cate_model.{k}"], hf[k]) and torch.equal(out[f"clip_binary_model.{k}"], other[k])
               and torch.equal(out[f"clip_unary_model.{k}"], base[k]) for k in hf)
    with pytest.raises(ValueError, match="base"):
        checkpoint.convert_state_dict(v1)
    with pytest.raises(ValueError, match="unrecognised"):
        checkpoint.convert_state_dict({"foo.weight": torch.zeros(1)})
    with_ids = dict(v3, **{"clip_cate_model.text_model.embeddings.position_ids": torch.arange(4)})
    assert _eq(checkpoint.convert_state_dict(with_ids), v3)


def test_files_are_read_without_unpickling_objects(tmp_path):
    v3 = predicate_model_state_dict(TINY, 5)
    p = os.path.join(tmp_path, "ensemble.3.model")
    torch.save(v3, p)
    assert _eq(checkpoint.read_state_dict(p), v3)
    from safetensors.torch import save_file
    q = os.path.join(tmp_path, "model.safetensors")
    save_file({k: v.contiguous() for k, v in random_state_dict(TINY, 7).items()}, q)
    assert checkpoint.layout_of(checkpoint.read_state_dict(q)) == "hf"
    obj = os.path.join(tmp_path, "module.1.model")
    torch.save(torch.nn.Linear(2, 2), obj)          # a pickled module object, as the reference's trainer writes
    with pytest.raises(ValueError, match="convert_model"):
        checkpoint.read_state_dict(obj)

"""Readers for the fixtures under tests/golden/ (written by make_golden.py)."""
from __future__ import annotations

import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _dec_key(k):
    if isinstance(k, list):
        return tuple(_dec_key(x) for x in k)
    return k


def load_forward_case(name: str):
    """-> (record, cate {vid:{key:val}}, unary {vid:{key:val}}, binary [ {key:val} ], dummy, arrays)."""
    index = json.load(open(os.path.join(GOLD, "forward_index.json")))
    rec = index[name]
    z = np.load(os.path.join(GOLD, f"forward_{name}.npz"))
    cate = {int(v): dict(zip([_dec_key(k) for k in keys], z[f"cate_vals_{v}"].tolist())) for v, keys in rec["cate_keys"].items()}
    unary = {int(v): dict(zip([_dec_key(k) for k in keys], z[f"unary_vals_{v}"].tolist())) for v, keys in rec["unary_keys"].items()}
    binary = [dict(zip([_dec_key(k) for k in keys], z[f"binary_vals_{i}"].tolist())) for i, keys in enumerate(rec["binary_keys"])]
    return rec, cate, unary, binary, float(z["dummy_prob"]), z


def load_functions():
    meta = json.load(open(os.path.join(GOLD, "functions.json")))
    z = np.load(os.path.join(GOLD, "functions.npz"))
    H, W = meta["H"], meta["W"]
    masks = []
    i = 0
    while f"mask_{i}" in z.files:
        masks.append(np.unpackbits(z[f"mask_{i}"])[: H * W].reshape(H, W, 1).astype(bool))
        i += 1
    return meta, z, masks


def object_box(meta_obj):
    """Rebuild the bbox argument exactly as make_golden.py passed it."""
    fmt, box = meta_obj["fmt"], meta_obj["box"]
    if fmt == "dict":
        return box
    if fmt == "list":
        return [int(x) for x in box]
    if fmt == "tuple_float":
        return tuple(box)
    return np.array(box)


def compare_prob_dicts(got, want, atol):
    """Same keys in the same order; values within atol.  Returns max abs diff."""
    assert list(got.keys()) == list(want.keys()), "dictionary keys/order differ"
    if not want:
        return 0.0
    g = np.array([float(got[k]) for k in want], dtype=np.float64)
    w = np.array([float(want[k]) for k in want], dtype=np.float64)
    d = float(np.abs(g - w).max())
    assert d <= atol, f"max abs diff {d} > {atol}"
    return d

"""Fact adapter, evaluation metrics and packed masks: CPU tests against the goldens produced by
the reference's own compute_metrics / bitmasks2bboxes (tests/golden/make_consumer_golden.py)."""
import json
import os

import numpy as np
import torch

from laser_amd import facts, masks
from oracle import facts_ref, masks_ref

HERE = os.path.dirname(os.path.abspath(__file__))


def _load():
    meta = json.load(open(os.path.join(HERE, "golden", "consumers.json")))
    z = np.load(os.path.join(HERE, "golden", "consumers.npz"))
    return meta, z


def metric_inputs(case, z, ci):
    names, rels, oids = case["names"], case["rels"], case["oids"]
    cols = [(f, tuple(p)) for f, p in case["cols"]]
    cate, binary = z[f"m{ci}_cate"], z[f"m{ci}_binary"]
    gt_objects = [tuple(g) for g in case["gt_objects"]]
    gt_rels = [[tuple(t) for t in ls] for ls in case["gt_rels"]]
    return names, rels, oids, cols, cate, binary, gt_objects, gt_rels


def as_expected(res):
    return {k: {kk: {str(t): v for t, v in vv.items()} for kk, vv in kv.items()} for k, kv in res.items()}


def test_compute_metrics_matches_reference_outputs():
    meta, z = _load()
    for ci, case in enumerate(meta["metrics"]):
        names, rels, oids, cols, cate, binary, gt_objects, gt_rels = metric_inputs(case, z, ci)
        cate_pred = {(oid, nm): torch.tensor(cate[r, g]) for g, oid in enumerate(oids) for r, nm in enumerate(names)}
        binary_pred = {(fid, pair, nm): torch.tensor(binary[r, c]) for r, nm in enumerate(rels)
                       for c, (fid, pair) in enumerate(cols)}
        for fn in (facts_ref.compute_metrics, facts.compute_metrics):   # oracle, and the product's dictionary route
            res = fn(gt_objects, cate_pred, gt_rels, binary_pred, precision_thres_ls=case["thres"],
                     recall_thres_ls=case["thres"])
            assert as_expected(res) == case["expected"], (ci, fn.__module__)


def test_bitmasks2bboxes_oracle_matches_reference_outputs():
    meta, z = _load()
    for bi, want in enumerate(meta["boxes"]):
        ms = [m[:, :, None] for m in z[f"b{bi}_masks"]]
        assert masks_ref.bitmasks2bboxes(ms) == want
    empty = [np.zeros((8, 8, 1), dtype=bool), np.ones((8, 8, 1), dtype=bool)]
    assert masks_ref.bitmasks2bboxes(empty) == meta["boxes_with_empty"]


def test_pack_masks_round_trip():
    rng = np.random.default_rng(3)
    ms = [rng.random((17, 29)) < 0.3, (rng.random((8, 64, 1)) < 0.5), rng.random((1, 5, 7)) < 0.5,
          np.zeros((4, 9), dtype=np.float32)]
    pk = masks.pack_masks(ms)
    assert len(pk) == 4 and pk.nbytes == 17 * 4 + 8 * 8 + 5 * 1 + 4 * 2
    assert np.array_equal(pk.unpack_host(0), ms[0])
    assert np.array_equal(pk.unpack_host(1), ms[1])
    assert np.array_equal(pk.unpack_host(2)[:, :, 0], ms[2][0])
    assert not pk.unpack_host(3).any()


def test_fact_lists_from_plain_dicts():
    cate = {(1, "cat"): torch.tensor(0.7), (1, "dog"): torch.tensor(0.3), (2, "cat"): torch.tensor(0.1)}
    unary = {(0, 1, "red"): torch.tensor(0.9), (1, 2, "tall"): torch.tensor(0.2)}
    binary = {(0, (1, 2), "on"): torch.tensor(0.6)}
    lookup = {"cat": -0, "CAT": -0, "dog": -1}
    want = facts_ref.scl_fact_lists(cate, unary, binary, lookup)
    got = facts.scl_facts({0: cate}, {0: unary}, [binary], [lookup])
    for w, g in zip(want, got):
        assert [t for _, t in w] == g[0].tuples
        assert [float(p) for p, _ in w] == [float(p) for p, _ in g[0].pairs()]
    assert facts.scl_facts({0: cate}, {0: unary}, [{}], [lookup])[2][0].tuples == []

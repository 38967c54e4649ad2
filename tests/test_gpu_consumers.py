"""GPU tests of the formats either side of forward(): the top-k / metrics route over the dense
tensors, the fact adapter, and bit-packed masks (unpack + boxes kernel), all through the C ABI."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cases import FORWARD_CASES, build_case_batch  # noqa: E402
from test_consumers import _load, as_expected, metric_inputs  # noqa: E402
from test_gpu_forward import _model  # noqa: E402

from laser_amd import facts, masks, ops  # noqa: E402
from laser_amd.prob_dict import ProbDict  # noqa: E402
from oracle import facts_ref, masks_ref  # noqa: E402

DEV = "cuda:0"


def test_topk_columns_order_matches_python_sorted():
    rng = np.random.default_rng(0)
    K, N, k = 37, 501, 10
    p = rng.random((K, N)).astype(np.float32)
    p[:, ::3] = np.round(p[:, ::3] * 3) / 3            # many exact ties
    names = [f"kw {rng.integers(1000):03d} {i}" for i in range(K)]
    order = sorted(range(K), key=lambda i: names[i])
    rank = np.empty(K, dtype=np.int32); rank[order] = np.arange(K)
    idx, val = ops.topk_columns(torch.from_numpy(p).to(DEV), k, torch.from_numpy(rank))
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    for n in range(N):
        want = sorted([(float(p[r, n]), names[r], r) for r in range(K)], reverse=True)[:k]
        assert [w[2] for w in want] == idx[:, n].tolist()
        assert [np.float32(w[0]) for w in want] == val[:, n].tolist()
    # fewer rows than k: the tail is -1 / 0; no tie ranks: lower row first
    q = torch.tensor([[0.5, 0.25], [0.5, 0.75]], device=DEV)
    idx, val = ops.topk_columns(q, 4)
    assert idx.cpu().tolist() == [[0, 1], [1, 0], [-1, -1], [-1, -1]]
    assert val.cpu().tolist() == [[0.5, 0.75], [0.5, 0.25], [0.0, 0.0], [0.0, 0.0]]


def test_compute_metrics_dense_route_matches_reference_outputs():
    meta, z = _load()
    for ci, case in enumerate(meta["metrics"]):
        names, rels, oids, cols, cate, binary, gt_objects, gt_rels = metric_inputs(case, z, ci)
        cate_pd = ProbDict(torch.from_numpy(cate).to(DEV), list(enumerate(names)), list(enumerate(oids)),
                           lambda oid, nm: (oid, nm), "col_major")
        bin_pd = ProbDict(torch.from_numpy(binary).to(DEV), list(enumerate(rels)), list(enumerate(cols)),
                          lambda fp, nm: (fp[0], fp[1], nm), "row_major")
        res = facts.compute_metrics(gt_objects, cate_pd, gt_rels, bin_pd, precision_thres_ls=case["thres"],
                                    recall_thres_ls=case["thres"])
        assert as_expected(res) == case["expected"], ci


def test_mask_unpack_and_boxes_match_reference_outputs():
    meta, z = _load()
    for bi, want in enumerate(meta["boxes"]):
        ms = [m[:, :, None] for m in z[f"b{bi}_masks"]]
        un = masks.unpack_to_device(masks.pack_masks(ms), DEV)
        flat = un.masks.cpu().numpy()
        for i, m in enumerate(ms):
            H, W = m.shape[:2]
            assert np.array_equal(flat[un.offsets[i]: un.offsets[i] + H * W].reshape(H, W), m[:, :, 0].astype(np.uint8))
        assert un.bbox_dicts() == want
        assert un.boxes[:, 4].tolist() == [masks_ref.mask_size(m) for m in ms]
    # mixed sizes in one call, widths that are and are not multiples of 8, an empty mask
    rng = np.random.default_rng(9)
    ms = [rng.random((33, 40)) < 0.2, rng.random((720, 1280, 1)) < 0.01, np.zeros((9, 13), dtype=bool),
          rng.random((5, 1)) < 0.9, rng.random((64, 72)) < 0.5]
    un = masks.unpack_to_device(masks.pack_masks(ms), DEV)
    flat = un.masks.cpu().numpy()
    dicts = un.bbox_dicts()
    for i, m in enumerate(ms):
        m2 = np.asarray(m).reshape(m.shape[0], m.shape[1])
        H, W = m2.shape
        assert np.array_equal(flat[un.offsets[i]: un.offsets[i] + H * W].reshape(H, W), m2.astype(np.uint8)), i
        ref = masks_ref.bitmasks2bboxes([m2[:, :, None]])
        assert dicts[i] == (ref[0] if ref else None), i
        assert int(un.boxes[i, 4]) == masks_ref.mask_size(m2)


def test_forward_from_packed_masks_equals_forward_from_byte_masks():
    name = "tiny_infer"
    cfg_name, wseed, _, fk = FORWARD_CASES[name]
    model = _model(cfg_name, wseed)
    kw = build_case_batch(name).forward_kwargs()
    want = model(**kw, **fk)
    # boxes in the reference's own convention (inclusive maxima) from its host function's restatement
    ref_boxes = masks_ref.bitmasks2bboxes([np.asarray(m).reshape(m.shape[0], m.shape[1], 1) for m in kw["batched_masks"]])
    kw_ref = dict(kw, batched_bboxes=ref_boxes)
    want_ref = model(**kw_ref, **fk)
    packed = masks.pack_masks(kw["batched_masks"])
    assert packed.nbytes * 7 < sum(np.asarray(m).size for m in kw["batched_masks"])
    di = model.stage_inputs(kw["batched_videos"], packed)
    assert di.bbox_dicts() == ref_boxes
    got_ref = model(**dict(kw, batched_masks=packed, batched_bboxes=di.bbox_dicts()), device_inputs=di, **fk)
    got = model(**dict(kw, batched_masks=packed), **fk)
    got_none = model(**dict(kw, batched_masks=packed, batched_bboxes=None), **fk)   # boxes derived on the GPU
    torch.cuda.synchronize()
    for a, b in ((want, got), (want_ref, got_ref), (want_ref, got_none)):
        for v in a[0]:
            assert list(a[0][v].keys()) == list(b[0][v].keys())
            assert torch.equal(a[0][v].values_tensor(), b[0][v].values_tensor())
            assert torch.equal(a[1][v].values_tensor(), b[1][v].values_tensor())
        for x, y in zip(a[2], b[2]):
            assert list(x.keys()) == list(y.keys()) and torch.equal(x.values_tensor(), y.values_tensor())


def test_fact_tables_equal_the_entry_by_entry_lists():
    name = "tiny_train"
    cfg_name, wseed, _, fk = FORWARD_CASES[name]
    model = _model(cfg_name, wseed)
    batch = build_case_batch(name)
    cate, unary, binary, _ = model(**batch.forward_kwargs(), **fk)
    lookups = []
    for vid in cate:
        names = sorted({nm for (_, nm) in cate[vid].keys()})
        lookups.append({nm: -k for k, nm in enumerate(names[::2])})     # every other name is a constant
    got = facts.scl_facts(cate, unary, binary, lookups)
    for vid in cate:
        want = facts_ref.scl_fact_lists(cate[vid].to_dict(), unary[vid].to_dict(), binary[vid].to_dict() if len(binary[vid]) else {},
                                        lookups[vid])
        for w, g in zip(want, (got[0][vid], got[1][vid], got[2][vid])):
            assert [t for _, t in w] == g.tuples
            assert torch.equal(torch.stack([p for p, _ in w]) if w else g.probs[:0], g.probs)
            assert len(g.pairs()) == len(w)


def test_load_checkpoint_file(tmp_path):
    from laser_amd import checkpoint
    from laser_amd.config import config_by_name
    from laser_amd.weights import predicate_model_state_dict
    name = "tiny_infer"
    cfg_name, wseed, _, fk = FORWARD_CASES[name]
    want_model = _model(cfg_name, wseed)
    sd = predicate_model_state_dict(config_by_name(cfg_name), wseed, float(np.log(100.0)))
    path = os.path.join(tmp_path, "ensemble.7.model")
    torch.save({"module." + k: v for k, v in sd.items()}, path)       # state dict of a DDP wrapper
    model = _model(cfg_name, wseed + 1)
    checkpoint.load_checkpoint(model, path)
    kw = build_case_batch(name).forward_kwargs()
    a, b = want_model(**kw, **fk), model(**kw, **fk)
    for v in a[0]:
        assert torch.equal(a[0][v].values_tensor(), b[0][v].values_tensor())
        assert torch.equal(a[1][v].values_tensor(), b[1][v].values_tensor())


@pytest.mark.parametrize("name,packed", [("tiny_infer", False), ("tiny_infer", True), ("tiny_no_pairs", False)])
def test_frame_sharded_forward_equals_the_whole_call(name, packed):
    """Two ranks' shards scored one after the other on this GPU, merged, against the unsharded call."""
    from laser_amd.distributed import FrameShardPlan, local_blocks, merge_frame_shards
    cfg_name, wseed, _, fk = FORWARD_CASES[name]
    model = _model(cfg_name, wseed)
    kw = dict(build_case_batch(name).forward_kwargs(), **fk)
    if packed:
        kw["batched_masks"] = masks.pack_masks(kw["batched_masks"])
    want = model(**kw)
    for world in (2, 3):
        plan = FrameShardPlan(kw["batched_object_ids"], kw["batched_obj_pairs"], world)
        assert sum(1 for r in range(world) if plan.obj_pos[r]) >= 2
        blocks = []
        for r in range(world):
            out = model(**plan.local_kwargs(kw, r)) if plan.obj_pos[r] else None
            blocks.append(local_blocks(out, kw, model.device))
        got = merge_frame_shards(blocks, plan, kw)
        torch.cuda.synchronize()
        for part in (0, 1):
            for v in want[part]:
                assert list(want[part][v].keys()) == list(got[part][v].keys())
                if len(want[part][v]):
                    assert torch.allclose(want[part][v].values_tensor(), got[part][v].values_tensor(), atol=2e-3)
        assert len(want[2]) == len(got[2])
        for x, y in zip(want[2], got[2]):
            assert list(x.keys()) == list(y.keys())
            if len(x):
                assert torch.allclose(x.values_tensor(), y.values_tensor(), atol=2e-3)
        assert abs(float(want[3]) - float(got[3])) < 2e-3


def test_forward_frame_sharded_single_rank_process_group():
    import torch.distributed as dist
    from laser_amd.distributed import forward_frame_sharded
    name = "tiny_infer"
    cfg_name, wseed, _, fk = FORWARD_CASES[name]
    model = _model(cfg_name, wseed)
    kw = dict(build_case_batch(name).forward_kwargs(), **fk)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        want, got = model(**kw), forward_frame_sharded(model, **kw)
        for v in want[0]:
            assert torch.allclose(want[0][v].values_tensor(), got[0][v].values_tensor(), atol=1e-5)
            assert torch.equal(want[1][v].values_tensor(), got[1][v].values_tensor())
        for x, y in zip(want[2], got[2]):
            assert list(x.keys()) == list(y.keys()) and torch.equal(x.values_tensor(), y.values_tensor())
    finally:
        dist.destroy_process_group()

"""GPU test at the size of the pair-explosion stress configuration: PredicateModel.forward() with the
ViT-L/14 geometry at 336 x 336 crops on one 1280 x 720 frame holding 64 objects and all 4 032 ordered
pairs (4 096 crops, 4 160 image encodes of 577 tokens), 35 + 200 + 50 keywords.

The CPU oracle needs about a second per L/14 encode, so the whole frame is checked through properties
that do not depend on its size (every softmax column sums to one, the call is deterministic, a crop's
probabilities do not depend on which other crops share the call) and a subset of its objects and
pairs is checked against the fp32 CPU restatement of the reference: probabilities within 2e-2
max-abs, top-1 predicate agreement >= 99 %."""
import numpy as np
import pytest
import torch

from laser_amd.config import config_by_name
from laser_amd.predicate_model import PredicateModel
from laser_amd.synthetic import SyntheticClipBatch, make_clip_batch
from laser_amd.tokenizer import HashTokenizer
from laser_amd.weights import TOWER_PREFIXES, split_predicate_state_dict

pytestmark = pytest.mark.gpu

OBJECTS = 64
SUBSET = (3, 17, 22, 38, 41, 55, 60, 63, 9, 30, 47)      # objects scored alone and by the oracle


def _object_subset(b: SyntheticClipBatch, oids) -> SyntheticClipBatch:
    """The same frame with only the objects `oids` and the ordered pairs among them (given order kept)."""
    keep = [i for i, (_, _, o) in enumerate(b.batched_object_ids) if o in oids]
    return SyntheticClipBatch(
        batched_video_ids=b.batched_video_ids, batched_videos=b.batched_videos,
        batched_masks=[b.batched_masks[i] for i in keep], batched_bboxes=[b.batched_bboxes[i] for i in keep],
        batched_names=b.batched_names, batched_object_ids=[b.batched_object_ids[i] for i in keep],
        batched_unary_kws=b.batched_unary_kws, batched_binary_kws=b.batched_binary_kws,
        batched_obj_pairs=[p for p in b.batched_obj_pairs if p[2][0] in oids and p[2][1] in oids],
        batched_video_splits=b.batched_video_splits, batched_binary_predicates=[None])


@pytest.fixture(scope="module")
def c4():
    cfg = config_by_name("vit-l-14-336")
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    model = PredicateModel(None, None, "cuda:0", "vit-l-14-336", tokenizer=tok, init_seed=13).eval()
    for p in TOWER_PREFIXES:
        model._modules[p].logit_scale.data.fill_(float(np.log(100.0)))
    model.invalidate_packed_weights()
    batch = make_clip_batch(43, n_videos=1, frames=1, objects=OBJECTS)
    assert len(batch.batched_obj_pairs) == OBJECTS * (OBJECTS - 1) == 4032
    out = model(**batch.forward_kwargs())
    torch.cuda.synchronize()
    yield cfg, tok, model, batch, out
    del model
    torch.cuda.empty_cache()


def test_c4_shapes_keys_and_softmax_columns(c4):
    cfg, tok, model, batch, (cate, unary, binary, dummy) = c4
    n_pair = OBJECTS * (OBJECTS - 1)
    assert len(cate[0]) == 35 * OBJECTS and len(unary[0]) == 200 * OBJECTS and len(binary[0]) == 50 * n_pair
    u, b, c = unary[0].dense, binary[0].dense, cate[0].dense
    assert u.shape == (200, OBJECTS) and b.shape == (50, n_pair) and c.shape == (35, OBJECTS)
    for t in (u, b, c):
        assert torch.isfinite(t).all() and (t >= 0).all()
        assert (t.sum(dim=0) - 1).abs().max().item() < 1e-4
    assert float(dummy) == 0.0
    kws = batch.batched_binary_kws[0]
    keys = list(binary[0].keys())
    assert keys[:3] == [(f, pr, kws[0]) for (_, f, pr) in batch.batched_obj_pairs[:3]]
    assert keys[-1] == (0, (OBJECTS - 1, OBJECTS - 2), kws[-1])


def test_c4_is_deterministic(c4):
    cfg, tok, model, batch, (cate, unary, binary, _) = c4
    c2, u2, b2, _ = model(**batch.forward_kwargs())
    torch.cuda.synchronize()
    assert torch.equal(u2[0].dense, unary[0].dense)
    assert torch.equal(b2[0].dense, binary[0].dense)
    assert torch.equal(c2[0].dense, cate[0].dense)


def test_c4_subset_alone_and_against_the_oracle(c4):
    """11 objects and the 110 ordered pairs among them: scored alone (small-grid kernels) they must reproduce their
    columns of the 4 096-crop call (bulk kernels), and both must match the fp32 oracle."""
    from oracle.forward_ref import OraclePredicateModel
    cfg, tok, model, batch, (cate, unary, binary, _) = c4
    oids = set(SUBSET)
    sub = _object_subset(batch, oids)
    c1, u1, b1, _ = model(**sub.forward_kwargs())
    torch.cuda.synchronize()
    ocols = [i for i, (_, _, o) in enumerate(batch.batched_object_ids) if o in oids]
    pcols = [i for i, p in enumerate(batch.batched_obj_pairs) if p[2][0] in oids and p[2][1] in oids]
    assert len(ocols) == len(SUBSET) and len(pcols) == len(SUBSET) * (len(SUBSET) - 1)
    full_u, full_b, full_c = unary[0].dense[:, ocols], binary[0].dense[:, pcols], cate[0].dense[:, ocols]
    # the same crops through differently tiled kernels: bf16 rounding of differently ordered sums only
    assert (u1[0].dense - full_u).abs().max().item() < 2e-3
    assert (b1[0].dense - full_b).abs().max().item() < 2e-3
    assert (c1[0].dense - full_c).abs().max().item() < 2e-3

    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.set_num_threads(16)
    want = OraclePredicateModel(cfg, split_predicate_state_dict(sd), tok)(**sub.forward_kwargs())
    assert list(u1[0].keys()) == list(want[1][0].keys())
    assert list(b1[0].keys()) == list(want[2][0].keys())
    assert list(c1[0].keys()) == list(want[0][0].keys())
    worst, agree, total = 0.0, 0, 0
    for got_dense, w, K in ((full_u, want[1][0], 200), (full_b, want[2][0], 50)):
        wv = np.array([float(x) for x in w.values()]).reshape(K, -1)      # name-major key order
        gv = got_dense.cpu().numpy()
        worst = max(worst, float(np.abs(gv - wv).max()))
        same = gv.argmax(0) == wv.argmax(0)
        agree += int(same.sum())
        total += gv.shape[1]
        for c in np.where(~same)[0]:
            top2 = np.sort(wv[:, c])[-2:]
            print(f"top-1 flip in column {c}: oracle top-2 probabilities {top2[1]:.4f} / {top2[0]:.4f}")
    cw = np.array([float(x) for x in want[0][0].values()])
    worst = max(worst, float(np.abs(c1[0].values_tensor().cpu().numpy() - cw).max()))
    assert worst <= 2e-2, worst
    assert agree / total >= 0.99, (agree, total)
    print(f"C4 L/14-336, {total} sampled crops: max |p - oracle| = {worst:.3e}, top-1 agreement {agree}/{total}")

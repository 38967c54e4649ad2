"""GPU tests at the full size of the headline workload: one synthetic clip of 32 frames x 16 objects
x 240 ordered pairs per frame at 1280 x 720, ViT-B/32, 35 + 200 + 50 keywords (8 192 crops).

The CPU oracle needs minutes for the whole clip, so the full-size run is checked through properties
that do not depend on the size -- every softmax column sums to one, the call is deterministic, a
crop's probabilities do not depend on which other crops share the call, the category mean is the
mean of the per-frame category probabilities -- and against the oracle on ONE frame of the clip
(16 object crops + 240 pair crops), scored alone by the fp32 CPU restatement of the reference."""
import numpy as np
import pytest
import torch

from laser_amd import _lib, ops
from laser_amd.config import config_by_name
from laser_amd.predicate_model import PredicateModel
from laser_amd.synthetic import SyntheticClipBatch, make_clip_batch
from laser_amd.tokenizer import HashTokenizer
from laser_amd.weights import predicate_model_state_dict, split_predicate_state_dict

pytestmark = pytest.mark.gpu

FRAMES, OBJECTS = 32, 16


def _frame_subset(b: SyntheticClipBatch, fid: int) -> SyntheticClipBatch:
    """The same clip cut down to one frame (frame index 0 in the new batch)."""
    keep = [i for i, (_, f, _) in enumerate(b.batched_object_ids) if f == fid]
    return SyntheticClipBatch(
        batched_video_ids=b.batched_video_ids, batched_videos=[b.batched_videos[fid]],
        batched_masks=[b.batched_masks[i] for i in keep], batched_bboxes=[b.batched_bboxes[i] for i in keep],
        batched_names=b.batched_names, batched_object_ids=[(0, 0, b.batched_object_ids[i][2]) for i in keep],
        batched_unary_kws=b.batched_unary_kws, batched_binary_kws=b.batched_binary_kws,
        batched_obj_pairs=[(0, 0, pr) for (_, f, pr) in b.batched_obj_pairs if f == fid],
        batched_video_splits=[1], batched_binary_predicates=[None])


@pytest.fixture(scope="module")
def full_clip():
    cfg = config_by_name("vit-b-32")
    sd = predicate_model_state_dict(cfg, 11, float(np.log(100.0)))
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    model = PredicateModel(None, None, "cuda:0", "vit-b-32", tokenizer=tok).eval()
    model.load_state_dict(sd)
    batch = make_clip_batch(41, n_videos=1, frames=FRAMES, objects=OBJECTS, distinct_frames=4)
    out = model(**batch.forward_kwargs())
    torch.cuda.synchronize()
    return cfg, sd, tok, model, batch, out


def test_full_clip_shapes_and_softmax_columns(full_clip):
    cfg, sd, tok, model, batch, (cate, unary, binary, dummy) = full_clip
    n_obj, n_pair = FRAMES * OBJECTS, FRAMES * OBJECTS * (OBJECTS - 1)
    assert len(unary[0]) == 200 * n_obj and len(binary[0]) == 50 * n_pair and len(cate[0]) == 35 * OBJECTS
    u, b, c = unary[0].dense, binary[0].dense, cate[0].dense
    assert u.shape == (200, n_obj) and b.shape == (50, n_pair) and c.shape == (35, OBJECTS)
    for t in (u, b, c):       # softmax over the keyword axis per crop; the mean of such columns also sums to one
        assert torch.isfinite(t).all() and (t >= 0).all()
        assert (t.sum(dim=0) - 1).abs().max().item() < 1e-4
    assert float(dummy) == 0.0
    # keys in the order the reference builds them: name-major, crops in the order given
    first = list(binary[0].keys())[:3]
    assert first == [(f, pr, batch.batched_binary_kws[0][0]) for (_, f, pr) in batch.batched_obj_pairs[:3]]


def test_full_clip_is_deterministic(full_clip):
    cfg, sd, tok, model, batch, (cate, unary, binary, _) = full_clip
    c2, u2, b2, _ = model(**batch.forward_kwargs())
    torch.cuda.synchronize()
    assert torch.equal(u2[0].dense, unary[0].dense)
    assert torch.equal(b2[0].dense, binary[0].dense)
    assert torch.equal(c2[0].dense, cate[0].dense)


@pytest.mark.parametrize("fid", [0, 19, 31])
def test_a_frame_scored_alone_matches_its_columns_of_the_full_clip(full_clip, fid):
    """Crop independence: the bulk kernels of the full clip (256 x 256 GEMM tiles, fused q/k/v + attention, concurrent
    towers) and the small-grid kernels of a 272-crop call must give the same probabilities for the same crop."""
    cfg, sd, tok, model, batch, (cate, unary, binary, _) = full_clip
    sub = _frame_subset(batch, fid)
    c1, u1, b1, _ = model(**sub.forward_kwargs())
    torch.cuda.synchronize()
    ucols = slice(fid * OBJECTS, (fid + 1) * OBJECTS)
    P = OBJECTS * (OBJECTS - 1)
    bcols = slice(fid * P, (fid + 1) * P)
    assert (u1[0].dense - unary[0].dense[:, ucols]).abs().max().item() < 1e-5
    assert (b1[0].dense - binary[0].dense[:, bcols]).abs().max().item() < 1e-5


def test_category_mean_is_the_mean_over_frames(full_clip):
    """(oid, name) probabilities are the per-object mean over the frames of the per-crop category softmax
    (llava_clip_model_v3.py:303-311): recombine it from the 32 one-frame calls' means."""
    cfg, sd, tok, model, batch, (cate, unary, binary, _) = full_clip
    acc = torch.zeros_like(cate[0].dense)
    for fid in range(FRAMES):
        c1 = model(**_frame_subset(batch, fid).forward_kwargs())[0]
        acc += c1[0].dense          # one frame: the "mean" over a single crop per object
    torch.cuda.synchronize()
    assert (acc / FRAMES - cate[0].dense).abs().max().item() < 1e-5


def test_one_frame_of_the_full_clip_against_the_oracle(full_clip):
    from oracle.forward_ref import OraclePredicateModel
    cfg, sd, tok, model, batch, (cate, unary, binary, _) = full_clip
    fid = 7
    sub = _frame_subset(batch, fid)
    torch.set_num_threads(16)
    want = OraclePredicateModel(cfg, split_predicate_state_dict(sd), tok)(**sub.forward_kwargs())
    P = OBJECTS * (OBJECTS - 1)
    worst, agree, total = 0.0, 0, 0
    for got_dense, w, cols, K in ((unary[0].dense[:, fid * OBJECTS:(fid + 1) * OBJECTS], want[1][0], OBJECTS, 200),
                                  (binary[0].dense[:, fid * P:(fid + 1) * P], want[2][0], P, 50)):
        wv = np.array([float(x) for x in w.values()]).reshape(K, cols)      # name-major key order
        gv = got_dense.cpu().numpy()
        worst = max(worst, float(np.abs(gv - wv).max()))
        agree += int((gv.argmax(0) == wv.argmax(0)).sum())
        total += cols
    assert worst <= 2e-2, worst                     # fp32 probabilities within 2e-2 max-abs under bf16
    assert agree / total >= 0.99, (agree, total)    # >= 99 % top-1 predicate agreement
    print(f"full-size clip, frame {fid}: max |p - oracle| = {worst:.3e}, top-1 agreement {agree}/{total}")

"""GPU parity tests of the full scoring path: laser_amd.PredicateModel.forward
against the golden outputs of the reference's forward() (tests/golden) and
against the CPU oracle on the same seeded inputs.

Tolerances (BASELINE.json north_star): crop boxes / pair indices / key order
bit-exact; probabilities within 2e-2 max-abs under bf16; >= 99 % top-1 agreement."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cases import FORWARD_CASES, build_case_batch  # noqa: E402
from golden_io import compare_prob_dicts, load_forward_case  # noqa: E402

from laser_amd.config import config_by_name  # noqa: E402
from laser_amd.predicate_model import PredicateModel  # noqa: E402
from laser_amd.tokenizer import HashTokenizer  # noqa: E402
from laser_amd.weights import predicate_model_state_dict  # noqa: E402

PROB_TOL = 2e-2


def _model(cfg_name, wseed):
    cfg = config_by_name(cfg_name)
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    m = PredicateModel(None, None, "cuda:0", cfg_name, tokenizer=tok).eval()
    m.load_state_dict(predicate_model_state_dict(cfg, wseed, float(np.log(100.0))))
    return m


def _top1_agreement(got_dense_cols, want_dense_cols):
    return float((got_dense_cols.argmax(0) == want_dense_cols.argmax(0)).float().mean())


@pytest.mark.parametrize("name", list(FORWARD_CASES))
def test_forward_matches_reference_golden(name):
    cfg_name, wseed, _, fk = FORWARD_CASES[name]
    rec, cate, unary, binary, dummy, z = load_forward_case(name)
    model = _model(cfg_name, wseed)
    out = model(**build_case_batch(name).forward_kwargs(), **fk)
    torch.cuda.synchronize()
    tol = 0.25 if fk.get("output_logit") else PROB_TOL     # logits are scaled by 100
    assert set(out[0].keys()) == set(cate.keys())
    worst = 0.0
    for v in cate:
        worst = max(worst, compare_prob_dicts(out[0][v], cate[v], tol))
        worst = max(worst, compare_prob_dicts(out[1][v], unary[v], tol))
    assert len(out[2]) == len(binary)
    for g, w in zip(out[2], binary):
        worst = max(worst, compare_prob_dicts(g, w, tol))
    assert abs(float(out[3]) - dummy) <= max(tol, 1e-3 * abs(dummy))
    if fk.get("output_embeddings"):
        emb = out[4]
        for v in range(len(binary)):
            assert [tuple(x) for x in rec["cate_object_ids"][str(v)]] == list(emb["cate_object_ids"][v])
            # the selected pairs and their order (training mode: list(set(...)) of the reference) are bit-exact
            assert [(p[0], p[1], tuple(p[2])) for p in rec["binary_object_pairs"][str(v)]] == list(emb["binary_object_pairs"][v])
            for nm in ("cate_obj_clip_features", "unary_obj_clip_features", "binary_obj_pair_features"):
                got, want = emb[nm][v].float().cpu().numpy(), z[f"{nm}_{v}"]
                cos = (got * want).sum(-1) / (np.linalg.norm(got, axis=-1) * np.linalg.norm(want, axis=-1))
                assert cos.min() > 0.999, (nm, cos.min())
    print(f"{name}: max |p - reference| = {worst:.3e}")


def test_c1_demo_clip_top1_agreement_with_the_reference():
    """BASELINE.json configs[0] whole: 8 frames x <= 5 masks at 1280 x 720, ViT-B/32, the evaluation driver's call
    (127 category names, no unary keywords, 57 relations), against the reference's own outputs for the same inputs:
    top-1 relation of every pair and top-1 category of every object."""
    name = "c1_demo_clip"
    cfg_name, wseed, _, fk = FORWARD_CASES[name]
    rec, cate, unary, binary, dummy, z = load_forward_case(name)
    model = _model(cfg_name, wseed)
    out = model(**build_case_batch(name).forward_kwargs(), **fk)
    torch.cuda.synchronize()
    agree = total = 0
    for got, want, order in ((out[2][0], binary[0], "row"), (out[0][0], cate[0], "col")):
        assert list(got.keys()) == list(want.keys())
        gv = got.values_tensor().cpu().numpy()
        wv = np.array(list(want.values()))
        K = len({k[-1] for k in want})
        gv, wv = (gv.reshape(K, -1), wv.reshape(K, -1)) if order == "row" else (gv.reshape(-1, K).T, wv.reshape(-1, K).T)
        assert np.abs(gv - wv).max() <= PROB_TOL
        same = gv.argmax(0) == wv.argmax(0)
        for c in np.where(~same)[0]:
            top2 = np.sort(wv[:, c])[-2:]
            print(f"top-1 flip in column {c}: reference top-2 probabilities {top2[1]:.4f} / {top2[0]:.4f}")
        agree += int(same.sum())
        total += gv.shape[1]
    assert agree / total >= 0.99, (agree, total)
    print(f"C1 demo clip: top-1 agreement with the reference {agree}/{total}")


def test_top1_agreement_b32():
    """>= 99 % top-1 predicate agreement with the fp32 oracle on a B/32 clip."""
    from oracle.forward_ref import OraclePredicateModel
    from laser_amd.synthetic import make_clip_batch
    from laser_amd.weights import split_predicate_state_dict
    cfg = config_by_name("vit-b-32")
    sd = predicate_model_state_dict(cfg, 5, float(np.log(100.0)))
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    model = PredicateModel(None, None, "cuda:0", "vit-b-32", tokenizer=tok).eval()
    model.load_state_dict(sd)
    b = make_clip_batch(2, n_videos=1, frames=6, objects=6, H=360, W=480, k_cate=12, k_unary=40, k_binary=20,
                        min_box=16, max_box=300)
    got = model(**b.forward_kwargs())
    torch.set_num_threads(16)
    want = OraclePredicateModel(cfg, split_predicate_state_dict(sd), tok)(**b.forward_kwargs())
    worst, agree, total = 0.0, 0, 0
    for g, w in ((got[1][0], want[1][0]), (got[2][0], want[2][0])):
        assert list(g.keys()) == list(w.keys())
        gv = g.values_tensor().cpu().numpy()
        wv = np.array([float(x) for x in w.values()])
        worst = max(worst, float(np.abs(gv - wv).max()))
        K = len({k[-1] for k in w})
        gv, wv = gv.reshape(K, -1), wv.reshape(K, -1)        # name-major key order
        same = gv.argmax(0) == wv.argmax(0)
        agree += int(same.sum())
        total += gv.shape[1]
        for c in np.where(~same)[0]:
            top2 = np.sort(wv[:, c])[-2:]
            print(f"top-1 flip in column {c}: oracle top-2 probabilities {top2[1]:.4f} / {top2[0]:.4f}")
    assert worst <= PROB_TOL, worst
    assert agree / total >= 0.99, (agree, total)
    print(f"b32: max |p - oracle| = {worst:.3e}, top-1 agreement {agree}/{total}")


def test_vit_l14_336_towers_match_oracle():
    """ViT-L/14@336 geometry (T = 577, d = 1024, patch K padded 588 -> 640, text d = 768):
    image and text features of the HIP towers against the fp32 oracle."""
    from laser_amd import _lib, ops
    from laser_amd.weights import random_state_dict
    from oracle import clip_ref, resample_ref
    cfg = config_by_name("vit-l-14-336")
    sd = random_state_dict(cfg, 17)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4)
    H, W = 200, 300
    frame = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    boxes = [(0, 0, W, H), (20, 30, 180, 170), (100, 10, 290, 60)]
    rows = [(0, -1, -1, W, H, x1, y1, x2, y2, _lib.CROP_PLAIN, 0) for (x1, y1, x2, y2) in boxes]
    from laser_amd.predicate_model import CROP_DTYPE
    frames_t = torch.from_numpy(ops.pad_bytes(frame)).to(dev)
    patches = ops.crop_preprocess(frames_t, None, np.array(rows, dtype=CROP_DTYPE), 336, layout=_lib.LAYOUT_PATCHES,
                                  patch=14, out_dtype=_lib.DTYPE_BF16)
    got = ops.Tower(cfg, "vision", sd, dev).encode_images(patches, len(boxes)).cpu()
    px = torch.from_numpy(np.stack([resample_ref.preprocess(frame[y1:y2, x1:x2], 336) for (x1, y1, x2, y2) in boxes]))
    torch.set_num_threads(16)
    want = clip_ref.image_features(sd, cfg, px)
    rel = ((got - want).norm(dim=-1) / want.norm(dim=-1)).max().item()
    assert rel < 1e-2, rel
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    t = tok(["a person riding a horse", "dog", "the quick brown fox jumps over the lazy dog " * 12])
    tgot = ops.Tower(cfg, "text", sd, dev).encode_text(t["input_ids"], t["attention_mask"]).cpu()
    twant = clip_ref.text_features(sd, cfg, t["input_ids"], t["attention_mask"])
    trel = ((tgot - twant).norm(dim=-1) / twant.norm(dim=-1)).max().item()
    assert trel < 1e-2, trel
    print(f"L/14-336: image feature rel err {rel:.2e}, text feature rel err {trel:.2e}")


def test_vit_b32_tower_in_bulk_matches_oracle():
    """A batch large enough (1 300 crops) for the bulk kernels of the B/32 image tower, the 256 x 256 pipelined GEMM and
    the fused q/k/v + attention kernel, which the small clips above do not reach: features of sampled crops against the
    fp32 oracle, and every crop against the same tower run in small batches (the small-grid kernels)."""
    from laser_amd import _lib, ops
    from laser_amd.weights import random_state_dict
    from laser_amd.predicate_model import CROP_DTYPE
    from oracle import clip_ref, resample_ref
    cfg = config_by_name("vit-b-32")
    sd = random_state_dict(cfg, 23)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(8)
    H, W, n = 240, 320, 1300
    frame = (rng.integers(0, 256, size=(H, W, 3)) // 2 + np.linspace(0, 127, W)[None, :, None]).astype(np.uint8)
    x1 = rng.integers(0, W - 40, n); y1 = rng.integers(0, H - 40, n)
    x2 = np.minimum(W, x1 + rng.integers(20, 200, n)); y2 = np.minimum(H, y1 + rng.integers(20, 200, n))
    rows = [(0, -1, -1, W, H, int(a), int(b), int(c), int(d), _lib.CROP_PLAIN, 0) for a, b, c, d in zip(x1, y1, x2, y2)]
    frames_t = torch.from_numpy(ops.pad_bytes(frame)).to(dev)
    patches = ops.crop_preprocess(frames_t, None, np.array(rows, dtype=CROP_DTYPE), 224, layout=_lib.LAYOUT_PATCHES,
                                  patch=32, out_dtype=_lib.DTYPE_BF16)
    tower = ops.Tower(cfg, "vision", sd, dev)
    got = tower.encode_images(patches, n)
    P = cfg.n_patches
    small = torch.cat([tower.encode_images(patches[i * P:(i + 100) * P], min(100, n - i)) for i in range(0, n, 100)])
    torch.cuda.synchronize()
    rel_small = ((got - small).norm(dim=-1) / small.norm(dim=-1)).max().item()
    assert rel_small < 5e-3, rel_small
    pick = [0, 1, 249, 250, 251, 777, 1298, 1299]       # incl. both sides of a 5-sequence row tile and the ragged last tile
    px = torch.from_numpy(np.stack([resample_ref.preprocess(frame[y1[i]:y2[i], x1[i]:x2[i]], 224) for i in pick]))
    torch.set_num_threads(16)
    want = clip_ref.image_features(sd, cfg, px)
    rel = ((got[pick].cpu() - want).norm(dim=-1) / want.norm(dim=-1)).max().item()
    assert rel < 1e-2, rel
    print(f"B/32 bulk tower: rel err vs oracle {rel:.2e}, vs small batches {rel_small:.2e}")


def test_vit_l14_336_tower_in_bulk_matches_small_batches_and_oracle():
    """ViT-L/14@336 with enough crops (200 x 577 token rows) for the 256 x 256 pipelined GEMM at K = 1024 / 4096 and the
    shared-K/V attention kernel in bulk: every crop against the same tower run 8 crops at a time (small-grid kernels),
    two crops against the fp32 oracle."""
    from laser_amd import _lib, ops
    from laser_amd.weights import random_state_dict
    from laser_amd.predicate_model import CROP_DTYPE
    from oracle import clip_ref, resample_ref
    cfg = config_by_name("vit-l-14-336")
    sd = random_state_dict(cfg, 29)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(12)
    H, W, n = 300, 400, 200
    frame = (rng.integers(0, 256, size=(H, W, 3)) // 2 + np.linspace(0, 127, W)[None, :, None]).astype(np.uint8)
    x1 = rng.integers(0, W - 60, n); y1 = rng.integers(0, H - 60, n)
    x2 = np.minimum(W, x1 + rng.integers(40, 300, n)); y2 = np.minimum(H, y1 + rng.integers(40, 250, n))
    rows = [(0, -1, -1, W, H, int(a), int(b), int(c), int(d), _lib.CROP_PLAIN, 0) for a, b, c, d in zip(x1, y1, x2, y2)]
    frames_t = torch.from_numpy(ops.pad_bytes(frame)).to(dev)
    patches = ops.crop_preprocess(frames_t, None, np.array(rows, dtype=CROP_DTYPE), 336, layout=_lib.LAYOUT_PATCHES,
                                  patch=14, out_dtype=_lib.DTYPE_BF16)
    tower = ops.Tower(cfg, "vision", sd, dev)
    got = tower.encode_images(patches, n)
    P = cfg.n_patches
    small = torch.cat([tower.encode_images(patches[i * P:(i + 8) * P], min(8, n - i)) for i in range(0, n, 8)])
    torch.cuda.synchronize()
    rel_small = ((got - small).norm(dim=-1) / small.norm(dim=-1)).max().item()
    assert rel_small < 5e-3, rel_small
    pick = [0, 199]
    px = torch.from_numpy(np.stack([resample_ref.preprocess(frame[y1[i]:y2[i], x1[i]:x2[i]], 336) for i in pick]))
    torch.set_num_threads(16)
    want = clip_ref.image_features(sd, cfg, px)
    rel = ((got[pick].cpu() - want).norm(dim=-1) / want.norm(dim=-1)).max().item()
    assert rel < 1e-2, rel
    print(f"L/14-336 bulk tower: rel err vs oracle {rel:.2e}, vs small batches {rel_small:.2e}")


@pytest.mark.parametrize("cfg_name,n", [("openai/clip-vit-base-patch16", 260), ("openai/clip-vit-large-patch14", 140)])
def test_other_clip_geometries_towers_match_oracle(cfg_name, n):
    """The remaining OpenAI CLIP geometries the reference's `--clip-model-name` reaches (scripts/eval_sgclip.py:34-95 ->
    laser/models/llava_clip_model_v3.py:15-23): ViT-B/16 (T = 197) and ViT-L/14 at 224 px (T = 257), sequence lengths
    between the fused q/k/v + attention kernel (T <= 128) and ViT-L/14-336's 577.  Enough crops for the bulk kernels:
    every crop against the same tower in small batches, sampled crops and the text tower against the fp32 oracle."""
    from laser_amd import _lib, ops
    from laser_amd.weights import random_state_dict
    from laser_amd.predicate_model import CROP_DTYPE
    from oracle import clip_ref, resample_ref
    cfg = config_by_name(cfg_name)
    sd = random_state_dict(cfg, 31)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(16)
    H, W = 300, 400
    frame = (rng.integers(0, 256, size=(H, W, 3)) // 2 + np.linspace(0, 127, W)[None, :, None]).astype(np.uint8)
    x1 = rng.integers(0, W - 60, n); y1 = rng.integers(0, H - 60, n)
    x2 = np.minimum(W, x1 + rng.integers(40, 300, n)); y2 = np.minimum(H, y1 + rng.integers(40, 250, n))
    rows = [(0, -1, -1, W, H, int(a), int(b), int(c), int(d), _lib.CROP_PLAIN, 0) for a, b, c, d in zip(x1, y1, x2, y2)]
    frames_t = torch.from_numpy(ops.pad_bytes(frame)).to(dev)
    patches = ops.crop_preprocess(frames_t, None, np.array(rows, dtype=CROP_DTYPE), cfg.image, layout=_lib.LAYOUT_PATCHES,
                                  patch=cfg.patch, out_dtype=_lib.DTYPE_BF16)
    tower = ops.Tower(cfg, "vision", sd, dev)
    got = tower.encode_images(patches, n)
    P = cfg.n_patches
    small = torch.cat([tower.encode_images(patches[i * P:(i + 8) * P], min(8, n - i)) for i in range(0, n, 8)])
    torch.cuda.synchronize()
    rel_small = ((got - small).norm(dim=-1) / small.norm(dim=-1)).max().item()
    assert rel_small < 5e-3, rel_small
    pick = [0, 1, n - 1]
    px = torch.from_numpy(np.stack([resample_ref.preprocess(frame[y1[i]:y2[i], x1[i]:x2[i]], cfg.image) for i in pick]))
    torch.set_num_threads(16)
    want = clip_ref.image_features(sd, cfg, px)
    rel = ((got[pick].cpu() - want).norm(dim=-1) / want.norm(dim=-1)).max().item()
    assert rel < 1e-2, rel
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    t = tok(["a person riding a horse", "dog", "the quick brown fox jumps over the lazy dog " * 12])
    tgot = ops.Tower(cfg, "text", sd, dev).encode_text(t["input_ids"], t["attention_mask"]).cpu()
    twant = clip_ref.text_features(sd, cfg, t["input_ids"], t["attention_mask"])
    trel = ((tgot - twant).norm(dim=-1) / twant.norm(dim=-1)).max().item()
    assert trel < 1e-2, trel
    print(f"{cfg_name}: image feature rel err vs oracle {rel:.2e}, vs small batches {rel_small:.2e}, text {trel:.2e}")


def test_concurrent_towers_give_the_single_stream_result():
    """forward() queues the unary tower and the halves of a large pair set on side streams; the result must be the
    bits of the single-stream call, twice in a row (the second call reuses every workspace and stream)."""
    from laser_amd.synthetic import make_clip_batch
    cfg = config_by_name("tiny")
    sd = predicate_model_state_dict(cfg, 31, float(np.log(100.0)))
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    model = PredicateModel(None, None, "cuda:0", "tiny", tokenizer=tok).eval()
    model.load_state_dict(sd)
    b = make_clip_batch(9, n_videos=1, frames=5, objects=16, H=96, W=128, k_cate=5, k_unary=6, k_binary=4,
                        min_box=12, max_box=90)
    kw = b.forward_kwargs()
    assert len(kw["batched_obj_pairs"]) >= 1024
    assert len(model._side_streams()) == 4    # unary tower, two pair halves, text towers
    runs = [model(**kw) for _ in range(2)]
    model.concurrent_towers = False           # single stream
    ref = model(**kw)
    torch.cuda.synchronize()
    for got in runs:
        for g, w in ((got[0][0], ref[0][0]), (got[1][0], ref[1][0]), (got[2][0], ref[2][0])):
            assert list(g.keys()) == list(w.keys())
            assert torch.equal(g.values_tensor(), w.values_tensor())


@pytest.mark.parametrize("seed", [101, 102, 103])
def test_forward_input_variants_vs_oracle(seed):
    """Seeded variations of the argument forms the reference accepts (list / tuple / ndarray boxes,
    corners in either order, float and uint8 masks, objects missing from some frames, a subset of
    the pairs, a repeated object entry, per-video vocabularies of different length): the HIP path
    against the CPU oracle on identical inputs; key order exact, probabilities within tolerance."""
    from oracle.forward_ref import OraclePredicateModel
    from laser_amd.synthetic import make_clip_batch
    from laser_amd.weights import split_predicate_state_dict
    cfg = config_by_name("tiny")
    sd = predicate_model_state_dict(cfg, seed, float(np.log(100.0)))
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    model = PredicateModel(None, None, "cuda:0", "tiny", tokenizer=tok).eval()
    model.load_state_dict(sd)
    b = make_clip_batch(seed, n_videos=2, frames=3, objects=4, H=96, W=128, k_cate=5, k_unary=6, k_binary=4,
                        min_box=12, max_box=90)
    kw = b.forward_kwargs()
    rng = np.random.default_rng(seed)
    keep = [i for i in range(len(kw["batched_object_ids"])) if rng.random() < 0.8 or i < 2]
    present = {kw["batched_object_ids"][i] for i in keep}
    boxes, masks_in = [], []
    for n, i in enumerate(keep):
        bb = kw["batched_bboxes"][i]
        x1, y1, x2, y2 = bb["x1"], bb["y1"], bb["x2"], bb["y2"]
        form = n % 4
        boxes.append(bb if form == 0 else [x2, y2, x1, y1] if form == 1 else (x1, y2, x2, y1) if form == 2
                     else np.array([x1, y1, x2, y2, 0.9]))
        m = np.asarray(kw["batched_masks"][i])
        masks_in.append(m.astype(np.float32) if n % 3 == 0 else m.astype(np.uint8) if n % 3 == 1 else m)
    kw["batched_object_ids"] = [kw["batched_object_ids"][i] for i in keep]
    kw["batched_bboxes"], kw["batched_masks"] = boxes, masks_in
    kw["batched_obj_pairs"] = [p for p in kw["batched_obj_pairs"]
                               if (p[0], p[1], p[2][0]) in present and (p[0], p[1], p[2][1]) in present and rng.random() < 0.7]
    # the same object listed twice (later entry wins in the dictionaries)
    kw["batched_object_ids"].append(kw["batched_object_ids"][0])
    kw["batched_bboxes"].append(kw["batched_bboxes"][0])
    kw["batched_masks"].append(kw["batched_masks"][0])
    kw["batched_names"] = [kw["batched_names"][0][:3], kw["batched_names"][1]]
    fk = dict(multi_class=bool(seed % 2), alpha=0.4, white_alpha=0.7)
    got = model(**kw, **fk)
    want = OraclePredicateModel(cfg, split_predicate_state_dict(sd), tok)(**kw, **fk)
    worst = 0.0
    for part in (0, 1):
        for v in want[part]:
            assert list(got[part][v].keys()) == list(want[part][v].keys())
            worst = max(worst, compare_prob_dicts(got[part][v], {k: float(x) for k, x in want[part][v].items()}, PROB_TOL))
    assert len(got[2]) == len(want[2])
    for g, w in zip(got[2], want[2]):
        assert list(g.keys()) == list(w.keys())
        worst = max(worst, compare_prob_dicts(g, {k: float(x) for k, x in w.items()}, PROB_TOL))
    assert abs(float(got[3]) - float(want[3])) <= PROB_TOL
    print(f"variants seed {seed}: max |p - oracle| = {worst:.3e}")


def test_mask_of_another_shape_than_its_frame_is_refused():
    """The reference's np.where / cv2.addWeighted raise when a mask and its frame differ in shape
    (model_utils.py:48-54); here the crop table would otherwise address the mask with the frame's geometry."""
    from laser_amd.masks import pack_masks
    model = _model("tiny", 11)
    kw = build_case_batch("tiny_infer").forward_kwargs()
    kw["batched_masks"] = list(kw["batched_masks"])
    kw["batched_masks"][2] = np.ascontiguousarray(np.asarray(kw["batched_masks"][2])[:-3])      # 3 rows short
    with pytest.raises(ValueError, match="its frame is"):
        model(**kw)
    kw["batched_masks"][2] = np.ascontiguousarray(np.asarray(kw["batched_masks"][2]).transpose(1, 0, 2))
    with pytest.raises(ValueError, match="its frame is"):
        model(**dict(kw, batched_masks=pack_masks(kw["batched_masks"])))

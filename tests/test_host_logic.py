"""CPU tests of the host side: box resolution, level tables, mapping views,
tokenizer, weight plumbing, sharding helpers, and the no-GPU failure mode."""
import numpy as np
import pytest
import torch

from laser_amd import ops
from laser_amd.config import TINY, VIT_B32, VIT_L14_336
from laser_amd.distributed import balance_by_cost, shard_range
from laser_amd.predicate_model import PredicateModel, resolve_union_box, slice_box
from laser_amd.prob_dict import ProbDict
from laser_amd.synthetic import make_clip_batch
from laser_amd.tokenizer import HashTokenizer
from laser_amd.weights import infer_config, predicate_model_state_dict, random_state_dict, split_predicate_state_dict
from oracle import crop_ref, resample_ref


def test_level_tables_equal_oracle():
    for a, wa in ((0.5, 0.8), (0.3, 0.65), (0.0, 1.0), (1.0, 0.0)):
        got = ops.blend_luts(a, wa)
        want = crop_ref.blend_luts(a, wa)
        assert np.array_equal(got[:256], want["white"]) and np.array_equal(got[256:512], want["keep"])
        assert np.array_equal(got[512:], want["tint"])
    assert np.array_equal(ops.normalize_lut().reshape(3, 256), resample_ref.normalize_lut())


def test_box_resolution_matches_oracle_and_numpy_slicing():
    rng = np.random.default_rng(0)
    img = np.arange(40 * 50).reshape(40, 50)
    for _ in range(300):
        boxes = []
        for _ in range(int(rng.integers(1, 3))):
            v = rng.integers(-10, 60, size=4).astype(float) + rng.random(4)
            kind = int(rng.integers(0, 4))
            boxes.append({"x1": int(v[0]), "y1": int(v[1]), "x2": int(v[2]), "y2": int(v[3])} if kind == 0 else
                         list(v) if kind == 1 else tuple(v) if kind == 2 else np.array(list(v) + [0.3]))
        try:
            want = crop_ref.union_box(boxes)
        except AssertionError:
            with pytest.raises(AssertionError, match="image bbox issue"):
                resolve_union_box(boxes)
            continue
        assert resolve_union_box(boxes) == tuple(int(x) for x in want)
        x1, y1, x2, y2 = want
        view = img[y1:y2, x1:x2]
        if view.size == 0:
            with pytest.raises(ValueError):
                slice_box(x1, y1, x2, y2, 50, 40)
        else:
            xa, ya, xb, yb = slice_box(x1, y1, x2, y2, 50, 40)
            assert np.array_equal(img[ya:yb, xa:xb], view)
    with pytest.raises(ValueError, match="Unsupported bbox format"):
        resolve_union_box(["nope"])


def test_prob_dict_semantics():
    dense = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    rows = [(0, "a"), (2, "b")]                       # row 1 is the dummy keyword
    cols = [(0, (0, 7)), (1, (0, 8)), (3, (1, 7))]    # column 2 skipped (fid == -1)
    d = ProbDict(dense, rows, cols, lambda fo, nm: (fo[0], fo[1], nm), "row_major")
    assert list(d.keys()) == [(0, 7, "a"), (0, 8, "a"), (1, 7, "a"), (0, 7, "b"), (0, 8, "b"), (1, 7, "b")]
    assert [float(v) for v in d.values()] == [0, 1, 3, 8, 9, 11]
    assert float(d[(1, 7, "b")]) == 11 and len(d) == 6 and (0, 8, "a") in d
    assert d[(0, 7, "a")].dim() == 0
    c = ProbDict(dense, rows, [(0, 5), (1, 9)], lambda oid, nm: (oid, nm), "col_major")
    assert list(c.keys()) == [(5, "a"), (5, "b"), (9, "a"), (9, "b")]
    assert [float(v) for k, v in c.items()] == [0, 8, 1, 9]
    # duplicate keys: first position, last value (what repeated dict assignment does)
    dup = ProbDict(dense, [(0, "a"), (1, "a")], [(0, (0, 1))], lambda fo, nm: (fo[0], fo[1], nm), "row_major")
    assert list(dup.keys()) == [(0, 1, "a")] and float(dup[(0, 1, "a")]) == 4
    import pickle
    back = pickle.loads(pickle.dumps(d))
    assert isinstance(back, dict) and list(back.keys()) == list(d.keys())
    empty = ProbDict(torch.zeros(0, 0), [], [], lambda a, b: (a, b), "row_major")
    assert len(empty) == 0 and list(empty.items()) == []


def test_hash_tokenizer_layout():
    tok = HashTokenizer(1024)
    t = tok(["a b c", "", "x " * 200], max_length=75)
    ids, am = t["input_ids"], t["attention_mask"]
    assert ids.shape == (3, 75) and am.shape == (3, 75)
    assert (ids[:, 0] == tok.bos_token_id).all()
    assert ids[0, 4] == tok.eos_token_id and am[0].sum() == 5 and (ids[0, 5:] == tok.pad_token_id).all()
    assert ids[1, 1] == tok.eos_token_id and am[1].sum() == 2
    assert ids[2, 74] == tok.eos_token_id and am[2].sum() == 75          # truncated, eos kept
    assert torch.equal(tok(["a b c"])["input_ids"], ids[:1])
    assert t.to("cpu")["input_ids"] is not None


def test_weights_roundtrip_and_geometry():
    for cfg in (TINY,):
        sd = random_state_dict(cfg, 3)
        assert torch.equal(sd["visual_projection.weight"], random_state_dict(cfg, 3)["visual_projection.weight"])
        assert infer_config(sd).to_dict() == {**cfg.to_dict(), "logit_scale_init": pytest.approx(cfg.logit_scale_init)}
        full = predicate_model_state_dict(cfg, 1)
        parts = split_predicate_state_dict(full)
        assert set(parts) == {"clip_cate_model", "clip_unary_model", "clip_binary_model"}
        assert sum(len(p) for p in parts.values()) == len(full)
        assert not torch.equal(parts["clip_cate_model"]["logit_scale"] + parts["clip_cate_model"]["visual_projection.weight"][0, 0],
                               parts["clip_unary_model"]["logit_scale"] + parts["clip_unary_model"]["visual_projection.weight"][0, 0])
    assert abs(VIT_B32.vit_flops_per_encode() / 1e9 - 8.82) < 0.01
    assert abs(VIT_L14_336.vit_flops_per_encode() / 1e9 - 381.9) < 0.1


def test_synthetic_batch_shape():
    b = make_clip_batch(3, n_videos=2, frames=2, objects=3, H=60, W=80, k_cate=2, k_unary=3, k_binary=2, min_box=8, max_box=40)
    assert len(b.batched_videos) == 4 and b.batched_video_splits == [2, 4]
    assert len(b.batched_masks) == len(b.batched_bboxes) == len(b.batched_object_ids) == 12
    assert len(b.batched_obj_pairs) == 2 * 2 * 6
    for m, bb in zip(b.batched_masks, b.batched_bboxes):
        assert m.shape == (60, 80, 1) and m.dtype == bool
        ys, xs = np.where(m[..., 0])
        assert (bb["x1"], bb["x2"], bb["y1"], bb["y2"]) == (xs.min(), xs.max(), ys.min(), ys.max())
    b2 = make_clip_batch(3, n_videos=2, frames=2, objects=3, H=60, W=80, k_cate=2, k_unary=3, k_binary=2, min_box=8, max_box=40)
    assert np.array_equal(b.batched_videos[1], b2.batched_videos[1])


def test_sharding_helpers():
    for n in (0, 1, 7, 256):
        for w in (1, 2, 8):
            got = [i for r in range(w) for i in shard_range(n, r, w)]
            assert got == list(range(n))
            sizes = [len(shard_range(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1
    costs = [272, 10, 500, 272, 33, 8, 120, 272]
    parts = balance_by_cost(costs, 3)
    assert sorted(i for p in parts for i in p) == list(range(len(costs)))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) <= sum(costs) / 3 + max(costs)


def test_no_gpu_means_loud_failure():
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no CPU path|no GPU"):
        PredicateModel(None, None, "cuda", "tiny", tokenizer=HashTokenizer(1024))


def test_missing_tokenizer_files_fail_loudly():
    """Without vocabulary files the reference's AutoTokenizer.from_pretrained raises
    (laser/models/llava_clip_model_v3.py:15); so does the default tokenizer here, naming where it looked."""
    with pytest.raises(OSError, match="(?s)no CLIP tokenizer files for .no-such-org/no-such-clip..*tokenizer="):
        PredicateModel._default_tokenizer(None, "no-such-org/no-such-clip")


def test_any_clip_geometry_from_a_name_a_snapshot_or_a_config_dict(tmp_path):
    """The reference builds whatever AutoModel.from_pretrained(model_name) names
    (laser/models/llava_clip_model_v3.py:15-23); here: the hub names with a known geometry, a local HF
    snapshot directory, a config.json path or its dict."""
    import json
    from laser_amd.config import VIT_B16, VIT_L14, config_by_name, from_hf_config
    assert config_by_name("openai/clip-vit-base-patch16") is VIT_B16 and VIT_B16.v_tokens == 197
    assert config_by_name("openai/clip-vit-large-patch14") is VIT_L14 and VIT_L14.v_tokens == 257
    assert config_by_name("openai/clip-vit-large-patch14-336") is VIT_L14_336
    hf = {"model_type": "clip", "projection_dim": 768, "logit_scale_init_value": 2.6592,
          "vision_config": {"hidden_size": 1024, "intermediate_size": 4096, "num_hidden_layers": 24, "num_attention_heads": 16,
                            "patch_size": 14, "image_size": 224, "hidden_act": "quick_gelu", "layer_norm_eps": 1e-5},
          "text_config": {"hidden_size": 768, "intermediate_size": 3072, "num_hidden_layers": 12, "num_attention_heads": 12,
                          "vocab_size": 49408, "max_position_embeddings": 77, "eos_token_id": 49407, "bos_token_id": 49406}}
    assert from_hf_config(hf) == VIT_L14
    assert from_hf_config({}) == VIT_B32                      # HF CLIPConfig() defaults are the B/32 geometry
    assert from_hf_config({"text_config": {"eos_token_id": 2}}).eos_token_id == 2     # the argmax pooling branch's marker
    (tmp_path / "config.json").write_text(json.dumps(hf))
    assert config_by_name(str(tmp_path)) == VIT_L14
    assert config_by_name(tmp_path / "config.json") == VIT_L14
    assert config_by_name(hf) == VIT_L14
    with pytest.raises(KeyError, match="local HF snapshot"):
        config_by_name("laion/CLIP-ViT-H-14-laion2B-s32B-b79K")
    bad = json.loads(json.dumps(hf))
    bad["vision_config"].update(hidden_size=1280, num_attention_heads=16)      # ViT-H: head width 80
    with pytest.raises(ValueError, match="head width"):
        from_hf_config(bad)
    bad = json.loads(json.dumps(hf))
    bad["text_config"]["hidden_act"] = "gelu"
    with pytest.raises(ValueError, match="quick_gelu"):
        from_hf_config(bad)

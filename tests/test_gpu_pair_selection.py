"""Training-mode pair selection (reference: laser/models/llava_clip_model_v3.py:303-314, 347-386) against the CPU
oracle over consecutive seeds, none chosen for its margin, and the fp32 towers behind the ranking guard against the
oracle's towers."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from selection_scan import make_pair, scan_seed  # noqa: E402

from laser_amd import _lib, ops  # noqa: E402
from laser_amd.config import config_by_name  # noqa: E402
from laser_amd.rescore import ExactTower, attention_exact, linear_exact  # noqa: E402
from laser_amd.tokenizer import HashTokenizer  # noqa: E402
from laser_amd.weights import predicate_model_state_dict, split_predicate_state_dict  # noqa: E402

SEEDS = list(range(400, 424))   # 24 consecutive seeds


def test_selected_pairs_equal_the_oracles_over_consecutive_seeds():
    """With the ranking guard the selected pair lists (content AND order) equal the oracle's for every seed whose
    boundary fp32 arithmetic can resolve; such a boundary (gap below the fp32 bound, reported by the model as
    undecided) may fall either way in any fp32 implementation and is only counted.  Without the guard the bf16
    ranking flips some seeds: printed, not asserted."""
    torch.set_num_threads(16)
    model, oracle = make_pair("vit-b-32", 22)
    rows = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for seed in SEEDS:
            rows.append(scan_seed(model, oracle, seed))
    for r in rows:
        print(f"seed {r['seed']}: oracle margin {r['margin']:.3e}  plain {'same' if r['same_plain'] else 'FLIP'}  guarded "
              f"{'same' if r['same_guarded'] else 'FLIP'}  rescored {r['rescored']} objects  undecided {r['undecided']}")
    flips_plain = sum(not r["same_plain"] for r in rows)
    print(f"bf16 ranking alone: {flips_plain}/{len(rows)} seeds select other pairs than the oracle")
    bad = [r for r in rows if not r["same_guarded"] and not r["undecided"]]
    assert not bad, bad
    assert sum(bool(r["undecided"]) for r in rows) <= 2
    assert all(r["same_guarded"] for r in rows if r["margin"] > 1e-6), [r for r in rows if not r["same_guarded"]]


@pytest.mark.parametrize("cfg_name,n", [("tiny", 5), ("vit-b-32", 3)])
def test_exact_towers_match_the_oracle(cfg_name, n):
    """fp32 towers of the ranking guard vs the oracle's fp32 towers: features to fp32 rounding."""
    from oracle import clip_ref
    cfg = config_by_name(cfg_name)
    sd = split_predicate_state_dict(predicate_model_state_dict(cfg, 3, float(np.log(100.0))))["clip_cate_model"]
    dev = torch.device("cuda:0")
    params = {k: v.to(dev) for k, v in sd.items()}
    g = torch.Generator().manual_seed(1)
    px = torch.randn(n, 3, cfg.image, cfg.image, generator=g)
    want = clip_ref.image_features(sd, cfg, px)
    p, G_ = cfg.patch, cfg.grid
    kpad = _lib.load().sgclip_patch_kpad(p)
    patches = torch.zeros(n * G_ * G_, kpad)
    patches[:, : 3 * p * p] = px.view(n, 3, G_, p, G_, p).permute(0, 2, 4, 1, 3, 5).reshape(n * G_ * G_, 3 * p * p)
    got = ExactTower(cfg, "vision", params).image_features(patches.to(dev), n).cpu()
    rel = float((got - want).norm() / want.norm())
    print(f"{cfg_name} image features: relative difference {rel:.3e}")
    assert rel < 2e-5
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    t = tok(["a dog", "person riding a horse on the beach", "x"], return_tensors="pt", max_length=75, truncation=True,
            padding="max_length")
    want = clip_ref.text_features(sd, cfg, t["input_ids"], t["attention_mask"])
    got = ExactTower(cfg, "text", params).text_features(t["input_ids"], t["attention_mask"]).cpu()
    rel = float((got - want).norm() / want.norm())
    print(f"{cfg_name} text features: relative difference {rel:.3e}")
    assert rel < 2e-5


def test_linear_exact_and_attention_f32_against_float64():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x, w, b = torch.randn(70, 136, generator=g), torch.randn(45, 136, generator=g), torch.randn(45, generator=g)
    res = torch.randn(70, 45, generator=g)
    ref = x.double() @ w.double().t() + b.double()
    got = linear_exact(x.to(dev), w.to(dev), b.to(dev)).cpu().double()
    assert float((got - ref).abs().max()) < 2e-5
    got = linear_exact(x.to(dev), w.to(dev), b.to(dev), _lib.F32_GELU).cpu().double()
    assert float((got - ref * torch.sigmoid(1.702 * ref)).abs().max()) < 2e-5
    got = linear_exact(x.to(dev), w.to(dev), b.to(dev), _lib.F32_RESID, aux=res.to(dev)).cpu().double()
    assert float((got - (ref + res.double())).abs().max()) < 2e-5
    for T, causal in ((50, False), (9, True), (197, False)):
        heads, n_seq = 2, 3
        qkv = torch.randn(n_seq * T, 3 * heads * 64, generator=g)
        km = (torch.rand(n_seq, T, generator=g) > 0.2).int()
        km[:, 0] = 1
        q, k, v = (qkv.double().view(n_seq, T, 3, heads, 64).permute(2, 0, 3, 1, 4)[i] for i in range(3))
        s = q @ k.transpose(-1, -2) / 8.0
        allowed = km.bool()[:, None, None, :].expand(n_seq, heads, T, T)
        if causal:
            allowed = allowed & torch.ones(T, T, dtype=torch.bool).tril()[None, None]
        s = s.masked_fill(~allowed, -1e30)
        ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(n_seq * T, heads * 64)
        got = attention_exact(qkv.to(dev), n_seq, T, heads, causal, km.to(dev)).cpu().double()
        assert float((got - ref).abs().max()) < 2e-5, (T, causal)

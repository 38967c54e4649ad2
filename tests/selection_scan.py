"""Training-mode pair selection of laser_amd.PredicateModel against the CPU oracle, seed by seed (shared by
tests/test_gpu_pair_selection.py and tools/scan_pair_selection.py).

The clip of a seed is `tests/golden/cases.py: tiny_batch(train=True)` at the geometry of the golden case
`b32_train`: 2 videos x 3 frames x 5 objects at 320 x 240, five category names, three predicates; with 5 objects and
topk_cate = 3 every name the predicates use has a third / fourth boundary."""
from __future__ import annotations

import numpy as np
import torch

from cases import tiny_batch

from laser_amd.config import config_by_name
from laser_amd.predicate_model import PredicateModel
from laser_amd.tokenizer import HashTokenizer
from laser_amd.weights import predicate_model_state_dict, split_predicate_state_dict


def make_pair(cfg_name: str, wseed: int):
    """(model on cuda:0, oracle) with the same seeded weights, logit scale ln 100."""
    from oracle.forward_ref import OraclePredicateModel
    cfg = config_by_name(cfg_name)
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    sd = predicate_model_state_dict(cfg, wseed, float(np.log(100.0)))
    model = PredicateModel(None, None, "cuda:0", cfg_name, tokenizer=tok).eval()
    model.load_state_dict(sd)
    oracle = OraclePredicateModel(cfg, split_predicate_state_dict(sd), tok)
    return model, oracle


def oracle_margin(want_cate, batch, topk=3):
    """Smallest gap between the topk-th and the next candidate over the names the predicates use (oracle values)."""
    worst = np.inf
    for vid, preds in enumerate(batch.batched_binary_predicates):
        used = {p[1] for p in preds} | {p[2] for p in preds}
        for nm in used:
            vals = sorted((float(v) for (oid, n), v in want_cate[vid].items() if n == nm), reverse=True)
            if len(vals) > topk:
                worst = min(worst, vals[topk - 1] - vals[topk])
    return worst


def scan_seed(model, oracle, seed: int):
    """-> dict(seed, margin, same_plain, same_guarded, rescored, undecided)."""
    b = tiny_batch(seed=seed, n_videos=2, frames=3, objects=5, H=240, W=320, train=True, max_box=220)
    kw = b.forward_kwargs()
    want = oracle(**kw, output_embeddings=True)
    want_pairs = {v: list(want[4]["binary_object_pairs"][v]) for v in range(2)}
    out = {}
    for label, guard in (("plain", False), ("guarded", True)):
        model.rank_guard = guard
        got = model(**kw, output_embeddings=True)
        torch.cuda.synchronize()
        got_pairs = {v: [(p[0], p[1], tuple(p[2])) for p in got[4]["binary_object_pairs"][v]] for v in range(2)}
        out[label] = got_pairs == {v: [(p[0], p[1], tuple(p[2])) for p in want_pairs[v]] for v in range(2)}
        if guard:
            out["rescored"] = sum(r["rescored"] for r in model.ranking_report)
            out["undecided"] = [u for r in model.ranking_report for u in r["undecided"]]
    model.rank_guard = True
    return dict(seed=seed, margin=oracle_margin(want[0], b), same_plain=out["plain"], same_guarded=out["guarded"],
                rescored=out["rescored"], undecided=out["undecided"])

"""Consumers of the scoring path's outputs, fed from the dense tensors.

forward() returns mappings with 10^5 entries per clip.  The reference's two
consumers walk them entry by entry:

* training turns them into Scallop input facts `(prob, tuple)`
  (scripts/train_sgclip.py:729-764, consumed by laser/utils.py:394-443
  `format_batched_facts`);
* evaluation ranks the keywords of every object / pair
  (scripts/eval_sgclip.py:97-203 `compute_metrics`).

Here both read the dense `[keywords, crops]` tensors behind a `ProbDict`: facts
are one gather + one tuple list, the ranking is one top-k kernel
(`sgclip_topk_columns`) whose tie order equals `sorted((p, name), reverse=True)`.
Plain dicts (the reference's own return type) are accepted too and take the
entry-by-entry route.
"""
from __future__ import annotations

from typing import Dict, Hashable, List, Mapping, Sequence, Tuple

import torch

from . import ops
from .prob_dict import ProbDict


class FactTable:
    """Columnar fact list: `probs[i]` is the probability of `tuples[i]`."""

    def __init__(self, probs: torch.Tensor, tuples: List[tuple]):
        assert probs.dim() == 1 and probs.shape[0] == len(tuples)
        self.probs, self.tuples = probs, tuples

    def __len__(self):
        return len(self.tuples)

    def pairs(self) -> List[Tuple[torch.Tensor, tuple]]:
        """The reference's form: a list of (0-dim tensor, tuple)."""
        return list(zip(self.probs.unbind(0), self.tuples))

    def __iter__(self):
        return iter(self.pairs())


def _keys_and_values(probs: Mapping):
    if isinstance(probs, ProbDict):
        return list(probs.keys()), probs.values_tensor()
    keys = list(probs.keys())
    vals = list(probs.values())
    if vals:
        return keys, torch.stack([v if torch.is_tensor(v) else torch.tensor(float(v)) for v in vals])
    return keys, torch.zeros(0)


def cate_facts(image_cate_probs: Mapping, const_lookup: Mapping[str, int]) -> FactTable:
    """(prob, (oid, const_id - 1)) for every (oid, name) whose name is a constant
    of the specification, in dictionary order (scripts/train_sgclip.py:733-735)."""
    keys, vals = _keys_and_values(image_cate_probs)
    sel = [i for i, (oid, name) in enumerate(keys) if name in const_lookup]
    tuples = [(keys[i][0], const_lookup[keys[i][1]] - 1) for i in sel]
    idx = torch.tensor(sel, dtype=torch.long, device=vals.device)
    return FactTable(vals[idx] if sel else vals[:0], tuples)


def unary_facts(image_unary_probs: Mapping) -> FactTable:
    """(prob, (name, fid, oid)) (scripts/train_sgclip.py:737-738)."""
    keys, vals = _keys_and_values(image_unary_probs)
    return FactTable(vals, [(name, fid, oid) for (fid, oid, name) in keys])


def binary_facts(image_binary_probs: Mapping) -> FactTable:
    """(prob, (name, fid, from, to)) (scripts/train_sgclip.py:761-762)."""
    keys, vals = _keys_and_values(image_binary_probs)
    return FactTable(vals, [(name, fid, pair[0], pair[1]) for (fid, pair, name) in keys])


def scl_facts(batched_image_cate_probs: Mapping, batched_image_unary_probs: Mapping,
              batched_image_binary_probs: Sequence[Mapping], const_lookup: Sequence[Mapping[str, int]]):
    """The three per-video fact lists of scripts/train_sgclip.py:729-764:
    (batched_cate_pred_scl, batched_unary_pred_scl, batched_binary_pred_scl), each a list of FactTable."""
    cate, unary, binary = [], [], []
    for vid, (c, u) in enumerate(zip(batched_image_cate_probs.values(), batched_image_unary_probs.values())):
        cate.append(cate_facts(c, const_lookup[vid]))
        unary.append(unary_facts(u))
    for b in batched_image_binary_probs:
        binary.append(binary_facts(b) if len(b) else FactTable(torch.zeros(0), []))
    return cate, unary, binary


# ---------------------------------------------------------------------------
# evaluation metrics
# ---------------------------------------------------------------------------
def _top_names(pred: ProbDict, k: int):
    """Per column key -> names of its k best keywords, best first (ProbDict route)."""
    rows = pred.row_names
    names = [nm for _, nm in rows]
    if len(set(names)) != len(names) or pred.dense.dim() != 2:
        return None  # duplicate names collapse in the dictionary: take the generic route
    if not rows or not pred.col_keys:
        return {}
    ridx = torch.tensor([r for r, _ in rows], dtype=torch.long, device=pred.dense.device)
    dense = pred.dense.detach().float()[ridx].contiguous()
    order = sorted(range(len(names)), key=lambda i: names[i])
    rank = [0] * len(names)
    for pos, i in enumerate(order):
        rank[i] = pos
    idx, _ = ops.topk_columns(dense, min(k, ops.TOPK_MAX), torch.tensor(rank, dtype=torch.int32))
    idx = idx.cpu().numpy()
    out = {}
    for c, ck in pred.col_keys:  # later duplicates overwrite, as dictionary insertion does
        out[ck] = [names[i] for i in idx[:, c] if i >= 0]
    return out


def _group_generic(pred: Mapping, key_of):
    grouped: Dict[Hashable, list] = {}
    for key, p in pred.items():
        ck, name = key_of(key)
        grouped.setdefault(ck, []).append((p, name))
    return {ck: [name for _, name in sorted(v, reverse=True)] for ck, v in grouped.items()}


def compute_metrics(gt_object_dict, cate_pred: Mapping, gt_object_rels, binary_pred: Mapping,
                    precision_thres_ls=(1, 5, 10), recall_thres_ls=(1, 5, 10)):
    """Same arguments and result as scripts/eval_sgclip.py:97-203."""
    kmax = max(list(precision_thres_ls) + list(recall_thres_ls) + [1])
    result_unary_precision = {t: [] for t in precision_thres_ls}
    result_binary_precision = {t: [] for t in precision_thres_ls}
    result_unary_recall = {t: [] for t in recall_thres_ls}
    result_binary_recall = {t: [] for t in recall_thres_ls}

    cate_top = None
    if isinstance(cate_pred, ProbDict) and kmax <= ops.TOPK_MAX:
        cate_top = _top_names(cate_pred, kmax)
    if cate_top is None:
        cate_top = _group_generic(cate_pred, lambda key: (key[0], key[1]))
    for vid, oid, gt_label in gt_object_dict:
        gt_label = gt_label.replace('/', ' ').replace('_', ' ')
        assert vid == 0
        ranked = cate_top.get(oid, [])
        for thres in precision_thres_ls:
            result_unary_precision[thres].append((1 if gt_label in ranked[:thres] else 0) / thres)
        for thres in recall_thres_ls:
            result_unary_recall[thres].append(1 if gt_label in ranked[:thres] else 0)

    gt_rel_dict = {}
    for fid, rel_ls in enumerate(gt_object_rels):
        for (from_id, to_id, rel) in rel_ls:
            gt_rel_dict[(fid, from_id, to_id)] = rel
    bin_top = None
    if isinstance(binary_pred, ProbDict) and kmax <= ops.TOPK_MAX:
        top = _top_names(binary_pred, kmax)
        if top is not None:
            bin_top = {(fid, pair[0], pair[1]): v for (fid, pair), v in top.items()}
    if bin_top is None:
        bin_top = _group_generic(binary_pred, lambda key: ((key[0], key[1][0], key[1][1]), key[2]))
    for key, gt_label in gt_rel_dict.items():
        gt_label = gt_label.replace('/', ' ').replace('_', ' ')
        ranked = bin_top.get(key, [])
        for thres in precision_thres_ls:
            result_binary_precision[thres].append((1 if gt_label in ranked[:thres] else 0) / thres)
        for thres in recall_thres_ls:
            result_binary_recall[thres].append(1 if gt_label in ranked[:thres] else 0)

    return {
        "precision": {"cate": result_unary_precision, "binary": result_binary_precision},
        "recall": {"cate": result_unary_recall, "binary": result_binary_recall},
    }

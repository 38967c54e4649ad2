"""TEST INFRASTRUCTURE - CPU restatement of the two consumers of forward()'s outputs.

Only tests/ may import this module; the product (laser_amd/) never does.

* `scl_fact_lists`  restates the fact construction inlined in the reference's training step,
  scripts/train_sgclip.py:729-764 (it is not a callable function there, so it cannot be run
  for goldens: parity for it is pinned by this restatement alone).
* `compute_metrics` restates scripts/eval_sgclip.py:97-203 loop for loop; it is pinned against
  outputs of the reference function itself (tests/golden/consumers.json).
"""
from __future__ import annotations


def scl_fact_lists(image_cate_probs, image_unary_probs, image_binary_probs, const_lookup):
    """One video's (cate, unary, binary) fact lists as lists of (prob, tuple)."""
    new_cate_pred_scl, unary_pred_scl, binary_pred_scl = [], [], []
    for (oid, cate_name), prob in image_cate_probs.items():            # train_sgclip.py:733-735
        if cate_name in const_lookup:
            new_cate_pred_scl.append((prob, (oid, const_lookup[cate_name] - 1)))
    for (fid, oid, unary_name), prob in image_unary_probs.items():     # :737-738
        unary_pred_scl.append((prob, (unary_name, fid, oid)))
    if len(image_binary_probs) != 0:                                   # :756-762
        for (fid, pair, binary_name), prob in image_binary_probs.items():
            binary_pred_scl.append((prob, (binary_name, fid, pair[0], pair[1])))
    return new_cate_pred_scl, unary_pred_scl, binary_pred_scl


def compute_metrics(gt_object_dict, cate_pred, gt_object_rels, binary_pred, precision_thres_ls=(1, 5, 10),
                    recall_thres_ls=(1, 5, 10)):
    up = {t: [] for t in precision_thres_ls}
    bp = {t: [] for t in precision_thres_ls}
    ur = {t: [] for t in recall_thres_ls}
    br = {t: [] for t in recall_thres_ls}
    new_cate_pred = {}
    for ((oid, name), p) in cate_pred.items():                          # eval_sgclip.py:112-115
        new_cate_pred.setdefault(oid, []).append((p, name))
    for vid, oid, gt_label in gt_object_dict:                           # :128-151
        gt_label = gt_label.replace('/', ' ').replace('_', ' ')
        assert vid == 0
        sorted_pred = sorted(new_cate_pred.get(oid, []), reverse=True)
        for thres in precision_thres_ls:
            top = [name for p, name in sorted_pred[:thres]]
            up[thres].append((1 if gt_label in top else 0) / thres)
        for thres in recall_thres_ls:
            top = [name for p, name in sorted_pred[:thres]]
            ur[thres].append(1 if gt_label in top else 0)
    gt_rel_dict = {}
    for fid, rel_ls in enumerate(gt_object_rels):                       # :154-157
        for (from_id, to_id, rel) in rel_ls:
            gt_rel_dict[(fid, from_id, to_id)] = rel
    new_binary_pred = {}
    for ((fid, (from_id, to_id), rel), p) in binary_pred.items():       # :160-164
        new_binary_pred.setdefault((fid, from_id, to_id), []).append((p, rel))
    for key, gt_label in gt_rel_dict.items():                           # :167-189
        gt_label = gt_label.replace('/', ' ').replace('_', ' ')
        sorted_pred = sorted(new_binary_pred.get(key, []), reverse=True)
        for thres in precision_thres_ls:
            top = [rel for p, rel in sorted_pred[:thres]]
            bp[thres].append((1 if gt_label in top else 0) / thres)
        for thres in recall_thres_ls:
            top = [rel for p, rel in sorted_pred[:thres]]
            br[thres].append(1 if gt_label in top else 0)
    return {"precision": {"cate": up, "binary": bp}, "recall": {"cate": ur, "binary": br}}

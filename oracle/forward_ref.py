"""TEST INFRASTRUCTURE ONLY - CPU oracle, never imported by the product path.

CPU restatement of `PredicateModel.forward`
(reference: laser/models/llava_clip_model_v3.py:86-487) on top of the other
oracle modules: whole-frame blends + box slices on the host (crop_ref), the
PIL-compatible preprocessing (resample_ref), fp32 CLIP towers (clip_ref).
Same argument list and the same return layout, except that dictionary values
are 0-dim fp32 torch tensors detached from any graph.

Pinned by tests/golden/forward_*.npz, which hold the outputs of the reference's
own forward() on the same seeded inputs (tests/golden/make_golden.py).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

from laser_amd.config import ClipConfig
from . import clip_ref, crop_ref, resample_ref


class OraclePredicateModel:
    def __init__(self, cfg: ClipConfig, towers: Dict[str, Dict[str, torch.Tensor]], tokenizer, image_size: int | None = None):
        """`towers`: {'clip_cate_model': sd, 'clip_unary_model': sd, 'clip_binary_model': sd}."""
        self.cfg = cfg
        self.cate, self.unary, self.binary = (towers[k] for k in ("clip_cate_model", "clip_unary_model", "clip_binary_model"))
        self.tok = tokenizer
        self.S = image_size or cfg.image
        self.timers: Dict[str, float] = {}
        # True: the returned values stay attached to the autograd graph of the state-dict tensors
        # (fp32 autograd reference for the train step, scripts/train_sgclip.py:682-694, 877-890)
        self.grad = False
        # True: the resize runs in Pillow's C code, as in the reference (CPU timing); False: the numpy restatement
        self.pillow = False

    # ----- helpers ---------------------------------------------------------
    def _text(self, sd, kws):
        t = self.tok(list(kws), return_tensors="pt", max_length=75, truncation=True, padding="max_length")
        fn = clip_ref.text_features.__wrapped__ if self.grad else clip_ref.text_features
        return fn(sd, self.cfg, t["input_ids"], t.get("attention_mask"))

    def _image(self, sd, px):
        fn = clip_ref.image_features.__wrapped__ if self.grad else clip_ref.image_features
        return fn(sd, self.cfg, px)

    def _pixels(self, crops: List[np.ndarray]) -> torch.Tensor:
        return torch.from_numpy(np.stack([resample_ref.preprocess(c, self.S, pillow=self.pillow) for c in crops]))

    # ----- forward ---------------------------------------------------------
    def forward(self, *args, **kwargs):
        with torch.set_grad_enabled(self.grad):
            return self._forward(*args, **kwargs)

    def _forward(self, batched_video_ids, batched_videos, batched_masks, batched_bboxes, batched_names,
                batched_object_ids, batched_unary_kws, batched_binary_kws, batched_obj_pairs,
                batched_video_splits, batched_binary_predicates, unary_segment_size=None,
                binary_segment_size=None, alpha=0.5, white_alpha=0.8, topk_cate=3, dummy_str="$$$",
                multi_class=False, output_logit=False, output_embeddings=False):
        import time
        fill = lambda b: [list(k) if len(k) else [dummy_str] for k in b]
        batched_names, batched_unary_kws, batched_binary_kws = fill(batched_names), fill(batched_unary_kws), fill(batched_binary_kws)
        dummy_prob = torch.tensor(0.0)
        B = len(batched_video_ids)
        t0 = time.perf_counter()
        name_feats = [self._text(self.cate, n) for n in batched_names]
        unary_feats = [self._text(self.unary, n) for n in batched_unary_kws]
        binary_feats = [self._text(self.binary, n) for n in batched_binary_kws]
        self.timers["text"] = time.perf_counter() - t0

        splits = [0] + list(batched_video_splits)
        assert len(batched_object_ids) > 0, f"No object bbox: {batched_video_ids}"
        frame_masks, frame_bboxes = {}, {}
        ids_lookup = {v: [] for v in range(B)}
        cropped_objs = {v: [] for v in range(B)}
        t0 = time.perf_counter()
        for (vid, fid, oid), mask, bbox in zip(batched_object_ids, batched_masks, batched_bboxes):
            frame = batched_videos[splits[vid] + fid]
            obj_img = crop_ref.extract_single_object(frame, mask, white_alpha)
            cropped_objs[vid].append(crop_ref.crop_image_contain_bboxes(obj_img, [bbox], batched_video_ids))
            frame_masks[vid, fid, oid] = mask
            frame_bboxes[vid, fid, oid] = bbox
            ids_lookup[vid].append((fid, oid))
        self.timers["obj_crop"] = time.perf_counter() - t0

        cate_probs, unary_probs, obj_per_cate_all = {}, {}, {}
        cate_feats_out, unary_feats_out = {}, {}
        self.timers["obj_pre"] = self.timers["obj_encode"] = 0.0
        for vid in range(B):
            cate_probs[vid], unary_probs[vid], obj_per_cate_all[vid] = {}, {}, {}
            cate_feats_out[vid], unary_feats_out[vid] = {}, {}
        # the reference zips this loop with batched_binary_predicates (:233-238),
        # so videos beyond its length keep the empty initial values
        for vid in range(min(B, len(batched_binary_predicates))):
            crops = cropped_objs[vid]
            object_ids = ids_lookup[vid]
            if len(crops):
                t0 = time.perf_counter()
                px = self._pixels(crops)
                self.timers["obj_pre"] += time.perf_counter() - t0
                t0 = time.perf_counter()
                cf = self._image(self.cate, px)
                uf = self._image(self.unary, px)
                self.timers["obj_encode"] += time.perf_counter() - t0
            else:
                cf = uf = torch.tensor([])
            cate_feats_out[vid], unary_feats_out[vid] = cf, uf
            if len(crops) == 0 or len(object_ids) == 0:
                cl = torch.tensor([])
            else:
                cl = clip_ref.clip_sim(name_feats[vid], cf, self.cate["logit_scale"])
                if not output_logit:
                    cl = cl.softmax(dim=0)
            assert len(object_ids) == 0 or (cl.dim() == 2 and cl.shape[1] == len(object_ids)), \
                f"Mismatched object id and cate logic: {batched_video_ids}"
            per_obj: Dict[int, Dict[str, list]] = {}
            for name, probs in zip(batched_names[vid], cl):
                if name == dummy_str:
                    dummy_prob = dummy_prob + probs.sum()
                else:
                    for p, (fid, oid) in zip(probs, object_ids):
                        per_obj.setdefault(oid, {}).setdefault(name, []).append(p)
            new_cate, obj_per_cate = {}, {}
            for oid, info in per_obj.items():
                for name, plist in info.items():
                    p = torch.mean(torch.stack(plist))
                    obj_per_cate.setdefault(name, []).append((p, oid))
                    new_cate[(oid, name)] = p
            for name in obj_per_cate:
                obj_per_cate[name] = sorted(obj_per_cate[name], reverse=True)
            if len(crops) == 0:
                ul = torch.tensor([])
            else:
                ul = clip_ref.clip_sim(unary_feats[vid], uf, self.unary["logit_scale"])
                if not output_logit:
                    ul = ul.softmax(dim=0)
            un = {}
            for name, probs in zip(batched_unary_kws[vid], ul):
                if name == dummy_str:
                    dummy_prob = dummy_prob + probs.sum()
                else:
                    for p, (fid, oid) in zip(probs, object_ids):
                        un[(fid, oid, name)] = p
            cate_probs[vid], unary_probs[vid], obj_per_cate_all[vid] = new_cate, un, obj_per_cate

        # pair selection
        topk = {v: {n: [oid for _, oid in lst[:topk_cate]] for n, lst in obj_per_cate_all[v].items()} for v in range(B)}
        pair_lookup = {v: {} for v in range(B)}
        for (vid, fid, (a, b)) in batched_obj_pairs:
            pair_lookup[vid].setdefault((a, b), []).append(fid)
        if batched_binary_predicates[0] is None:
            selected = list(batched_obj_pairs)
        else:
            sel = set()
            for vid, preds in enumerate(batched_binary_predicates):
                cands = topk[vid]
                for (rel, fn, tn) in preds:
                    if fn in cands and tn in cands:
                        for a in cands[fn]:
                            for b in cands[tn]:
                                if (a, b) in pair_lookup[vid]:
                                    for fid in pair_lookup[vid][(a, b)]:
                                        sel.add((vid, fid, (a, b)))
            selected = list(sel)
        new_select = {v: [] for v in range(B)}
        for (vid, fid, pr) in selected:
            new_select[vid].append((vid, fid, pr))
        pair_crops = {v: [] for v in range(B)}
        t0 = time.perf_counter()
        for (vid, fid, (a, b)) in selected:
            frame = batched_videos[splits[vid] + fid]
            img = crop_ref.extract_object_subject(frame, frame_masks[vid, fid, a], frame_masks[vid, fid, b],
                                                  alpha=alpha, white_alpha=white_alpha)
            pair_crops[vid].append(crop_ref.crop_image_contain_bboxes(
                img, [frame_bboxes[vid, fid, a], frame_bboxes[vid, fid, b]], batched_video_ids))
        self.timers["pair_crop"] = time.perf_counter() - t0
        if len(selected) == 0:
            selected.append((0, -1, (-1, -1)))
            new_select[0] = [(0, -1, (-1, -1))]
            pair_crops[0] = [batched_videos[0]]

        binary_probs = []
        pair_feats_out = {v: torch.tensor([]) for v in range(B)}
        self.timers["pair_pre"] = self.timers["pair_encode"] = 0.0
        for vid in range(B):
            crops = pair_crops[vid]
            if len(crops) == 0:
                binary_probs.append({})
                continue
            t0 = time.perf_counter()
            px = self._pixels(crops)
            self.timers["pair_pre"] += time.perf_counter() - t0
            t0 = time.perf_counter()
            pf = self._image(self.binary, px)
            self.timers["pair_encode"] += time.perf_counter() - t0
            pair_feats_out[vid] = pf
            bl = clip_ref.clip_sim(binary_feats[vid], pf, self.binary["logit_scale"])
            if not output_logit:
                bl = bl.sigmoid() if multi_class else bl.softmax(dim=0)
            bp = {}
            for name, probs in zip(batched_binary_kws[vid], bl):
                if name == dummy_str:
                    dummy_prob = dummy_prob + probs.sum()
                else:
                    for p, (_, fid, pr) in zip(probs, new_select[vid]):
                        if fid == -1:
                            dummy_prob = dummy_prob + p
                        else:
                            bp[(fid, pr, name)] = p
            binary_probs.append(bp)
        if output_embeddings:
            emb = {"cate_obj_clip_features": cate_feats_out, "cate_object_ids": ids_lookup,
                   "unary_obj_clip_features": unary_feats_out, "unary_object_ids": ids_lookup,
                   "binary_obj_pair_features": pair_feats_out, "binary_object_pairs": new_select}
            return cate_probs, unary_probs, binary_probs, dummy_prob, emb
        return cate_probs, unary_probs, binary_probs, dummy_prob

    __call__ = forward

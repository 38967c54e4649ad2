"""Data-parallel sharding of the scoring path over the GPUs of one node.

Clips (and, inside a clip, frames) are independent through crop -> encode ->
similarity, so they shard with no data-path collective; the only exchange is the
all-gather of the dense probability blocks (the reference gathers pickled metric
dicts with all_gather_object every iteration, scripts/eval_sgclip.py:495-496).
One process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI) on
the GPUs, "gloo" in CPU tests.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> range:
    """Contiguous balanced shard of range(n): the first n % world ranks get one extra."""
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def balance_by_cost(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time assignment of work units (clips or frames, cost = image
    encodes = 2 * objects + pairs) to ranks; returns the unit indices per rank, each sorted."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += costs[i]
    return [sorted(x) for x in out]


def all_gather_blocks(blocks: Sequence[torch.Tensor], group=None) -> List[List[torch.Tensor]]:
    """All-gather a list of fp32 tensors whose shapes differ between ranks.

    Returns, for every rank, its list of blocks.  One shape exchange (int64, tiny) and
    one padded payload all-gather per call, whatever the number of blocks."""
    world = dist.get_world_size(group)
    dev = blocks[0].device if blocks else torch.device("cpu")
    nb = len(blocks)
    shapes = torch.zeros((nb, 3), dtype=torch.int64, device=dev)   # ndim, d0, d1 (blocks are <= 2-D)
    for i, b in enumerate(blocks):
        if b.dim() > 2:
            raise ValueError("probability blocks are at most 2-D")
        shapes[i, 0] = b.dim()
        for d in range(b.dim()):
            shapes[i, 1 + d] = b.shape[d]
    all_shapes = [torch.zeros_like(shapes) for _ in range(world)]
    dist.all_gather(all_shapes, shapes, group=group)
    sizes = []
    for s in all_shapes:
        tot = 0
        for i in range(nb):
            nd = int(s[i, 0])
            tot += int(s[i, 1]) * (int(s[i, 2]) if nd == 2 else 1) if nd else 1
        sizes.append(tot)
    cap = max(max(sizes), 1)
    flat = torch.zeros(cap, dtype=torch.float32, device=dev)
    if blocks:
        cat = torch.cat([b.reshape(-1).float() for b in blocks])
        flat[: cat.numel()] = cat
    gathered = torch.empty(world * cap, dtype=torch.float32, device=dev)
    if dev.type == "cuda":
        dist.all_gather_into_tensor(gathered, flat, group=group)
    else:
        parts = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(parts, flat, group=group)
        gathered = torch.cat(parts)
    out: List[List[torch.Tensor]] = []
    for r, s in enumerate(all_shapes):
        off = r * cap
        lst = []
        for i in range(nb):
            nd = int(s[i, 0])
            shape = tuple(int(s[i, 1 + d]) for d in range(nd))
            n = 1
            for v in shape:
                n *= v
            lst.append(gathered[off: off + n].reshape(shape))
            off += n
        out.append(lst)
    return out


# ---------------------------------------------------------------------------
# frame-parallel scoring of one forward() call
# ---------------------------------------------------------------------------
class FrameShardPlan:
    """Which rank scores which frames of one forward() call (inference mode).

    Frames are the unit: every object crop and every pair crop of a frame goes to the rank that
    owns the frame, so no crop needs data from another rank.  The one cross-frame coupling of
    the inference path, the per-object mean of category probabilities over frames
    (laser/models/llava_clip_model_v3.py:296-311), is recombined from per-rank means and counts."""

    def __init__(self, batched_object_ids, batched_obj_pairs, world: int):
        units = sorted({(vid, fid) for (vid, fid, _) in batched_object_ids})
        cost = {u: 0.0 for u in units}
        for (vid, fid, _) in batched_object_ids:
            cost[(vid, fid)] += 2.0                       # category + unary encode
        for (vid, fid, _) in batched_obj_pairs:
            if (vid, fid) in cost:
                cost[(vid, fid)] += 1.0
        owner = {}
        for r, idxs in enumerate(balance_by_cost([cost[u] for u in units], world)):
            for i in idxs:
                owner[units[i]] = r
        self.world = world
        self.obj_pos = [[i for i, (vid, fid, _) in enumerate(batched_object_ids) if owner[(vid, fid)] == r]
                        for r in range(world)]
        # a pair whose frame has no object at all cannot be scored anywhere; it goes to rank 0, which
        # fails on it exactly as the unsharded call does
        self.pair_pos = [[i for i, (vid, fid, _) in enumerate(batched_obj_pairs) if owner.get((vid, fid), 0) == r]
                         for r in range(world)]
        self.has_pairs = len(batched_obj_pairs) > 0
        with_objects = [r for r in range(world) if self.obj_pos[r]]
        self.dummy_owner = with_objects[0] if with_objects else 0

    def local_kwargs(self, kw: dict, rank: int) -> dict:
        """forward() arguments of `rank`: the parallel object lists and the pair list filtered to its frames."""
        from .masks import PackedMasks
        op, pp = self.obj_pos[rank], self.pair_pos[rank]
        out = dict(kw)
        out["batched_object_ids"] = [kw["batched_object_ids"][i] for i in op]
        masks = kw["batched_masks"]
        out["batched_masks"] = masks.subset(op) if isinstance(masks, PackedMasks) else [masks[i] for i in op]
        if kw.get("batched_bboxes") is not None:
            out["batched_bboxes"] = [kw["batched_bboxes"][i] for i in op]
        out["batched_obj_pairs"] = [kw["batched_obj_pairs"][i] for i in pp]
        out["allow_dummy_pair"] = (not self.has_pairs) and rank == self.dummy_owner
        return out


def _vids(kw) -> int:
    return len(kw["batched_video_ids"])


def local_blocks(out, kw: dict, device) -> List[torch.Tensor]:
    """Dense blocks of one rank's forward() result, in a fixed order: per video the per-object
    category means, the unary block and the binary block, then the dummy probability.
    `out` is None for a rank that owns no frame."""
    blocks = []
    for vid in range(_vids(kw)):
        for part in (0, 1, 2):
            pd = None if out is None else (out[part][vid] if part < 2 else (out[2][vid] if vid < len(out[2]) else None))
            dense = getattr(pd, "dense", None)
            blocks.append(dense.detach().float() if dense is not None and dense.dim() == 2
                          else torch.zeros((0, 0), dtype=torch.float32, device=device))
    blocks.append((out[3].detach().float() if out is not None else torch.zeros((), device=device)).reshape(1, 1))
    return blocks


def merge_frame_shards(per_rank_blocks: Sequence[Sequence[torch.Tensor]], plan: FrameShardPlan, kw: dict,
                       dummy_str: str = "$$$"):
    """forward()'s four results for the whole call from every rank's `local_blocks`."""
    from .prob_dict import ProbDict
    B = _vids(kw)
    object_ids, pairs = kw["batched_object_ids"], kw["batched_obj_pairs"]
    dev = per_rank_blocks[0][-1].device
    cate_out, unary_out, binary_out = {v: {} for v in range(B)}, {v: {} for v in range(B)}, []
    for vid in range(B):
        names = list(kw["batched_names"][vid]) or [dummy_str]
        ukws = list(kw["batched_unary_kws"][vid]) or [dummy_str]
        bkws = list(kw["batched_binary_kws"][vid]) or [dummy_str]
        g_objs = [i for i, (v, _, _) in enumerate(object_ids) if v == vid]          # global column order
        col_of = {i: c for c, i in enumerate(g_objs)}
        ids_vid = [(object_ids[i][1], object_ids[i][2]) for i in g_objs]
        group_of, oids = {}, []
        for (_, oid) in ids_vid:
            if oid not in group_of:
                group_of[oid] = len(oids)
                oids.append(oid)
        unary_dense = cate_sum = cate_cnt = None
        for r in range(plan.world):
            mine = [i for i in plan.obj_pos[r] if object_ids[i][0] == vid]
            if not mine:
                continue
            cm, ud = per_rank_blocks[r][3 * vid], per_rank_blocks[r][3 * vid + 1]
            if unary_dense is None:
                unary_dense = torch.zeros((ud.shape[0], len(g_objs)), dtype=torch.float32, device=dev)
                cate_sum = torch.zeros((cm.shape[0], len(oids)), dtype=torch.float32, device=dev)
                cate_cnt = torch.zeros(len(oids), dtype=torch.float32, device=dev)
            unary_dense[:, torch.tensor([col_of[i] for i in mine], device=dev)] = ud
            l_group, l_oids, l_cnt = {}, [], []
            for i in mine:                                  # the rank's own groups, in its order of appearance
                oid = object_ids[i][2]
                if oid not in l_group:
                    l_group[oid] = len(l_oids)
                    l_oids.append(oid)
                    l_cnt.append(0)
                l_cnt[l_group[oid]] += 1
            gidx = torch.tensor([group_of[o] for o in l_oids], device=dev)
            cnt = torch.tensor(l_cnt, dtype=torch.float32, device=dev)
            cate_sum[:, gidx] += cm * cnt
            cate_cnt[gidx] += cnt
        if unary_dense is not None:
            uniq: Dict[str, list] = {}
            for r_, nm in enumerate(names):
                if nm != dummy_str:
                    uniq.setdefault(nm, []).append(r_)
            c_rows = (list(enumerate(uniq.keys())) if any(len(v) > 1 for v in uniq.values())
                      else [(r_, nm) for r_, nm in enumerate(names) if nm != dummy_str])
            cate_out[vid] = ProbDict(cate_sum / cate_cnt, c_rows, list(enumerate(oids)), lambda oid, nm: (oid, nm), "col_major")
            u_rows = [(r_, nm) for r_, nm in enumerate(ukws) if nm != dummy_str]
            unary_out[vid] = ProbDict(unary_dense, u_rows, list(enumerate(ids_vid)),
                                      lambda fo, nm: (fo[0], fo[1], nm), "row_major")
        g_pairs = [i for i, (v, _, _) in enumerate(pairs) if v == vid]
        if g_pairs:
            pcol = {i: c for c, i in enumerate(g_pairs)}
            dense = None
            for r in range(plan.world):
                mine = [i for i in plan.pair_pos[r] if pairs[i][0] == vid]
                if not mine:
                    continue
                bd = per_rank_blocks[r][3 * vid + 2]
                if dense is None:
                    dense = torch.zeros((bd.shape[0], len(g_pairs)), dtype=torch.float32, device=dev)
                dense[:, torch.tensor([pcol[i] for i in mine], device=dev)] = bd
            b_rows = [(r_, nm) for r_, nm in enumerate(bkws) if nm != dummy_str]
            cols = [(c, (pairs[i][1], pairs[i][2])) for c, i in enumerate(g_pairs)]
            binary_out.append(ProbDict(dense, b_rows, cols, lambda fp, nm: (fp[0], fp[1], nm), "row_major"))
        else:
            binary_out.append({})
    dummy = sum(per_rank_blocks[r][-1].reshape(()) for r in range(plan.world))
    return cate_out, unary_out, binary_out, dummy


def forward_frame_sharded(model, group=None, **kw):
    """One inference-mode forward() call with its frames spread over the ranks of `group`
    (one process per GPU); every rank returns the results of the whole call.  One all-gather
    of the dense probability blocks; no other data-path communication."""
    preds = kw.get("batched_binary_predicates")
    if preds and preds[0] is not None:
        raise ValueError("frame sharding covers the inference path (batched_binary_predicates=[None]); training "
                         "shards clips, as the reference does")
    if kw.get("output_embeddings"):
        raise ValueError("output_embeddings is not available from the frame-sharded call")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    plan = FrameShardPlan(kw["batched_object_ids"], kw["batched_obj_pairs"], world)
    out = model(**plan.local_kwargs(kw, rank)) if plan.obj_pos[rank] else None
    device = getattr(model, "device", torch.device("cpu"))
    gathered = all_gather_blocks(local_blocks(out, kw, device), group=group)
    return merge_frame_shards(gathered, plan, kw, kw.get("dummy_str", "$$$"))

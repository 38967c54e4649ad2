"""Data-parallel sharding of the scoring path over the GPUs of one node.

Clips (and, inside a clip, frames) are independent through crop -> encode ->
similarity, so they shard with no data-path collective; the only exchange is the
all-gather of the dense probability blocks (the reference gathers pickled metric
dicts with all_gather_object every iteration, scripts/eval_sgclip.py:495-496).
One process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI) on
the GPUs, "gloo" in CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence

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

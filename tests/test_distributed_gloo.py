"""world_size-2 test of the multi-GPU exchange on CPU (gloo): ranks hold probability
blocks of different shapes, all-gather them, and every rank reconstructs all blocks."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from laser_amd.distributed import all_gather_blocks, balance_by_cost, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _blocks_for(rank):
    g = torch.Generator().manual_seed(100 + rank)
    n_obj, n_pair = 5 + 3 * rank, 11 - 4 * rank
    return [torch.rand(4, 3 + rank, generator=g), torch.rand(7, n_obj, generator=g), torch.rand(6, n_pair, generator=g),
            torch.rand((), generator=g)]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        got = all_gather_blocks(_blocks_for(rank))
        ok = len(got) == world
        for r in range(world):
            want = _blocks_for(r)
            ok = ok and len(got[r]) == len(want) and all(g.shape == w.shape and torch.equal(g, w) for g, w in zip(got[r], want))
        # clip sharding: every clip scored exactly once across ranks
        mine = list(shard_range(5, rank, world))
        counts = torch.zeros(5)
        counts[mine] = 1
        dist.all_reduce(counts)
        ok = ok and bool((counts == 1).all())
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_all_gather_blocks_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
    assert balance_by_cost([1, 1, 1, 1], 2) == [[0, 2], [1, 3]]

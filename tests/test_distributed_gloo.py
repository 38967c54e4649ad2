"""world_size-2 test of the multi-GPU exchange on CPU (gloo): ranks hold probability
blocks of different shapes, all-gather them, and every rank reconstructs all blocks."""
import os
import socket
import zlib

import pytest
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


# ---------------------------------------------------------------------------
# frame-parallel forward(): split / all-gather / merge logic with a stand-in scorer on CPU
# ---------------------------------------------------------------------------
class _FakeScorer:
    """Stands in for PredicateModel on CPU: same argument list and result layout as forward()
    (ProbDicts over dense tensors), with probabilities that are pure functions of the keys."""
    device = torch.device("cpu")

    @staticmethod
    def _p(*key):
        h = 0
        for k in key:
            h = (h * 1000003 + (zlib.crc32(k.encode()) if isinstance(k, str) else int(k) + 17)) % 2147483647
        return (h % 10007) / 10007.0

    def __call__(self, batched_video_ids, batched_videos, batched_masks, batched_bboxes, batched_names, batched_object_ids,
                 batched_unary_kws, batched_binary_kws, batched_obj_pairs, batched_video_splits, batched_binary_predicates,
                 allow_dummy_pair=True, **_):
        from laser_amd.prob_dict import ProbDict
        B = len(batched_video_ids)
        assert len(batched_masks) == len(batched_object_ids) == len(batched_bboxes)
        cate, unary, binary = {v: {} for v in range(B)}, {v: {} for v in range(B)}, []
        dummy = torch.zeros(())
        for vid in range(B):
            objs = [(f, o) for (v, f, o) in batched_object_ids if v == vid]
            if objs:
                names, ukws = batched_names[vid], batched_unary_kws[vid]
                per_crop = torch.tensor([[self._p(n, vid, f, o) for (f, o) in objs] for n in names])
                oids = []
                for _, o in objs:
                    if o not in oids:
                        oids.append(o)
                mean = torch.stack([per_crop[:, [i for i, (_, o2) in enumerate(objs) if o2 == o]].mean(dim=1) for o in oids], dim=1)
                cate[vid] = ProbDict(mean, list(enumerate(names)), list(enumerate(oids)), lambda oid, nm: (oid, nm), "col_major")
                ud = torch.tensor([[self._p("u", n, vid, f, o) for (f, o) in objs] for n in ukws])
                unary[vid] = ProbDict(ud, list(enumerate(ukws)), list(enumerate(objs)), lambda fo, nm: (fo[0], fo[1], nm), "row_major")
                dummy = dummy + 0.001 * len(objs)
            prs = [(f, p) for (v, f, p) in batched_obj_pairs if v == vid]
            if prs:
                bk = batched_binary_kws[vid]
                bd = torch.tensor([[self._p("b", n, vid, f, p[0], p[1]) for (f, p) in prs] for n in bk])
                binary.append(ProbDict(bd, list(enumerate(bk)), list(enumerate(prs)), lambda fp, nm: (fp[0], fp[1], nm), "row_major"))
            else:
                binary.append({})
        if not batched_obj_pairs and allow_dummy_pair:
            dummy = dummy + 0.5
        return cate, unary, binary, dummy


def _sharding_batch(with_pairs=True):
    rng = torch.Generator().manual_seed(4)
    object_ids, pairs = [], []
    for vid, frames in enumerate((4, 3)):
        for fid in range(frames):
            n = int(torch.randint(1, 5, (1,), generator=rng))
            oids = torch.randperm(6, generator=rng)[:n].tolist()
            object_ids += [(vid, fid, o) for o in oids]
            if with_pairs:
                pairs += [(vid, fid, (a, b)) for a in oids for b in oids if a != b][:5]
    n = len(object_ids)
    return dict(batched_video_ids=[0, 1], batched_videos=[None] * 7, batched_masks=[f"mask{i}" for i in range(n)],
                batched_bboxes=[{"x1": i, "y1": 0, "x2": i + 1, "y2": 1} for i in range(n)],
                batched_names=[["cat", "dog", "tree"], ["cat", "car"]], batched_object_ids=object_ids,
                batched_unary_kws=[["red", "tall"], ["small"]], batched_binary_kws=[["on", "near"], ["under", "on", "by"]],
                batched_obj_pairs=pairs, batched_video_splits=[4, 7], batched_binary_predicates=[None, None])


def _same_results(a, b, tol=1e-6):
    ok = True
    for part in (0, 1):
        for v in a[part]:
            ok = ok and list(a[part][v].keys()) == list(b[part][v].keys())
            if len(a[part][v]):
                ok = ok and torch.allclose(a[part][v].values_tensor(), b[part][v].values_tensor(), atol=tol)
    ok = ok and len(a[2]) == len(b[2])
    for x, y in zip(a[2], b[2]):
        ok = ok and list(x.keys()) == list(y.keys())
        if len(x):
            ok = ok and torch.allclose(x.values_tensor(), y.values_tensor(), atol=tol)
    return ok and abs(float(a[3]) - float(b[3])) <= tol


def _shard_worker(rank, world, port, q):
    from laser_amd.distributed import forward_frame_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for with_pairs in (True, False):
            kw = _sharding_batch(with_pairs)
            want = _FakeScorer()(**kw)
            got = forward_frame_sharded(_FakeScorer(), **kw)
            ok = ok and _same_results(want, got)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_forward_frame_sharded_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_frame_shard_plan_covers_every_crop_once():
    from laser_amd.distributed import FrameShardPlan
    kw = _sharding_batch()
    for world in (1, 2, 3, 8):
        plan = FrameShardPlan(kw["batched_object_ids"], kw["batched_obj_pairs"], world)
        assert sorted(i for r in plan.obj_pos for i in r) == list(range(len(kw["batched_object_ids"])))
        assert sorted(i for r in plan.pair_pos for i in r) == list(range(len(kw["batched_obj_pairs"])))
        for r in range(world):                      # a pair lives with both of its objects
            frames = {kw["batched_object_ids"][i][:2] for i in plan.obj_pos[r]}
            assert all(kw["batched_obj_pairs"][i][:2] in frames for i in plan.pair_pos[r])


def _grad_worker(rank, world, port, q):
    from laser_amd.training import allreduce_gradients
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(n)) for n in (5, 300, 1, 64)]
        params.append(torch.nn.Parameter(torch.zeros(7), requires_grad=False))
        for i, p in enumerate(params[:4]):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        if rank == 1:
            params[2].grad = None        # a parameter this rank's batch never touched
        allreduce_gradients(params, bucket_bytes=1024)     # several buckets
        want = [(1 + 2) / 2 * (i + 1) for i in range(4)]
        want[2] = 1 * 3 / 2
        ok = all(torch.allclose(p.grad, torch.full_like(p, w)) for p, w in zip(params[:4], want))
        ok = ok and params[4].grad is None
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_allreduce_gradients_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


class _StubTower:
    """What GradientReducer needs from a tower: a bucket key and the pending-backward count it resets."""

    def __init__(self, key):
        self.bucket_key, self.reducer, self.backward_pending = key, None, 0


def _reducer_worker(rank, world, port, q):
    from laser_amd.training import FlatParameters, GradientReducer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sizes = {"a.vision": 301, "a.text": 64, "b.vision": 1000, "b.text": 7}
        named = [(k, torch.nn.Parameter(torch.zeros(n))) for k, n in sizes.items()]
        flat = FlatParameters(named)
        buckets, off = [], 0
        for k, _ in named:
            lo, cnt, _ = flat.index[k]
            end = lo + (cnt + flat.ALIGN - 1) // flat.ALIGN * flat.ALIGN
            buckets.append((k, lo, end))
            off = end
        assert off == flat.total
        towers = [_StubTower(k) for k in sizes]
        ok, logs = True, []
        for wire in (None, torch.bfloat16):
            red = GradientReducer(flat, buckets, max_bucket_bytes=1024, wire_dtype=wire, towers=towers)   # b.vision: 4 pieces
            assert all(t.reducer is red for t in towers)
            for step in range(3):
                flat.zero_grad()
                red.begin()
                for i, (k, p) in enumerate(named):
                    p.grad.fill_(float((rank + 1) * (i + 1) * (step + 1)))
                # backward finishes the towers in an order of its own, different per rank; rank 1's batch never
                # touches b.text (its gradient stays zero and the bucket leaves in finish())
                done = ["b.vision", "a.text", "b.text", "a.vision"] if rank == 0 else ["a.text", "a.vision", "b.vision"]
                if rank == 1:
                    named[3][1].grad.zero_()
                for k in done:
                    red.ready(k)
                red.finish()
                logs.append(tuple(red.launch_log))
                for i, (k, p) in enumerate(named):
                    want = (1 + 2) / 2 * (i + 1) * (step + 1)
                    if k == "b.text":
                        want = 1 * (i + 1) * (step + 1) / 2
                    ok = ok and torch.allclose(p.grad, torch.full_like(p, want), rtol=1e-2 if wire is not None else 1e-6)
                if step >= 1:    # rank 0's completion order of the first step, on every rank
                    ok = ok and logs[-1] == ("b.vision", "a.text", "b.text", "a.vision")
        gathered = [None] * world
        dist.all_gather_object(gathered, logs)
        ok = ok and all(g == gathered[0] for g in gathered)      # same collectives in the same order everywhere
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_gradient_reducer_bucket_per_tower_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_gradient_reducer_single_process_is_a_no_op():
    from laser_amd.training import FlatParameters, GradientReducer
    named = [("x", torch.nn.Parameter(torch.zeros(5))), ("y", torch.nn.Parameter(torch.zeros(8)))]
    flat = FlatParameters(named)
    red = GradientReducer(flat, [("x", 0, 8), ("y", 8, 16)])
    named[0][1].grad.fill_(3.0)
    red.begin()
    red.ready("y")
    assert red.launch_log == ["y"]          # layout order reversed: y leaves first
    red.finish()
    assert red.launch_log == ["y", "x"] and float(named[0][1].grad[0]) == 3.0
    with pytest.raises(AssertionError):
        GradientReducer(flat, [("x", 0, 8)])

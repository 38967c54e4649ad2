#!/usr/bin/env python3
"""Headline benchmark of the SGCLIP scoring path on MI355X.

Metric (BASELINE.json): object+pair crops scored per second.  One "crop scored"
is one object crop or one ordered-pair crop taken through
crop/blend/resize/normalize -> CLIP ViT encode (object crops through the
category AND the unary tower, pair crops through the binary tower) -> cosine
similarity against the vocabulary -> softmax.  Workload = BASELINE.json
configs[1]: one synthetic clip of 32 frames x 16 objects (240 ordered pairs per
frame), 224^2 crops, 35 category + 200 unary + 50 binary keywords, ViT-B/32,
bf16 MFMA with fp32 accumulation.  A step is ONE call of PredicateModel.forward()
on one clip per GPU (8 192 crops, 8 704 image encodes, 285 text encodes: the
vocabulary cache is off, so the text towers run every step as in the reference);
frames, masks and weights are resident in HBM when the timed region starts, the
crop tables are built and uploaded inside the call.  With N > 1 every rank scores
its own clip (weak scaling, clips are independent) and the dense probability
blocks are all-gathered over RCCL each step.

Usage: python bench.py --gpus N --steps K --warmup W.  With N > 1 and no launcher environment
(WORLD_SIZE unset) the script starts `python -m torch.distributed.run --nproc-per-node N` on itself,
one process per GPU, as the reference's drivers spawn theirs (scripts/eval_sgclip.py:643); under a
launcher it is one of the N ranks.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from laser_amd import _lib, ops  # noqa: E402
from laser_amd.config import config_by_name  # noqa: E402
from laser_amd.predicate_model import PredicateModel, resolve_union_box, slice_box  # noqa: E402
from laser_amd.synthetic import make_clip_batch  # noqa: E402
from laser_amd.tokenizer import HashTokenizer  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak
MFMA_PEAK_TFLOPS = 2500.0    # MI355X dense bf16 MFMA spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="vit-b-32")
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--objects", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-pairs", type=int, default=240)
    ap.add_argument("--cpu-sample-frames", type=int, default=3, help="frames of the clip the CPU baseline scores (5-6 s each)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the all-gather even with one rank (rehearsal)")
    ap.add_argument("--no-forward-api", action="store_true",
                    help="skip timing PredicateModel.forward() from host numpy inputs (PCIe + host index build included)")
    ap.add_argument("--no-c4", action="store_true",
                    help="skip the ViT-L/14-336 sub-result (1 frame x 64 objects x 4 032 pairs)")
    ap.add_argument("--no-train", action="store_true",
                    help="skip the train-step sub-result (two clips of the headline shape, forward + backward + Adam)")
    ap.add_argument("--distinct-frames", type=int, default=0,
                    help="frame pixel pool size (0 = every frame distinct, as the workload states)")
    return ap.parse_args()


def self_launch(args):
    """--gpus N > 1 without a launcher: start one process per GPU through torch.distributed.run and hand back its
    exit code.  Runs before anything touches the GPU (a process that has initialised HIP must not be replaced)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


class ClipPlan:
    """One synthetic clip staged in HBM (frames and masks), its forward() arguments and the byte / count accounting
    of its crops."""

    def __init__(self, model: PredicateModel, batch, device):
        cfg = model.cfg
        self.batch = batch
        self.kw = batch.forward_kwargs()
        self.di = model.stage_inputs(batch.batched_videos, batch.batched_masks)
        di = self.di
        self.n_obj, self.n_pair = len(batch.batched_object_ids), len(batch.batched_obj_pairs)
        index = {}
        self.algo_crop_bytes = 0.0
        for pos, ((vid, fid, oid), bbox) in enumerate(zip(batch.batched_object_ids, batch.batched_bboxes)):
            H, W = di.frame_hw[fid]
            xa, ya, xb, yb = slice_box(*resolve_union_box([bbox]), W, H)
            index[(fid, oid)] = pos
            self.algo_crop_bytes += (3 + 1) * float(xb - xa) * (yb - ya) + 2 * 3 * cfg.image * cfg.image
        for (vid, fid, (a, b)) in batch.batched_obj_pairs:
            H, W = di.frame_hw[fid]
            pa, pb = index[(fid, a)], index[(fid, b)]
            xa, ya, xb, yb = slice_box(*resolve_union_box([batch.batched_bboxes[pa], batch.batched_bboxes[pb]]), W, H)
            self.algo_crop_bytes += (3 + 2) * float(xb - xa) * (yb - ya) + 2 * 3 * cfg.image * cfg.image
        tok = model.clip_tokenizer
        self.text_positions = []
        for prefix, kws in (("clip_cate_model", batch.batched_names[0]), ("clip_unary_model", batch.batched_unary_kws[0]),
                            ("clip_binary_model", batch.batched_binary_kws[0])):
            t = tok(list(kws), return_tensors="pt", max_length=75, truncation=True, padding="max_length")
            # positions after the longest keyword's end-of-text token cannot reach any pooled output (causal tower)
            self.text_positions.append(model._packed()[prefix][1].pooled_length(t["input_ids"]))


def score_clip(model: PredicateModel, plan: ClipPlan, concurrent: bool = True):
    """One step: PredicateModel.forward() on the staged clip, the call a user of the reference makes
    (scripts/eval_sgclip.py:339-352), text towers included (the vocabulary cache is off in this benchmark).
    Returns the dense blocks behind the returned mappings.  concurrent=False: every kernel on one stream."""
    model.concurrent_towers = concurrent and os.environ.get("BENCH_STREAMS", "1") != "0"
    cate, unary, binary, _ = model(**plan.kw, device_inputs=plan.di)
    model.concurrent_towers = True
    return cate[0].dense, unary[0].dense, binary[0].dense


def pmc_traffic(substr):
    """Average HBM bytes per launch of the kernels whose name contains `substr`, from the
    rocprofv3 PMC passes over this same command (the newest profiles/r*_pmc_bench_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate runs, FETCH_SIZE doubled as gfx950
    requires for wide streaming reads).  None when no such file exists."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_bench_traffic.json")))
    if not found:
        return None
    path = found[-1]
    ks = json.load(open(path))["kernels"]
    subs = (substr,) if isinstance(substr, str) else tuple(substr)
    tot = sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for k, v in ks.items() if any(s in k for s in subs))
    n = sum(v["launches"] for k, v in ks.items() if any(s in k for s in subs))
    return tot / n if n else None


def pmc_traffic_per_call(substrs, calls_per_step):
    """HBM bytes per library call for a call that launches several kernels (the crop call: one launch per class and
    kernel variant): the bytes of all kernels whose name contains one of `substrs` in the newest PMC file, divided
    by the passes that profile made over the step and by the calls one step makes.  None without such a file."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_bench_traffic.json")))
    if not found or not calls_per_step:
        return None
    d = json.load(open(found[-1]))
    tot = sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for k, v in d["kernels"].items() if any(s in k for s in substrs))
    return tot / d.get("passes", 3) / calls_per_step if tot else None


def timing_read(kind):
    lib = _lib.load()
    ms, n, work = C.c_double(), C.c_int64(), C.c_double()
    _lib.check(lib.sgclip_timing_read(kind, C.byref(ms), C.byref(n), C.byref(work)))
    return ms.value, n.value, work.value


def calibrate(device):
    """What this box sustains on the two roofline axes: a device-to-device stream copy and a large
    square bf16 GEMM of this library (random operands), next to the vendor peaks used for `frac`."""
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=device).random_(0, 255)
    dst = torch.empty_like(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dst.copy_(src)
    e0.record()
    for _ in range(5):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 5 * 2 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    M = 8192
    A = torch.randn(M, M, device=device).to(torch.bfloat16)
    W = (torch.randn(M, M, device=device) * M ** -0.5).to(torch.bfloat16)
    bias = torch.zeros(M, device=device)
    C = torch.empty(M, M, device=device, dtype=torch.bfloat16)
    ops.gemm_bf16(A, W, C, _lib.EPI_BIAS_BF16, bias=bias)
    e0.record()
    for _ in range(5):
        ops.gemm_bf16(A, W, C, _lib.EPI_BIAS_BF16, bias=bias)
    e1.record()
    torch.cuda.synchronize()
    tf = 5 * 2.0 * M ** 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    return {"stream_copy_GBps": copy_gbs, "hbm_peak_GBps": HBM_PEAK_GBS, "gemm_8192_bf16_TFLOPs": tf,
            "mfma_peak_TFLOPs": MFMA_PEAK_TFLOPS}


def cpu_baseline(args, cfg, sd, tok, batch):
    """The CPU oracle (a port of the reference path) on a bounded sample of the same clip: its first
    --cpu-sample-frames frames with all their objects and the first --cpu-sample-pairs pairs of each."""
    from laser_amd.weights import split_predicate_state_dict
    from oracle.forward_ref import OraclePredicateModel
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    O = args.objects
    kw = batch.forward_kwargs()
    nf = max(1, min(args.cpu_sample_frames, args.frames))
    kw["batched_videos"] = kw["batched_videos"][:nf]
    kw["batched_masks"] = kw["batched_masks"][:nf * O]
    kw["batched_bboxes"] = kw["batched_bboxes"][:nf * O]
    kw["batched_object_ids"] = kw["batched_object_ids"][:nf * O]
    kw["batched_obj_pairs"] = [p for f in range(nf)
                               for p in [q for q in kw["batched_obj_pairs"] if q[1] == f][: args.cpu_sample_pairs]]
    kw["batched_video_splits"] = [nf]
    oracle = OraclePredicateModel(cfg, split_predicate_state_dict(sd), tok)
    try:
        import PIL  # noqa: F401
        oracle.pillow = True        # the reference's processor resizes with Pillow's C code; bit-equal to the restatement
    except ImportError:
        pass
    t0 = time.perf_counter()
    oracle(**kw)
    dt = time.perf_counter() - t0
    crops = (nf * O + len(kw["batched_obj_pairs"])) / nf        # per frame
    # the text towers run once per forward (per clip), the rest scales with the frames:
    # clip estimate = text + frames * (mean frame time without text), crops scaled the same way
    t_text = oracle.timers["text"]
    per_frame = (dt - t_text) / nf
    clip_s = t_text + args.frames * per_frame
    return {"value": args.frames * crops / clip_s, "unit": "crops/s", "cores": cores, "kind": "port",
            "sample": f"the first {nf} frames of the same clip: per frame {O} object crops (2 encodes each) + "
                      f"{len(kw['batched_obj_pairs']) // nf} pair crops in {per_frame:.1f} s, plus the once-per-clip text encode of "
                      f"{len(kw['batched_names'][0]) + len(kw['batched_unary_kws'][0]) + len(kw['batched_binary_kws'][0])} keywords "
                      f"in {t_text:.1f} s; clip rate = {args.frames} x crops / (text + {args.frames} x frame time); fp32 torch CPU, "
                      f"resize by {'Pillow (C)' if oracle.pillow else 'the numpy restatement of Pillow'}",
            "stage_seconds": {k: round(v, 3) for k, v in oracle.timers.items()}}


def kernel_pass(model, plan, tsteps):
    """Per-kernel durations, live, with HIP events on the launch stream: everything queued on ONE stream so that an
    event pair brackets its kernel alone.  -> dict of (ms, launches, work) per timing kind, per `tsteps` steps."""
    lib = _lib.load()
    _lib.check(lib.sgclip_timing_enable(1, 65536))
    for _ in range(tsteps):
        score_clip(model, plan, concurrent=False)
    torch.cuda.synchronize()
    kinds = {k: timing_read(k) for k in (0, 1, 2, 3, 4, 8, 9)}
    _lib.check(lib.sgclip_timing_enable(0, 0))
    return kinds


def c4_sub_result(device, tok_cls):
    """BASELINE.json configs[3] on one GPU: ViT-L/14 at 336^2, one 1280 x 720 frame with 64 objects and all 4 032
    ordered pairs (4 096 crops = 4 160 image encodes of 577 tokens), inputs resident in HBM."""
    cfg = config_by_name("vit-l-14-336")
    tok = tok_cls(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    model = PredicateModel(None, None, device, "vit-l-14-336", tokenizer=tok, init_seed=9).eval()
    for p in ("clip_cate_model", "clip_unary_model", "clip_binary_model"):
        model._modules[p].logit_scale.data.fill_(float(np.log(100.0)))
    model.invalidate_packed_weights()
    model.cache_text_features = False
    batch = make_clip_batch(300, n_videos=1, frames=1, objects=64)
    plan = ClipPlan(model, batch, device)
    score_clip(model, plan)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 2
    for _ in range(n):
        score_clip(model, plan)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    kinds = kernel_pass(model, plan, 1)
    g_ms = sum(kinds[k][0] for k in (0, 1, 2, 3, 4))
    g_fl = sum(kinds[k][2] for k in (0, 1, 2, 3, 4))
    a_ms, _, a_fl = kinds[9]
    crops = plan.n_obj + plan.n_pair
    out = {"workload": "1 frame x 64 objects x 4032 ordered pairs at 1280x720, 336^2 crops, 35 + 200 + 50 keywords, "
                       "vit-l-14-336, random-init weights",
           "ms_per_frame": ms, "crops_per_s": crops / (ms * 1e-3), "crops": crops,
           "image_encodes": 2 * plan.n_obj + plan.n_pair,
           "gemm_ms": g_ms, "gemm_tflops": g_fl / (g_ms * 1e-3) / 1e12 if g_ms else 0.0,
           "attention_ms": a_ms, "attention_tflops": a_fl / (a_ms * 1e-3) / 1e12 if a_ms else 0.0,
           "crop_ms": kinds[8][0], "crop_GBps": kinds[8][2] / (kinds[8][0] * 1e-3) / 1e9 if kinds[8][0] else 0.0,
           "executed_tflops": (g_fl + a_fl) / (ms * 1e-3) / 1e12,
           "note": "kernel times from a one-stream pass with HIP events; ms_per_frame with the towers on side streams"}
    del model, plan
    torch.cuda.empty_cache()
    return out


def train_sub_result(args, device, tok):
    """BASELINE.json configs[4] on one GPU: one train step (forward under autograd, backward, Adam) over two clips
    of the headline shape.  The Scallop-derived loss of scripts/train_sgclip.py:781-836 needs scallopy on the host
    (absent here): a binary cross-entropy of every returned probability against seeded random targets stands in,
    so that every output carries a gradient."""
    from laser_amd import training
    model = PredicateModel(None, None, device, args.model, tokenizer=tok, init_seed=5).enable_training()
    flat = model.flatten_parameters()
    opt = training.Adam(flat, lr=1e-6)
    model.set_recompute_chunk(2048)
    batch = make_clip_batch(500, n_videos=2, frames=args.frames, objects=args.objects, distinct_frames=8)
    kw = batch.forward_kwargs()
    di = model.stage_inputs(kw["batched_videos"], kw["batched_masks"])
    n_crops = len(kw["batched_object_ids"]) + len(kw["batched_obj_pairs"])
    gen = torch.Generator(device=device).manual_seed(0)

    def step():
        cate, unary, binary, dummy = model(**kw, device_inputs=di)
        loss = 0.0 * dummy
        for d in [cate[v] for v in cate] + [unary[v] for v in unary] + list(binary):
            p = d.dense.clamp(1e-6, 1 - 1e-6)
            t = (torch.rand(p.shape, device=device, generator=gen) < 0.1).float()
            loss = loss + torch.nn.functional.binary_cross_entropy(p, t)
        opt.zero_grad()
        loss.backward()
        opt.step()

    step()
    torch.cuda.synchronize()
    n = 2
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    lib = _lib.load()
    _lib.check(lib.sgclip_timing_enable(1, 65536))
    step()
    torch.cuda.synchronize()
    kinds = {k: timing_read(k) for k in (0, 1, 2, 3, 4, 5, 9)}
    _lib.check(lib.sgclip_timing_enable(0, 0))
    g_ms = sum(kinds[k][0] for k in (0, 1, 2, 3, 4))
    g_fl = sum(kinds[k][2] for k in (0, 1, 2, 3, 4))
    w_ms, _, w_fl = kinds[5]
    out = {"workload": f"2 clips x {args.frames} frames x {args.objects} objects x {args.objects * (args.objects - 1)} ordered "
                       f"pairs/frame at 1280x720, {args.model}, forward + backward + Adam, image towers in chunks of 2048 crops "
                       f"(a chunk's activations stay in HBM while it has room, else its forward is recomputed during "
                       f"backward); LOSS IS A STAND-IN: binary cross-entropy of every returned probability against random "
                       f"targets, not the reference's Scallop-derived contrastive loss (scripts/train_sgclip.py:729-836, host "
                       f"CPU, scallopy absent here)",
           "ms_per_step": ms, "crops_per_s": n_crops / (ms * 1e-3), "crops": n_crops,
           "gemm_nt_ms": g_ms, "gemm_nt_tflops": g_fl / (g_ms * 1e-3) / 1e12 if g_ms else 0.0,
           "gemm_tn_weight_grad_ms": w_ms, "gemm_tn_weight_grad_tflops": w_fl / (w_ms * 1e-3) / 1e12 if w_ms else 0.0,
           "roofline_frac_mfma": (g_fl + w_fl) / ((g_ms + w_ms) * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS if g_ms + w_ms else 0.0,
           "peak_mem_gib": torch.cuda.max_memory_allocated() / 2 ** 30,
           "note": "GEMM times from HIP events around every launch of one step (forward, any recompute, data and "
                   "weight gradients)"}
    del model, flat, opt, di
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.force_dist:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        # N ranks share the node's host cores: an even share each, at most 16 (os.cpu_count() shows the whole machine,
        # and N x the default thread pools oversubscribe it)
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // max(world, 8))))
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:   # --force-dist rehearsal without a launcher
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)

    cfg = config_by_name(args.model)
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    model = PredicateModel(None, None, device, args.model, tokenizer=tok, init_seed=7).eval()
    for p in ("clip_cate_model", "clip_unary_model", "clip_binary_model"):
        model._modules[p].logit_scale.data.fill_(float(np.log(100.0)))
    model.invalidate_packed_weights()
    model.cache_text_features = False     # the text towers run every step, as in the reference (llava_clip_model_v3.py:137-186)
    # every rank scores its own clip (different seed); every frame has its own pixels unless --distinct-frames
    batch = make_clip_batch(100 + rank, n_videos=1, frames=args.frames, objects=args.objects,
                            distinct_frames=args.distinct_frames or None)
    plan = ClipPlan(model, batch, device)
    crops_per_step = plan.n_obj + plan.n_pair
    encodes_per_step = 2 * plan.n_obj + plan.n_pair

    gathered = None

    def step():
        """PredicateModel.forward() on the resident clip (+ the all-gather of its dense blocks with N > 1)."""
        nonlocal gathered
        cate_mean, unary, binary = score_clip(model, plan)
        if dist is not None:
            flat = torch.cat([cate_mean.reshape(-1), unary.reshape(-1), binary.reshape(-1)])
            if gathered is None:
                gathered = torch.empty(world * flat.numel(), dtype=flat.dtype, device=device)
            dist.all_gather_into_tensor(gathered, flat)
        return cate_mean, unary, binary

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * crops_per_step * args.steps / elapsed

    # p50 latency of one clip scored alone (each step fenced on both sides), outside the timed region above
    lat = []
    for _ in range(max(5, min(args.steps, 9))):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        lat.append(1e3 * (time.perf_counter() - t1))
    lat.sort()
    latency_p50_ms = lat[len(lat) // 2]

    # per-kernel durations in a separate pass, so that event recording does not perturb the number above
    tsteps = min(args.steps, 2)
    kinds = kernel_pass(model, plan, tsteps)
    g_ms = sum(kinds[k][0] for k in (0, 1, 2, 3, 4))
    g_n = sum(kinds[k][1] for k in (0, 1, 2, 3, 4))
    g_fl = sum(kinds[k][2] for k in (0, 1, 2, 3, 4))
    gemm_tflops = g_fl / (g_ms * 1e-3) / 1e12 if g_ms else 0.0
    crop_ms, crop_n, crop_bytes = kinds[8]
    crop_gbs = crop_bytes / (crop_ms * 1e-3) / 1e9 if crop_ms else 0.0
    step_ms_timed = {"gemm": g_ms / tsteps, "attention": kinds[9][0] / tsteps, "crop": crop_ms / tsteps}
    executed_flops_per_step = (g_fl + kinds[9][2]) / tsteps

    result = {
        "metric": "object+pair crops scored/sec", "value": value, "unit": "crops/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"synthetic clip {args.frames} frames x {args.objects} objects x {args.objects * (args.objects - 1)} "
                               f"ordered pairs/frame at 1280x720, {cfg.image}^2 crops, 35 category + 200 unary + 50 binary "
                               f"keywords, {args.model}, one clip per GPU per step, random-init weights",
                   "crops_per_step_per_gpu": crops_per_step, "image_encodes_per_step_per_gpu": encodes_per_step,
                   "text_encodes_per_step_per_gpu": 285,
                   "text_positions_run": plan.text_positions,
                   "timed_call": "PredicateModel.forward(..., device_inputs=staged clip): host index build, crop tables, "
                                 "towers on four HIP streams, lazy mappings; vocabulary cache off", "parallelism": f"dp{world} (clips)"},
        "frames_per_s": world * args.frames * args.steps / elapsed,
        "clip_latency_ms": ms_per_step, "clip_latency_p50_ms": latency_p50_ms,
        "roofline": {"kernel": "sgclip::gemm_*_kernel + sgclip::qkv_attn_kernel (bf16 MFMA GEMMs of the three ViT towers + text "
                               "towers with all epilogues; the q/k/v GEMM carries the fused attention)",
                     "bound": "mfma", "achieved": gemm_tflops, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": gemm_tflops / MFMA_PEAK_TFLOPS, "traffic": pmc_traffic(("gemm", "qkv_attn")),
                     "algorithmic_flops_per_launch": g_fl / max(g_n, 1),
                     "launches_per_step": g_n / tsteps, "avg_launch_ms": g_ms / max(g_n, 1),
                     "note": "launch durations from HIP events in a pass that queues the towers on ONE stream, so an "
                             "event pair brackets its kernel alone (rocprof summary of that mode: BENCH_STREAMS=0); the "
                             "timed steps behind `value` overlap the towers on four streams"},
        "roofline_crop": {"kernel": "sgclip::crop_mfma_kernel (+ sgclip::crop_kernel for the crops it does not take), per "
                                    "sgclip_crop_preprocess call",
                          "bound": "hbm", "achieved": crop_gbs, "peak": HBM_PEAK_GBS,
                          "unit": "GB/s", "frac": crop_gbs / HBM_PEAK_GBS,
                          "traffic": pmc_traffic_per_call(("crop_mfma_kernel", "crop_kernel"), crop_n / max(tsteps, 1)),
                          "algorithmic_bytes_per_launch": crop_bytes / max(crop_n, 1),
                          "avg_launch_ms": crop_ms / max(crop_n, 1), "algorithmic_bytes_per_step": plan.algo_crop_bytes},
        "kernel_ms_per_step": step_ms_timed,
        # FLOPs the kernels executed (GEMM + attention launches of all towers; the last layer runs on the pooled
        # rows only, so this is below encodes x the reference's FLOPs per encode) over the timed step
        "whole_step_tflops": executed_flops_per_step / (ms_per_step * 1e-3) / 1e12,
        "executed_tflop_per_step": executed_flops_per_step / 1e12,
        "reference_tflop_per_step": encodes_per_step * cfg.vit_flops_per_encode() / 1e12,
    }
    if rank == 0 and world == 1:
        result["calibration"] = calibrate(device)
    if not args.no_forward_api and world == 1:
        # what a caller that holds host numpy inputs gets (PCIe + host packing included; never `value`); the
        # vocabulary is cached here, as in an evaluation loop over one dataset (model.eval())
        model.cache_text_features = True
        kw = plan.kw
        from laser_amd.masks import pack_masks
        packed = pack_masks(batch.batched_masks)            # done once, off line (the on-disk format)
        kwp = dict(kw, batched_masks=packed, batched_bboxes=None)

        def timed(fn, n=3):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n

        dt = timed(lambda: model(**kw))
        dt2 = timed(lambda: model(**kw, device_inputs=plan.di))
        dt3 = timed(lambda: model(**kwp))

        def pipelined(kwargs, masks, n=6):
            """A stream of n clips from host arrays: clip i + 1 is packed and uploaded (stage_inputs_async: worker
            thread + copy stream, double-buffered) while clip i is scored."""
            pending = model.stage_inputs_async(batch.batched_videos, masks)
            model(**kwargs, device_inputs=pending)                      # warm-up (allocates both buffer sets)
            pending = model.stage_inputs_async(batch.batched_videos, masks)
            model(**kwargs, device_inputs=pending)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pending = model.stage_inputs_async(batch.batched_videos, masks)
            for i in range(n):
                nxt = model.stage_inputs_async(batch.batched_videos, masks) if i + 1 < n else None
                model(**kwargs, device_inputs=pending)
                pending = nxt
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n

        dt4 = pipelined(kw, batch.batched_masks)
        dt5 = pipelined(kwp, packed)
        result["forward_api_packed_masks"] = {"host_inputs_ms": dt3 * 1e3, "host_inputs_crops_per_s": crops_per_step / dt3,
                                              "pipelined_host_inputs_ms": dt5 * 1e3,
                                              "pipelined_host_inputs_crops_per_s": crops_per_step / dt5,
                                              "mask_bytes_uploaded": packed.nbytes,
                                              "note": "masks arrive bit-packed; unpack + boxes on the GPU"}
        result["forward_api"] = {"host_inputs_ms": dt * 1e3, "host_inputs_crops_per_s": crops_per_step / dt,
                                 "staged_inputs_ms": dt2 * 1e3, "staged_inputs_crops_per_s": crops_per_step / dt2,
                                 "pipelined_host_inputs_ms": dt4 * 1e3,
                                 "pipelined_host_inputs_crops_per_s": crops_per_step / dt4,
                                 "pipelined_vs_staged": dt2 / dt4,
                                 "note": "PredicateModel.forward() from host numpy inputs, text features cached (eval): "
                                         "host_inputs = stage_inputs inside the call (pack + PCIe upload, then score); "
                                         "pipelined = stage_inputs_async of clip i + 1 while clip i is scored "
                                         "(scripts/eval_sgclip.py:339-352 loops over clips this way); staged = inputs "
                                         "already in HBM"}
        model.cache_text_features = False
    if rank == 0 and world == 1 and not args.no_c4:
        result["c4_l14_336"] = c4_sub_result(device, HashTokenizer)
    if rank == 0 and world == 1 and not args.no_train:
        result["train_step"] = train_sub_result(args, device, tok)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        result["cpu_baseline"] = cpu_baseline(args, cfg, sd, tok, batch)
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

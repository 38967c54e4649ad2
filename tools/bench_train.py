"""Times one SGCLIP train step (forward with saved activations, backward, Adam) on synthetic clips.

    python tools/bench_train.py [--model vit-b-32] [--videos 2] [--frames 8] [--objects 8] [--steps 5] [--warmup 2]

Prints one JSON line: ms per step, crops (object + pair) per second through forward + backward + update.
The loss is a stand-in for the Scallop-derived one (scripts/train_sgclip.py:781-836): binary cross-entropy of
every returned probability against seeded random targets, so every output carries a gradient."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from laser_amd import training  # noqa: E402
from laser_amd.config import config_by_name  # noqa: E402
from laser_amd.predicate_model import PredicateModel  # noqa: E402
from laser_amd.synthetic import make_clip_batch  # noqa: E402
from laser_amd.tokenizer import HashTokenizer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="vit-b-32")
    ap.add_argument("--videos", type=int, default=2)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--objects", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--loose", action="store_true", help="separate parameter tensors instead of the flat buffers")
    ap.add_argument("--recompute", type=int, default=0, help="recompute the image towers in backward, N crops at a time")
    ap.add_argument("--no-keep", action="store_true", help="with --recompute: recompute every chunk, keep no activations")
    ap.add_argument("--H", type=int, default=360)
    ap.add_argument("--W", type=int, default=640)
    a = ap.parse_args()
    cfg = config_by_name(a.model)
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    model = PredicateModel(None, None, "cuda:0", a.model, tokenizer=tok, init_seed=3).enable_training()
    flat = None if a.loose else model.flatten_parameters()
    opt = training.Adam(flat if flat is not None else model.parameters(), lr=1e-6)
    if a.recompute:
        model.set_recompute_chunk(a.recompute, keep_activations=not a.no_keep)
    b = make_clip_batch(5, n_videos=a.videos, frames=a.frames, objects=a.objects, H=a.H, W=a.W, k_cate=35, k_unary=50,
                        k_binary=50, min_box=24, max_box=min(a.H, a.W) - 8)
    kw = b.forward_kwargs()
    di = model.stage_inputs(kw["batched_videos"], kw["batched_masks"])
    n_crops = len(kw["batched_object_ids"]) + len(kw["batched_obj_pairs"])
    rng = np.random.default_rng(0)

    def step():
        cate, unary, binary, dummy = model(**kw, device_inputs=di)
        loss = 0.0 * dummy
        for v in cate:
            for d in (cate[v], unary[v]):
                p = d.dense.clamp(1e-6, 1 - 1e-6)
                t = torch.from_numpy((rng.random(p.shape) < 0.1).astype(np.float32)).to(p.device)
                loss = loss + torch.nn.functional.binary_cross_entropy(p, t)
        for d in binary:
            p = d.dense.clamp(1e-6, 1 - 1e-6)
            t = torch.from_numpy((rng.random(p.shape) < 0.1).astype(np.float32)).to(p.device)
            loss = loss + torch.nn.functional.binary_cross_entropy(p, t)
        opt.zero_grad()
        tf = time.perf_counter()
        loss.backward()
        opt.step()
        return tf

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fwd = 0.0
    for _ in range(a.steps):
        ts = time.perf_counter()
        tf = step()
        fwd += tf - ts
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"metric": "train step (forward + backward + Adam)", "ms_per_step": dt * 1e3,
                      "crops_per_s": n_crops / dt, "crops": n_crops, "host_ms_until_backward": fwd / a.steps * 1e3,
                      "config": {"model": a.model, "videos": a.videos, "frames": a.frames, "objects": a.objects},
                      "peak_mem_gib": torch.cuda.max_memory_allocated() / 2 ** 30}))


if __name__ == "__main__":
    main()

"""Golden vectors for the consumers either side of forward(), from the reference's own functions.

Runs ONLY in the build container (needs /root/reference).  Two reference functions are pure
(numpy / builtins only) but live in modules whose imports are not installed here, so each is
compiled on its own from its source file with `ast` and executed unmodified:

  * compute_metrics    scripts/eval_sgclip.py:97-203
  * bitmasks2bboxes    laser/loading.py:84-104

Writes consumers.npz / consumers.json next to this script.
Usage:  python tests/golden/make_consumer_golden.py
"""
from __future__ import annotations

import ast
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_function(rel_path: str, name: str, namespace: dict):
    src = open(os.path.join(REF, rel_path)).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, os.path.join(REF, rel_path), "exec"), namespace)
            return namespace[name]
    raise KeyError(name)


def metrics_case(rng, n_names, n_obj, n_frames, pairs_per_frame, n_rel, ties):
    names = [f"cat {i:02d}" for i in range(n_names)]
    rels = [f"rel {chr(97 + i % 26)}{i:02d}" for i in range(n_rel)]
    oids = list(range(1, n_obj + 1))
    cate = rng.random((n_names, n_obj)).astype(np.float32)
    cols = []
    for fid in range(n_frames):
        seen = set()
        while len(seen) < pairs_per_frame:
            a, b = (int(v) for v in rng.choice(oids, size=2, replace=False))
            seen.add((a, b))
        cols += [(fid, p) for p in sorted(seen)]
    binary = rng.random((n_rel, len(cols))).astype(np.float32)
    if ties:  # exact ties: the order then depends on the name comparison of sorted((p, name), reverse=True)
        cate[: min(6, n_names), :] = np.float32(0.5)
        binary[:, ::2] = np.round(binary[:, ::2] * 4) / 4
    gt_objects = [(0, oid, names[int(rng.integers(n_names))].replace(" ", "_")) for oid in oids]
    gt_objects.append((0, 999, names[0]))            # an object that was never scored
    gt_rels = []
    for fid in range(n_frames):
        here = [p for f, p in cols if f == fid]
        ls = [(a, b, rels[int(rng.integers(n_rel))].replace(" ", "/")) for a, b in here[::2]]
        ls.append((998, 999, rels[0]))               # a pair that was never scored
        gt_rels.append(ls)
    return names, rels, oids, cols, cate, binary, gt_objects, gt_rels


def main():
    compute_metrics = load_function("scripts/eval_sgclip.py", "compute_metrics", {})
    bitmasks2bboxes = load_function("laser/loading.py", "bitmasks2bboxes", {"np": np})
    rng = np.random.default_rng(2024)
    arrays, meta = {}, {"metrics": [], "boxes": []}

    for ci, (nn, no, nf, pp, nr, ties, thres) in enumerate([(12, 5, 3, 4, 7, False, [1, 5, 10]),
                                                            (20, 8, 4, 6, 14, True, [1, 5, 10]),
                                                            (3, 2, 2, 2, 2, True, [1, 3])]):
        names, rels, oids, cols, cate, binary, gt_objects, gt_rels = metrics_case(rng, nn, no, nf, pp, nr, ties)
        cate_pred = {(oid, nm): torch.tensor(cate[r, g]) for g, oid in enumerate(oids) for r, nm in enumerate(names)}
        binary_pred = {(fid, pair, nm): torch.tensor(binary[r, c]) for r, nm in enumerate(rels)
                       for c, (fid, pair) in enumerate(cols)}
        res = compute_metrics(gt_objects, cate_pred, gt_rels, binary_pred, precision_thres_ls=thres, recall_thres_ls=thres)
        arrays[f"m{ci}_cate"], arrays[f"m{ci}_binary"] = cate, binary
        meta["metrics"].append({"names": names, "rels": rels, "oids": oids, "cols": [[f, list(p)] for f, p in cols],
                                "gt_objects": [list(g) for g in gt_objects],
                                "gt_rels": [[list(t) for t in ls] for ls in gt_rels], "thres": thres,
                                "expected": {k: {kk: {str(t): v for t, v in vv.items()} for kk, vv in kv.items()}
                                             for k, kv in res.items()}})

    # masks in the layout the dataset hands to bitmasks2bboxes: [H, W, 1] bool
    for bi, (H, W) in enumerate([(37, 53), (64, 64), (45, 120)]):
        masks = []
        for _ in range(6):
            m = np.zeros((H, W, 1), dtype=bool)
            y1, x1 = int(rng.integers(0, H)), int(rng.integers(0, W))
            y2, x2 = int(rng.integers(y1, H)), int(rng.integers(x1, W))
            m[y1:y2 + 1, x1:x2 + 1, 0] = rng.random((y2 - y1 + 1, x2 - x1 + 1)) < 0.6
            m[y1, x1, 0] = True
            masks.append(m)
        masks.append(np.ones((H, W, 1), dtype=bool))
        single = np.zeros((H, W, 1), dtype=bool); single[H - 1, W - 1, 0] = True
        masks.append(single)
        boxes = bitmasks2bboxes(masks)
        assert len(boxes) == len(masks)
        arrays[f"b{bi}_masks"] = np.stack([m[:, :, 0] for m in masks])
        meta["boxes"].append([{k: int(v) for k, v in b.items()} for b in boxes])
    # an empty mask is silently dropped by the reference function (the list gets shorter)
    empty = [np.zeros((8, 8, 1), dtype=bool), np.ones((8, 8, 1), dtype=bool)]
    meta["boxes_with_empty"] = [{k: int(v) for k, v in b.items()} for b in bitmasks2bboxes(empty)]

    np.savez_compressed(os.path.join(HERE, "consumers.npz"), **arrays)
    json.dump(meta, open(os.path.join(HERE, "consumers.json"), "w"), indent=1)
    print("wrote consumers.npz / consumers.json:", len(meta["metrics"]), "metric cases,", len(meta["boxes"]), "box cases")


if __name__ == "__main__":
    main()

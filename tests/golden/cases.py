"""Seeded input batches of the golden forward cases (shared by the generator
and the tests, so both see byte-identical inputs)."""
from __future__ import annotations

from laser_amd.synthetic import make_clip_batch


def tiny_batch(seed, n_videos=2, frames=3, objects=4, H=150, W=200, train=False):
    b = make_clip_batch(seed, n_videos=n_videos, frames=frames, objects=objects, H=H, W=W,
                        k_cate=5, k_unary=7, k_binary=6, min_box=12, max_box=140)
    if train:
        names = b.batched_names[0]
        kws = b.batched_binary_kws[0]
        b.batched_binary_predicates = [
            [(kws[0], names[0], names[1]), (kws[1], names[2], names[0])],
            [(kws[2], names[1], names[3])],
        ][:n_videos]
    return b



FORWARD_CASES = {
    # name: (cfg name, seed, batch kwargs, forward kwargs)
    "tiny_infer": ("tiny", 11, dict(seed=5, n_videos=2, frames=3, objects=4), dict()),
    "tiny_train": ("tiny", 11, dict(seed=6, n_videos=2, frames=3, objects=4, train=True), dict()),
    "tiny_multiclass_emb": ("tiny", 12, dict(seed=7, n_videos=1, frames=2, objects=3), dict(multi_class=True, output_embeddings=True)),
    "tiny_logit": ("tiny", 12, dict(seed=8, n_videos=1, frames=2, objects=3), dict(output_logit=True)),
    "tiny_empty_kws": ("tiny", 13, dict(seed=9, n_videos=1, frames=2, objects=3), dict()),
    "tiny_no_pairs": ("tiny", 13, dict(seed=10, n_videos=1, frames=2, objects=2), dict()),
    "b32_infer": ("vit-b-32", 21, dict(seed=31, n_videos=1, frames=2, objects=3, H=240, W=320), dict(output_embeddings=True)),
}



def build_case_batch(name: str):
    _, _, bk, _ = FORWARD_CASES[name]
    b = tiny_batch(**bk)
    if name == "tiny_empty_kws":
        b.batched_unary_kws = [[]]
    if name == "tiny_no_pairs":
        b.batched_obj_pairs = []
    return b

"""stage_inputs_async: clips packed and uploaded by the worker thread on the copy stream while the previous clip is
scored give bit-identical results to stage_inputs inside the call (the caller loop of scripts/eval_sgclip.py:339-352,
one clip after the other)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from laser_amd.config import config_by_name  # noqa: E402
from laser_amd.masks import pack_masks  # noqa: E402
from laser_amd.predicate_model import PredicateModel  # noqa: E402
from laser_amd.synthetic import make_clip_batch  # noqa: E402
from laser_amd.tokenizer import HashTokenizer  # noqa: E402


def _dense(out):
    return [out[0][0].dense.clone(), out[1][0].dense.clone(), out[2][0].dense.clone()]


@pytest.mark.parametrize("packed", [False, True])
def test_pipelined_staging_gives_the_same_results(packed):
    cfg = config_by_name("tiny")
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    model = PredicateModel(None, None, "cuda:0", "tiny", tokenizer=tok, init_seed=4).eval()
    # clips of different sizes, so that the buffer sets grow and are reused with other layouts
    clips = [make_clip_batch(50 + i, n_videos=1, frames=2 + (i % 3), objects=3 + (i % 2), H=120 + 16 * (i % 2), W=160,
                             k_cate=4, k_unary=5, k_binary=3, min_box=12, max_box=100) for i in range(6)]

    def args(b):
        kw = b.forward_kwargs()
        if packed:
            kw = dict(kw, batched_masks=pack_masks(b.batched_masks), batched_bboxes=None)
        return kw
    want = []
    for b in clips:
        want.append(_dense(model(**args(b))))
    torch.cuda.synchronize()
    kws = [args(b) for b in clips]
    got = []
    pending = model.stage_inputs_async(kws[0]["batched_videos"], kws[0]["batched_masks"])
    for i, kw in enumerate(kws):
        nxt = model.stage_inputs_async(kws[i + 1]["batched_videos"], kws[i + 1]["batched_masks"]) if i + 1 < len(kws) else None
        got.append(_dense(model(**kw, device_inputs=pending)))
        pending = nxt
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        for a, b in zip(g, w):
            assert torch.equal(a, b)


def test_a_failed_staging_frees_its_buffer_set():
    cfg = config_by_name("tiny")
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    model = PredicateModel(None, None, "cuda:0", "tiny", tokenizer=tok, init_seed=4).eval()
    b = make_clip_batch(60, n_videos=1, frames=2, objects=3, H=120, W=160, k_cate=4, k_unary=5, k_binary=3, min_box=12,
                        max_box=100)
    kw = b.forward_kwargs()
    for _ in range(3):    # more failures than buffer sets: a leaked set would block the next submission for good
        bad = model.stage_inputs_async([np.zeros((4, 4), dtype=np.uint8)], kw["batched_masks"])
        with pytest.raises(ValueError):
            bad.result()
    ok = model.stage_inputs_async(kw["batched_videos"], kw["batched_masks"])
    out = model(**kw, device_inputs=ok)
    assert torch.equal(out[1][0].dense, model(**kw)[1][0].dense)


def test_a_dropped_clip_is_discarded_or_reported():
    cfg = config_by_name("tiny")
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    model = PredicateModel(None, None, "cuda:0", "tiny", tokenizer=tok, init_seed=4).eval()
    b = make_clip_batch(61, n_videos=1, frames=2, objects=3, H=120, W=160, k_cate=4, k_unary=5, k_binary=3, min_box=12,
                        max_box=100)
    kw = b.forward_kwargs()
    for _ in range(3):    # staged and dropped, more often than there are buffer sets
        model.stage_inputs_async(kw["batched_videos"], kw["batched_masks"]).discard()
    ok = model.stage_inputs_async(kw["batched_videos"], kw["batched_masks"])
    assert torch.equal(model(**kw, device_inputs=ok)[1][0].dense, model(**kw)[1][0].dense)
    # without discard() the third submission finds its buffer set still held and says so instead of waiting for ever
    model._stager.reuse_timeout_s = 0.5
    held = [model.stage_inputs_async(kw["batched_videos"], kw["batched_masks"]) for _ in range(2)]
    for h in held:
        h.result()
    with pytest.raises(RuntimeError, match="discard"):
        model.stage_inputs_async(kw["batched_videos"], kw["batched_masks"]).result()
    for h in held:
        h.discard()
    ok = model.stage_inputs_async(kw["batched_videos"], kw["batched_masks"])
    assert torch.equal(model(**kw, device_inputs=ok)[1][0].dense, model(**kw)[1][0].dense)

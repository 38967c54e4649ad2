"""GPU tests of the train step through PredicateModel.forward(): probabilities and parameter
gradients against fp32 autograd of the CPU oracle on the same inputs, and an optimisation run."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from laser_amd import training  # noqa: E402
from laser_amd.config import config_by_name  # noqa: E402
from laser_amd.predicate_model import PredicateModel  # noqa: E402
from laser_amd.synthetic import make_clip_batch  # noqa: E402
from laser_amd.tokenizer import HashTokenizer  # noqa: E402
from laser_amd.weights import TOWER_PREFIXES, predicate_model_state_dict, split_predicate_state_dict  # noqa: E402


def _setup(seed, logit_scale):
    cfg = config_by_name("tiny")
    sd = predicate_model_state_dict(cfg, seed, logit_scale)
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    model = PredicateModel(None, None, "cuda:0", "tiny", tokenizer=tok)
    model.load_state_dict(sd)
    return cfg, sd, tok, model


def _loss(out, weights_seed):
    """A fixed random linear functional of every returned probability (same key order on both sides)."""
    vals = []
    for v in sorted(out[0]):
        for d in (out[0][v], out[1][v]):
            if len(d):
                vals.append(d.values_tensor() if hasattr(d, "values_tensor") else torch.stack(list(d.values())))
    for d in out[2]:
        if len(d):
            vals.append(d.values_tensor() if hasattr(d, "values_tensor") else torch.stack(list(d.values())))
    flat = torch.cat([v.reshape(-1) for v in vals])
    w = torch.from_numpy(np.random.default_rng(weights_seed).standard_normal(flat.numel()).astype(np.float32))
    return (flat * w.to(flat.device)).sum() + 0.3 * out[3].to(flat.device), flat


def test_forward_with_gradients_matches_fp32_autograd_of_the_oracle():
    cfg, sd, tok, model = _setup(21, float(np.log(8.0)))
    b = make_clip_batch(9, n_videos=2, frames=3, objects=3, H=96, W=128, k_cate=5, k_unary=6, k_binary=4, min_box=12,
                        max_box=90)
    kw = b.forward_kwargs()
    kw["batched_unary_kws"] = [list(kw["batched_unary_kws"][0]), []]        # a dummy keyword row in video 1
    model.enable_training()
    out = model(**kw)
    loss, flat = _loss(out, 1)
    loss.backward()
    torch.cuda.synchronize()

    towers, want, wloss, wflat = _oracle_gradients(cfg, sd, tok, kw, 1)

    assert [list(out[0][v].keys()) for v in out[0]] == [list(want[0][v].keys()) for v in want[0]]
    assert (flat.detach().cpu() - wflat.detach()).abs().max().item() <= 2e-2
    assert abs(loss.item() - wloss.item()) <= 5e-2 * max(1.0, abs(wloss.item()))
    total = _compare_gradients(model, towers, 0.1, 5e-2)
    print(f"train forward: max |p - oracle| = {(flat.detach().cpu() - wflat.detach()).abs().max().item():.2e}, "
          f"all-parameter gradient rel. error {total:.2e}")


def test_inference_after_a_step_reads_the_updated_weights_and_matches_the_training_forward():
    cfg, sd, tok, model = _setup(22, float(np.log(20.0)))
    b = make_clip_batch(10, n_videos=1, frames=2, objects=3, H=96, W=128, k_cate=4, k_unary=5, k_binary=3, min_box=12,
                        max_box=90)
    kw = b.forward_kwargs()
    with torch.no_grad():
        before = model(**kw)[1][0].values_tensor().clone()
    model.enable_training()
    opt = training.Adam(model.parameters(), lr=1e-3)
    out = model(**kw)
    assert out[1][0].dense.requires_grad
    (-torch.log(out[1][0].values_tensor()[0])).backward()
    opt.step()
    opt.zero_grad()
    train_view = model(**kw)[1][0].values_tensor().detach()
    with torch.no_grad():
        after = model(**kw)[1][0].values_tensor()
    torch.cuda.synchronize()
    assert (after - before).abs().max().item() > 1e-4           # the step changed the scores
    assert (after - train_view).abs().max().item() <= 2e-2      # both forms of the forward agree


def test_a_few_adam_steps_raise_the_target_probabilities():
    cfg, sd, tok, model = _setup(23, float(np.log(20.0)))
    b = make_clip_batch(11, n_videos=1, frames=2, objects=4, H=96, W=128, k_cate=4, k_unary=6, k_binary=4, min_box=12,
                        max_box=90)
    kw = b.forward_kwargs()
    model.enable_training()
    opt = training.Adam(model.parameters(), lr=2e-4)
    losses = []
    for _ in range(6):
        cate, unary, binary, dummy = model(**kw)
        # push the first keyword of every vocabulary up for every crop
        u, bn = unary[0].dense[0], binary[0].dense[0]
        c = cate[0].dense[0]
        loss = -(torch.log(u + 1e-6).mean() + torch.log(bn + 1e-6).mean() + torch.log(c + 1e-6).mean()) + 0.0 * dummy
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0] * 0.8, losses


def _grads_of(model):
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def _tower_nodes(outputs):
    """The image/text tower nodes of the autograd graph behind a forward()'s outputs."""
    seen, found, stack = set(), [], [getattr(outputs[3], "grad_fn", None)]
    for d in list(outputs[0].values()) + list(outputs[1].values()) + list(outputs[2]):
        stack.append(d.dense.grad_fn)
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        if hasattr(fn, "chunks"):
            found.append(fn)
        stack += [nf for nf, _ in fn.next_functions]
    return found


def test_flat_parameters_and_chunked_recompute_give_the_same_gradients():
    """The same batch through (a) loose parameters, (b) flat buffers with in-place gradient accumulation,
    (c) flat buffers + image towers run 5 crops at a time, every chunk recomputed in backward, (d) the same with the
    chunks' activations kept while memory allows: same loss, same gradients (up to the order of the fp32 atomic
    additions); two backward passes accumulate; state_dict survives flattening."""
    b = make_clip_batch(12, n_videos=1, frames=3, objects=3, H=96, W=128, k_cate=4, k_unary=5, k_binary=3, min_box=12,
                        max_box=90)
    kw = b.forward_kwargs()
    results = []
    for mode in ("loose", "flat", "flat+recompute", "flat+chunks kept", "flat+kept within a budget"):
        cfg, sd, tok, model = _setup(24, float(np.log(10.0)))
        model.enable_training()
        if mode != "loose":
            flat = model.flatten_parameters()
            assert all(torch.equal(v.cpu(), sd[k]) for k, v in model.state_dict().items())
        if mode == "flat+recompute":
            model.set_recompute_chunk(5, keep_activations=False)
        if mode == "flat+chunks kept":
            model.set_recompute_chunk(5)
        if mode == "flat+kept within a budget":      # all towers together may keep 3 MB: some chunks kept, some recomputed
            model.set_recompute_chunk(5, max_kept_bytes=3 << 20)
        out = model(**kw)
        if mode == "flat+kept within a budget":
            fns = [f for f in _tower_nodes(out) if len(f.chunks) > 1]
            kept, total = sum(len(f.kept) for f in fns), sum(len(f.chunks) for f in fns)
            held = sum(h for f in fns for _, h in f.kept.values())
            assert 0 < kept < total and held <= 3 << 20 and fns[0].tower.kept_budget.used == held, (kept, total, held)
        elif mode.startswith("flat+"):
            fns = [f for f in _tower_nodes(out) if len(f.chunks) > 1]
            assert fns and all(len(f.kept) == (len(f.chunks) if mode == "flat+chunks kept" else 0) for f in fns)
        loss, _ = _loss(out, 3)
        loss.backward()
        torch.cuda.synchronize()
        results.append((loss.item(), _grads_of(model)))
        if mode == "flat":
            once = flat.grad.clone()
            loss2, _ = _loss(model(**kw), 3)
            loss2.backward()
            assert ((flat.grad - 2 * once).norm() / (2 * once).norm()).item() <= 1e-4     # accumulation
            flat.zero_grad()
            assert float(flat.grad.abs().max()) == 0.0
    (l0, g0), (l1, g1), (l2, g2), (l3, g3), (l4, g4) = results
    assert all(abs(l0 - l) <= 1e-5 * max(1, abs(l0)) for l in (l1, l2, l3, l4))
    for other in (g1, g2, g3, g4):
        assert set(other) == set(g0)
        a = torch.cat([g0[k].reshape(-1) for k in sorted(g0)])
        o = torch.cat([other[k].reshape(-1) for k in sorted(g0)])
        assert ((a - o).norm() / a.norm()).item() <= 1e-4


def test_flat_adam_matches_per_parameter_adam():
    """One Adam launch over the flat buffers against one per parameter.  The first step comes from a real backward
    pass (so the flat path's in-place gradient accumulation is what the optimizer reads); the gradients of both modes
    are then set to the same values: the weight gradients are sums of fp32 atomics whose order differs from run to
    run, and Adam turns a last-bit difference in a near-zero gradient into a full step of the learning rate."""
    b = make_clip_batch(13, n_videos=1, frames=2, objects=3, H=96, W=128, k_cate=4, k_unary=5, k_binary=3, min_box=12,
                        max_box=90)
    kw = b.forward_kwargs()
    after, shared = [], {}
    for use_flat in (False, True):
        cfg, sd, tok, model = _setup(25, float(np.log(10.0)))
        model.enable_training()
        opt = training.Adam(model.flatten_parameters() if use_flat else model.parameters(), lr=1e-3)
        for step in range(2):
            opt.zero_grad()
            loss, _ = _loss(model(**kw), 4)
            loss.backward()
            torch.cuda.synchronize()
            for n, p in model.named_parameters():
                if p.grad is None:
                    continue
                if not use_flat:
                    shared[(step, n)] = p.grad.detach().clone()
                else:
                    ref = shared[(step, n)]
                    assert ((p.grad - ref).norm() / ref.norm().clamp_min(1e-20)).item() <= 1e-3, n
                    p.grad.copy_(ref)       # a view of the flat gradient buffer
            opt.step()
        torch.cuda.synchronize()
        after.append({k: v.detach().clone() for k, v in model.state_dict().items()})
    a = torch.cat([after[0][k].reshape(-1) for k in sorted(after[0])])
    o = torch.cat([after[1][k].reshape(-1) for k in sorted(after[0])])
    assert (a - o).abs().max().item() <= 1e-6


def _oracle_gradients(cfg, sd, tok, kw, weights_seed):
    from oracle.forward_ref import OraclePredicateModel
    towers = {p: {k: (v.clone().float().requires_grad_(True) if v.is_floating_point() else v) for k, v in t.items()}
              for p, t in split_predicate_state_dict(sd).items()}
    oracle = OraclePredicateModel(cfg, towers, tok)
    oracle.grad = True
    torch.set_num_threads(16)
    want = oracle(**kw)
    wloss, wflat = _loss(want, weights_seed)
    wloss.backward()
    return towers, want, wloss, wflat


def _compare_gradients(model, towers, per_tensor_tol, total_tol):
    named = dict(model.named_parameters())
    got_all, want_all, bad = [], [], {}
    for prefix in TOWER_PREFIXES:
        for k, ref in towers[prefix].items():
            if not ref.is_floating_point():
                continue
            g = named[f"{prefix}.{k}"].grad
            assert g is not None, (prefix, k)
            g, r = g.detach().cpu().reshape(-1), (ref.grad if ref.grad is not None else torch.zeros_like(ref)).reshape(-1)
            got_all.append(g)
            want_all.append(r)
            if "k_proj.bias" in k:        # exact gradient is zero (softmax shift invariance)
                continue
            rel = ((g - r).norm() / r.norm().clamp_min(1e-12)).item()
            if g.numel() == 1:
                # logit_scale: sum over all logits of dlogit * logit, terms of both signs (the softmax gradient of a
                # column sums to zero), so the error is judged against the gradient scale of the whole model
                if abs(float(g - r)) > 0.05 * max(abs(float(r)), 1.0):
                    bad[f"{prefix}.{k}"] = (float(g), float(r))
            elif rel > per_tensor_tol:
                bad[f"{prefix}.{k}"] = rel
    assert not bad, bad
    g, r = torch.cat(got_all), torch.cat(want_all)
    total = ((g - r).norm() / r.norm()).item()
    assert total <= total_tol, total
    return total


@pytest.mark.parametrize("cfg_name,objects,frames", [("vit-b-32", 5, 2), ("vit-l-14-336", 2, 1), ("vit-b-16", 3, 1), ("vit-l-14", 2, 1)])
def test_train_gradients_at_real_width_match_the_oracle(cfg_name, objects, frames):
    """The train step of scripts/train_sgclip.py:682-694, 877-890 (forward under autograd with the spec's binary
    predicates selecting the pairs, loss.backward()) at the real CLIP geometries: ViT-B/32, ViT-L/14 at 336 px
    whose 577-token sequences run the blocked attention adjoint, and the two geometries in between (ViT-B/16 with 197
    tokens, ViT-L/14 at 224 px with 257).  Selected pair keys bit-exact, probabilities within
    2e-2, every parameter gradient against fp32 autograd of the CPU oracle."""
    cfg = config_by_name(cfg_name)
    sd = predicate_model_state_dict(cfg, 27, float(np.log(20.0)))
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    model = PredicateModel(None, None, "cuda:0", cfg_name, tokenizer=tok)
    model.load_state_dict(sd)
    b = make_clip_batch(14, n_videos=1, frames=frames, objects=objects, H=240, W=320, k_cate=4, k_unary=5, k_binary=3,
                        min_box=16, max_box=200)
    names, kws = b.batched_names[0], b.batched_binary_kws[0]
    b.batched_binary_predicates = [[(kws[0], names[0], names[1]), (kws[1], names[2], names[0])]]
    kw = b.forward_kwargs()
    model.enable_training()
    out = model(**kw)
    loss, flat = _loss(out, 5)
    loss.backward()
    torch.cuda.synchronize()
    towers, want, wloss, wflat = _oracle_gradients(cfg, sd, tok, kw, 5)
    assert list(out[2][0].keys()) == list(want[2][0].keys())          # the selected pairs, in the reference's order
    assert list(out[1][0].keys()) == list(want[1][0].keys())
    assert (flat.detach().cpu() - wflat.detach()).abs().max().item() <= 2e-2
    total = _compare_gradients(model, towers, 0.15, 5e-2)
    print(f"{cfg_name} train step: {len(out[2][0])} pair probabilities, all-parameter gradient rel. error {total:.2e}")
    del model
    torch.cuda.empty_cache()


def test_c5_train_step_at_full_size():
    """BASELINE.json configs[4] at its size: two clips of 32 frames x 16 objects x 240 ordered pairs at 1280 x 720
    (16 384 crops, 17 408 image encodes), ViT-B/32, forward + backward + Adam with the image towers recomputed in
    chunks during backward.  The CPU oracle cannot run this size, so the check is a property that does not depend on
    it: clips are independent, hence the gradient of the two-clip batch equals the sum of the gradients of each clip
    run alone (the loss is a sum over clips); then one Adam step must change what inference scores."""
    cfg = config_by_name("vit-b-32")
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    model = PredicateModel(None, None, "cuda:0", "vit-b-32", tokenizer=tok, init_seed=31).enable_training()
    for p in TOWER_PREFIXES:
        model._modules[p].logit_scale.data.fill_(float(np.log(20.0)))
    flat = model.flatten_parameters()
    model.set_recompute_chunk(2048)
    b = make_clip_batch(15, n_videos=2, frames=32, objects=16, distinct_frames=8)
    kw = b.forward_kwargs()
    assert len(kw["batched_object_ids"]) + len(kw["batched_obj_pairs"]) == 16384

    def clip_loss(out, vid):
        u, bn, c = out[1][vid].dense, out[2][vid].dense, out[0][vid].dense
        return -(torch.log(u[0] + 1e-6).mean() + torch.log(bn[1] + 1e-6).mean() + torch.log(c[2] + 1e-6).mean())

    out = model(**kw)
    loss = clip_loss(out, 0) + clip_loss(out, 1) + 0.0 * out[3]
    loss.backward()
    torch.cuda.synchronize()
    both = flat.grad.clone()
    assert torch.isfinite(both).all() and torch.isfinite(loss.detach()).item()
    flat.zero_grad()

    def one_clip(vid):
        n_f = 32
        keep = [i for i, (v, _, _) in enumerate(kw["batched_object_ids"]) if v == vid]
        sub = dict(kw, batched_video_ids=[kw["batched_video_ids"][vid]],
                   batched_videos=kw["batched_videos"][vid * n_f:(vid + 1) * n_f],
                   batched_masks=[kw["batched_masks"][i] for i in keep], batched_bboxes=[kw["batched_bboxes"][i] for i in keep],
                   batched_names=[kw["batched_names"][vid]], batched_unary_kws=[kw["batched_unary_kws"][vid]],
                   batched_binary_kws=[kw["batched_binary_kws"][vid]],
                   batched_object_ids=[(0, f, o) for (v, f, o) in kw["batched_object_ids"] if v == vid],
                   batched_obj_pairs=[(0, f, pr) for (v, f, pr) in kw["batched_obj_pairs"] if v == vid],
                   batched_video_splits=[n_f], batched_binary_predicates=[None])
        o = model(**sub)
        l = clip_loss(o, 0) + 0.0 * o[3]
        l.backward()
        return float(l.detach())

    l0, l1 = one_clip(0), one_clip(1)            # gradients accumulate in the flat buffer
    torch.cuda.synchronize()
    assert abs((l0 + l1) - float(loss.detach())) <= 1e-3 * max(1.0, abs(float(loss.detach())))
    rel = ((flat.grad - both).norm() / both.norm()).item()
    assert rel <= 5e-3, rel
    for prefix in TOWER_PREFIXES:                 # every tower received a gradient
        g = dict(model.named_parameters())[f"{prefix}.vision_model.encoder.layers.0.mlp.fc1.weight"].grad
        assert g is not None and float(g.abs().max()) > 0
    with torch.no_grad():
        before = model(**kw)[1][0].dense.clone()
    opt = training.Adam(flat, lr=1e-4)
    opt.step()
    with torch.no_grad():
        after = model(**kw)[1][0].dense
    torch.cuda.synchronize()
    assert (after - before).abs().max().item() > 1e-5
    print(f"C5 train step: loss {float(loss.detach()):.4f}, batch vs per-clip gradient rel. diff {rel:.2e}, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB")
    del model, flat, opt
    torch.cuda.empty_cache()


def test_gradient_reducer_is_told_by_every_tower_during_backward():
    """One process (no collective runs): the towers report their buckets while backward is still going, every bucket
    exactly once, the buckets tile the flat buffer, and a scoring pass between steps does not confuse the counts."""
    b = make_clip_batch(17, n_videos=1, frames=2, objects=3, H=96, W=128, k_cate=4, k_unary=5, k_binary=3, min_box=12,
                        max_box=90)
    kw = b.forward_kwargs()
    cfg, sd, tok, model = _setup(31, float(np.log(10.0)))
    model.enable_training()
    flat = model.flatten_parameters()
    red = model.gradient_reducer()
    spans = sorted(red.buckets.values())
    assert spans[0][0] == 0 and spans[-1][1] == flat.total and all(a[1] == c[0] for a, c in zip(spans, spans[1:]))
    assert set(red.buckets) >= {f"{p}.{k}" for p in TOWER_PREFIXES for k in ("vision", "text")}
    for step in range(2):
        with torch.no_grad():
            model(**kw)                       # forward passes that never get a backward
        flat.zero_grad()
        red.begin()
        loss, _ = _loss(model(**kw), 4)
        seen_during_backward = []
        ready = red.ready
        red.ready = lambda key: (seen_during_backward.append(key), ready(key))[1]
        loss.backward()
        red.ready = ready
        assert sorted(seen_during_backward) == sorted(k for k in red.buckets if k != "rest"), seen_during_backward
        red.finish()
        assert sorted(red.launch_log) == sorted(red.buckets)
    torch.cuda.synchronize()
    del model, flat, red
    torch.cuda.empty_cache()

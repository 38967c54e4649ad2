"""GPU tests of the train-step building blocks: every backward op against torch autograd of
the same fp32 op, then whole towers / similarity against the fp32 CLIP restatement in oracle/."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from laser_amd import _lib, grad_ops as G, ops, training  # noqa: E402
from laser_amd.config import TINY  # noqa: E402
from laser_amd.weights import random_state_dict  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _lib.load()
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def _randn(shape, dev, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev)


@pytest.mark.parametrize("R,C,ld", [(64, 64, 64), (100, 37, 128), (850, 768, 896), (3, 200, 64)])
def test_transpose(dev, R, C, ld):
    x = _randn((R, C), dev, 1, dtype=torch.bfloat16)
    y = G.transpose_bf16(x, ld)
    torch.cuda.synchronize()
    assert torch.equal(y[:, :R], x.t())
    assert (y[:, R:] == 0).all()


@pytest.mark.parametrize("M,N", [(1, 5), (333, 130), (5000, 768)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_colsum(dev, M, N, dtype):
    x = _randn((M, N), dev, 2, dtype=dtype)
    out = G.colsum(x)
    torch.cuda.synchronize()
    ref = x.double().sum(0)
    assert (out.double() - ref).abs().max().item() <= 1e-4 * max(1.0, M ** 0.5) * 4


def test_quick_gelu_and_its_gradient(dev):
    u = _randn((257, 96), dev, 3, 2.0, torch.bfloat16)
    dg = _randn((257, 96), dev, 4, 1.0, torch.bfloat16)
    g = G.quick_gelu(u)
    du = G.quick_gelu_bwd(u, dg)
    uf = u.float().requires_grad_(True)
    ref = uf * torch.sigmoid(1.702 * uf)
    ref.backward(dg.float())
    torch.cuda.synchronize()
    assert (g.float() - ref).abs().max().item() <= 2e-2
    assert (du.float() - uf.grad).abs().max().item() <= 3e-2


def test_quick_gelu_tail(dev):
    u = _randn((1003,), dev, 3, 2.0, torch.bfloat16)
    dg = _randn((1003,), dev, 4, 1.0, torch.bfloat16)
    uf = u.float()
    sg = torch.sigmoid(1.702 * uf)
    torch.cuda.synchronize()
    assert (G.quick_gelu(u).float() - uf * sg).abs().max().item() <= 2e-2
    assert (G.quick_gelu_bwd(u, dg).float() - dg.float() * (sg + 1.702 * uf * sg * (1 - sg))).abs().max().item() <= 3e-2


@pytest.mark.parametrize("M,N,K", [(768, 768, 6400), (3072, 768, 44800), (100, 130, 640), (512, 2048, 960), (200, 256, 64)])
def test_gemm_split_k_accumulates_atomically(dev, M, N, K):
    A = _randn((M, K), dev, 50, 1.0, torch.bfloat16)
    W = _randn((N, K), dev, 51, 1.0, torch.bfloat16)
    base = _randn((M, N), dev, 52)
    out = base.clone()
    ops.gemm_bf16(A, W, out, _lib.EPI_F32_ATOMIC)
    ref = base.double() + A.double() @ W.double().t()
    torch.cuda.synchronize()
    assert (out.double() - ref).abs().max().item() <= 2e-3 * K ** 0.5


@pytest.mark.parametrize("Kd,M,N", [(64, 256, 256), (128, 16, 8), (51200, 768, 768), (6400, 2304, 768), (1425, 520, 264),
                                    (37, 64, 128), (1000, 3072, 768), (4099, 768, 3072), (127, 264, 520), (129, 264, 520),
                                    (191, 256, 256), (3648, 256, 256), (3648 + 9, 768, 768)])
def test_gemm_tn_accumulates_a_transposed_product(dev, Kd, M, N):
    """sgclip_gemm_tn_bf16: C += A^T B from row-major operands, any reduction length, partial tiles, strided rows.
    Whole 64-row stages run on one kernel and the remaining rows on another; 3648 rows would leave a last reduction
    slice of a single stage without the host's adjustment."""
    a = _randn((Kd, M + 8), dev, 61, 1.0, torch.bfloat16)[:, :M]
    b = _randn((Kd, N), dev, 62, 1.0, torch.bfloat16)
    c0 = _randn((M, N), dev, 63, 1.0)
    c = c0.clone()
    from laser_amd import _lib
    from laser_amd.ops import _stream
    _lib.check(_lib.load().sgclip_gemm_tn_bf16(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), c.stride(0),
                                               M, N, Kd, _stream()))
    torch.cuda.synchronize()
    want = c0.double() + a.double().t() @ b.double()
    assert ((c.double() - want).abs().max() / want.abs().max()).item() <= 2e-5


def test_gemm_tn_exact_small_integers(dev):
    """Integer operands: every product and sum is exact in fp32, so the fragment maps are checked bit for bit."""
    g = torch.Generator().manual_seed(4)
    a = torch.randint(-3, 4, (200, 264), generator=g).to(torch.bfloat16).to(dev)
    b = torch.randint(-3, 4, (200, 520), generator=g).to(torch.bfloat16).to(dev)
    c = torch.zeros((264, 520), dtype=torch.float32, device=dev)
    from laser_amd import _lib
    from laser_amd.ops import _stream
    _lib.check(_lib.load().sgclip_gemm_tn_bf16(a.data_ptr(), 264, b.data_ptr(), 520, c.data_ptr(), 520, 264, 520, 200, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(c, a.float().t() @ b.float())


@pytest.mark.parametrize("M,out_f,in_f", [(6400, 768, 768), (850, 3072, 768), (100, 64, 128)])
def test_linear_wgrad(dev, M, out_f, in_f):
    dy = _randn((M, out_f), dev, 53, 1.0, torch.bfloat16)
    x = _randn((M, in_f), dev, 54, 1.0, torch.bfloat16)
    dw = G.linear_wgrad(dy, x)
    torch.cuda.synchronize()
    assert _rel(dw, dy.float().t() @ x.float()) <= 1e-3


def test_cast(dev):
    x = _randn((1000,), dev, 5, 10.0)
    assert torch.equal(G.cast_bf16(x), x.to(torch.bfloat16))


@pytest.mark.parametrize("M,N", [(1, 8), (50, 768), (40000, 768), (3001, 1024), (70, 2048), (33, 520), (9, 100)])
def test_cast_with_column_sums_in_the_same_pass(dev, M, N):
    x = _randn((M, N), dev, 6, 3.0)
    acc = torch.full((N,), 0.5, dtype=torch.float32, device=dev)
    y = G.cast_bf16_colsum(x, acc)
    torch.cuda.synchronize()
    assert torch.equal(y, x.to(torch.bfloat16))
    ref = 0.5 + x.double().sum(0)
    assert float((acc.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(x.abs().sum(0).max()))


# (70000, 3072, 768) runs on the 256 x 256 kernel (column sums in its epilogue), the others on a pass over the result
@pytest.mark.parametrize("M,N,K", [(70000, 3072, 768), (300, 512, 256), (65, 264, 64), (66000, 1032, 512)])
def test_fc2_dgrad_through_gelu_with_fc1_bias_gradient(dev, M, N, K):
    dy = _randn((M, K), dev, 7, 0.5).to(torch.bfloat16)
    w_t = _randn((N, K), dev, 8, 0.05).to(torch.bfloat16)
    u = _randn((M, N), dev, 9, 1.5).to(torch.bfloat16)
    zb = torch.zeros(N, dtype=torch.float32, device=dev)
    acc = torch.full((N,), -2.0, dtype=torch.float32, device=dev)
    du = G.fc2_dgrad_through_gelu(dy, w_t, u, zb, acc)
    plain = torch.empty_like(u)
    ops.gemm_bf16(dy, w_t, plain, _lib.EPI_BIAS_MULGELUGRAD_BF16, bias=zb, aux=u)
    torch.cuda.synchronize()
    assert torch.equal(du, plain)
    uf = u.float()
    sig = torch.sigmoid(1.702 * uf)
    ref = (dy.float() @ w_t.float().t()) * (sig * (1 + 1.702 * uf * (1 - sig)))
    assert _rel(du.float(), ref) <= 1e-2
    want = -2.0 + du.double().sum(0)
    assert float((acc.double() - want).abs().max()) <= 2e-5 * max(1.0, float(du.float().abs().sum(0).max()))


@pytest.mark.parametrize("rows,d", [(1, 128), (50, 768), (1030, 512), (7, 1024)])
@pytest.mark.parametrize("dy_bf16", [False, True])
def test_layernorm_bwd(dev, rows, d, dy_bf16):
    x = _randn((rows, d), dev, 6, 2.0) + 0.5
    w = _randn((d,), dev, 7) * 0.2 + 1.0
    b = _randn((d,), dev, 8) * 0.1
    dy = _randn((rows, d), dev, 9, dtype=torch.bfloat16 if dy_bf16 else torch.float32)
    base = _randn((rows, d), dev, 10)
    dw = torch.zeros(d, device=dev)
    db = torch.zeros(d, device=dev)
    dx = G.layernorm_bwd(x, w, 1e-5, dy, dw, db)
    dx_acc = base.clone()
    dw2, db2 = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    G.layernorm_bwd(x, w, 1e-5, dy, dw2, db2, dx=dx_acc, accumulate=True)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.layer_norm(xr, (d,), wr, br, 1e-5).backward(dy.float())
    torch.cuda.synchronize()
    assert _rel(dx, xr.grad) <= 1e-4
    assert _rel(dx_acc - base, xr.grad) <= 1e-4
    assert _rel(dw, wr.grad) <= 1e-4 and _rel(db, br.grad) <= 1e-4
    assert _rel(dw2, wr.grad) <= 1e-4


def _attention_ref(qkv, n_seq, T, heads, causal, km):
    d = heads * 64
    q, k, v = [t.reshape(n_seq, T, heads, 64).transpose(1, 2) for t in qkv.split(d, dim=1)]
    s = q @ k.transpose(-1, -2) * 0.125
    allowed = torch.ones(n_seq, 1, T, T, dtype=torch.bool, device=qkv.device)
    if causal:
        allowed = allowed & torch.ones(T, T, dtype=torch.bool, device=qkv.device).tril()
    if km is not None:
        allowed = allowed & km.bool()[:, None, None, :]
    s = s.masked_fill(~allowed, float("-inf"))
    return (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(n_seq * T, d)


@pytest.mark.parametrize("n_seq,T,heads,causal,masked", [(3, 50, 2, False, False), (5, 17, 1, False, False),
                                                          (4, 20, 2, True, True), (2, 77, 8, True, True),
                                                          (2, 128, 1, False, False), (9, 7, 12, True, False),
                                                          (2, 577, 2, False, False), (3, 129, 1, True, True),
                                                          (1, 200, 3, True, False), (2, 320, 1, False, True)])
def test_attention_bwd(dev, n_seq, T, heads, causal, masked):
    d = heads * 64
    qkv = _randn((n_seq * T, 3 * d), dev, 11, 1.0, torch.bfloat16)
    dout = _randn((n_seq * T, d), dev, 12, 1.0, torch.bfloat16)
    km = None
    if masked:
        lens = torch.randint(1, T + 1, (n_seq,), generator=torch.Generator().manual_seed(13))
        km = (torch.arange(T)[None, :] < lens[:, None]).to(torch.int32).to(dev)
    dqkv = G.attention_bwd(qkv, dout, n_seq, T, heads, causal, km)
    ref_in = qkv.float().requires_grad_(True)
    out = _attention_ref(ref_in, n_seq, T, heads, causal, km)
    out.backward(dout.float())
    torch.cuda.synchronize()
    # the forward kernel of the library agrees with the same reference
    fwd = ops.attention(qkv, n_seq, T, heads, causal, km)
    assert (fwd.float() - out).abs().max().item() <= 3e-2
    err = (dqkv.float() - ref_in.grad).abs().max().item()
    assert err <= 2e-2 * max(1.0, ref_in.grad.abs().max().item()), err
    assert _rel(dqkv, ref_in.grad) <= 1e-2
    # with the bias gradient riding along: the same dqkv bit for bit, and its column sums ADDED to what dbias held
    dbias = torch.full((3 * d,), 0.5, dtype=torch.float32, device=dev)
    dqkv2 = G.attention_bwd(qkv, dout, n_seq, T, heads, causal, km, dbias=dbias)
    torch.cuda.synchronize()
    assert torch.equal(dqkv2, dqkv)
    want = 0.5 + dqkv.double().sum(dim=0)
    assert (dbias.double() - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("M,N,K,tA,tB", [(5, 7, 3, False, False), (130, 70, 33, True, False), (64, 512, 768, False, True),
                                         (1, 1, 5000, False, True), (100, 65, 64, True, True)])
def test_matmul_f32(dev, M, N, K, tA, tB):
    A = _randn((K, M) if tA else (M, K), dev, 14)
    B = _randn((N, K) if tB else (K, N), dev, 15)
    C0 = _randn((M, N), dev, 16)
    ref = 0.5 * ((A.t() if tA else A).double() @ (B.t() if tB else B).double())
    out = G.matmul_f32(A, B, tA, tB, alpha=0.5)
    acc = G.matmul_f32(A, B, tA, tB, alpha=0.5, out=C0.clone(), accumulate=True)
    torch.cuda.synchronize()
    assert (out.double() - ref).abs().max().item() <= 1e-4 * K ** 0.5
    assert (acc.double() - ref - C0.double()).abs().max().item() <= 1e-4 * K ** 0.5


@pytest.mark.parametrize("mode", [_lib.SIM_LOGIT, _lib.SIM_SOFTMAX, _lib.SIM_SIGMOID])
def test_similarity_function_gradients(dev, mode):
    K, N, E = 13, 301, 512
    text = _randn((K, E), dev, 17).requires_grad_(True)
    img = _randn((N, E), dev, 18).requires_grad_(True)
    ls = torch.tensor(2.0, device=dev, requires_grad=True)
    dout = _randn((K, N), dev, 19)
    out = training.similarity(text, img, ls, mode)
    out.backward(dout)
    t2, i2, l2 = [t.detach().clone().requires_grad_(True) for t in (text, img, ls)]
    logits = (t2 / t2.norm(dim=-1, keepdim=True)) @ (i2 / i2.norm(dim=-1, keepdim=True)).t() * l2.exp()
    ref = logits if mode == _lib.SIM_LOGIT else (logits.softmax(dim=0) if mode == _lib.SIM_SOFTMAX else logits.sigmoid())
    ref.backward(dout)
    torch.cuda.synchronize()
    assert (out - ref).abs().max().item() <= 1e-4
    assert _rel(text.grad, t2.grad) <= 2e-3 and _rel(img.grad, i2.grad) <= 2e-3
    assert abs(ls.grad.item() - l2.grad.item()) <= 2e-3 * max(1.0, abs(l2.grad.item()))


def test_segment_mean_gradient(dev):
    K, N, Gn = 6, 40, 7
    seg_h = np.random.default_rng(0).integers(0, Gn, N)
    seg_h[:Gn] = np.arange(Gn)
    seg = torch.from_numpy(seg_h.astype(np.int32)).to(dev)
    probs = _randn((K, N), dev, 20).requires_grad_(True)
    dmean = _randn((K, Gn), dev, 21)
    out = training.segment_mean(probs, seg, Gn, seg_h)
    out.backward(dmean)
    p2 = probs.detach().clone().requires_grad_(True)
    ref = torch.stack([p2[:, seg.long() == g].mean(dim=1) for g in range(Gn)], dim=1)
    ref.backward(dmean)
    torch.cuda.synchronize()
    assert (out - ref).abs().max().item() <= 1e-5
    assert (probs.grad - p2.grad).abs().max().item() <= 1e-6


def test_rows_scatter_add(dev):
    src = _randn((300, 128), dev, 22)
    idx = torch.randint(0, 20, (300,), generator=torch.Generator().manual_seed(23)).to(torch.int32).to(dev)
    dst = torch.zeros((20, 128), device=dev)
    G.rows_scatter_add(src, idx, dst)
    ref = torch.zeros((20, 128), device=dev).index_add_(0, idx.long(), src)
    torch.cuda.synchronize()
    assert (dst - ref).abs().max().item() <= 1e-4


def test_adam_matches_torch(dev):
    p = torch.nn.Parameter(_randn((1000,), dev, 24))
    q = torch.nn.Parameter(p.detach().clone())
    mine = training.Adam([p], lr=1e-2)
    ref = torch.optim.Adam([q], lr=1e-2)
    for step in range(4):
        g = _randn((1000,), dev, 30 + step)
        p.grad, q.grad = g.clone(), g.clone()
        v0 = p._version
        mine.step()
        ref.step()
        assert p._version > v0
    torch.cuda.synchronize()
    assert (p - q).abs().max().item() <= 1e-5


# ---------------------------------------------------------------------------------------------
# whole towers against fp32 autograd of the CLIP restatement
# ---------------------------------------------------------------------------------------------
def _params(cfg, seed, dev):
    sd = random_state_dict(cfg, seed)
    return {k: torch.nn.Parameter(v.float().to(dev)) for k, v in sd.items() if v.is_floating_point()}


def _check_grads(tower, ref_params, tol):
    bad = {}
    for name in tower.names:
        g, r = tower.params[name].grad.cpu(), ref_params[name].grad.cpu()
        if "k_proj.bias" in name:
            # softmax ignores a shift of all scores of a row: the exact gradient is zero and the
            # reference holds rounding noise; ours must be small against the value bias gradient
            scale = ref_params[name.replace("k_proj", "v_proj")].grad.norm().item()
            if g.norm().item() > tol * scale:
                bad[name] = g.norm().item() / scale
            continue
        if _rel(g, r) > tol:
            bad[name] = _rel(g, r)
    assert not bad, bad


def test_vision_tower_gradients(dev):
    from oracle import clip_ref
    cfg = TINY
    n = 6
    params = _params(cfg, 3, dev)
    tower = training.TrainTower(cfg, "vision", params)
    pv = _randn((n, 3, cfg.image, cfg.image), dev, 40)
    p = cfg.patch
    patches = pv.reshape(n, 3, cfg.grid, p, cfg.grid, p).permute(0, 2, 4, 1, 3, 5).reshape(n * cfg.n_patches, 3 * p * p)
    patches = patches.to(torch.bfloat16).contiguous()
    dfeat = _randn((n, cfg.proj), dev, 41)
    feats = training.encode_images(tower, patches, n)
    feats.backward(dfeat)
    ref_params = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    pv_rounded = patches.float().reshape(n, cfg.grid, cfg.grid, 3, p, p).permute(0, 3, 1, 4, 2, 5).reshape(pv.shape)
    ref = clip_ref.image_features.__wrapped__(ref_params, cfg, pv_rounded)
    ref.backward(dfeat)
    torch.cuda.synchronize()
    assert _rel(feats, ref) <= 2e-2
    _check_grads(tower, ref_params, 5e-2)


def test_text_tower_gradients(dev):
    from oracle import clip_ref
    cfg = TINY
    K, T = 5, 12
    params = _params(cfg, 4, dev)
    tower = training.TrainTower(cfg, "text", params)
    rng = np.random.default_rng(5)
    ids = rng.integers(1, cfg.vocab - 2, (K, T))
    lens = [12, 5, 9, 3, 7]
    am = np.zeros((K, T), dtype=np.int64)
    for k, ln in enumerate(lens):
        ids[k, 0] = cfg.bos_token_id
        ids[k, ln - 1] = cfg.eos_token_id
        ids[k, ln:] = 0
        am[k, :ln] = 1
    ids_t, am_t = torch.from_numpy(ids), torch.from_numpy(am)
    dfeat = _randn((K, cfg.proj), dev, 42)
    feats = training.encode_text(tower, ids_t, am_t)
    feats.backward(dfeat)
    ref_params = {k: v.detach().clone().cpu().requires_grad_(True) for k, v in params.items()}
    ref = clip_ref.text_features.__wrapped__(ref_params, cfg, ids_t, am_t)
    ref.backward(dfeat.cpu())
    torch.cuda.synchronize()
    assert _rel(feats.cpu(), ref) <= 2e-2
    _check_grads(tower, ref_params, 5e-2)


def test_tower_refreshes_weights_after_an_optimizer_step(dev):
    cfg = TINY
    params = _params(cfg, 6, dev)
    tower = training.TrainTower(cfg, "vision", params)
    patches = _randn((2 * cfg.n_patches, 3 * cfg.patch ** 2), dev, 43, dtype=torch.bfloat16)
    f0 = training.encode_images(tower, patches, 2)
    f0.sum().backward()
    opt = training.Adam(tower.param_list(), lr=1e-2)
    opt.step()
    f1 = training.encode_images(tower, patches, 2).detach()
    torch.cuda.synchronize()
    assert (f1 - f0.detach()).abs().max().item() > 1e-3

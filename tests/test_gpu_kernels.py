"""GPU unit tests: every C-ABI building block against a plain fp32 reference of
the same op (torch on the same device) or, for the byte-exact crop kernel,
against the CPU oracle and the golden vectors."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from laser_amd import _lib, ops  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _lib.load()
    return torch.device("cuda:0")


def _rand_bf16(shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(dev)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 192, 128), (1000, 768, 768), (333, 64, 256), (4096, 2304, 768),
                                   (50, 3072, 768), (777, 768, 3072)])
def test_gemm_bias(dev, M, N, K):
    A = _rand_bf16((M, K), dev, 1.0, 1)
    W = _rand_bf16((N, K), dev, K ** -0.5, 2)
    bias = torch.randn(N, device=dev)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16(A, W, out, _lib.EPI_BIAS_BF16, bias=bias)
    ref = A.float() @ W.float().t() + bias
    torch.cuda.synchronize()
    err = (out.float() - ref).abs().max().item()
    assert err <= 2e-2 * max(1.0, ref.abs().max().item()), err


def test_gemm_asymmetric_identity(dev):
    """A = I catches a transposed or permuted fragment map."""
    K = 128
    A = torch.eye(K, dtype=torch.bfloat16, device=dev)
    W = (torch.arange(256 * K, device=dev).reshape(256, K) % 251).to(torch.bfloat16)
    out = torch.empty((K, 256), dtype=torch.float32, device=dev)
    ops.gemm_bf16(A, W, out, _lib.EPI_F32)
    torch.cuda.synchronize()
    assert torch.equal(out, W.float().t())


def test_gemm_epilogues(dev):
    M, N, K = 300, 256, 192
    A = _rand_bf16((M, K), dev, 1.0, 3)
    W = _rand_bf16((N, K), dev, K ** -0.5, 4)
    bias = torch.randn(N, device=dev)
    ref = A.float() @ W.float().t()
    # quick-GELU
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16(A, W, out, _lib.EPI_BIAS_GELU_BF16, bias=bias)
    z = ref + bias
    assert (out.float() - z * torch.sigmoid(1.702 * z)).abs().max().item() < 3e-2
    # pre-activation and activation together (train step)
    g = torch.empty_like(out)
    ops.gemm_bf16(A, W, out, _lib.EPI_BIAS_GELU_BOTH_BF16, bias=bias, aux=g)
    assert (out.float() - z).abs().max().item() < 3e-2
    u = out.float()
    assert ((g.float() - u * torch.sigmoid(1.702 * u)).abs() <= (u * torch.sigmoid(1.702 * u)).abs() * 2.0 ** -7 + 1e-6).all()
    # residual, fp32 in place
    x0 = torch.randn(M, N, device=dev)
    x = x0.clone()
    ops.gemm_bf16(A, W, x, _lib.EPI_BIAS_RESID_F32, bias=bias)
    assert (x - (x0 + ref + bias)).abs().max().item() < 1e-3
    # plain fp32
    o = torch.empty((M, N), dtype=torch.float32, device=dev)
    ops.gemm_bf16(A, W, o, _lib.EPI_F32)
    assert (o - ref).abs().max().item() < 1e-3
    # patch embedding remap: row m -> (m // P) * (P + 1) + 1 + m % P, plus pos[1 + m % P]
    P, imgs = 4, 75
    pos = torch.randn(P + 1, N, device=dev)
    xo = torch.full((imgs * (P + 1), N), 7.0, device=dev)
    ops.gemm_bf16(A, W, xo, _lib.EPI_PATCH_F32, aux=pos, aux_rows=P)
    want = torch.full_like(xo, 7.0)
    rows = torch.arange(M, device=dev)
    want[(rows // P) * (P + 1) + 1 + rows % P] = ref + pos[1 + rows % P]
    torch.cuda.synchronize()
    assert (xo - want).abs().max().item() < 1e-3


@pytest.mark.parametrize("M,N,K", [(70001 // 8 * 8 + 8, 768, 768), (66000, 1024, 192), (65536, 1024, 128), (20000, 3328, 320)])
def test_gemm_many_tiles_per_workgroup(dev, M, N, K):
    """More 256 x 256 tiles than CUs: the persistent kernel's tile loop (odd and even stage counts, two-stage tiles, a ragged
    last tile row, a ragged last tile column) with every epilogue it serves."""
    A = _rand_bf16((M, K), dev, 1.0, 11)
    W = _rand_bf16((N, K), dev, K ** -0.5, 12)
    bias = torch.randn(N, device=dev)
    ref = A.float() @ W.float().t()
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16(A, W, out, _lib.EPI_BIAS_BF16, bias=bias)
    assert (out.float() - (ref + bias)).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    ops.gemm_bf16(A, W, out, _lib.EPI_BIAS_GELU_BF16, bias=bias)
    z = ref + bias
    assert (out.float() - z * torch.sigmoid(1.702 * z)).abs().max().item() < 4e-2
    # both halves of the train step's MLP activation in one epilogue: C = bf16(u), aux = bf16(quick_gelu(C))
    g = torch.empty_like(out)
    ops.gemm_bf16(A, W, out, _lib.EPI_BIAS_GELU_BOTH_BF16, bias=bias, aux=g)
    assert (out.float() - (ref + bias)).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    u = out.float()
    assert torch.equal(g, (u * torch.sigmoid(1.702 * u)).to(torch.bfloat16)) or \
        ((g.float() - u * torch.sigmoid(1.702 * u)).abs() <= (u * torch.sigmoid(1.702 * u)).abs() * 2.0 ** -7 + 1e-6).all()
    del out, z, g, u
    x0 = torch.randn(M, N, device=dev)
    x = x0.clone()
    ops.gemm_bf16(A, W, x, _lib.EPI_BIAS_RESID_F32, bias=bias)
    assert (x - (x0 + ref + bias)).abs().max().item() < 2e-3
    ops.gemm_bf16(A, W, x, _lib.EPI_F32)
    assert (x - ref).abs().max().item() < 2e-3
    P = 8
    imgs = M // P
    pos = torch.randn(P + 1, N, device=dev)
    xo = torch.full((imgs * (P + 1), N), 7.0, device=dev)
    ops.gemm_bf16(A, W, xo, _lib.EPI_PATCH_F32, aux=pos, aux_rows=P)
    want = torch.full_like(xo, 7.0)
    rows = torch.arange(M, device=dev)
    want[(rows // P) * (P + 1) + 1 + rows % P] = ref + pos[1 + rows % P]
    torch.cuda.synchronize()
    assert (xo - want).abs().max().item() < 2e-3


@pytest.mark.parametrize("d", [128, 512, 768, 1024])
def test_layernorm(dev, d):
    x = torch.randn(777, d, device=dev) * 3 + 1
    w, b = torch.randn(d, device=dev), torch.randn(d, device=dev)
    ref = torch.nn.functional.layer_norm(x, (d,), w, b, 1e-5)
    y = ops.layernorm(x, w, b, 1e-5, _lib.DTYPE_F32)
    assert (y - ref).abs().max().item() < 2e-5
    yb = ops.layernorm(x, w, b, 1e-5, _lib.DTYPE_BF16)
    # bf16 keeps 8 significant bits: half an ulp is 2^-9 of the value
    assert bool(((yb.float() - ref).abs() <= ref.abs() * 2.0 ** -8 + 1e-3).all())
    idx = torch.tensor([5, 0, 776, 13], dtype=torch.int32, device=dev)
    yi = ops.layernorm(x, w, b, 1e-5, _lib.DTYPE_F32, row_idx=idx)
    assert (yi - ref[idx.long()]).abs().max().item() < 2e-5
    # strided rows (class-token pooling): every 7th row
    ys = ops.layernorm(x, w, b, 1e-5, _lib.DTYPE_F32, rows=111, x_stride=7 * d)
    assert (ys - ref[::7]).abs().max().item() < 2e-5


def _attn_ref(qkv, n_seq, T, heads, causal, key_mask):
    d = heads * 64
    q, k, v = (qkv.float().view(n_seq, T, 3, heads, 64)[:, :, i].transpose(1, 2) for i in range(3))
    s = q @ k.transpose(-1, -2) * 0.125
    ok = torch.ones(n_seq, 1, T, T, dtype=torch.bool, device=qkv.device)
    if causal:
        ok = ok & torch.ones(T, T, dtype=torch.bool, device=qkv.device).tril()
    if key_mask is not None:
        ok = ok & key_mask.bool()[:, None, None, :]
    s = s.masked_fill(~ok, float("-inf"))
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(n_seq * T, d)


@pytest.mark.parametrize("n_seq,T,heads,causal", [(3, 50, 12, False), (2, 75, 8, True), (1, 577, 4, False), (5, 17, 2, False),
                                                  (2, 32, 2, True), (7, 64, 3, False),
                                                  # T >= 128: the kernel that shares K / V blocks between four query tiles
                                                  (3, 128, 2, False), (2, 200, 3, True), (1, 130, 1, True), (2, 577, 16, False),
                                                  (2, 257, 2, True),
                                                  # no masks, T >= 128: query groups that end in one tile, one row, a full group
                                                  (3, 257, 2, False), (2, 197, 3, False), (2, 129, 1, False), (1, 160, 2, False),
                                                  (2, 192, 2, False), (1, 256, 1, False), (1, 1025, 1, False)])
def test_attention(dev, n_seq, T, heads, causal):
    qkv = _rand_bf16((n_seq * T, 3 * heads * 64), dev, 1.0, 5)
    key_mask = None
    if causal:
        lens = torch.tensor([max(2, T - 9 * i) for i in range(n_seq)])
        key_mask = (torch.arange(T)[None, :] < lens[:, None]).to(torch.int32).to(dev)
    out = ops.attention(qkv, n_seq, T, heads, causal, key_mask)
    ref = _attn_ref(qkv, n_seq, T, heads, causal, key_mask)
    torch.cuda.synchronize()
    err = (out.float() - ref).abs().max().item()
    assert err < 3e-2, err


@pytest.mark.parametrize("n_seq,T,heads,causal", [(5, 50, 12, False), (7, 75, 8, True), (2, 577, 4, False)])
def test_attention_pooled_equals_the_full_kernel_rows(dev, n_seq, T, heads, causal):
    d = heads * 64
    qkv = _rand_bf16((n_seq * T, 3 * d), dev, seed=T)
    g = torch.Generator().manual_seed(T)
    pos = torch.randint(0, T, (n_seq,), generator=g)
    rows = (torch.arange(n_seq) * T + pos).to(torch.int32)
    key_mask = None
    if causal:
        key_mask = torch.ones((n_seq, T), dtype=torch.int32)
        for s in range(n_seq):
            key_mask[s, int(pos[s]) + 1:] = 0           # padding after the pooled (end-of-text) token
        key_mask = key_mask.to(dev)
    full = ops.attention(qkv, n_seq, T, heads, causal, key_mask)
    got = ops.attention_pooled(qkv, rows, n_seq, T, heads, causal, key_mask)
    assert torch.equal(got, full[rows.long().to(dev)])


@pytest.mark.parametrize("n_seq,T,heads", [(5, 50, 12), (23, 50, 4), (1, 50, 2), (7, 37, 3), (4, 64, 2), (3, 100, 2),
                                           (6, 128, 2), (300, 50, 12),
                                           # the three-slot kernel (d >= 256) at its smallest depth, at ViT-L width with the
                                           # heads in four groups over several row tiles, and with a ragged last tile
                                           (23, 50, 16), (2, 128, 8), (11, 50, 5), (1, 3, 4)])
def test_fused_qkv_attention_equals_the_separate_kernels(dev, n_seq, T, heads):
    """sgclip_qkv_attention against sgclip_gemm_bf16(BIAS_BF16) + sgclip_attention and against fp32 torch."""
    d = heads * 64
    x = _rand_bf16((n_seq * T, d), dev, 1.0, seed=T + heads)
    w = _rand_bf16((3 * d, d), dev, d ** -0.5, seed=n_seq)
    bias = torch.randn(3 * d, generator=torch.Generator().manual_seed(3)).to(dev)
    got = ops.qkv_attention(x, w, bias, n_seq, T, heads)
    qkv = torch.empty((n_seq * T, 3 * d), dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16(x, w, qkv, _lib.EPI_BIAS_BF16, bias=bias)
    sep = ops.attention(qkv, n_seq, T, heads, False, None)
    torch.cuda.synchronize()
    # both round q, k, v to bf16 before the attention; only the summation order inside the GEMM differs
    assert (got.float() - sep.float()).abs().max().item() < 2e-2
    ref = _attn_ref((x.float() @ w.float().t() + bias).to(torch.bfloat16), n_seq, T, heads, False, None)
    assert (got.float() - ref).abs().max().item() < 3e-2


def test_fused_qkv_attention_rejects_what_it_does_not_cover(dev):
    lib = _lib.load()
    assert lib.sgclip_qkv_attention_supported(50, 12) == 1
    assert lib.sgclip_qkv_attention_supported(577, 16) == 0
    x = _rand_bf16((577, 128), dev)
    rc = lib.sgclip_qkv_attention(x.data_ptr(), 128, x.data_ptr(), 128, x.data_ptr(), x.data_ptr(), 1, 577, 2, None)
    assert rc == _lib.EINVAL


def test_attention_exact_integers(dev):
    """Small-integer q/k/v with one-hot softmax rows: checks the fragment maps exactly."""
    n_seq, T, heads = 2, 50, 2
    g = torch.Generator().manual_seed(9)
    qkv = torch.zeros(n_seq * T, 3 * heads * 64)
    v = torch.randint(-8, 9, (n_seq * T, heads * 64), generator=g).float()
    qkv[:, 2 * heads * 64:] = v
    # q_t . k_s = 800 * [s == perm(t)] so softmax is one-hot at perm(t)
    perm = torch.randperm(T, generator=g)
    for s in range(n_seq):
        for hd in range(heads):
            for t in range(T):
                qkv[s * T + t, hd * 64 + (t % 64)] = 80.0
                qkv[s * T + perm[t], heads * 64 + hd * 64 + (t % 64)] = 80.0
    qkv = qkv.to(torch.bfloat16).to(dev)
    out = ops.attention(qkv, n_seq, T, heads, False, None).float().cpu()
    want = torch.stack([v[s * T + perm[t]] for s in range(n_seq) for t in range(T)])
    assert torch.equal(out, want)


@pytest.mark.parametrize("K,N,E,mode", [(7, 5, 64, _lib.SIM_SOFTMAX), (200, 513, 512, _lib.SIM_SOFTMAX),
                                        (50, 1000, 512, _lib.SIM_SIGMOID), (35, 64, 768, _lib.SIM_LOGIT), (1, 3, 64, _lib.SIM_SOFTMAX)])
def test_similarity(dev, K, N, E, mode):
    text, img = torch.randn(K, E, device=dev) * 3, torch.randn(N, E, device=dev) * 0.2
    out = ops.similarity(text, img, 100.0, mode)
    t, i = text.double(), img.double()
    logits = (t / t.norm(dim=-1, keepdim=True)) @ (i / i.norm(dim=-1, keepdim=True)).t() * 100.0
    ref = {_lib.SIM_LOGIT: logits, _lib.SIM_SOFTMAX: logits.softmax(0), _lib.SIM_SIGMOID: logits.sigmoid()}[mode]
    torch.cuda.synchronize()
    tol = 2e-3 if mode == _lib.SIM_LOGIT else 2e-4
    assert (out.double() - ref).abs().max().item() < tol


def test_segment_mean(dev):
    K, N, G = 9, 100, 7
    p = torch.rand(K, N, device=dev)
    seg = torch.randint(0, G - 1, (N,), device=dev, dtype=torch.int32)  # group G-1 stays empty
    out = ops.segment_mean(p, seg, G)
    for g in range(G):
        sel = seg == g
        want = p[:, sel].mean(dim=1) if sel.any() else torch.zeros(K, device=dev)
        assert (out[:, g] - want).abs().max().item() < 1e-6


# ---------------------------------------------------------------------------
# crop kernel: bit-exact against the oracle and the golden vectors
# ---------------------------------------------------------------------------
def _entries(rows):
    arr = (_lib.CropEntry * len(rows))()
    for e, r in zip(arr, rows):
        (e.frame_off, e.mask_a_off, e.mask_b_off, e.frame_w, e.frame_h, e.x1, e.y1, e.x2, e.y2, e.mode) = r
    return arr


def test_crop_matches_golden_and_oracle(dev):
    from golden_io import load_functions, object_box
    from oracle import crop_ref, resample_ref
    meta, z, masks = load_functions()
    frame = z["frame"]
    H, W = meta["H"], meta["W"]
    frames_t = torch.from_numpy(ops.pad_bytes(frame)).to(dev)
    masks_np = np.stack([m[..., 0].astype(np.uint8) for m in masks])
    masks_t = torch.from_numpy(ops.pad_bytes(masks_np)).to(dev)
    rows, want64 = [], []
    for i, mo in enumerate(meta["objects"]):
        x1, y1, x2, y2 = crop_ref.union_box([object_box(mo)])
        rows.append((0, i * H * W, -1, W, H, x1, y1, x2, y2, _lib.CROP_OBJECT))
        want64.append(z[f"obj_px64_{i}"])
    for mp in meta["pairs"]:
        a, c = mp["from"], mp["to"]
        x1, y1, x2, y2 = crop_ref.union_box([meta["bboxes"][a], meta["bboxes"][c]])
        rows.append((0, a * H * W, c * H * W, W, H, x1, y1, x2, y2, _lib.CROP_PAIR))
        want64.append(z[f"pair_px64_{a}_{c}"])
    rows.append((0, -1, -1, W, H, 0, 0, W, H, _lib.CROP_PLAIN))
    want64.append(z["frame_px64"])
    ent = _entries(rows)
    out = ops.crop_preprocess(frames_t, masks_t, ent, 64).cpu().numpy()
    for i, w in enumerate(want64):
        assert np.array_equal(out[i], w), f"crop {i}: {np.abs(out[i] - w).max()} max diff, {(out[i] != w).sum()} px"
    # S = 224 against the golden vectors and the oracle (down- and up-scaling)
    ent224 = _entries(rows[:4])
    out224 = ops.crop_preprocess(frames_t, masks_t, ent224, 224).cpu().numpy()
    for i in range(2):
        assert np.array_equal(out224[i], z[f"obj_px224_{i}"])
    for i in (2, 3):
        full = crop_ref.extract_single_object(frame, masks[i], 0.8)
        _, _, _, _, _, x1, y1, x2, y2, _ = rows[i]
        assert np.array_equal(out224[i], resample_ref.preprocess(full[y1:y2, x1:x2], 224))
    # patch (im2col) layout in bf16 equals the CHW result regrouped
    p = 16
    outp = ops.crop_preprocess(frames_t, masks_t, ent, 64, layout=_lib.LAYOUT_PATCHES, patch=p, out_dtype=_lib.DTYPE_BF16)
    G = 64 // p
    chw = torch.from_numpy(out).to(torch.bfloat16).float()
    want = chw.view(len(rows), 3, G, p, G, p).permute(0, 2, 4, 1, 3, 5).reshape(len(rows) * G * G, 3 * p * p)
    assert torch.equal(outp.float().cpu(), want)


@pytest.mark.parametrize("bands", ["auto", "mid", "all"])
def test_crop_random_boxes_vs_oracle(dev, bands, monkeypatch):
    """Seeded random frames / boxes incl. heavy down-scaling, odd alignments, 1-2 px wide crops; S = 64 and 336;
    one, a few and all bands of a crop per workgroup; f32 stores and the LDS-transposed bf16 stores."""
    from oracle import crop_ref, resample_ref
    rng = np.random.default_rng(5)
    H, W = 333, 517
    frames = rng.integers(0, 256, size=(2, H, W, 3), dtype=np.uint8)
    masks = rng.random((3, H, W)) < 0.5
    frames_t = torch.from_numpy(ops.pad_bytes(frames)).to(dev)
    masks_t = torch.from_numpy(ops.pad_bytes(masks.astype(np.uint8))).to(dev)
    boxes = [(0, 0, W, H), (1, 2, 3, H), (5, 7, 505, 9), (100, 50, 101, 51), (13, 17, 517, 333), (250, 100, 314, 164),
             (3, 3, 227, 227), (7, 300, 500, 333)]
    for _ in range(12):
        x1, y1 = int(rng.integers(0, W - 2)), int(rng.integers(0, H - 2))
        boxes.append((x1, y1, int(rng.integers(x1 + 1, W + 1)), int(rng.integers(y1 + 1, H + 1))))
    rows = []
    for i, (x1, y1, x2, y2) in enumerate(boxes):
        f, a, b = i % 2, i % 3, (i + 1) % 3
        mode = (_lib.CROP_OBJECT, _lib.CROP_PAIR, _lib.CROP_PLAIN)[i % 3]
        rows.append((f * H * W * 3, a * H * W, b * H * W, W, H, x1, y1, x2, y2, mode))
    ent = _entries(rows)
    for S, alpha, wa in ((64, 0.5, 0.8), (336, 0.3, 0.65)):
        if bands != "auto":
            monkeypatch.setenv("SGCLIP_CROP_BANDS", str(S // 16 if bands == "all" else (2 if S == 64 else 3)))
        out_t = ops.crop_preprocess(frames_t, masks_t, ent, S, alpha=alpha, white_alpha=wa)
        out16 = ops.crop_preprocess(frames_t, masks_t, ent, S, alpha=alpha, white_alpha=wa, out_dtype=_lib.DTYPE_BF16)
        assert torch.equal(out16, out_t.to(torch.bfloat16))
        out = out_t.cpu().numpy()
        for i, r in enumerate(rows):
            foff, aoff, boff, _, _, x1, y1, x2, y2, mode = r
            fr = frames[foff // (H * W * 3)]
            ma, mb = masks[aoff // (H * W)][..., None], masks[boff // (H * W)][..., None]
            if mode == _lib.CROP_OBJECT:
                full = crop_ref.extract_single_object(fr, ma, wa)
            elif mode == _lib.CROP_PAIR:
                full = crop_ref.extract_object_subject(fr, ma, mb, alpha, wa)
            else:
                full = fr
            want = resample_ref.preprocess(full[y1:y2, x1:x2], S)
            assert np.array_equal(out[i], want), f"S={S} crop {i} box {boxes[i]}: {(out[i] != want).sum()} values differ"


def test_crop_of_4k_frames_beyond_the_lds_staging(dev):
    """Source spans too wide for one LDS staging (the reference slices any box of any frame, model_utils.py:117, and
    uses the whole first frame as the stand-in pair, llava_clip_model_v3.py:416-420): a whole 3840 x 2160 frame, a
    large pair union box and an object box on it, mixed in one call with small crops; S = 224 and 336, both layouts'
    bits against the oracle."""
    from oracle import crop_ref, resample_ref
    rng = np.random.default_rng(21)
    H, W = 2160, 3840
    frame = (rng.integers(0, 256, size=(H, W, 3)) // 2 + np.linspace(0, 127, W)[None, :, None]).astype(np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    masks = np.stack([((xx - 1500) / 1400.0) ** 2 + ((yy - 1000) / 900.0) ** 2 <= 1.0,
                      ((xx - 2600) / 1100.0) ** 2 + ((yy - 1300) / 800.0) ** 2 <= 1.0])
    frames_t = torch.from_numpy(ops.pad_bytes(frame)).to(dev)
    masks_t = torch.from_numpy(ops.pad_bytes(masks.astype(np.uint8))).to(dev)
    boxes = [(0, 0, W, H, _lib.CROP_PLAIN), (100, 100, 3700, 2100, _lib.CROP_PAIR), (100, 100, 2900, 1900, _lib.CROP_OBJECT),
             (17, 33, 3840, 2160, _lib.CROP_PAIR), (1000, 1000, 1200, 1300, _lib.CROP_PAIR), (5, 5, 3001, 120, _lib.CROP_OBJECT)]
    rows = [(0, 0, H * W, W, H, x1, y1, x2, y2, mode) for (x1, y1, x2, y2, mode) in boxes]
    ent = _entries(rows)
    for S in (224, 336):
        out = ops.crop_preprocess(frames_t, masks_t, ent, S).cpu().numpy()
        for i, (x1, y1, x2, y2, mode) in enumerate(boxes):
            if mode == _lib.CROP_OBJECT:
                full = crop_ref.extract_single_object(frame, masks[0][..., None], 0.8)
            elif mode == _lib.CROP_PAIR:
                full = crop_ref.extract_object_subject(frame, masks[0][..., None], masks[1][..., None], 0.5, 0.8)
            else:
                full = frame
            want = resample_ref.preprocess(full[y1:y2, x1:x2], S)
            assert np.array_equal(out[i], want), f"S={S} crop {i} box {boxes[i]}: {(out[i] != want).sum()} values differ"
    p = 32
    outp = ops.crop_preprocess(frames_t, masks_t, ent, 224, layout=_lib.LAYOUT_PATCHES, patch=p, out_dtype=_lib.DTYPE_BF16)
    G = 224 // p
    chw = ops.crop_preprocess(frames_t, masks_t, ent, 224).to(torch.bfloat16).float().cpu()
    want = chw.view(len(rows), 3, G, p, G, p).permute(0, 2, 4, 1, 3, 5).reshape(len(rows) * G * G, 3 * p * p)
    assert torch.equal(outp.float().cpu(), want)


def test_crop_errors(dev):
    frames_t = torch.zeros(100 * 100 * 3 + ops.PAD, dtype=torch.uint8, device=dev)
    with pytest.raises(AssertionError, match="image bbox issue"):
        ops.crop_preprocess(frames_t, None, _entries([(0, -1, -1, 100, 100, 10, 10, 10, 20, _lib.CROP_PLAIN)]), 64)
    with pytest.raises(_lib.SgclipError):
        ops.crop_preprocess(frames_t, None, _entries([(0, -1, -1, 100, 100, 10, 10, 120, 20, _lib.CROP_PLAIN)]), 64)
    with pytest.raises(_lib.SgclipError):
        ops.crop_preprocess(frames_t, None, _entries([(0, -1, -1, 100, 100, 10, 10, 50, 20, _lib.CROP_OBJECT)]), 64)


def test_crop_order_hint_changes_the_schedule_not_the_result(dev):
    """The order field moves crops between workgroups (largest first); every crop's output stays where its index says,
    bit for bit.  Anything but all zeros or a permutation of 1 .. n is refused."""
    from laser_amd.predicate_model import CROP_DTYPE
    rng = np.random.default_rng(5)
    H, W, n = 360, 640, 300
    frames = rng.integers(0, 256, size=(2, H, W, 3), dtype=np.uint8)
    masks = rng.random((3, H, W)) < 0.5
    frames_t = torch.from_numpy(ops.pad_bytes(frames)).to(dev)
    masks_t = torch.from_numpy(ops.pad_bytes(masks.astype(np.uint8))).to(dev)
    rows = []
    for i in range(n):
        w, h = int(np.exp(rng.uniform(np.log(8), np.log(W)))), int(np.exp(rng.uniform(np.log(8), np.log(H))))
        x1, y1 = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
        rows.append(((i % 2) * H * W * 3, (i % 3) * H * W, ((i + 1) % 3) * H * W, W, H, x1, y1, x1 + w, y1 + h,
                     (_lib.CROP_OBJECT, _lib.CROP_PAIR, _lib.CROP_PLAIN)[i % 3], 0))
    table = np.array(rows, dtype=CROP_DTYPE)
    kw = dict(layout=_lib.LAYOUT_PATCHES, patch=32, out_dtype=_lib.DTYPE_BF16)
    old = ops.CROP_ORDER
    try:
        ops.CROP_ORDER = False
        plain = ops.crop_preprocess(frames_t, masks_t, table.copy(), 224, **kw)
        ops.CROP_ORDER = True
        hinted_table = table.copy()
        hinted = ops.crop_preprocess(frames_t, masks_t, hinted_table, 224, **kw)
    finally:
        ops.CROP_ORDER = old
    torch.cuda.synchronize()
    assert sorted(hinted_table["order"].tolist()) == list(range(1, n + 1))
    first = int(hinted_table["order"][0]) - 1
    area = (table["x2"] - table["x1"]).astype(np.int64) * (table["y2"] - table["y1"])
    assert area[first] == area.max()
    assert torch.equal(plain, hinted)
    # a reversed and a random permutation given by the caller (the table uploaded as it is)
    for perm in (np.arange(n)[::-1], rng.permutation(n)):
        t = table.copy()
        t["order"] = perm.astype(np.int32) + 1
        tdev = torch.from_numpy(t.view(np.uint8).reshape(-1).copy()).to(dev)
        got = ops.crop_preprocess(frames_t, masks_t, t, 224, entries_dev=tdev, entries_dev_ready=True, **kw)
        torch.cuda.synchronize()
        assert torch.equal(plain, got)
    bad = table.copy()
    bad["order"] = 1                      # not a permutation
    tdev = torch.from_numpy(bad.view(np.uint8).reshape(-1).copy()).to(dev)
    with pytest.raises(_lib.SgclipError, match="permutation"):
        ops.crop_preprocess(frames_t, masks_t, bad, 224, entries_dev=tdev, entries_dev_ready=True, **kw)


def test_crop_sweep_at_720p_vs_oracle(dev):
    """Boxes drawn log-uniformly in size on 1280 x 720 frames (the benchmark's distribution) plus thin, tiny and
    frame-sized ones, all crop modes, S = 224: every variant of the matrix-core crop kernel and the vector kernel it
    leaves the rest to, bit for bit against the oracle (tools/fuzz_crop.py runs the same sweep at any length)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_crop", os.path.join(ROOT, "tools", "fuzz_crop.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    from oracle import crop_ref, resample_ref
    rng = np.random.default_rng(77)
    H, W = 720, 1280
    frames = (rng.integers(0, 256, size=(2, H, W, 3)) // 2 + np.linspace(0, 127, W)[None, None, :, None]).astype(np.uint8)
    masks = rng.random((3, H, W)) < 0.4
    frames_t = torch.from_numpy(ops.pad_bytes(frames)).to(dev)
    masks_t = torch.from_numpy(ops.pad_bytes(masks.astype(np.uint8))).to(dev)
    boxes = fz.draw_boxes(rng, 160, W, H)
    rows = []
    for i, (x1, y1, x2, y2) in enumerate(boxes):
        mode = (_lib.CROP_OBJECT, _lib.CROP_PAIR, _lib.CROP_PAIR, _lib.CROP_PLAIN)[i % 4]
        rows.append(((i % 2) * H * W * 3, (i % 3) * H * W, ((i + 1) % 3) * H * W, W, H, x1, y1, x2, y2, mode))
    out = ops.crop_preprocess(frames_t, masks_t, _entries(rows), 224).cpu().numpy()
    blended = {}
    for i, (foff, aoff, boff, _, _, x1, y1, x2, y2, mode) in enumerate(rows):
        key = (foff, aoff, boff, mode)
        if key not in blended:
            fr = frames[foff // (H * W * 3)]
            ma, mb = masks[aoff // (H * W)][..., None], masks[boff // (H * W)][..., None]
            blended[key] = (crop_ref.extract_single_object(fr, ma, 0.8) if mode == _lib.CROP_OBJECT else
                            crop_ref.extract_object_subject(fr, ma, mb, 0.5, 0.8) if mode == _lib.CROP_PAIR else fr)
        want = resample_ref.preprocess(blended[key][y1:y2, x1:x2], 224)
        assert np.array_equal(out[i], want), f"crop {i} box {boxes[i]} mode {mode}: {(out[i] != want).sum()} values differ"


def test_gemm_many_tiles_with_padded_leading_dimensions(dev):
    """The persistent kernel's lane-offset addressing with operands whose rows are further apart than K (views into
    wider buffers, as the towers' strided operands are), ragged in both tile directions."""
    M, N, K, pad = 66000 + 40, 768 + 128, 448, 64
    A = _rand_bf16((M, K + pad), dev, 1.0, 21)[:, :K]
    W = _rand_bf16((N, K + 2 * pad), dev, K ** -0.5, 22)[:, pad:pad + K]
    bias = torch.randn(N, device=dev)
    ref = A.float() @ W.float().t() + bias
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16(A, W, out, _lib.EPI_BIAS_BF16, bias=bias)
    assert (out.float() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    x0 = torch.randn(M, N, device=dev)
    x = x0.clone()
    ops.gemm_bf16(A, W, x, _lib.EPI_BIAS_RESID_F32, bias=bias)
    assert (x - (x0 + ref)).abs().max().item() < 2e-3

"""Torch-facing wrappers of the backward building blocks of libsgclip_hip.so.

Torch supplies device memory, the stream and views; every arithmetic step of the
train step's backward pass is one of the calls below (or `ops.gemm_bf16`).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib
from .ops import _ptr, _stream, gemm_bf16

BF16, F32 = torch.bfloat16, torch.float32


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == BF16:
        return _lib.DTYPE_BF16
    if t.dtype == F32:
        return _lib.DTYPE_F32
    raise TypeError(f"expected bf16 or f32, got {t.dtype}")


_TN_WGRAD = os.environ.get("SGCLIP_TN_WGRAD", "1") != "0"   # 0: transposed copies + the NT kernel (comparison)


def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def transpose_bf16(x: torch.Tensor, ld_out: Optional[int] = None) -> torch.Tensor:
    """bf16 [R, C] -> [C, ld_out] with out[c, r] = x[r, c]; columns R..ld_out-1 are zero."""
    assert x.dtype == BF16 and x.dim() == 2 and x.stride(1) == 1
    R, Cc = x.shape
    ld_out = ld_out or R
    out = torch.empty((Cc, ld_out), dtype=BF16, device=x.device)
    _lib.check(_lib.load().sgclip_transpose_bf16(x.data_ptr(), x.stride(0), R, Cc, out.data_ptr(), ld_out, _stream(x)))
    return out


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """f32 [N] (+)= column sums of x [M, N] (bf16 or f32)."""
    assert x.dim() == 2 and x.stride(1) == 1
    M, N = x.shape
    if out is None:
        out = torch.zeros(N, dtype=F32, device=x.device)
    _lib.check(_lib.load().sgclip_colsum(x.data_ptr(), _dtype_code(x), x.stride(0), M, N, out.data_ptr(), _stream(x)))
    return out


def quick_gelu(u: torch.Tensor) -> torch.Tensor:
    assert u.dtype == BF16 and u.is_contiguous()
    g = torch.empty_like(u)
    _lib.check(_lib.load().sgclip_quick_gelu(u.data_ptr(), g.data_ptr(), u.numel(), _stream(u)))
    return g


def quick_gelu_bwd(u: torch.Tensor, dg: torch.Tensor) -> torch.Tensor:
    assert u.dtype == BF16 and dg.dtype == BF16 and u.is_contiguous() and dg.is_contiguous() and u.shape == dg.shape
    du = torch.empty_like(u)
    _lib.check(_lib.load().sgclip_quick_gelu_bwd(u.data_ptr(), dg.data_ptr(), du.data_ptr(), u.numel(), _stream(u)))
    return du


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    assert x.dtype == F32 and x.is_contiguous()
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    _lib.check(_lib.load().sgclip_cast_f32_bf16(x.data_ptr(), y.data_ptr(), x.numel(), _stream(x)))
    return y


def cast_bf16_colsum(x: torch.Tensor, colsum_out: torch.Tensor) -> torch.Tensor:
    """bf16 copy of x (f32 [M, N]) with colsum_out (f32 [N]) += column sums of x, in one pass over x."""
    assert x.dtype == F32 and x.dim() == 2 and x.is_contiguous()
    M, N = x.shape
    if N % 8 or N > 2048:
        colsum(x, colsum_out)
        return cast_bf16(x)
    assert colsum_out.dtype == F32 and colsum_out.is_contiguous() and colsum_out.numel() == N
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    _lib.check(_lib.load().sgclip_cast_colsum_f32_bf16(x.data_ptr(), y.data_ptr(), M, N, colsum_out.data_ptr(), _stream(x)))
    return y


def fc2_dgrad_through_gelu(dy: torch.Tensor, w_t: torch.Tensor, u: torch.Tensor, zero_bias: torch.Tensor,
                           colsum_out: torch.Tensor) -> torch.Tensor:
    """du = (dy w_t^T) * quick_gelu'(u) (bf16 [M, N]) with colsum_out (f32 [N]) += column sums of du: the gradient at
    fc1's output and fc1's bias gradient from one GEMM (dy bf16 [M, K], w_t bf16 [N, K] = fc2's weight transposed)."""
    assert dy.dtype == BF16 and w_t.dtype == BF16 and u.dtype == BF16 and u.is_contiguous()
    assert dy.stride(1) == 1 and w_t.stride(1) == 1
    M, K = dy.shape
    N = w_t.shape[0]
    assert w_t.shape[1] == K and u.shape == (M, N)
    assert colsum_out.dtype == F32 and colsum_out.is_contiguous() and colsum_out.numel() == N and zero_bias.numel() >= N
    du = torch.empty_like(u)
    _lib.check(_lib.load().sgclip_gemm_mulgelugrad_bf16(dy.data_ptr(), dy.stride(0), w_t.data_ptr(), w_t.stride(0), du.data_ptr(),
                                                        N, M, N, K, zero_bias.data_ptr(), u.data_ptr(), colsum_out.data_ptr(),
                                                        _stream(dy)))
    return du


def layernorm_bwd(x: torch.Tensor, w: torch.Tensor, eps: float, dy: torch.Tensor, dw: torch.Tensor, db: torch.Tensor,
                  dx: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """dx (f32 [rows, d]; added to `dx` when accumulate) and dw / db (+=) of y = LayerNorm(x) w + b."""
    assert x.dtype == F32 and x.dim() == 2 and x.stride(1) == 1 and dy.is_contiguous() and dy.shape == x.shape
    rows, d = x.shape
    if dx is None:
        assert not accumulate
        dx = torch.empty((rows, d), dtype=F32, device=x.device)
    assert dx.dtype == F32 and dx.is_contiguous() and dx.shape == x.shape
    _lib.check(_lib.load().sgclip_layernorm_bwd(x.data_ptr(), x.stride(0), rows, d, w.data_ptr(), eps, dy.data_ptr(),
                                                _dtype_code(dy), dx.data_ptr(), 1 if accumulate else 0, dw.data_ptr(),
                                                db.data_ptr(), _stream(x)))
    return dx


def attention_bwd(qkv: torch.Tensor, dout: torch.Tensor, n_seq: int, T: int, heads: int, causal: bool,
                  key_mask: Optional[torch.Tensor] = None, dbias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dqkv; with dbias (f32 [3d]) the column sums of dqkv are ADDED to it (the q | k | v bias gradient)."""
    assert qkv.dtype == BF16 and dout.dtype == BF16 and qkv.is_contiguous() and dout.is_contiguous()
    assert qkv.shape == (n_seq * T, 3 * heads * 64) and dout.shape == (n_seq * T, heads * 64)
    dqkv = torch.empty_like(qkv)
    if dbias is None:
        _lib.check(_lib.load().sgclip_attention_bwd(qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), n_seq, T, heads,
                                                    1 if causal else 0, _ptr(key_mask), _stream(qkv)))
    else:
        assert dbias.dtype == F32 and dbias.is_contiguous() and dbias.numel() == 3 * heads * 64
        _lib.check(_lib.load().sgclip_attention_bwd_bias(qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), dbias.data_ptr(), n_seq,
                                                         T, heads, 1 if causal else 0, _ptr(key_mask), _stream(qkv)))
    return dqkv


def matmul_f32(A: torch.Tensor, B: torch.Tensor, transA: bool = False, transB: bool = False, alpha: float = 1.0,
               out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """alpha op(A) op(B) in fp32; A, B row-major 2-D."""
    assert A.dtype == F32 and B.dtype == F32 and A.stride(1) == 1 and B.stride(1) == 1
    M, K = (A.shape[1], A.shape[0]) if transA else A.shape
    Kb, N = (B.shape[1], B.shape[0]) if transB else B.shape
    assert K == Kb, (A.shape, B.shape, transA, transB)
    if out is None:
        assert not accumulate
        out = torch.empty((M, N), dtype=F32, device=A.device)
    _lib.check(_lib.load().sgclip_matmul_f32(A.data_ptr(), A.stride(0), 1 if transA else 0, B.data_ptr(), B.stride(0),
                                             1 if transB else 0, out.data_ptr(), out.stride(0), M, N, K, float(alpha),
                                             1 if accumulate else 0, _stream(A)))
    return out


def dot_f32(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """0-dim f32 <a, b>."""
    assert a.dtype == F32 and b.dtype == F32 and a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
    out = torch.zeros((), dtype=F32, device=a.device)
    _lib.check(_lib.load().sgclip_dot_f32(a.data_ptr(), b.data_ptr(), a.numel(), out.data_ptr(), _stream(a)))
    return out


def prob_bwd(probs: torch.Tensor, dprobs: torch.Tensor, mode: int) -> torch.Tensor:
    assert probs.dtype == F32 and probs.is_contiguous() and dprobs.is_contiguous() and probs.shape == dprobs.shape
    K, N = probs.shape
    dl = torch.empty_like(probs)
    _lib.check(_lib.load().sgclip_prob_bwd(probs.data_ptr(), dprobs.data_ptr(), K, N, mode, dl.data_ptr(), _stream(probs)))
    return dl


def normalize_rows(x: torch.Tensor) -> torch.Tensor:
    assert x.dtype == F32 and x.is_contiguous() and x.dim() == 2
    y = torch.empty_like(x)
    _lib.check(_lib.load().sgclip_normalize_rows(x.data_ptr(), x.shape[0], x.shape[1], y.data_ptr(), _stream(x)))
    return y


def normalize_bwd(x: torch.Tensor, dxhat: torch.Tensor) -> torch.Tensor:
    assert x.dtype == F32 and x.is_contiguous() and dxhat.is_contiguous() and x.shape == dxhat.shape
    dx = torch.empty_like(x)
    _lib.check(_lib.load().sgclip_normalize_bwd(x.data_ptr(), dxhat.data_ptr(), x.shape[0], x.shape[1], dx.data_ptr(),
                                                _stream(x)))
    return dx


def segment_mean_bwd(dmean: torch.Tensor, seg: torch.Tensor, inv_count: torch.Tensor, N: int) -> torch.Tensor:
    assert dmean.dtype == F32 and dmean.is_contiguous() and seg.dtype == torch.int32 and inv_count.dtype == F32
    K, G = dmean.shape
    out = torch.empty((K, N), dtype=F32, device=dmean.device)
    _lib.check(_lib.load().sgclip_segment_mean_bwd(dmean.data_ptr(), seg.data_ptr(), inv_count.data_ptr(), K, N, G,
                                                   out.data_ptr(), _stream(dmean)))
    return out


def embed_text(table: torch.Tensor, pos: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """f32 [K*T, d] = table[ids] + pos[:T]; ids int32 [K, T] on the device."""
    assert table.dtype == F32 and pos.dtype == F32 and ids.dtype == torch.int32 and ids.is_contiguous()
    K, T = ids.shape
    d = table.shape[1]
    x = torch.empty((K * T, d), dtype=F32, device=table.device)
    _lib.check(_lib.load().sgclip_embed_text(table.data_ptr(), pos.data_ptr(), ids.data_ptr(), K, T, d, x.data_ptr(), _stream(table)))
    return x


def embed_cls(cls: torch.Tensor, pos: torch.Tensor, x: torch.Tensor, n_images: int, T: int):
    """Writes cls + pos[0] into row 0 of every image of x (f32 [n_images*T, d])."""
    d = x.shape[1]
    _lib.check(_lib.load().sgclip_embed_cls(cls.data_ptr(), pos.data_ptr(), n_images, T, d, x.data_ptr(), _stream(x)))


def rows_scatter_add(src: torch.Tensor, idx: torch.Tensor, dst: torch.Tensor):
    """dst[idx[r]] += src[r] (f32 rows)."""
    assert src.dtype == F32 and dst.dtype == F32 and src.is_contiguous() and dst.is_contiguous()
    assert idx.dtype == torch.int32 and idx.numel() == src.shape[0] and src.shape[1] == dst.shape[1]
    _lib.check(_lib.load().sgclip_rows_scatter_add(src.data_ptr(), idx.data_ptr(), src.shape[0], src.shape[1], dst.data_ptr(),
                                                   _stream(src)))


def adamw(param: torch.Tensor, grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
          beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.0):
    for t in (param, grad, m, v):
        assert t.dtype == F32 and t.is_contiguous() and t.numel() == param.numel()
    _lib.check(_lib.load().sgclip_adamw(param.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), param.numel(), lr,
                                        beta1, beta2, eps, weight_decay, step, _stream(param)))


# -- linear layers on the MFMA GEMM ---------------------------------------------------------
def linear_dgrad(dy: torch.Tensor, w_t: torch.Tensor, zero_bias: torch.Tensor) -> torch.Tensor:
    """dx bf16 [M, in] = dy [M, out] W, with w_t = W^T stored [in, out] (bf16)."""
    M = dy.shape[0]
    dx = torch.empty((M, w_t.shape[0]), dtype=BF16, device=dy.device)
    return gemm_bf16(dy, w_t, dx, _lib.EPI_BIAS_BF16, bias=zero_bias)


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out f32 [out, in] += dy^T x for dy bf16 [M, out], x bf16 [M, in] (out: zeros when omitted); M is zero-padded
    to a multiple of 64.  The reduction over M (all tokens of the batch) is split over workgroups that add into
    `out` atomically, so repeated calls accumulate."""
    M = dy.shape[0]
    if out is None:
        out = torch.zeros((dy.shape[1], x.shape[1]), dtype=F32, device=dy.device)
    assert out.dtype == F32 and out.shape == (dy.shape[1], x.shape[1]) and out.stride(1) == 1
    if _TN_WGRAD and dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0 and dy.stride(1) == 1 and x.stride(1) == 1 \
            and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0:
        # both operands as they lie in memory (rows = the reduction index); fragments are transposed on the LDS read
        _lib.check(_lib.load().sgclip_gemm_tn_bf16(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), out.data_ptr(),
                                                   out.stride(0), dy.shape[1], x.shape[1], M, _stream(dy)))
        return out
    Mp = pad64(M)
    dyt = transpose_bf16(dy, Mp)
    xt = transpose_bf16(x, Mp)
    return gemm_bf16(dyt, xt, out, _lib.EPI_F32_ATOMIC)

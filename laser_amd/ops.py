"""Torch-facing wrappers over the C ABI (torch supplies device memory and the
stream; all arithmetic happens in libsgclip_hip.so)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .config import ClipConfig

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _stream(t: Optional[torch.Tensor] = None) -> int:
    """The HIP stream a call is queued on: torch's current stream of the current device.  `t` is a tensor the
    kernel will touch: its device must be the current one (HIP launches on the current device; a pointer of
    another GPU would fault there), so callers with tensors elsewhere run under `torch.cuda.device(...)`,
    as PredicateModel does for its own device."""
    if t is not None and t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"tensor on {t.device} but the current HIP device is cuda:{torch.cuda.current_device()}: "
                           f"run this call under torch.cuda.device({t.device.index})")
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------
# level tables (host, 256 entries each)
# ---------------------------------------------------------------------------
def blend_luts(alpha: float = 0.5, white_alpha: float = 0.8) -> np.ndarray:
    """u8[768] = white | keep | tint.

    white[p]: pixel outside the mask(s), the 8-bit weighted sum
              p*(1-white_alpha) + 255*white_alpha in float32, rounded to nearest even
              (laser/models/model_utils.py:54 and :138);
    keep[p]:  pixel inside the object mask, p*(1-white_alpha) + p*white_alpha (:51-54);
    tint[p]:  tinted channel, trunc(p + (255-p)*alpha) (:129,:132)."""
    p = np.arange(256, dtype=np.float32)
    a, b = np.float32(1 - white_alpha), np.float32(white_alpha)
    white = np.clip(np.rint(p * a + np.float32(255) * b), 0, 255).astype(np.uint8)
    keep = np.clip(np.rint(p * a + p * b), 0, 255).astype(np.uint8)
    lv = np.arange(256, dtype=np.uint8)
    tint = np.clip(lv + (255 - lv) * alpha, 0, 255).astype(np.uint8)
    return np.concatenate([white, keep, tint])


def normalize_lut(mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """f32[768]: level v of channel c -> ((v * (1/255)) - mean[c]) / std[c] with the
    dtype sequence of the HF processor (float64 rescale -> float32 -> float32 normalise)."""
    lv = np.arange(256, dtype=np.uint8)
    x = (lv.astype(np.float64) * (1 / 255)).astype(np.float32)
    m, s = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32)
    return np.concatenate([(x - m[c]) / s[c] for c in range(3)]).astype(np.float32)


# ---------------------------------------------------------------------------
# crop kernel
# ---------------------------------------------------------------------------
def crop_preprocess(frames: torch.Tensor, masks: Optional[torch.Tensor], entries, S: int, *, layout=_lib.LAYOUT_CHW,
                    patch: int = 0, out_dtype=_lib.DTYPE_F32, alpha: float = 0.5, white_alpha: float = 0.8,
                    mean=CLIP_MEAN, std=CLIP_STD, out: Optional[torch.Tensor] = None,
                    entries_dev: Optional[torch.Tensor] = None, entries_dev_ready: bool = False) -> torch.Tensor:
    """frames / masks: flat u8 CUDA buffers (with PAD = 16 bytes of slack at the end, see
    `pad_bytes`); entries: ctypes array of _lib.CropEntry.  Returns [N,3,S,S] (CHW) or
    [N*P, kpad] (patch layout)."""
    lib = _lib.load()
    n = len(entries)
    dev = frames.device
    tdt = torch.bfloat16 if out_dtype == _lib.DTYPE_BF16 else torch.float32
    kpad = lib.sgclip_patch_kpad(patch) if layout == _lib.LAYOUT_PATCHES else 0
    if out is None:
        if layout == _lib.LAYOUT_PATCHES:
            P = (S // patch) ** 2
            # pad columns are never written by the kernel and must stay zero
            out = torch.zeros((n * P, kpad), dtype=tdt, device=dev)
        else:
            out = torch.empty((n, 3, S, S), dtype=tdt, device=dev)
    if entries_dev is None and isinstance(entries, np.ndarray) and entries.dtype.names and "order" in entries.dtype.names:
        # scheduling hint: largest boxes first, so that the launch does not end on a few big crops (their
        # workgroups take ten times as long as a small crop's); few crops: table order
        if n >= CROP_ORDER_MIN and CROP_ORDER:
            area = (entries["x2"] - entries["x1"]).astype(np.int64) * (entries["y2"] - entries["y1"])
            entries["order"] = np.argsort(-area, kind="stable").astype(np.int32) + 1
        else:
            entries["order"] = 0
    pinned = None
    if entries_dev is None and n and PINNED_TABLES:
        # the table goes up from pinned memory with an asynchronous copy queued on this stream: a copy from pageable
        # memory may make the host wait for the work queued before it, which ends the host's run ahead of the GPU
        if isinstance(entries, np.ndarray):
            pinned = torch.from_numpy(entries.view(np.uint8).reshape(-1)).pin_memory()
        else:
            pinned = torch.frombuffer(bytearray(C.string_at(C.addressof(entries), n * C.sizeof(_lib.CropEntry))),
                                      dtype=torch.uint8).pin_memory()
        entries_dev = pinned.to(dev, non_blocking=True)
        entries_dev_ready = True
    elif entries_dev is None:
        entries_dev = torch.empty(max(n, 1) * C.sizeof(_lib.CropEntry), dtype=torch.uint8, device=dev)
    lut_dev = _crop_luts(dev, alpha, white_alpha, tuple(mean), tuple(std))
    rc = lib.sgclip_crop_preprocess(
        frames.data_ptr(), frames.numel() - PAD, _ptr(masks), (masks.numel() - PAD) if masks is not None else 0,
        _host_table_ptr(entries) if n else None, entries_dev.data_ptr(), 1 if entries_dev_ready else 0, n,
        None, None, lut_dev.data_ptr(), S, layout, patch, kpad, out_dtype, out.data_ptr(), _stream(frames))
    _lib.check(rc)
    # entries_dev is consumed asynchronously: keep it (and its pinned source) alive until the stream has run
    out._sgclip_keepalive = (lut_dev, entries_dev, pinned)
    return out


_LUT_CACHE: Dict[tuple, torch.Tensor] = {}


def _crop_luts(dev, alpha, white_alpha, mean, std) -> torch.Tensor:
    """The crop kernel's level tables (blend tables, then the normalisation table) in HBM, uploaded once per
    device and parameter set with a blocking copy, so that every later call on any stream finds them in place."""
    key = (str(dev), float(alpha), float(white_alpha), mean, std)
    t = _LUT_CACHE.get(key)
    if t is None:
        host = np.concatenate([blend_luts(alpha, white_alpha), normalize_lut(mean, std).view(np.uint8)])
        assert host.size == _lib.CROP_LUT_BYTES
        t = _LUT_CACHE[key] = torch.from_numpy(host).to(dev)
        torch.cuda.current_stream(dev).synchronize()
    return t


def _host_table_ptr(entries) -> int:
    if isinstance(entries, np.ndarray):
        if entries.dtype.itemsize != C.sizeof(_lib.CropEntry) or not entries.flags["C_CONTIGUOUS"]:
            raise ValueError("crop table must be a C-contiguous array of 56-byte records")
        return entries.ctypes.data
    return C.addressof(entries)


PINNED_TABLES = os.environ.get("SGCLIP_PINNED_TABLES", "1") != "0"   # 0: small tables go up from pageable memory (comparison)
CROP_ORDER = os.environ.get("SGCLIP_CROP_ORDER", "1") != "0"   # crop tables get the largest-first scheduling hint
CROP_ORDER_MIN = 256                                           # ... from this many crops on
PAD = 16  # slack bytes appended to frame / mask buffers (the kernel copies rows in 16-byte pieces)


def pad_bytes(flat_u8: np.ndarray) -> np.ndarray:
    return np.concatenate([flat_u8.reshape(-1), np.zeros(PAD, dtype=np.uint8)])


# ---------------------------------------------------------------------------
# towers
# ---------------------------------------------------------------------------
def tower_desc(cfg: ClipConfig, kind: str) -> _lib.TowerDesc:
    if kind == "vision":
        return _lib.TowerDesc(0, cfg.v_hidden, cfg.v_ffn, cfg.v_layers, cfg.v_heads, cfg.v_tokens, cfg.proj, cfg.patch,
                              cfg.image, 0, 0, cfg.ln_eps)
    return _lib.TowerDesc(1, cfg.t_hidden, cfg.t_ffn, cfg.t_layers, cfg.t_heads, cfg.max_pos, cfg.proj, 0, 0, cfg.vocab,
                          cfg.eos_token_id, cfg.ln_eps)


class Tower:
    """One CLIP tower's weights packed into the library's two device blobs."""

    def __init__(self, cfg: ClipConfig, kind: str, sd: Dict[str, torch.Tensor], device):
        self.cfg, self.kind, self.device = cfg, kind, torch.device(device)
        self.desc = tower_desc(cfg, kind)
        lib = _lib.load()
        nb, nf, isb = C.c_int64(), C.c_int64(), C.c_int()
        _lib.check(lib.sgclip_tower_layout(C.byref(self.desc), -1, 0, C.byref(nb), C.byref(nf), C.byref(isb)))
        wb = torch.zeros(nb.value, dtype=torch.bfloat16)
        wf = torch.zeros(nf.value, dtype=torch.float32)

        def put(tid, layer, t):
            off, cnt = C.c_int64(), C.c_int64()
            _lib.check(lib.sgclip_tower_layout(C.byref(self.desc), tid, layer, C.byref(off), C.byref(cnt), C.byref(isb)))
            t = t.detach().to("cpu", torch.float32).reshape(-1)
            if t.numel() != cnt.value:
                raise ValueError(f"tensor {tid} layer {layer}: {t.numel()} elements, layout expects {cnt.value}")
            (wb if isb.value else wf)[off.value: off.value + cnt.value] = t.to(torch.bfloat16) if isb.value else t

        L = _lib
        pre = "vision_model" if kind == "vision" else "text_model"
        d = cfg.v_hidden if kind == "vision" else cfg.t_hidden
        layers = cfg.v_layers if kind == "vision" else cfg.t_layers
        if kind == "vision":
            kpad = lib.sgclip_patch_kpad(cfg.patch)
            pw = torch.zeros(d, kpad)
            pw[:, : 3 * cfg.patch * cfg.patch] = sd["vision_model.embeddings.patch_embedding.weight"].float().reshape(d, -1)
            put(L.T_PATCH_W, 0, pw)
            put(L.T_CLS, 0, sd["vision_model.embeddings.class_embedding"])
            put(L.T_POS, 0, sd["vision_model.embeddings.position_embedding.weight"])
            put(L.T_PRE_LN_W, 0, sd["vision_model.pre_layrnorm.weight"])
            put(L.T_PRE_LN_B, 0, sd["vision_model.pre_layrnorm.bias"])
            put(L.T_POST_LN_W, 0, sd["vision_model.post_layernorm.weight"])
            put(L.T_POST_LN_B, 0, sd["vision_model.post_layernorm.bias"])
            put(L.T_PROJ_W, 0, sd["visual_projection.weight"])
        else:
            put(L.T_TOK_EMB, 0, sd["text_model.embeddings.token_embedding.weight"])
            put(L.T_POS, 0, sd["text_model.embeddings.position_embedding.weight"])
            put(L.T_POST_LN_W, 0, sd["text_model.final_layer_norm.weight"])
            put(L.T_POST_LN_B, 0, sd["text_model.final_layer_norm.bias"])
            put(L.T_PROJ_W, 0, sd["text_projection.weight"])
        for i in range(layers):
            b = f"{pre}.encoder.layers.{i}"
            put(L.T_LN1_W, i, sd[f"{b}.layer_norm1.weight"])
            put(L.T_LN1_B, i, sd[f"{b}.layer_norm1.bias"])
            put(L.T_QKV_W, i, torch.cat([sd[f"{b}.self_attn.{n}_proj.weight"].float() for n in "qkv"], dim=0))
            put(L.T_QKV_B, i, torch.cat([sd[f"{b}.self_attn.{n}_proj.bias"].float() for n in "qkv"], dim=0))
            put(L.T_O_W, i, sd[f"{b}.self_attn.out_proj.weight"])
            put(L.T_O_B, i, sd[f"{b}.self_attn.out_proj.bias"])
            put(L.T_LN2_W, i, sd[f"{b}.layer_norm2.weight"])
            put(L.T_LN2_B, i, sd[f"{b}.layer_norm2.bias"])
            put(L.T_FC1_W, i, sd[f"{b}.mlp.fc1.weight"])
            put(L.T_FC1_B, i, sd[f"{b}.mlp.fc1.bias"])
            put(L.T_FC2_W, i, sd[f"{b}.mlp.fc2.weight"])
            put(L.T_FC2_B, i, sd[f"{b}.mlp.fc2.bias"])
        self.wb = wb.to(self.device)
        self.wf = wf.to(self.device)
        self._ws: Dict[int, torch.Tensor] = {}

    def _workspace(self, nbytes: int, slot: int = 0) -> torch.Tensor:
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < nbytes:
            ws = self._ws[slot] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return ws

    def encode_images(self, patches: torch.Tensor, n_images: int, out: Optional[torch.Tensor] = None,
                      ws_slot: int = 0) -> torch.Tensor:
        """ws_slot: calls that run concurrently on different streams need different workspaces."""
        lib = _lib.load()
        need = lib.sgclip_vit_workspace_bytes(C.byref(self.desc), n_images)
        ws = self._workspace(max(need, 256), ws_slot)
        if out is None:
            out = torch.empty((n_images, self.cfg.proj), dtype=torch.float32, device=self.device)
        _lib.check(lib.sgclip_vit_encode(C.byref(self.desc), self.wb.data_ptr(), self.wf.data_ptr(), patches.data_ptr(),
                                         n_images, ws.data_ptr(), ws.numel(), out.data_ptr(), _stream(patches)))
        return out

    def pooled_length(self, input_ids: torch.Tensor) -> int:
        """Positions that can influence the pooled (end-of-text) token of any row: the tower is
        causal, so everything after the pooled position of the longest row is dead computation.
        Same pooling rule as the reference's text model (modeling_clip.py:561-581)."""
        ids = input_ids.detach().cpu().numpy()
        if self.cfg.eos_token_id == 2:
            pos = ids.argmax(axis=1)
        else:
            hit = ids == self.cfg.eos_token_id
            pos = np.where(hit.any(axis=1), hit.argmax(axis=1), 0)
        return int(pos.max()) + 1 if len(pos) else 1

    def encode_text(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor],
                    seq_len: Optional[int] = None) -> torch.Tensor:
        """seq_len: leading positions to run (default: `pooled_length` when the ids are on the host,
        else all of them; callers with device-resident ids pass the value they computed once)."""
        lib = _lib.load()
        if seq_len is None and input_ids.device.type == "cpu":
            seq_len = self.pooled_length(input_ids)
        if seq_len is not None and seq_len < input_ids.shape[1]:
            input_ids = input_ids[:, :seq_len]
            attention_mask = None if attention_mask is None else attention_mask[:, :seq_len]
        K, T = input_ids.shape

        def up(t):
            # host tensors go up from pinned memory without blocking: a blocking copy waits for everything queued on
            # the stream before it and ends the host's run ahead of the GPU
            if t.device.type == "cpu" and PINNED_TABLES:
                return t.to(torch.int32).contiguous().pin_memory().to(self.device, non_blocking=True)
            return t.to(self.device, torch.int32).contiguous()
        ids = up(input_ids)
        am = None if attention_mask is None else up(attention_mask)
        need = lib.sgclip_text_workspace_bytes(C.byref(self.desc), K, T)
        ws = self._workspace(max(need, 256))
        out = torch.empty((K, self.cfg.proj), dtype=torch.float32, device=self.device)
        _lib.check(lib.sgclip_text_encode(C.byref(self.desc), self.wb.data_ptr(), self.wf.data_ptr(), ids.data_ptr(),
                                          _ptr(am), K, T, ws.data_ptr(), ws.numel(), out.data_ptr(), _stream(ids)))
        out._sgclip_keepalive = (ids, am)
        return out


# ---------------------------------------------------------------------------
# similarity / softmax / segment mean
# ---------------------------------------------------------------------------
def similarity(text: torch.Tensor, img: torch.Tensor, exp_logit_scale: float, mode: int,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    K, E = text.shape
    N = img.shape[0]
    text, img = text.contiguous(), img.contiguous()
    if out is None:
        out = torch.empty((K, N), dtype=torch.float32, device=text.device)
    ws = torch.empty(max(lib.sgclip_similarity_workspace_bytes(K, N), 256), dtype=torch.uint8, device=text.device)
    _lib.check(lib.sgclip_similarity(text.data_ptr(), img.data_ptr(), K, N, E, float(exp_logit_scale), mode,
                                     out.data_ptr(), ws.data_ptr(), ws.numel(), _stream(img)))
    out._sgclip_keepalive = (ws, text, img)
    return out


def segment_mean(probs: torch.Tensor, seg_ids: torch.Tensor, G: int) -> torch.Tensor:
    lib = _lib.load()
    K, N = probs.shape
    out = torch.empty((K, G), dtype=torch.float32, device=probs.device)
    seg = seg_ids.to(probs.device, torch.int32).contiguous()
    _lib.check(lib.sgclip_segment_mean(probs.data_ptr(), seg.data_ptr(), K, N, G, out.data_ptr(), _stream(probs)))
    out._sgclip_keepalive = (seg,)
    return out


TOPK_MAX = 16  # largest k of sgclip_topk_columns


def topk_columns(probs: torch.Tensor, k: int, tie_rank: Optional[torch.Tensor] = None):
    """k best rows of every column of probs [K, N], best first: (idx int32 [k, N], val f32 [k, N]).
    Ties rank by tie_rank[row] descending (None: lower row first)."""
    lib = _lib.load()
    if probs.dim() != 2 or probs.dtype != torch.float32 or probs.stride(1) != 1:
        raise ValueError("topk_columns expects a row-major f32 [K, N] tensor")
    K, N = probs.shape
    idx = torch.empty((k, N), dtype=torch.int32, device=probs.device)
    val = torch.empty((k, N), dtype=torch.float32, device=probs.device)
    tr = tie_rank.to(probs.device, torch.int32).contiguous() if tie_rank is not None else None
    _lib.check(lib.sgclip_topk_columns(probs.data_ptr(), probs.stride(0) if K > 1 else max(N, 1), K, N, _ptr(tr), k,
                                       idx.data_ptr(), val.data_ptr(), _stream()))
    idx._sgclip_keepalive = (tr, probs)
    return idx, val


# ---------------------------------------------------------------------------
# building blocks (tests / profiling)
# ---------------------------------------------------------------------------
def gemm_bf16(A: torch.Tensor, W: torch.Tensor, C_out: torch.Tensor, epilogue: int, bias=None, aux=None, aux_rows=0):
    lib = _lib.load()
    M, K = A.shape
    N = W.shape[0]
    if aux is not None and epilogue != _lib.EPI_PATCH_F32:
        # the second operand of an epilogue is addressed with C's row stride: residual (f32, read), pre-activation
        # (bf16, read) or activation (bf16, WRITTEN by EPI_BIAS_GELU_BOTH_BF16)
        want = torch.float32 if epilogue == _lib.EPI_BIAS_RESID_F32 else torch.bfloat16
        if aux.dtype != want or aux.dim() != 2 or aux.stride(1) != 1 or aux.stride(0) != C_out.stride(0) or aux.shape != C_out.shape:
            raise ValueError(f"epilogue {epilogue}: aux must be {want} of C's shape and row stride, got {aux.dtype} "
                             f"{tuple(aux.shape)} stride {aux.stride()} for C {tuple(C_out.shape)} stride {C_out.stride()}")
    _lib.check(lib.sgclip_gemm_bf16(A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), C_out.data_ptr(),
                                    C_out.stride(0), M, N, K, epilogue, _ptr(bias), _ptr(aux), aux_rows, _stream(A)))
    return C_out


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float, out_dtype=_lib.DTYPE_BF16,
              row_idx: Optional[torch.Tensor] = None, rows: Optional[int] = None, x_stride: Optional[int] = None):
    lib = _lib.load()
    d = x.shape[-1]
    rows = rows if rows is not None else (row_idx.numel() if row_idx is not None else x.shape[0])
    y = torch.empty((rows, d), dtype=torch.bfloat16 if out_dtype == _lib.DTYPE_BF16 else torch.float32, device=x.device)
    _lib.check(lib.sgclip_layernorm(x.data_ptr(), x_stride if x_stride is not None else x.stride(0), _ptr(row_idx), rows, d,
                                    w.data_ptr(), b.data_ptr(), eps, out_dtype, y.data_ptr(), _stream()))
    return y


def qkv_attention(x: torch.Tensor, w_qkv: torch.Tensor, bias: torch.Tensor, n_seq: int, T: int, heads: int):
    """Fused q/k/v projection + attention: x bf16 [n_seq*T, d], w_qkv bf16 [3d, d], bias f32 [3d] -> bf16 [n_seq*T, d]."""
    lib = _lib.load()
    out = torch.empty((n_seq * T, heads * 64), dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.sgclip_qkv_attention(x.data_ptr(), x.stride(0), w_qkv.data_ptr(), w_qkv.stride(0), bias.data_ptr(),
                                        out.data_ptr(), n_seq, T, heads, _stream()))
    return out


def attention(qkv: torch.Tensor, n_seq: int, T: int, heads: int, causal: bool, key_mask: Optional[torch.Tensor] = None):
    lib = _lib.load()
    out = torch.empty((n_seq * T, heads * 64), dtype=torch.bfloat16, device=qkv.device)
    _lib.check(lib.sgclip_attention(qkv.data_ptr(), out.data_ptr(), n_seq, T, heads, 1 if causal else 0, _ptr(key_mask),
                                    _stream()))
    return out


def attention_pooled(qkv: torch.Tensor, q_rows: torch.Tensor, n_seq: int, T: int, heads: int, causal: bool,
                     key_mask: Optional[torch.Tensor] = None):
    """Attention output of one query row per sequence (q_rows: int32 [n_seq], global row indices): bf16 [n_seq, d]."""
    lib = _lib.load()
    out = torch.empty((n_seq, heads * 64), dtype=torch.bfloat16, device=qkv.device)
    qr = q_rows.to(qkv.device, torch.int32).contiguous()
    _lib.check(lib.sgclip_attention_pooled(qkv.data_ptr(), qr.data_ptr(), out.data_ptr(), n_seq, T, heads,
                                           1 if causal else 0, _ptr(key_mask), _stream()))
    out._sgclip_keepalive = (qr,)
    return out

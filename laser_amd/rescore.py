"""fp32 scoring of a few crops, for the one place where the reference's result is a discrete
function of a floating-point ranking.

In training mode (`batched_binary_predicates` given) the reference keeps, per category name, the
`topk_cate` objects with the highest mean category probability and scores only the pairs among
them (laser/models/llava_clip_model_v3.py:303-314, 347-386).  The bf16 towers reproduce the
probabilities to a few 1e-3, which is enough for every returned value but not for a ranking whose
boundary candidates are closer than that.  `ExactTower` runs one CLIP tower in the reference's own
fp32 arithmetic (transformers models/clip/modeling_clip.py:269-277, 298-350, 613-656, 513-586)
straight from the fp32 master parameters, on the fp32 matrix cores; `PredicateModel` uses it to score
the objects on the boundary again before it ranks.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, grad_ops as G, ops
from .config import ClipConfig

F32 = torch.float32


def linear_exact(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, mode: int = _lib.F32_PLAIN,
                 aux: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, aux_rows: int = 0,
                 out_tokens: int = 0) -> torch.Tensor:
    """epi(x w^T + bias) in fp32 (x [rows, in], w [out, in], both row-major f32)."""
    assert x.dtype == F32 and w.dtype == F32 and x.stride(1) == 1 and w.stride(1) == 1 and x.shape[1] == w.shape[1]
    rows, out_f = x.shape[0], w.shape[0]
    if out is None:
        out = torch.empty((rows, out_f), dtype=F32, device=x.device)
    _lib.check(_lib.load().sgclip_linear_exact_f32(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), ops._ptr(bias), rows,
                                                   out_f, x.shape[1], mode, ops._ptr(aux), aux_rows, out_tokens, out.data_ptr(),
                                                   out.stride(0), ops._stream(x)))
    return out


def attention_exact(qkv: torch.Tensor, n_seq: int, T: int, heads: int, causal: bool,
                    key_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert qkv.dtype == F32 and qkv.is_contiguous() and qkv.shape == (n_seq * T, 3 * heads * 64)
    out = torch.empty((n_seq * T, heads * 64), dtype=F32, device=qkv.device)
    _lib.check(_lib.load().sgclip_attention_f32(qkv.data_ptr(), out.data_ptr(), n_seq, T, heads, 1 if causal else 0,
                                                ops._ptr(key_mask), ops._stream(qkv)))
    return out


def probabilities_exact(text: torch.Tensor, img: torch.Tensor, exp_logit_scale: float, softmax: bool) -> torch.Tensor:
    """clip_sim (llava_clip_model_v3.py:41-47) and the softmax over the names behind it, in fp32: [K, N]."""
    th, ih = G.normalize_rows(text.contiguous()), G.normalize_rows(img.contiguous())
    out = linear_exact(th, ih)
    _lib.check(_lib.load().sgclip_scale_softmax_columns(out.data_ptr(), out.shape[0], out.shape[1], float(exp_logit_scale),
                                                        1 if softmax else 0, ops._stream(out)))
    return out


class ExactTower:
    """One CLIP tower (vision or text) in fp32.  `params`: HF parameter name -> fp32 tensor on the device."""

    def __init__(self, cfg: ClipConfig, kind: str, params: Dict[str, torch.Tensor]):
        self.cfg, self.vision = cfg, kind == "vision"
        self.pre = "vision_model" if self.vision else "text_model"
        self.d = cfg.v_hidden if self.vision else cfg.t_hidden
        self.L = cfg.v_layers if self.vision else cfg.t_layers
        self.heads = cfg.v_heads if self.vision else cfg.t_heads
        self.eps = cfg.ln_eps
        self.params = params
        self._cache: Dict[str, torch.Tensor] = {}
        self._versions = None

    def _p(self, name) -> torch.Tensor:
        return self.params[name].detach()

    def _fused(self):
        """q | k | v weights and biases stacked per layer and the patch matrix padded to the crop kernel's row
        length; rebuilt when a parameter changed."""
        versions = tuple(p._version for p in self.params.values())
        if versions != self._versions:
            c = {}
            for i in range(self.L):
                b = f"{self.pre}.encoder.layers.{i}.self_attn"
                c[f"{i}.w"] = torch.cat([self._p(f"{b}.{n}_proj.weight") for n in "qkv"], dim=0).contiguous()
                c[f"{i}.b"] = torch.cat([self._p(f"{b}.{n}_proj.bias") for n in "qkv"], dim=0).contiguous()
            if self.vision:
                kpad = _lib.load().sgclip_patch_kpad(self.cfg.patch)
                pw = torch.zeros((self.d, kpad), dtype=F32, device=c["0.w"].device)
                pw[:, : 3 * self.cfg.patch ** 2] = self._p("vision_model.embeddings.patch_embedding.weight").reshape(self.d, -1)
                c["patch"] = pw
            self._cache, self._versions = c, versions
        return self._cache

    def _layers(self, x: torch.Tensor, n_seq: int, T: int, causal: bool, km) -> torch.Tensor:
        c = self._fused()
        for i in range(self.L):
            b = f"{self.pre}.encoder.layers.{i}"
            h = ops.layernorm(x, self._p(f"{b}.layer_norm1.weight"), self._p(f"{b}.layer_norm1.bias"), self.eps,
                              out_dtype=_lib.DTYPE_F32)
            qkv = linear_exact(h, c[f"{i}.w"], c[f"{i}.b"])
            a = attention_exact(qkv, n_seq, T, self.heads, causal, km)
            x = linear_exact(a, self._p(f"{b}.self_attn.out_proj.weight"), self._p(f"{b}.self_attn.out_proj.bias"),
                             _lib.F32_RESID, aux=x)
            h = ops.layernorm(x, self._p(f"{b}.layer_norm2.weight"), self._p(f"{b}.layer_norm2.bias"), self.eps,
                              out_dtype=_lib.DTYPE_F32)
            g = linear_exact(h, self._p(f"{b}.mlp.fc1.weight"), self._p(f"{b}.mlp.fc1.bias"), _lib.F32_GELU)
            x = linear_exact(g, self._p(f"{b}.mlp.fc2.weight"), self._p(f"{b}.mlp.fc2.bias"), _lib.F32_RESID, aux=x)
        return x

    def image_features(self, patches: torch.Tensor, n: int) -> torch.Tensor:
        """patches: f32 [n * P, kpad] (the crop kernel's patch layout in fp32) -> f32 [n, proj]."""
        assert self.vision and patches.dtype == F32
        cfg, d = self.cfg, self.d
        T = cfg.v_tokens
        pos = self._p("vision_model.embeddings.position_embedding.weight")
        emb = torch.empty((n * T, d), dtype=F32, device=patches.device)
        linear_exact(patches, self._fused()["patch"], None, _lib.F32_PATCH, aux=pos, out=emb, aux_rows=T - 1, out_tokens=T)
        G.embed_cls(self._p("vision_model.embeddings.class_embedding"), pos, emb, n, T)
        x = ops.layernorm(emb, self._p("vision_model.pre_layrnorm.weight"), self._p("vision_model.pre_layrnorm.bias"),
                          self.eps, out_dtype=_lib.DTYPE_F32)
        x = self._layers(x, n, T, False, None)
        rows = torch.arange(n, dtype=torch.int32, device=x.device) * T
        pooled = ops.layernorm(x, self._p("vision_model.post_layernorm.weight"), self._p("vision_model.post_layernorm.bias"),
                               self.eps, out_dtype=_lib.DTYPE_F32, row_idx=rows)
        return linear_exact(pooled, self._p("visual_projection.weight"))

    def text_features(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor]) -> torch.Tensor:
        """input_ids / attention_mask as the tokenizer returns them ([K, T], host or device) -> f32 [K, proj]."""
        assert not self.vision
        cfg = self.cfg
        ids_h = input_ids.detach().cpu().numpy()
        if cfg.eos_token_id == 2:
            pos = ids_h.argmax(axis=1)
        else:
            hit = ids_h == cfg.eos_token_id
            pos = np.where(hit.any(axis=1), hit.argmax(axis=1), 0)
        T = int(pos.max()) + 1 if len(pos) else 1   # the tower is causal: later positions cannot reach a pooled token
        dev = self._p("text_projection.weight").device
        ids = torch.from_numpy(np.ascontiguousarray(ids_h[:, :T]).astype(np.int32)).to(dev)
        am = None if attention_mask is None else attention_mask[:, :T].to(dev, torch.int32).contiguous()
        x = G.embed_text(self._p("text_model.embeddings.token_embedding.weight"),
                         self._p("text_model.embeddings.position_embedding.weight"), ids)
        x = self._layers(x, ids.shape[0], T, True, am)
        rows = torch.from_numpy((np.arange(len(pos)) * T + pos).astype(np.int32)).to(dev)
        pooled = ops.layernorm(x, self._p("text_model.final_layer_norm.weight"), self._p("text_model.final_layer_norm.bias"),
                               self.eps, out_dtype=_lib.DTYPE_F32, row_idx=rows)
        return linear_exact(pooled, self._p("text_projection.weight"))

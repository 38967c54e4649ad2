"""TEST INFRASTRUCTURE ONLY - CPU oracle, never imported by the product path.

Plain PyTorch fp32 restatement of the two CLIP towers the reference reaches
through `model.get_image_features` / `model.get_text_features`
(reference call sites: laser/models/llava_clip_model_v3.py:52-83; arithmetic
from third-party `transformers==4.46.2` modeling_clip.py, the same at the
5.15.0 installed here: vision `CLIPVisionModel.forward` :613-656 + projection
:751, text `CLIPTextModel.forward` :513-586 + projection :713, encoder layer
:362-383, attention :298-335 with fp32 softmax, MLP with quick_gelu :346-350).

Operates on a plain state dict with HF key names.  Pinned against HF's own
modules in tests/test_oracle_clip.py and through the golden forward fixtures.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from laser_amd.config import ClipConfig


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(1.702 * x)


def _encoder(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str, layers: int, heads: int,
             eps: float, attn_bias: torch.Tensor | None) -> torch.Tensor:
    B, T, d = x.shape
    hd = d // heads
    for i in range(layers):
        p = f"{prefix}.encoder.layers.{i}"
        h = F.layer_norm(x, (d,), sd[f"{p}.layer_norm1.weight"], sd[f"{p}.layer_norm1.bias"], eps)
        q = F.linear(h, sd[f"{p}.self_attn.q_proj.weight"], sd[f"{p}.self_attn.q_proj.bias"])
        k = F.linear(h, sd[f"{p}.self_attn.k_proj.weight"], sd[f"{p}.self_attn.k_proj.bias"])
        v = F.linear(h, sd[f"{p}.self_attn.v_proj.weight"], sd[f"{p}.self_attn.v_proj.bias"])
        q = q.view(B, T, heads, hd).transpose(1, 2)
        k = k.view(B, T, heads, hd).transpose(1, 2)
        v = v.view(B, T, heads, hd).transpose(1, 2)
        s = torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5)
        if attn_bias is not None:
            s = s + attn_bias
        a = torch.softmax(s, dim=-1, dtype=torch.float32)
        o = torch.matmul(a, v).transpose(1, 2).reshape(B, T, d)
        x = x + F.linear(o, sd[f"{p}.self_attn.out_proj.weight"], sd[f"{p}.self_attn.out_proj.bias"])
        h = F.layer_norm(x, (d,), sd[f"{p}.layer_norm2.weight"], sd[f"{p}.layer_norm2.bias"], eps)
        h = quick_gelu(F.linear(h, sd[f"{p}.mlp.fc1.weight"], sd[f"{p}.mlp.fc1.bias"]))
        x = x + F.linear(h, sd[f"{p}.mlp.fc2.weight"], sd[f"{p}.mlp.fc2.bias"])
    return x


@torch.no_grad()
def image_features(sd: Dict[str, torch.Tensor], cfg: ClipConfig, pixel_values: torch.Tensor,
                   batch: int = 64) -> torch.Tensor:
    """f32[N,3,S,S] -> f32[N,E] (un-normalised projected CLS embedding)."""
    outs = []
    d = cfg.v_hidden
    for s in range(0, pixel_values.shape[0], batch):
        pv = pixel_values[s:s + batch].float()
        pe = F.conv2d(pv, sd["vision_model.embeddings.patch_embedding.weight"], stride=cfg.patch)
        pe = pe.flatten(2).transpose(1, 2)
        cls = sd["vision_model.embeddings.class_embedding"].expand(pe.shape[0], 1, -1)
        x = torch.cat([cls, pe], dim=1) + sd["vision_model.embeddings.position_embedding.weight"][None]
        x = F.layer_norm(x, (d,), sd["vision_model.pre_layrnorm.weight"], sd["vision_model.pre_layrnorm.bias"], cfg.ln_eps)
        x = _encoder(x, sd, "vision_model", cfg.v_layers, cfg.v_heads, cfg.ln_eps, None)
        pooled = F.layer_norm(x[:, 0, :], (d,), sd["vision_model.post_layernorm.weight"],
                              sd["vision_model.post_layernorm.bias"], cfg.ln_eps)
        outs.append(F.linear(pooled, sd["visual_projection.weight"]))
    return torch.cat(outs, dim=0)


@torch.no_grad()
def text_features(sd: Dict[str, torch.Tensor], cfg: ClipConfig, input_ids: torch.Tensor,
                  attention_mask: torch.Tensor | None) -> torch.Tensor:
    """i64[K,T] (+ 0/1 mask) -> f32[K,E].  Causal mask plus key-padding mask;
    pooled at the first eos position (argmax-of-ids when eos_token_id == 2)."""
    K, T = input_ids.shape
    d = cfg.t_hidden
    x = sd["text_model.embeddings.token_embedding.weight"][input_ids] + \
        sd["text_model.embeddings.position_embedding.weight"][:T][None]
    neg = torch.finfo(torch.float32).min
    bias = torch.zeros(K, 1, T, T)
    causal = torch.ones(T, T, dtype=torch.bool).tril()
    allowed = causal[None, None]
    if attention_mask is not None:
        allowed = allowed & attention_mask.bool()[:, None, None, :]
    bias = torch.where(allowed, torch.zeros(()), torch.full((), neg)).expand(K, 1, T, T)
    x = _encoder(x, sd, "text_model", cfg.t_layers, cfg.t_heads, cfg.ln_eps, bias)
    x = F.layer_norm(x, (d,), sd["text_model.final_layer_norm.weight"], sd["text_model.final_layer_norm.bias"], cfg.ln_eps)
    if cfg.eos_token_id == 2:
        pos = input_ids.to(torch.int).argmax(dim=-1)
    else:
        pos = (input_ids.to(torch.int) == cfg.eos_token_id).int().argmax(dim=-1)
    pooled = x[torch.arange(K), pos]
    return F.linear(pooled, sd["text_projection.weight"])


def clip_sim(nl_feat: torch.Tensor, img_feat: torch.Tensor, logit_scale: torch.Tensor) -> torch.Tensor:
    """llava_clip_model_v3.py:41-47: logits_per_text [K,N]."""
    img_feat = img_feat / img_feat.norm(p=2, dim=-1, keepdim=True)
    nl_feat = nl_feat / nl_feat.norm(p=2, dim=-1, keepdim=True)
    return torch.matmul(nl_feat, img_feat.t()) * logit_scale.exp()

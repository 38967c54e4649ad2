"""CLIP weights: seeded random initialisation and state-dict plumbing.

Key names are the HuggingFace `CLIPModel.state_dict()` names, which is what the
reference checkpoints carry under the prefixes `clip_cate_model.`,
`clip_unary_model.` and `clip_binary_model.` (reference:
scripts/eval_sgclip.py:257-265, laser/models/llava_clip_model_v3.py:19-21).

`random_state_dict` is deterministic for a given (config, seed) on any machine
with the same torch version (CPU generator), so tests regenerate weights
instead of shipping them.
"""
from __future__ import annotations

from typing import Dict

import torch

from .config import ClipConfig

TOWER_PREFIXES = ("clip_cate_model", "clip_unary_model", "clip_binary_model")


def _layer_keys(prefix: str, i: int):
    base = f"{prefix}.encoder.layers.{i}"
    return base


def random_state_dict(cfg: ClipConfig, seed: int, logit_scale: float | None = None) -> Dict[str, torch.Tensor]:
    """Seeded fp32 weights with the std's of HF's CLIP initialiser
    (transformers/models/clip/modeling_clip.py:404-452).  Unlike HF (zero
    biases, unit LayerNorm) biases and LayerNorm affine terms are perturbed so
    every term of every kernel is exercised by parity tests."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)

    def nrm(*shape, std):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * std

    sd: Dict[str, torch.Tensor] = {}

    def tower(prefix, d, f, L):
        in_std = d ** -0.5 * (2 * L) ** -0.5
        out_std = d ** -0.5
        fc_std = (2 * d) ** -0.5
        for i in range(L):
            b = f"{prefix}.encoder.layers.{i}"
            for nm in ("q_proj", "k_proj", "v_proj"):
                sd[f"{b}.self_attn.{nm}.weight"] = nrm(d, d, std=in_std)
                sd[f"{b}.self_attn.{nm}.bias"] = nrm(d, std=0.02)
            sd[f"{b}.self_attn.out_proj.weight"] = nrm(d, d, std=out_std)
            sd[f"{b}.self_attn.out_proj.bias"] = nrm(d, std=0.02)
            sd[f"{b}.layer_norm1.weight"] = 1.0 + nrm(d, std=0.05)
            sd[f"{b}.layer_norm1.bias"] = nrm(d, std=0.02)
            sd[f"{b}.mlp.fc1.weight"] = nrm(f, d, std=fc_std)
            sd[f"{b}.mlp.fc1.bias"] = nrm(f, std=0.02)
            sd[f"{b}.mlp.fc2.weight"] = nrm(d, f, std=in_std)
            sd[f"{b}.mlp.fc2.bias"] = nrm(d, std=0.02)
            sd[f"{b}.layer_norm2.weight"] = 1.0 + nrm(d, std=0.05)
            sd[f"{b}.layer_norm2.bias"] = nrm(d, std=0.02)

    d, dt = cfg.v_hidden, cfg.t_hidden
    sd["logit_scale"] = torch.tensor(cfg.logit_scale_init if logit_scale is None else logit_scale, dtype=torch.float32)
    # text tower
    sd["text_model.embeddings.token_embedding.weight"] = nrm(cfg.vocab, dt, std=0.02)
    sd["text_model.embeddings.position_embedding.weight"] = nrm(cfg.max_pos, dt, std=0.02)
    tower("text_model", dt, cfg.t_ffn, cfg.t_layers)
    sd["text_model.final_layer_norm.weight"] = 1.0 + nrm(dt, std=0.05)
    sd["text_model.final_layer_norm.bias"] = nrm(dt, std=0.02)
    # vision tower
    sd["vision_model.embeddings.class_embedding"] = nrm(d, std=d ** -0.5)
    sd["vision_model.embeddings.patch_embedding.weight"] = nrm(d, 3, cfg.patch, cfg.patch, std=0.02)
    sd["vision_model.embeddings.position_embedding.weight"] = nrm(cfg.v_tokens, d, std=0.02)
    sd["vision_model.pre_layrnorm.weight"] = 1.0 + nrm(d, std=0.05)
    sd["vision_model.pre_layrnorm.bias"] = nrm(d, std=0.02)
    tower("vision_model", d, cfg.v_ffn, cfg.v_layers)
    sd["vision_model.post_layernorm.weight"] = 1.0 + nrm(d, std=0.05)
    sd["vision_model.post_layernorm.bias"] = nrm(d, std=0.02)
    sd["visual_projection.weight"] = nrm(cfg.proj, d, std=d ** -0.5)
    sd["text_projection.weight"] = nrm(cfg.proj, dt, std=dt ** -0.5)
    return sd


def predicate_model_state_dict(cfg: ClipConfig, seed: int, logit_scale: float | None = None) -> Dict[str, torch.Tensor]:
    """Three independently seeded CLIP models under the reference's prefixes."""
    out: Dict[str, torch.Tensor] = {}
    for k, prefix in enumerate(TOWER_PREFIXES):
        for name, t in random_state_dict(cfg, seed * 1000 + k, logit_scale).items():
            out[f"{prefix}.{name}"] = t
    return out


def split_predicate_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
    """{'clip_cate_model': {...}, ...} from a flat reference-style state dict.
    Non-persistent `position_ids` buffers that older checkpoints carry are dropped."""
    out = {p: {} for p in TOWER_PREFIXES}
    for k, v in sd.items():
        for p in TOWER_PREFIXES:
            if k.startswith(p + "."):
                name = k[len(p) + 1:]
                if name.endswith("position_ids"):
                    break
                out[p][name] = v
                break
    return out


def infer_config(sd: Dict[str, torch.Tensor], base: ClipConfig | None = None) -> ClipConfig:
    """Geometry from the tensor shapes of one CLIP state dict."""
    pe = sd["vision_model.embeddings.patch_embedding.weight"]
    d, _, p, _ = pe.shape
    T = sd["vision_model.embeddings.position_embedding.weight"].shape[0]
    grid = int(round((T - 1) ** 0.5))
    vL = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("vision_model.encoder.layers."))
    tL = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("text_model.encoder.layers."))
    vocab, dt = sd["text_model.embeddings.token_embedding.weight"].shape
    base = base or ClipConfig()
    return ClipConfig(
        v_hidden=d, v_ffn=sd["vision_model.encoder.layers.0.mlp.fc1.weight"].shape[0], v_layers=vL,
        v_heads=d // 64, patch=p, image=grid * p,
        t_hidden=dt, t_ffn=sd["text_model.encoder.layers.0.mlp.fc1.weight"].shape[0], t_layers=tL,
        t_heads=dt // 64, vocab=vocab,
        max_pos=sd["text_model.embeddings.position_embedding.weight"].shape[0],
        eos_token_id=base.eos_token_id if vocab == base.vocab else vocab - 1,
        bos_token_id=base.bos_token_id if vocab == base.vocab else vocab - 2,
        proj=sd["visual_projection.weight"].shape[0], ln_eps=base.ln_eps,
        logit_scale_init=float(sd["logit_scale"]),
    )

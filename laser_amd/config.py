"""Geometry of one CLIP model as used by the SGCLIP scoring path.

Field names follow HuggingFace `CLIPVisionConfig` / `CLIPTextConfig`
(transformers/models/clip/configuration_clip.py) so that reference
checkpoints map one to one.  The reference instantiates three such models
per PredicateModel (reference: laser/models/llava_clip_model_v3.py:19-21).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, asdict
from typing import Mapping, Union


@dataclass(frozen=True)
class ClipConfig:
    # vision tower
    v_hidden: int = 768
    v_ffn: int = 3072
    v_layers: int = 12
    v_heads: int = 12
    patch: int = 32
    image: int = 224
    # text tower
    t_hidden: int = 512
    t_ffn: int = 2048
    t_layers: int = 12
    t_heads: int = 8
    vocab: int = 49408
    max_pos: int = 77
    eos_token_id: int = 49407
    bos_token_id: int = 49406
    # joint space
    proj: int = 512
    ln_eps: float = 1e-5
    logit_scale_init: float = 2.6592

    @property
    def grid(self) -> int:
        return self.image // self.patch

    @property
    def n_patches(self) -> int:
        return self.grid * self.grid

    @property
    def v_tokens(self) -> int:
        return self.n_patches + 1

    def to_dict(self):
        return asdict(self)

    def vit_flops_per_encode(self) -> float:
        """Algorithmic FLOPs of one image encode:
        2*P*3p^2*d + L*[2*T*(4d^2+2df) + 4*T^2*d] + 2*d*E with T = P+1."""
        P, p, d, f, L, E = self.n_patches, self.patch, self.v_hidden, self.v_ffn, self.v_layers, self.proj
        T = P + 1
        return 2.0 * P * 3 * p * p * d + L * (2.0 * T * (4 * d * d + 2 * d * f) + 4.0 * T * T * d) + 2.0 * d * E


# openai/clip-vit-base-patch32 geometry (HF `CLIPConfig()` defaults).
VIT_B32 = ClipConfig()

# openai/clip-vit-base-patch16 geometry (T = 197).
VIT_B16 = ClipConfig(patch=16)

# openai/clip-vit-large-patch14 geometry (224 px, T = 257).
VIT_L14 = ClipConfig(
    v_hidden=1024, v_ffn=4096, v_layers=24, v_heads=16, patch=14, image=224,
    t_hidden=768, t_ffn=3072, t_layers=12, t_heads=12, proj=768,
)

# openai/clip-vit-large-patch14-336 geometry.
VIT_L14_336 = ClipConfig(
    v_hidden=1024, v_ffn=4096, v_layers=24, v_heads=16, patch=14, image=336,
    t_hidden=768, t_ffn=3072, t_layers=12, t_heads=12, proj=768,
)

# Small geometry for fast parity tests (head_dim stays 64, the only head size
# the attention kernel is built for; the joint space keeps the 512 dimensions of
# ViT-B/32 because the logit error of a given feature error scales with 1/sqrt(E)).
TINY = ClipConfig(
    v_hidden=128, v_ffn=256, v_layers=2, v_heads=2, patch=16, image=64,
    t_hidden=128, t_ffn=256, t_layers=2, t_heads=2, vocab=1024, max_pos=77,
    eos_token_id=1023, bos_token_id=1022, proj=512,
)

# Names the reference's `--clip-model-name` flag takes (scripts/eval_sgclip.py:34-95 passes it to
# AutoModel.from_pretrained, laser/models/llava_clip_model_v3.py:15-23) whose geometry is known
# without the hub, plus short aliases.
KNOWN_GEOMETRIES = {
    "openai/clip-vit-base-patch32": VIT_B32,
    "openai/clip-vit-base-patch16": VIT_B16,
    "openai/clip-vit-large-patch14": VIT_L14,
    "openai/clip-vit-large-patch14-336": VIT_L14_336,
    "vit-b-32": VIT_B32,
    "vit-b-16": VIT_B16,
    "vit-l-14": VIT_L14,
    "vit-l-14-336": VIT_L14_336,
    "tiny": TINY,
}

# what the kernels are built for; anything else is refused when the configuration is made, not at the first launch
HEAD_DIM = 64
MAX_HIDDEN = 1024


def validate(cfg: ClipConfig) -> ClipConfig:
    """Raises ValueError for a geometry the HIP towers cannot run."""
    for side, d, h, f in (("vision", cfg.v_hidden, cfg.v_heads, cfg.v_ffn), ("text", cfg.t_hidden, cfg.t_heads, cfg.t_ffn)):
        if h <= 0 or d != h * HEAD_DIM:
            raise ValueError(f"{side} tower: hidden {d} / heads {h} is head width {d / max(h, 1):g}; the attention "
                             f"kernels are built for head width {HEAD_DIM}")
        if d > MAX_HIDDEN:
            raise ValueError(f"{side} tower: hidden {d} exceeds {MAX_HIDDEN}")
        if f % 64 or f <= 0:
            raise ValueError(f"{side} tower: intermediate size {f} must be a positive multiple of 64")
    if cfg.patch <= 0 or cfg.image % cfg.patch:
        raise ValueError(f"image size {cfg.image} is not a multiple of patch size {cfg.patch}")
    if min(cfg.v_layers, cfg.t_layers, cfg.proj, cfg.vocab, cfg.max_pos) <= 0:
        raise ValueError("layers, projection_dim, vocab_size and max_position_embeddings must be positive")
    return cfg


def from_hf_config(hf: Mapping) -> ClipConfig:
    """ClipConfig from the dict form of a HuggingFace CLIP `config.json` (`CLIPConfig.to_dict()`): keys
    `vision_config` / `text_config` / `projection_dim` / `logit_scale_init_value`; missing entries take
    `CLIPVisionConfig` / `CLIPTextConfig`'s defaults (transformers/models/clip/configuration_clip.py)."""
    if hf.get("model_type", "clip") != "clip":
        raise ValueError(f"not a CLIP configuration: model_type {hf.get('model_type')!r}")
    v = dict(hf.get("vision_config") or hf.get("vision_config_dict") or {})
    t = dict(hf.get("text_config") or hf.get("text_config_dict") or {})
    for side, c in (("vision", v), ("text", t)):
        act = c.get("hidden_act", "quick_gelu")
        if act != "quick_gelu":
            raise ValueError(f"{side} tower: hidden_act {act!r}; the MLP epilogues implement quick_gelu only")
    if v.get("num_channels", 3) != 3:
        raise ValueError(f"vision tower: {v.get('num_channels')} input channels, need 3")
    eps_v, eps_t = v.get("layer_norm_eps", 1e-5), t.get("layer_norm_eps", 1e-5)
    if eps_v != eps_t:
        raise ValueError(f"layer_norm_eps differs between the towers ({eps_v} / {eps_t})")
    return validate(ClipConfig(
        v_hidden=v.get("hidden_size", 768), v_ffn=v.get("intermediate_size", 3072), v_layers=v.get("num_hidden_layers", 12),
        v_heads=v.get("num_attention_heads", 12), patch=v.get("patch_size", 32), image=v.get("image_size", 224),
        t_hidden=t.get("hidden_size", 512), t_ffn=t.get("intermediate_size", 2048), t_layers=t.get("num_hidden_layers", 12),
        t_heads=t.get("num_attention_heads", 8), vocab=t.get("vocab_size", 49408), max_pos=t.get("max_position_embeddings", 77),
        eos_token_id=t.get("eos_token_id", 49407), bos_token_id=t.get("bos_token_id", 49406),
        proj=hf.get("projection_dim", 512), ln_eps=float(eps_v), logit_scale_init=float(hf.get("logit_scale_init_value", 2.6592)),
    ))


def config_by_name(name: Union[str, "os.PathLike", Mapping, ClipConfig]) -> ClipConfig:
    """The geometry behind a model name as the reference's constructor takes it
    (laser/models/llava_clip_model_v3.py:11-23): a known hub name or alias, a directory holding an HF
    `config.json` (a local snapshot of any CLIP checkpoint), the path of such a json file, its dict, or a
    ClipConfig.  There is no hub access: an unknown name raises KeyError and says what would be accepted."""
    if isinstance(name, ClipConfig):
        return validate(name)
    if isinstance(name, Mapping):
        return from_hf_config(name)
    key = os.fspath(name)
    if key in KNOWN_GEOMETRIES:
        return KNOWN_GEOMETRIES[key]
    path = os.path.join(key, "config.json") if os.path.isdir(key) else key
    if os.path.isfile(path):
        with open(path) as fh:
            return from_hf_config(json.load(fh))
    raise KeyError(f"unknown CLIP geometry {key!r}: not one of {sorted(KNOWN_GEOMETRIES)} and not a local HF snapshot "
                   f"(a directory with config.json, or the json file itself); the hub cannot be reached from here")

"""Geometry of one CLIP model as used by the SGCLIP scoring path.

Field names follow HuggingFace `CLIPVisionConfig` / `CLIPTextConfig`
(transformers/models/clip/configuration_clip.py) so that reference
checkpoints map one to one.  The reference instantiates three such models
per PredicateModel (reference: laser/models/llava_clip_model_v3.py:19-21).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict


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


def config_by_name(name: str) -> ClipConfig:
    table = {
        "openai/clip-vit-base-patch32": VIT_B32,
        "vit-b-32": VIT_B32,
        "openai/clip-vit-large-patch14-336": VIT_L14_336,
        "vit-l-14-336": VIT_L14_336,
        "tiny": TINY,
    }
    if name not in table:
        raise KeyError(f"unknown CLIP geometry {name!r}; known: {sorted(table)}")
    return table[name]

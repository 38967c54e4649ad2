"""Tokenizer plumbing for the SGCLIP text path.

The reference tokenises keyword lists with the HF CLIP tokenizer
(`max_length=75, truncation=True, padding='max_length'`, reference:
laser/models/llava_clip_model_v3.py:144-148,159-165,175-181).  BPE
tokenisation is host work and stays with the caller's tokenizer object: any
callable with that signature that returns `input_ids` / `attention_mask` works
(an HF `CLIPTokenizer` when its vocabulary files are available).

`HashTokenizer` is a self-contained deterministic stand-in with the same call
signature for machines without vocabulary files (tests, benchmarks).
"""
from __future__ import annotations

import zlib
from typing import Iterable, List

import numpy as np
import torch


class TokenBatch(dict):
    """dict with the two BatchEncoding behaviours the scoring path uses."""

    def to(self, device):
        return TokenBatch({k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in self.items()})


class HashTokenizer:
    """Whitespace words -> crc32-derived ids, `<bos> w1 .. wn <eos> <pad>..`.

    Layout matches CLIP's: bos first, first eos marks the pooled position, pad
    token equals eos by default (as `openai/clip-vit-base-patch32` does), and
    the attention mask is 0 on padding.
    """

    def __init__(self, vocab: int = 49408, bos_token_id: int | None = None, eos_token_id: int | None = None):
        self.vocab = vocab
        self.bos_token_id = vocab - 2 if bos_token_id is None else bos_token_id
        self.eos_token_id = vocab - 1 if eos_token_id is None else eos_token_id
        self.pad_token = "<|endoftext|>"
        self.pad_token_id = self.eos_token_id
        self.unk_token = "<|endoftext|>"
        self.eos_token = "<|endoftext|>"

    def _word_id(self, w: str) -> int:
        return zlib.crc32(w.encode("utf-8")) % (self.vocab - 2)

    def encode(self, text: str, max_length: int) -> List[int]:
        ids = [self.bos_token_id] + [self._word_id(w) for w in text.lower().split()]
        ids = ids[: max_length - 1] + [self.eos_token_id]
        return ids

    def __call__(self, texts: Iterable[str], return_tensors: str = "pt", max_length: int = 75,
                 truncation: bool = True, padding: str = "max_length") -> TokenBatch:
        if isinstance(texts, str):
            texts = [texts]
        texts = list(texts)
        ids = np.full((len(texts), max_length), self.pad_token_id, dtype=np.int64)
        mask = np.zeros((len(texts), max_length), dtype=np.int64)
        for i, t in enumerate(texts):
            e = self.encode(t, max_length)
            ids[i, : len(e)] = e
            mask[i, : len(e)] = 1
        return TokenBatch(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask))

"""Train-step path of the scoring module: forward with saved activations and a
hand-written backward through the same HIP library.

The reference trains by running `PredicateModel.forward()` under autograd
(scripts/train_sgclip.py:682-694), turning the returned probabilities into
Scallop facts, and calling `loss.backward()` / `Adam.step()` over all CLIP
parameters (scripts/train_sgclip.py:232-241, 875-890).  Here the graph between the
fp32 master parameters and the dense probability tensors is made of four
`torch.autograd.Function`s - text tower, image tower, similarity (+ softmax /
sigmoid), per-object mean - whose forward and backward are sequences of C-ABI
calls; torch's autograd only connects them to the loss the caller builds.

Numerics: bf16 MFMA GEMMs with fp32 accumulation for the encoder layers (forward
as in inference; gradients cast to bf16 before each data / weight-gradient
product), fp32 residual stream, layer norms, projection, similarity and all
parameter gradients.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from torch import nn

from . import _lib, grad_ops as G, ops
from .config import ClipConfig

BF16, F32 = torch.bfloat16, torch.float32


class TrainTower:
    """One CLIP tower (vision or text) of one of the three models, in training form.

    `params` maps HF parameter names (as in the model's state dict, e.g.
    `vision_model.encoder.layers.0.mlp.fc1.weight`) to the fp32 master parameters."""

    def __init__(self, cfg: ClipConfig, kind: str, params: Dict[str, nn.Parameter]):
        self.cfg, self.kind = cfg, kind
        self.vision = kind == "vision"
        self.pre = "vision_model" if self.vision else "text_model"
        self.d = cfg.v_hidden if self.vision else cfg.t_hidden
        self.ffn = cfg.v_ffn if self.vision else cfg.t_ffn
        self.L = cfg.v_layers if self.vision else cfg.t_layers
        self.heads = cfg.v_heads if self.vision else cfg.t_heads
        self.eps = cfg.ln_eps
        names: List[str] = []
        if self.vision:
            names += ["vision_model.embeddings.patch_embedding.weight", "vision_model.embeddings.class_embedding",
                      "vision_model.embeddings.position_embedding.weight", "vision_model.pre_layrnorm.weight",
                      "vision_model.pre_layrnorm.bias", "vision_model.post_layernorm.weight",
                      "vision_model.post_layernorm.bias", "visual_projection.weight"]
        else:
            names += ["text_model.embeddings.token_embedding.weight", "text_model.embeddings.position_embedding.weight",
                      "text_model.final_layer_norm.weight", "text_model.final_layer_norm.bias", "text_projection.weight"]
        for i in range(self.L):
            b = f"{self.pre}.encoder.layers.{i}"
            names += [f"{b}.layer_norm1.weight", f"{b}.layer_norm1.bias"]
            names += [f"{b}.self_attn.{n}_proj.weight" for n in "qkv"]      # adjacent: one [3d, d] view when flat
            names += [f"{b}.self_attn.{n}_proj.bias" for n in "qkv"]
            names += [f"{b}.self_attn.out_proj.weight", f"{b}.self_attn.out_proj.bias", f"{b}.layer_norm2.weight",
                      f"{b}.layer_norm2.bias", f"{b}.mlp.fc1.weight", f"{b}.mlp.fc1.bias", f"{b}.mlp.fc2.weight",
                      f"{b}.mlp.fc2.bias"]
        self.names = names
        self.params = {n: params[n] for n in names}
        self.device = next(iter(self.params.values())).device
        self._versions = None
        self._w: Dict[str, torch.Tensor] = {}
        self._zero_bias = torch.zeros(max(3 * self.d, self.ffn), dtype=F32, device=self.device)
        self.flat: Optional["FlatParameters"] = None   # set by FlatParameters.attach
        self.reducer: Optional["GradientReducer"] = None   # set by GradientReducer: told when this tower's backward is done
        self.bucket_key: Optional[str] = None
        self.backward_pending = 0                      # forward passes of this step whose backward has not run yet
        self.flat_prefix = ""
        self.recompute_chunk: Optional[int] = None      # images per recomputed chunk (None: keep all activations)
        self.keep_activations = True    # chunked: keep the activations of as many chunks as free memory allows
        self.kept_budget: Optional["KeptBudget"] = None   # shared by the towers of a model: bytes kept chunks may hold

    def param_list(self) -> List[nn.Parameter]:
        return [self.params[n] for n in self.names]

    # -- bf16 working copies of the GEMM weights and their transposes -----------------------
    def refresh(self):
        versions = tuple(p._version for p in self.params.values())
        if versions == self._versions:
            return
        P = {n: p.detach() for n, p in self.params.items()}
        w: Dict[str, torch.Tensor] = {}
        if self.vision:
            kpad = _lib.load().sgclip_patch_kpad(self.cfg.patch)
            pw = torch.zeros((self.d, kpad), dtype=F32, device=self.device)
            pw[:, : 3 * self.cfg.patch ** 2] = P["vision_model.embeddings.patch_embedding.weight"].reshape(self.d, -1)
            w["patch"] = G.cast_bf16(pw)
        for i in range(self.L):
            b = f"{self.pre}.encoder.layers.{i}"
            qkv = torch.cat([P[f"{b}.self_attn.{n}_proj.weight"] for n in "qkv"], dim=0).contiguous()
            w[f"{i}.qkv"] = G.cast_bf16(qkv)
            w[f"{i}.qkv_b"] = torch.cat([P[f"{b}.self_attn.{n}_proj.bias"] for n in "qkv"], dim=0).contiguous()
            w[f"{i}.o"] = G.cast_bf16(P[f"{b}.self_attn.out_proj.weight"].contiguous())
            w[f"{i}.fc1"] = G.cast_bf16(P[f"{b}.mlp.fc1.weight"].contiguous())
            w[f"{i}.fc2"] = G.cast_bf16(P[f"{b}.mlp.fc2.weight"].contiguous())
            for k in ("qkv", "o", "fc1", "fc2"):
                w[f"{i}.{k}_t"] = G.transpose_bf16(w[f"{i}.{k}"])
        self._w, self._versions = w, versions

    def _p(self, name):
        return self.params[name].detach()

    # -- encoder layers -----------------------------------------------------------------------
    def _layer_forward(self, i, x, n_seq, T, causal, km, save_extras: bool = False):
        b, w, d = f"{self.pre}.encoder.layers.{i}", self._w, self.d
        M = x.shape[0]
        h1 = ops.layernorm(x, self._p(f"{b}.layer_norm1.weight"), self._p(f"{b}.layer_norm1.bias"), self.eps)
        qkv = torch.empty((M, 3 * d), dtype=BF16, device=x.device)
        ops.gemm_bf16(h1, w[f"{i}.qkv"], qkv, _lib.EPI_BIAS_BF16, bias=w[f"{i}.qkv_b"])
        a = ops.attention(qkv, n_seq, T, self.heads, causal, km)
        # residual adds read the old stream from `aux` and write the new one: no copy of x
        x_mid = torch.empty_like(x)
        ops.gemm_bf16(a, w[f"{i}.o"], x_mid, _lib.EPI_BIAS_RESID_F32, bias=self._p(f"{b}.self_attn.out_proj.bias"), aux=x)
        h2 = ops.layernorm(x_mid, self._p(f"{b}.layer_norm2.weight"), self._p(f"{b}.layer_norm2.bias"), self.eps)
        # fc1 leaves both the pre-activation u (kept for backward) and quick_gelu(u) (fc2's input) in one pass
        u = torch.empty((M, self.ffn), dtype=BF16, device=x.device)
        g = torch.empty((M, self.ffn), dtype=BF16, device=x.device)
        ops.gemm_bf16(h2, w[f"{i}.fc1"], u, _lib.EPI_BIAS_GELU_BOTH_BF16, bias=self._p(f"{b}.mlp.fc1.bias"), aux=g)
        x_out = torch.empty_like(x)
        ops.gemm_bf16(g, w[f"{i}.fc2"], x_out, _lib.EPI_BIAS_RESID_F32, bias=self._p(f"{b}.mlp.fc2.bias"), aux=x_mid)
        # save_extras: also keep what one elementwise pass rebuilds (LayerNorm outputs, the activation).  A recomputed
        # chunk's saved set lives for that chunk's backward only, so it keeps them; otherwise backward rebuilds them
        extra = (h1, h2, g) if save_extras else None
        return x_out, (x, qkv, a, x_mid, u, extra)

    def _layer_backward(self, i, saved, dx, n_seq, T, causal, km, sink: "GradSink"):
        """dx: f32 [M, d] gradient of the layer's output, overwritten with that of its input.
        Parameter gradients are ADDED to the sink's tensors."""
        b, w, d = f"{self.pre}.encoder.layers.{i}", self._w, self.d
        x, qkv, a, x_mid, u, extra = saved
        zb = self._zero_bias
        h1, h2, g = extra if extra is not None else (None, None, None)
        # MLP
        dxb = G.cast_bf16_colsum(dx, sink.get(f"{b}.mlp.fc2.bias"))   # the cast's pass over dx is the bias gradient too
        if g is None:
            g = G.quick_gelu(u)
        G.linear_wgrad(dxb, g, sink.get(f"{b}.mlp.fc2.weight"))
        # du = (dx W2) * quick_gelu'(u): the activation's adjoint and fc1's bias gradient ride in the data-gradient
        # GEMM's epilogue
        du = G.fc2_dgrad_through_gelu(dxb, w[f"{i}.fc2_t"], u, zb, sink.get(f"{b}.mlp.fc1.bias"))
        del g
        if h2 is None:
            h2 = ops.layernorm(x_mid, self._p(f"{b}.layer_norm2.weight"), self._p(f"{b}.layer_norm2.bias"), self.eps)
        G.linear_wgrad(du, h2, sink.get(f"{b}.mlp.fc1.weight"))
        dh2 = G.linear_dgrad(du, w[f"{i}.fc1_t"], zb)
        G.layernorm_bwd(x_mid, self._p(f"{b}.layer_norm2.weight"), self.eps, dh2, sink.get(f"{b}.layer_norm2.weight"),
                        sink.get(f"{b}.layer_norm2.bias"), dx=dx, accumulate=True)
        del du, h2, dh2
        # attention
        dxb = G.cast_bf16_colsum(dx, sink.get(f"{b}.self_attn.out_proj.bias"))
        G.linear_wgrad(dxb, a, sink.get(f"{b}.self_attn.out_proj.weight"))
        da = G.linear_dgrad(dxb, w[f"{i}.o_t"], zb)
        wq, bq = sink.qkv(b, d)
        dqkv = G.attention_bwd(qkv, da, n_seq, T, self.heads, causal, km, dbias=bq)   # the bias gradient rides along
        if h1 is None:
            h1 = ops.layernorm(x, self._p(f"{b}.layer_norm1.weight"), self._p(f"{b}.layer_norm1.bias"), self.eps)
        G.linear_wgrad(dqkv, h1, wq)
        dh1 = G.linear_dgrad(dqkv, w[f"{i}.qkv_t"], zb)
        G.layernorm_bwd(x, self._p(f"{b}.layer_norm1.weight"), self.eps, dh1, sink.get(f"{b}.layer_norm1.weight"),
                        sink.get(f"{b}.layer_norm1.bias"), dx=dx, accumulate=True)

    # -- whole towers -------------------------------------------------------------------------
    def forward_images(self, patches: torch.Tensor, n: int, save_extras: bool = False):
        """patches: bf16 [n * P, kpad] from the crop kernel.  Returns (features f32 [n, proj], saved)."""
        assert self.vision
        self.refresh()
        cfg, d = self.cfg, self.d
        T = cfg.v_tokens
        emb = torch.empty((n * T, d), dtype=F32, device=self.device)
        pos = self._p("vision_model.embeddings.position_embedding.weight")
        ops.gemm_bf16(patches, self._w["patch"], emb, _lib.EPI_PATCH_F32, aux=pos, aux_rows=T - 1)
        G.embed_cls(self._p("vision_model.embeddings.class_embedding"), pos, emb, n, T)
        x = ops.layernorm(emb, self._p("vision_model.pre_layrnorm.weight"), self._p("vision_model.pre_layrnorm.bias"),
                          self.eps, out_dtype=_lib.DTYPE_F32)
        layers = []
        for i in range(self.L):
            x, s = self._layer_forward(i, x, n, T, False, None, save_extras)
            layers.append(s)
        rows = torch.from_numpy(np.arange(n, dtype=np.int32) * T).to(self.device)
        xp = x.index_select(0, rows.long())
        pl = ops.layernorm(xp, self._p("vision_model.post_layernorm.weight"), self._p("vision_model.post_layernorm.bias"),
                           self.eps, out_dtype=_lib.DTYPE_F32)
        feats = self._project(pl, self._p("visual_projection.weight"))
        return feats, dict(patches=patches, emb=emb, layers=layers, rows=rows, xp=xp, pl=pl, n=n, T=T, km=None)

    def forward_text(self, ids: torch.Tensor, attention_mask: Optional[torch.Tensor], rows: torch.Tensor,
                     save_extras: bool = False):
        """ids int32 [K, T] and attention_mask int32 [K, T] (or None) on the device; rows int32 [K]: k * T + position
        of the end-of-text token of row k (modeling_clip.py:561-581)."""
        assert not self.vision
        self.refresh()
        K, T = ids.shape
        x = G.embed_text(self._p("text_model.embeddings.token_embedding.weight"),
                         self._p("text_model.embeddings.position_embedding.weight"), ids)
        layers = []
        for i in range(self.L):
            x, s = self._layer_forward(i, x, K, T, True, attention_mask, save_extras)
            layers.append(s)
        xp = x.index_select(0, rows.long())
        pl = ops.layernorm(xp, self._p("text_model.final_layer_norm.weight"), self._p("text_model.final_layer_norm.bias"),
                           self.eps, out_dtype=_lib.DTYPE_F32)
        feats = self._project(pl, self._p("text_projection.weight"))
        return feats, dict(ids=ids, layers=layers, rows=rows, xp=xp, pl=pl, n=K, T=T, km=attention_mask)

    def _project(self, pl, wproj):
        out = torch.empty((pl.shape[0], wproj.shape[0]), dtype=F32, device=pl.device)
        _lib.check(_lib.load().sgclip_linear_f32(pl.data_ptr(), wproj.data_ptr(), pl.shape[0], wproj.shape[0], wproj.shape[1],
                                                 out.data_ptr(), ops._stream()))
        return out

    def backward(self, saved, dfeat: torch.Tensor, sink: "GradSink"):
        """Adds the gradient of every parameter of the tower, given d(loss)/d(features) [n, proj], to `sink`."""
        d, n, T = self.d, saved["n"], saved["T"]
        dev = self.device
        proj_name = "visual_projection.weight" if self.vision else "text_projection.weight"
        ln_name = "vision_model.post_layernorm" if self.vision else "text_model.final_layer_norm"
        dfeat = dfeat.to(F32).contiguous()
        G.matmul_f32(dfeat, saved["pl"], transA=True, out=sink.get(proj_name), accumulate=True)
        dpl = G.matmul_f32(dfeat, self._p(proj_name))
        dxp = G.layernorm_bwd(saved["xp"], self._p(f"{ln_name}.weight"), self.eps, dpl, sink.get(f"{ln_name}.weight"),
                              sink.get(f"{ln_name}.bias"))
        dx = torch.zeros((n * T, d), dtype=F32, device=dev)
        G.rows_scatter_add(dxp, saved["rows"], dx)
        for i in reversed(range(self.L)):
            self._layer_backward(i, saved["layers"][i], dx, n, T, not self.vision, saved["km"], sink)
            saved["layers"][i] = None
        if self.vision:
            demb = G.layernorm_bwd(saved["emb"], self._p("vision_model.pre_layrnorm.weight"), self.eps, dx,
                                   sink.get("vision_model.pre_layrnorm.weight"), sink.get("vision_model.pre_layrnorm.bias"))
            flat = demb.view(n, T * d)
            G.colsum(flat, sink.get("vision_model.embeddings.position_embedding.weight").view(-1))
            G.colsum(flat[:, :d], sink.get("vision_model.embeddings.class_embedding"))   # row 0 of every image
            dpatch = G.cast_bf16(demb.view(n, T, d)[:, 1:, :].contiguous().view(n * (T - 1), d))
            gp = sink.get("vision_model.embeddings.patch_embedding.weight")
            k_real = 3 * self.cfg.patch ** 2
            if saved["patches"].shape[1] == k_real:
                G.linear_wgrad(dpatch, saved["patches"], gp.view(d, k_real))
            else:                                   # padded patch rows (ViT-L/14: 588 -> 640)
                gp.view(d, k_real).add_(G.linear_wgrad(dpatch, saved["patches"])[:, :k_real])
        else:
            G.rows_scatter_add(dx, saved["ids"].reshape(-1), sink.get("text_model.embeddings.token_embedding.weight"))
            gpos = sink.get("text_model.embeddings.position_embedding.weight")
            G.colsum(dx.view(n, T * d), gpos[:T].view(-1))


class GradSink:
    """Where a tower's backward adds its parameter gradients: the persistent views of a `FlatParameters`
    gradient buffer, or zero-initialised tensors created on demand and handed back to autograd."""

    def __init__(self, tower: TrainTower):
        self.tower = tower
        self.flat = tower.flat
        self.loose: Dict[str, torch.Tensor] = {}

    def get(self, name: str) -> torch.Tensor:
        if self.flat is not None:
            return self.flat.grad_view(self.tower.flat_prefix + name)
        if name not in self.loose:
            p = self.tower.params[name]
            self.loose[name] = torch.zeros(p.shape, dtype=F32, device=p.device)
        return self.loose[name]

    def qkv(self, layer: str, d: int):
        """[3d, d] and [3d] gradients of the fused q, k, v projection of `layer`."""
        wn = [f"{layer}.self_attn.{n}_proj.weight" for n in "qkv"]
        bn = [f"{layer}.self_attn.{n}_proj.bias" for n in "qkv"]
        if self.flat is not None:
            pre = self.tower.flat_prefix
            return (self.flat.grad_span([pre + n for n in wn]).view(3 * d, d), self.flat.grad_span([pre + n for n in bn]))
        if wn[0] not in self.loose:
            dev = self.tower.device
            w, b = torch.zeros((3 * d, d), dtype=F32, device=dev), torch.zeros(3 * d, dtype=F32, device=dev)
            for j in range(3):
                self.loose[wn[j]], self.loose[bn[j]] = w[j * d:(j + 1) * d], b[j * d:(j + 1) * d]
            self.loose[layer + ".qkv_w"], self.loose[layer + ".qkv_b"] = w, b
        return self.loose[layer + ".qkv_w"], self.loose[layer + ".qkv_b"]

    def autograd_outputs(self):
        """One entry per tower parameter for autograd: None where the gradient already went into the flat buffer."""
        out = []
        for name, p in zip(self.tower.names, self.tower.param_list()):
            g = None if self.flat is not None else self.loose.get(name)
            out.append(g.reshape(p.shape) if (g is not None and p.requires_grad) else None)
        return out


class FlatParameters:
    """All parameters of a module in ONE fp32 buffer and their gradients in a second one.

    Each parameter's `.data` / `.grad` become views of the two buffers (state_dict / load_state_dict keep
    working).  The towers' backward adds weight gradients straight into the gradient buffer (split-K atomics of
    the weight-gradient GEMM land there), the optimizer is one kernel launch over the whole buffer, and the
    gradient all-reduce is one collective per bucket of the same buffer with no packing copies."""

    ALIGN = 4   # elements (16 bytes); q / k / v tensors are multiples of it, so fused views stay gap-free

    def __init__(self, named_params):
        named_params = [(n, p) for n, p in named_params]
        off, self.index = 0, {}
        for n, p in named_params:
            self.index[n] = (off, p.numel(), tuple(p.shape))
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        dev = named_params[0][1].device
        self.total = off
        self.data = torch.zeros(off, dtype=F32, device=dev)
        self.grad = torch.zeros(off, dtype=F32, device=dev)
        self.params = [p for _, p in named_params]
        with torch.no_grad():
            for n, p in named_params:
                o, cnt, shape = self.index[n]
                self.data[o:o + cnt].copy_(p.detach().reshape(-1))
                p.data = self.data[o:o + cnt].view(shape)
                p.grad = self.grad[o:o + cnt].view(shape)

    def grad_view(self, name: str) -> torch.Tensor:
        o, cnt, shape = self.index[name]
        return self.grad[o:o + cnt].view(shape)

    def grad_span(self, names) -> torch.Tensor:
        """1-D view over the gradients of parameters that lie back to back."""
        o0 = self.index[names[0]][0]
        o = o0
        for n in names:
            assert self.index[n][0] == o, f"{n} is not adjacent to its predecessor in the flat buffer"
            o += self.index[n][1]
        return self.grad[o0:o]

    def zero_grad(self):
        self.grad.zero_()
        for n, p in zip(self.index, self.params):    # re-attach views a foreign zero_grad(set_to_none=True) dropped
            if p.grad is None:
                p.grad = self.grad_view(n)

    def mark_updated(self):
        for p in self.params:
            torch.autograd.graph.increment_version(p)


def _headroom(dev) -> int:
    """Bytes a torch allocation on `dev` can still get: the card's free memory plus what torch's allocator holds unused."""
    free, _ = torch.cuda.mem_get_info(dev)
    return free + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)


class _TowerFn(torch.autograd.Function):
    """features = tower(inputs).  The tower's parameters are inputs of the node so that autograd knows the output
    depends on them; their gradients are returned to autograd, or added to the flat gradient buffer directly."""

    @staticmethod
    def forward(ctx, tower: TrainTower, run, chunks, *params):
        """run(lo, hi, save_extras) -> (features, saved) for items lo..hi; chunks: list of (lo, hi).  With more than
        one chunk backward recomputes the forward of the chunks whose activations were not kept (the reference wraps
        its feature calls in torch.utils.checkpoint the same way, llava_clip_model_v3.py:52-83)."""
        ctx.tower, ctx.run, ctx.chunks = tower, run, chunks
        ctx.kept = {}
        tower.backward_pending += 1
        if len(chunks) == 1:
            feats, ctx.saved = run(*chunks[0], bool(tower.recompute_chunk))
            return feats
        # Recompute costs a second forward, a quarter of the step's GEMM work, and is only worth it for the memory it
        # frees: with 288 GB on the card most batches need little of it.  Chunks keep their saved sets (without the
        # cheap extras) while the card's free memory stays above what the backward of a recomputed chunk needs AND the
        # model's kept-bytes budget (all towers together) has room.  keep_activations = False is the bounded-memory mode:
        # one chunk's activations at a time, whatever the card holds.
        ctx.saved = None
        dev, feats, per = tower.device, [], 0
        budget = tower.kept_budget
        for idx, (lo, hi) in enumerate(chunks):
            before = torch.cuda.memory_allocated(dev)
            f, saved = run(lo, hi, False)
            feats.append(f)
            held = torch.cuda.memory_allocated(dev) - before
            per = max(per, held)
            # 2.5 x: a recomputed chunk's backward holds its saved set with the extras (+ 50 %) and its gradients
            if tower.keep_activations and _headroom(dev) > 2.5 * per + (12 << 30) and (budget is None or budget.take(held)):
                ctx.kept[idx] = (saved, held)
            del saved
        return torch.cat(feats, dim=0)

    @staticmethod
    def backward(ctx, dfeat):
        sink = GradSink(ctx.tower)
        if ctx.saved is not None:
            ctx.tower.backward(ctx.saved, dfeat, sink)
            ctx.saved = None
        else:
            budget = ctx.tower.kept_budget
            for idx, (lo, hi) in enumerate(ctx.chunks):
                saved, held = ctx.kept.pop(idx, (None, 0))
                if saved is None:
                    _, saved = ctx.run(lo, hi, True)
                ctx.tower.backward(saved, dfeat[lo:hi], sink)
                del saved
                if budget is not None and held:
                    budget.give_back(held)
        ctx.tower.backward_pending -= 1
        if ctx.tower.reducer is not None and ctx.tower.backward_pending <= 0:
            ctx.tower.reducer.ready(ctx.tower.bucket_key)     # every gradient of this tower lies in the flat buffer
        return (None, None, None) + tuple(sink.autograd_outputs())


class KeptBudget:
    """Bytes of activations that kept (not recomputed) chunks may hold, shared by every tower of a model so that what the
    first towers keep counts when the largest one runs.  limit None: bounded by the card's free memory only."""

    def __init__(self, limit: Optional[int] = None):
        self.limit, self.used = limit, 0

    def take(self, n: int) -> bool:
        if self.limit is not None and self.used + n > self.limit:
            return False
        self.used += n
        return True

    def give_back(self, n: int):
        self.used = max(0, self.used - n)


def encode_images(tower: TrainTower, patches: torch.Tensor, n: int) -> torch.Tensor:
    P = tower.cfg.n_patches
    c = tower.recompute_chunk
    chunks = [(0, n)] if not c or n <= c else [(lo, min(n, lo + c)) for lo in range(0, n, c)]
    return _TowerFn.apply(tower, lambda lo, hi, extras: tower.forward_images(patches[lo * P: hi * P], hi - lo, extras), chunks,
                          *tower.param_list())


def encode_text(tower: TrainTower, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """input_ids / attention_mask: host or device integer tensors [K, T] as the tokenizer returns them.  Only the
    positions up to the longest row's end-of-text token are run (the tower is causal)."""
    cfg = tower.cfg
    ids_h = input_ids.detach().cpu().numpy()
    if cfg.eos_token_id == 2:
        pos = ids_h.argmax(axis=1)
    else:
        hit = ids_h == cfg.eos_token_id
        pos = np.where(hit.any(axis=1), hit.argmax(axis=1), 0)
    T = int(pos.max()) + 1 if len(pos) else 1
    dev = tower.device
    ids = torch.from_numpy(np.ascontiguousarray(ids_h[:, :T]).astype(np.int32)).to(dev)
    am = None
    if attention_mask is not None:
        am = attention_mask[:, :T].to(dev, torch.int32).contiguous()
    rows = torch.from_numpy((np.arange(len(pos)) * T + pos).astype(np.int32)).to(dev)
    return _TowerFn.apply(tower, lambda lo, hi, extras: tower.forward_text(ids, am, rows, extras), [(0, ids.shape[0])],
                          *tower.param_list())


class _SimilarityFn(torch.autograd.Function):
    """out[k][n] = f(exp(logit_scale) <text_k, img_n> / (|text_k| |img_n|)) (llava_clip_model_v3.py:41-47 and the
    softmax(dim=0) / sigmoid that follow it)."""

    @staticmethod
    def forward(ctx, text, img, logit_scale, mode: int):
        text, img = text.contiguous(), img.contiguous()
        s = float(logit_scale.detach().exp())
        out = ops.similarity(text, img, s, mode)
        ctx.save_for_backward(text, img, out)
        ctx.s, ctx.mode = s, mode
        return out

    @staticmethod
    def backward(ctx, dout):
        text, img, out = ctx.saved_tensors
        s = ctx.s
        dl = G.prob_bwd(out, dout.to(F32).contiguous(), ctx.mode)
        th, ih = G.normalize_rows(text), G.normalize_rows(img)
        dth = G.matmul_f32(dl, ih, alpha=s)                  # [K, E]
        dih = G.matmul_f32(dl, th, transA=True, alpha=s)     # [N, E]
        # d/d(logit_scale) of s <th_k, ih_n> is the logit itself: sum_kn dl logit = sum_k <dth_k, th_k>
        dls = G.dot_f32(dth, th)
        return G.normalize_bwd(text, dth), G.normalize_bwd(img, dih), dls, None


def similarity(text: torch.Tensor, img: torch.Tensor, logit_scale: torch.Tensor, mode: int) -> torch.Tensor:
    return _SimilarityFn.apply(text, img, logit_scale, mode)


class _SegmentMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, probs, seg, inv_count, n_groups: int):
        ctx.save_for_backward(seg, inv_count)
        ctx.N = probs.shape[1]
        return ops.segment_mean(probs.contiguous(), seg, n_groups)

    @staticmethod
    def backward(ctx, dmean):
        seg, inv_count = ctx.saved_tensors
        return G.segment_mean_bwd(dmean.to(F32).contiguous(), seg, inv_count, ctx.N), None, None, None


def segment_mean(probs: torch.Tensor, seg: torch.Tensor, n_groups: int, seg_host: Optional[Sequence[int]] = None) -> torch.Tensor:
    """Per-group column means (llava_clip_model_v3.py:296-311) with gradient; seg: int32 [N] on the device
    (seg_host: the same ids on the host, saves reading them back)."""
    ids = np.asarray(seg_host if seg_host is not None else seg.cpu().numpy(), dtype=np.int64)
    counts = np.maximum(np.bincount(ids, minlength=n_groups), 1)
    inv = torch.from_numpy((1.0 / counts).astype(np.float32)).to(probs.device)
    return _SegmentMeanFn.apply(probs, seg, inv, n_groups)


# ---------------------------------------------------------------------------------------------
# optimizer and gradient exchange
# ---------------------------------------------------------------------------------------------
class Adam:
    """torch.optim.Adam's update (scripts/train_sgclip.py:232-241 uses its defaults: betas (0.9, 0.999),
    eps 1e-8, no weight decay) as HIP kernel launches: one per parameter, or ONE over the whole buffer when
    given a `FlatParameters`."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        self.flat = params if isinstance(params, FlatParameters) else None
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        if self.flat is not None:
            self.params = list(self.flat.params)
            self.m, self.v = torch.zeros_like(self.flat.data), torch.zeros_like(self.flat.data)
        else:
            self.params = [p for p in params if p.requires_grad]
            self.m = [torch.zeros_like(p, dtype=F32) for p in self.params]
            self.v = [torch.zeros_like(p, dtype=F32) for p in self.params]

    def zero_grad(self, set_to_none: bool = True):
        if self.flat is not None:
            self.flat.zero_grad()
            return
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        self.step_count += 1
        b1, b2 = self.betas
        if self.flat is not None:
            G.adamw(self.flat.data, self.flat.grad, self.m, self.v, self.step_count, self.lr, b1, b2, self.eps,
                    self.weight_decay)
            self.flat.mark_updated()
            return
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            G.adamw(p.data, g, m, v, self.step_count, self.lr, b1, b2, self.eps, self.weight_decay)
            torch.autograd.graph.increment_version(p)   # the towers rebuild their bf16 copies when this changes


def allreduce_gradients(params, group=None, bucket_bytes: int = 256 << 20):
    """Average `.grad` over the ranks of `group` (RCCL on GPUs): gradients are packed into flat fp32
    buckets of about `bucket_bytes` so that each all-reduce is large enough to run at link bandwidth.
    Parameters without a gradient on this rank contribute zeros (every rank walks the same list)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    if isinstance(params, FlatParameters):
        # the gradients already lie in one buffer: all-reduce it in place, bucket by bucket
        flat, step = params.grad, max(1, bucket_bytes // 4)
        avg = dist.get_backend(group) == "nccl"
        for lo in range(0, flat.numel(), step):
            piece = flat[lo: lo + step]
            dist.all_reduce(piece, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=group)
            if not avg:
                piece.div_(world)
        return
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(F32) for p in bucket])
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
        else:
            dist.all_reduce(flat, group=group)
            flat.div_(world)
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off: off + n].view(p.shape)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
        bucket, size = [], 0

    for p in params:
        if not p.requires_grad:
            continue
        bucket.append(p)
        size += p.numel() * 4
        if size >= bucket_bytes:
            flush()
    flush()


class GradientReducer:
    """Gradient averaging over the ranks that overlaps with backward: the flat gradient buffer is cut into one bucket per
    tower, and a bucket's all-reduce is queued as soon as that tower's backward has finished, while the towers that
    come later in backward are still computing.  The reference gets the same effect from DistributedDataParallel's
    bucketed reducer (scripts/train_sgclip.py:203-213 wraps the model in DDP); here the buckets follow the towers
    because a tower's backward is one node of the autograd graph, and they are views of the flat buffer, so nothing is
    packed or copied.

    Every rank issues the same collectives in the same order whatever its batch looks like: buckets are launched in
    the fixed sequence `order`, a bucket goes out when it is ready AND all buckets before it have gone out, and
    `finish()` launches whatever is left (a tower this rank's batch never used contributes zeros).  `order` starts as
    the reverse of the buffer layout; after the first step rank 0's observed completion order is broadcast and used
    from then on, so that buckets leave in the order backward really finishes them.

    On GPUs (backend nccl = RCCL) the collectives run asynchronously on the process group's own stream, ordered after
    the kernels queued so far on the calling stream; `finish()` makes the calling stream wait for them.  With
    `wire_dtype=torch.bfloat16` a bucket is averaged in bf16 (half the bytes over xGMI, about three significant digits
    per gradient element) and written back to the fp32 buffer.

        reducer = model.gradient_reducer()
        for batch in loader:
            opt.zero_grad(); reducer.begin()
            loss(model(**batch)).backward()
            reducer.finish(); opt.step()
    """

    def __init__(self, flat: FlatParameters, buckets, group=None, max_bucket_bytes: int = 256 << 20, wire_dtype=None,
                 towers: Sequence["TrainTower"] = ()):
        """buckets: list of (key, lo, hi) element ranges of flat.grad that together cover the buffer once.
        towers: the TrainTowers that report to this reducer (`bucket_key` set); they call ready(key) themselves."""
        import torch.distributed as dist
        self.flat, self.group, self.wire_dtype = flat, group, wire_dtype
        self.towers = list(towers)
        for t in self.towers:
            t.reducer = self
        self.buckets = {k: (lo, hi) for k, lo, hi in buckets}
        spans = sorted(self.buckets.values())
        assert spans and spans[0][0] == 0 and spans[-1][1] == flat.total and \
            all(a[1] == b[0] for a, b in zip(spans, spans[1:])), "buckets must tile the flat gradient buffer"
        self.order = [k for k, _, _ in reversed(list(buckets))]
        self.max_elems = max(1, max_bucket_bytes // 4)
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.active else 1
        self.native_avg = self.active and dist.get_backend(group) == "nccl"
        self._order_agreed = False
        self.launch_log: List[str] = []       # bucket keys in the order the last step launched them
        self.begin()

    def begin(self):
        """Start of a step (before forward)."""
        self._ready, self._launched, self._pending = set(), 0, []
        for t in self.towers:
            t.backward_pending = 0     # forward passes without a backward (scoring under no_grad) count for nothing
        self._seen: List[str] = []
        self.launch_log = []

    def ready(self, key: str):
        """The gradients of bucket `key` are final on the calling stream (called by the towers' backward)."""
        if key not in self.buckets or key in self._ready:
            return
        self._ready.add(key)
        self._seen.append(key)
        self._launch_ready()

    def _launch_ready(self):
        while self._launched < len(self.order) and self.order[self._launched] in self._ready:
            self._launch(self.order[self._launched])
            self._launched += 1

    def _launch(self, key: str):
        import torch.distributed as dist
        self.launch_log.append(key)
        if not self.active:
            return
        lo, hi = self.buckets[key]
        op = dist.ReduceOp.AVG if self.native_avg else dist.ReduceOp.SUM
        for a in range(lo, hi, self.max_elems):
            piece = self.flat.grad[a: min(hi, a + self.max_elems)]
            wire = piece.to(self.wire_dtype) if self.wire_dtype is not None and self.wire_dtype != piece.dtype else piece
            work = dist.all_reduce(wire, op=op, group=self.group, async_op=True)
            self._pending.append((work, piece, wire))

    def finish(self):
        """After backward: launch what is left, wait for every bucket, leave the averages in flat.grad."""
        import torch.distributed as dist
        for key in self.order[self._launched:]:
            self._launch(key)
        self._launched = len(self.order)
        for work, piece, wire in self._pending:
            work.wait()               # nccl: the calling stream waits; gloo: the host does
            if wire is not piece:
                piece.copy_(wire)
            if not self.native_avg:
                piece.div_(self.world)
        self._pending = []
        if self.active and not self._order_agreed:
            # rank 0's completion order of this first step (buckets its batch never touched go last, as laid out)
            seen = self._seen + [k for k in self.order if k not in self._seen]
            box = [seen]
            dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                                       group=self.group)
            self.order = list(box[0])
            self._order_agreed = True

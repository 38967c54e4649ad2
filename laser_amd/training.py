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
            for n in "qkv":
                names += [f"{b}.self_attn.{n}_proj.weight", f"{b}.self_attn.{n}_proj.bias"]
            names += [f"{b}.self_attn.out_proj.weight", f"{b}.self_attn.out_proj.bias", f"{b}.layer_norm2.weight",
                      f"{b}.layer_norm2.bias", f"{b}.mlp.fc1.weight", f"{b}.mlp.fc1.bias", f"{b}.mlp.fc2.weight",
                      f"{b}.mlp.fc2.bias"]
        self.names = names
        self.params = {n: params[n] for n in names}
        self.device = next(iter(self.params.values())).device
        self._versions = None
        self._w: Dict[str, torch.Tensor] = {}
        self._zero_bias = torch.zeros(max(3 * self.d, self.ffn), dtype=F32, device=self.device)

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
    def _layer_forward(self, i, x, n_seq, T, causal, km):
        b, w, d = f"{self.pre}.encoder.layers.{i}", self._w, self.d
        M = x.shape[0]
        h1 = ops.layernorm(x, self._p(f"{b}.layer_norm1.weight"), self._p(f"{b}.layer_norm1.bias"), self.eps)
        qkv = torch.empty((M, 3 * d), dtype=BF16, device=x.device)
        ops.gemm_bf16(h1, w[f"{i}.qkv"], qkv, _lib.EPI_BIAS_BF16, bias=w[f"{i}.qkv_b"])
        a = ops.attention(qkv, n_seq, T, self.heads, causal, km)
        x_mid = x.clone()
        ops.gemm_bf16(a, w[f"{i}.o"], x_mid, _lib.EPI_BIAS_RESID_F32, bias=self._p(f"{b}.self_attn.out_proj.bias"))
        h2 = ops.layernorm(x_mid, self._p(f"{b}.layer_norm2.weight"), self._p(f"{b}.layer_norm2.bias"), self.eps)
        u = torch.empty((M, self.ffn), dtype=BF16, device=x.device)
        ops.gemm_bf16(h2, w[f"{i}.fc1"], u, _lib.EPI_BIAS_BF16, bias=self._p(f"{b}.mlp.fc1.bias"))
        g = G.quick_gelu(u)
        x_out = x_mid.clone()
        ops.gemm_bf16(g, w[f"{i}.fc2"], x_out, _lib.EPI_BIAS_RESID_F32, bias=self._p(f"{b}.mlp.fc2.bias"))
        return x_out, (x, qkv, a, x_mid, u)

    def _layer_backward(self, i, saved, dx, n_seq, T, causal, km, grads):
        """dx: f32 [M, d] gradient of the layer's output, overwritten with that of its input."""
        b, w, d = f"{self.pre}.encoder.layers.{i}", self._w, self.d
        x, qkv, a, x_mid, u = saved
        zb = self._zero_bias

        def ln_grads(name):
            dw = torch.zeros(d, dtype=F32, device=dx.device)
            db = torch.zeros(d, dtype=F32, device=dx.device)
            grads[f"{b}.{name}.weight"], grads[f"{b}.{name}.bias"] = dw, db
            return dw, db

        # MLP
        dxb = G.cast_bf16(dx)
        g = G.quick_gelu(u)
        grads[f"{b}.mlp.fc2.weight"] = G.linear_wgrad(dxb, g)
        grads[f"{b}.mlp.fc2.bias"] = G.colsum(dx)
        dg = G.linear_dgrad(dxb, w[f"{i}.fc2_t"], zb)
        du = G.quick_gelu_bwd(u, dg)
        del g, dg
        h2 = ops.layernorm(x_mid, self._p(f"{b}.layer_norm2.weight"), self._p(f"{b}.layer_norm2.bias"), self.eps)
        grads[f"{b}.mlp.fc1.weight"] = G.linear_wgrad(du, h2)
        grads[f"{b}.mlp.fc1.bias"] = G.colsum(du)
        dh2 = G.linear_dgrad(du, w[f"{i}.fc1_t"], zb)
        dw, db = ln_grads("layer_norm2")
        G.layernorm_bwd(x_mid, self._p(f"{b}.layer_norm2.weight"), self.eps, dh2, dw, db, dx=dx, accumulate=True)
        del du, h2, dh2
        # attention
        dxb = G.cast_bf16(dx)
        grads[f"{b}.self_attn.out_proj.weight"] = G.linear_wgrad(dxb, a)
        grads[f"{b}.self_attn.out_proj.bias"] = G.colsum(dx)
        da = G.linear_dgrad(dxb, w[f"{i}.o_t"], zb)
        dqkv = G.attention_bwd(qkv, da, n_seq, T, self.heads, causal, km)
        h1 = ops.layernorm(x, self._p(f"{b}.layer_norm1.weight"), self._p(f"{b}.layer_norm1.bias"), self.eps)
        dwqkv = G.linear_wgrad(dqkv, h1)
        dbqkv = G.colsum(dqkv)
        for j, n in enumerate("qkv"):
            grads[f"{b}.self_attn.{n}_proj.weight"] = dwqkv[j * d:(j + 1) * d]
            grads[f"{b}.self_attn.{n}_proj.bias"] = dbqkv[j * d:(j + 1) * d]
        dh1 = G.linear_dgrad(dqkv, w[f"{i}.qkv_t"], zb)
        dw, db = ln_grads("layer_norm1")
        G.layernorm_bwd(x, self._p(f"{b}.layer_norm1.weight"), self.eps, dh1, dw, db, dx=dx, accumulate=True)

    # -- whole towers -------------------------------------------------------------------------
    def forward_images(self, patches: torch.Tensor, n: int):
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
            x, s = self._layer_forward(i, x, n, T, False, None)
            layers.append(s)
        rows = torch.from_numpy(np.arange(n, dtype=np.int32) * T).to(self.device)
        xp = x.index_select(0, rows.long())
        pl = ops.layernorm(xp, self._p("vision_model.post_layernorm.weight"), self._p("vision_model.post_layernorm.bias"),
                           self.eps, out_dtype=_lib.DTYPE_F32)
        feats = self._project(pl, self._p("visual_projection.weight"))
        return feats, dict(patches=patches, emb=emb, layers=layers, rows=rows, xp=xp, pl=pl, n=n, T=T, km=None)

    def forward_text(self, ids: torch.Tensor, attention_mask: Optional[torch.Tensor], rows: torch.Tensor):
        """ids int32 [K, T] and attention_mask int32 [K, T] (or None) on the device; rows int32 [K]: k * T + position
        of the end-of-text token of row k (modeling_clip.py:561-581)."""
        assert not self.vision
        self.refresh()
        K, T = ids.shape
        x = G.embed_text(self._p("text_model.embeddings.token_embedding.weight"),
                         self._p("text_model.embeddings.position_embedding.weight"), ids)
        layers = []
        for i in range(self.L):
            x, s = self._layer_forward(i, x, K, T, True, attention_mask)
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

    def backward(self, saved, dfeat: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Gradients of every parameter of the tower given d(loss)/d(features) [n, proj]."""
        grads: Dict[str, torch.Tensor] = {}
        d, n, T = self.d, saved["n"], saved["T"]
        dev = self.device
        proj_name = "visual_projection.weight" if self.vision else "text_projection.weight"
        ln_name = "vision_model.post_layernorm" if self.vision else "text_model.final_layer_norm"
        dfeat = dfeat.to(F32).contiguous()
        grads[proj_name] = G.matmul_f32(dfeat, saved["pl"], transA=True)
        dpl = G.matmul_f32(dfeat, self._p(proj_name))
        dw, db = torch.zeros(d, dtype=F32, device=dev), torch.zeros(d, dtype=F32, device=dev)
        grads[f"{ln_name}.weight"], grads[f"{ln_name}.bias"] = dw, db
        dxp = G.layernorm_bwd(saved["xp"], self._p(f"{ln_name}.weight"), self.eps, dpl, dw, db)
        dx = torch.zeros((n * T, d), dtype=F32, device=dev)
        G.rows_scatter_add(dxp, saved["rows"], dx)
        for i in reversed(range(self.L)):
            self._layer_backward(i, saved["layers"][i], dx, n, T, not self.vision, saved["km"], grads)
            saved["layers"][i] = None
        if self.vision:
            dw, db = torch.zeros(d, dtype=F32, device=dev), torch.zeros(d, dtype=F32, device=dev)
            grads["vision_model.pre_layrnorm.weight"], grads["vision_model.pre_layrnorm.bias"] = dw, db
            demb = G.layernorm_bwd(saved["emb"], self._p("vision_model.pre_layrnorm.weight"), self.eps, dx, dw, db)
            dpos = G.colsum(demb.view(n, T * d)).view(T, d)
            grads["vision_model.embeddings.position_embedding.weight"] = dpos
            grads["vision_model.embeddings.class_embedding"] = dpos[0]
            dpatch = G.cast_bf16(demb.view(n, T, d)[:, 1:, :].contiguous().view(n * (T - 1), d))
            dwp = G.linear_wgrad(dpatch, saved["patches"])
            pshape = self.params["vision_model.embeddings.patch_embedding.weight"].shape
            grads["vision_model.embeddings.patch_embedding.weight"] = dwp[:, : 3 * self.cfg.patch ** 2].reshape(pshape)
        else:
            tok = self.params["text_model.embeddings.token_embedding.weight"]
            dtok = torch.zeros(tok.shape, dtype=F32, device=dev)
            G.rows_scatter_add(dx, saved["ids"].reshape(-1), dtok)
            grads["text_model.embeddings.token_embedding.weight"] = dtok
            posw = self.params["text_model.embeddings.position_embedding.weight"]
            dpos = torch.zeros(posw.shape, dtype=F32, device=dev)
            dpos[:T] = G.colsum(dx.view(n, T * d)).view(T, d)
            grads["text_model.embeddings.position_embedding.weight"] = dpos
        return grads


class _TowerFn(torch.autograd.Function):
    """features = tower(inputs); the tower's parameters are inputs of the node so that autograd
    delivers their gradients."""

    @staticmethod
    def forward(ctx, tower: TrainTower, run, *params):
        feats, saved = run()
        ctx.tower, ctx.saved = tower, saved
        return feats

    @staticmethod
    def backward(ctx, dfeat):
        grads = ctx.tower.backward(ctx.saved, dfeat)
        ctx.saved = None
        out = []
        for name, p in zip(ctx.tower.names, ctx.tower.param_list()):
            g = grads.get(name)
            out.append(g.reshape(p.shape) if (g is not None and p.requires_grad) else None)
        return (None, None) + tuple(out)


def encode_images(tower: TrainTower, patches: torch.Tensor, n: int) -> torch.Tensor:
    return _TowerFn.apply(tower, lambda: tower.forward_images(patches, n), *tower.param_list())


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
    return _TowerFn.apply(tower, lambda: tower.forward_text(ids, am, rows), *tower.param_list())


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
        dls = G.matmul_f32(dth.view(1, -1), th.view(1, -1), transB=True).view(())
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
    eps 1e-8, no weight decay) as one HIP kernel launch per parameter."""

    def __init__(self, params: Sequence[nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.m = [torch.zeros_like(p, dtype=F32) for p in self.params]
        self.v = [torch.zeros_like(p, dtype=F32) for p in self.params]

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        self.step_count += 1
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            G.adamw(p.data, g, m, v, self.step_count, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay)
            torch.autograd.graph.increment_version(p)   # the towers rebuild their bf16 copies when this changes


def allreduce_gradients(params: Sequence[nn.Parameter], group=None, bucket_bytes: int = 256 << 20):
    """Average `.grad` over the ranks of `group` (RCCL on GPUs): gradients are packed into flat fp32
    buckets of about `bucket_bytes` so that each all-reduce is large enough to run at link bandwidth.
    Parameters without a gradient on this rank contribute zeros (every rank walks the same list)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
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

"""ViT-B/16 frame + sketch feature extractor on the HIP kernels (reference
lib/modeling/backbone.py:11-62,116-132).

The reference wraps Hugging Face ``ViTModel`` (``google/vit-base-patch16-224-in21k``) and runs it frame by frame in a
Python loop after PIL preprocessing, keeping only the [CLS] state; its path is broken (``device`` is undefined,
backbone.py:30).  Here ``ViTExtractor`` IS the model — same parameter names as ``transformers.ViTModel(config,
add_pooling_layer=False)`` (5.x naming: ``embeddings.patch_embeddings.projection``, ``layers.{i}.attention.q_proj``
..., ``layernorm``; ``load_hf_state_dict`` also accepts the 4.x ``encoder.layer.{i}.attention.attention.query`` style)
— and it takes already-normalised ``pixel_values``, all frames of the batch at once:

    patchify (im2col kernel) -> ONE MFMA GEMM (768 x 768) -> [CLS] + position embeddings ->
    12 x { LN -> q|k|v GEMMs -> short-sequence attention (197 tokens, 12 heads, d_h = 64) -> out-proj GEMM + fp32 residual
           -> LN -> fc1 GEMM + erf-GELU -> fc2 GEMM + fp32 residual } -> LN

bf16 MFMA operands, fp32 residual stream, inference only (the extractor is frozen in the reference: its features
are normally pre-extracted, preprocess/sketch_vit_feature_extractor.py).  ``ViTBackbone.forward`` returns what the head
consumes at BASELINE configs[3]: the sketch's [CLS] state [B,1,768] and ALL 196 patch tokens of every frame
[B, T*196, 768].
LayerNorm uses the kernels' eps = 1e-5 where HF uses 1e-12: a relative change of 5e-6 at unit variance, three orders
below the bf16 operand rounding.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
from torch import nn

from .. import _lib, ops
from ..ops import _DT, _ptr, _stream


def vit_base_config(**over) -> SimpleNamespace:
    a = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, image_size=224,
             patch_size=16, num_channels=3, layer_norm_eps=1e-12)
    a.update(over)
    return SimpleNamespace(**a)


class _Attention(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = (nn.Linear(d, d) for _ in range(4))


class _MLP(nn.Module):
    def __init__(self, d, f):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(d, f), nn.Linear(f, d)


class _Layer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d = cfg.hidden_size
        self.attention = _Attention(d)
        self.layernorm_before = nn.LayerNorm(d, eps=cfg.layer_norm_eps)
        self.layernorm_after = nn.LayerNorm(d, eps=cfg.layer_norm_eps)
        self.mlp = _MLP(d, cfg.intermediate_size)


class _PatchEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.projection = nn.Conv2d(cfg.num_channels, cfg.hidden_size, kernel_size=cfg.patch_size, stride=cfg.patch_size)


class _Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        n_tok = (cfg.image_size // cfg.patch_size) ** 2 + 1
        self.cls_token = nn.Parameter(torch.randn(1, 1, cfg.hidden_size))
        self.position_embeddings = nn.Parameter(torch.randn(1, n_tok, cfg.hidden_size))
        self.patch_embeddings = _PatchEmbeddings(cfg)


class ViTExtractor(nn.Module):
    def __init__(self, cfg=None, compute_dtype='bf16'):
        super().__init__()
        cfg = cfg or vit_base_config()
        if compute_dtype != 'bf16':
            raise NotImplementedError('the short-sequence attention kernel is bf16 (fp32 residual stream)')
        self.cfg = cfg
        self.embeddings = _Embeddings(cfg)
        self.layers = nn.ModuleList([_Layer(cfg) for _ in range(cfg.num_hidden_layers)])
        self.layernorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.requires_grad_(False)
        self._wcache = {}

    def load_hf_state_dict(self, sd):
        """state dict of transformers.ViTModel, 5.x names or the 4.x ``encoder.layer.N.attention.attention.query`` style."""
        ren = (('encoder.layer.', 'layers.'), ('attention.attention.query', 'attention.q_proj'),
               ('attention.attention.key', 'attention.k_proj'), ('attention.attention.value', 'attention.v_proj'),
               ('attention.output.dense', 'attention.o_proj'), ('intermediate.dense', 'mlp.fc1'), ('output.dense', 'mlp.fc2'))
        out = {}
        for k, v in sd.items():
            if k.startswith('vit.'):
                k = k[4:]
            if k.startswith('pooler.'):
                continue
            for a, b in ren:
                k = k.replace(a, b)
            out[k] = v
        self._wcache.clear()
        return self.load_state_dict(out, strict=True)

    def _w(self, src):
        """bf16 [out, in] copy of a frozen fp32 weight (refreshed if the parameter storage or version changed)."""
        ent = self._wcache.get(id(src))
        tag = (src.data_ptr(), src._version)
        if ent is None or ent[0] != tag:
            ent = (tag, ops.cast(src.detach().reshape(src.shape[0], -1).contiguous(), torch.bfloat16))
            self._wcache[id(src)] = ent
        return ent[1]

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor, return_pre_norm: bool = False):
        """[n, C, H, W] fp32 (normalised) -> last_hidden_state [n, 1 + P, d] fp32."""
        if not pixel_values.is_cuda:
            raise RuntimeError('svol_amd ViTExtractor runs on the MI355X HIP kernels only (no CPU path)')
        cfg = self.cfg
        dt = torch.bfloat16
        n, C, Hh, Ww = pixel_values.shape
        p, d, H = cfg.patch_size, cfg.hidden_size, cfg.num_attention_heads
        P = (Hh // p) * (Ww // p)
        L = P + 1
        dev = pixel_values.device
        L_ = _lib.lib()
        pix = pixel_values.float().contiguous()
        patches = torch.empty((n * P, C * p * p), dtype=dt, device=dev)
        _lib.check(L_.svol_patchify(_ptr(pix), _ptr(patches), n, C, Hh, Ww, p, _DT[dt], _stream()), 'svol_patchify')
        emb = self.embeddings
        proj = ops.gemm_nt(patches, self._w(emb.patch_embeddings.projection.weight), emb.patch_embeddings.projection.bias,
                           out_f32=True)
        x32 = torch.empty((n * L, d), dtype=torch.float32, device=dev)
        _lib.check(L_.svol_vit_embed(_ptr(proj), _ptr(emb.cls_token), _ptr(emb.position_embeddings), _ptr(x32), 0, n, P, d,
                                     _DT[dt], _stream()), 'svol_vit_embed')
        del patches, proj
        scale = 1.0 / math.sqrt(d // H)
        for lyr in self.layers:
            a = lyr.attention
            _, y, _, _, _ = ops.layernorm_fwd(x32, lyr.layernorm_before.weight, lyr.layernorm_before.bias, dt)
            qkv = torch.empty((n * L, 3 * d), dtype=dt, device=dev)
            for j, lin in enumerate((a.q_proj, a.k_proj, a.v_proj)):
                ops.gemm_nt(y, self._w(lin.weight), lin.bias, out=qkv[:, j * d:(j + 1) * d])
            o = torch.empty((n * L, d), dtype=dt, device=dev)
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
            _lib.check(L_.svol_attn_small_fwd(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0), _ptr(o),
                                              o.stride(0), n, H, L, d // H, scale, _DT[dt], _stream()), 'svol_attn_small_fwd')
            x32 = ops.gemm_nt(o, self._w(a.o_proj.weight), a.o_proj.bias, residual=x32, out_f32=True)
            _, y, _, _, _ = ops.layernorm_fwd(x32, lyr.layernorm_after.weight, lyr.layernorm_after.bias, dt)
            hmid = ops.gemm_nt(y, self._w(lyr.mlp.fc1.weight), lyr.mlp.fc1.bias, ops.ACT_GELU)
            x32 = ops.gemm_nt(hmid, self._w(lyr.mlp.fc2.weight), lyr.mlp.fc2.bias, residual=x32, out_f32=True)
            del qkv, o, hmid, y
        last, _, _, _, _ = ops.layernorm_fwd(x32, self.layernorm.weight, self.layernorm.bias, dt, want32=True, want_t=False)
        if return_pre_norm:
            return last.view(n, L, d), x32.view(n, L, d)
        return last.view(n, L, d)


class ViTBackbone(nn.Module):
    """backbone.py:11-62 on device: (src_sketch [B,1,C,H,W], src_video [B,T,C,H,W]) -> ([B,1,d], [B,T*P,d])."""

    def __init__(self, video_backbone: ViTExtractor, sketch_backbone: ViTExtractor, use_sketch_cls_token: bool = True,
                 frames_per_launch: int = 512):
        super().__init__()
        self.video_backbone = video_backbone
        self.sketch_backbone = sketch_backbone
        self.use_sketch_cls_token = use_sketch_cls_token
        self.frames_per_launch = frames_per_launch

    @torch.no_grad()
    def forward(self, src_sketch, src_video):
        B, T = src_video.shape[:2]
        sk = self.sketch_backbone(src_sketch.reshape(-1, *src_sketch.shape[2:]))
        sk = sk[:, :1] if self.use_sketch_cls_token else sk[:, 1:].mean(1, keepdim=True)  # backbone.py:35-38
        frames = src_video.reshape(-1, *src_video.shape[2:])
        outs = [self.video_backbone(frames[i:i + self.frames_per_launch])[:, 1:]
                for i in range(0, frames.shape[0], self.frames_per_launch)]
        vd = outs[0] if len(outs) == 1 else torch.cat(outs)
        return sk.reshape(B, -1, sk.shape[-1]).contiguous(), vd.reshape(B, -1, vd.shape[-1])

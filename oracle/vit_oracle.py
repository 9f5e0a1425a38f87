"""ORACLE — test infrastructure only (see oracle/__init__.py).

CPU restatement of the Hugging Face ``ViTModel`` forward that the reference's ViT backbone calls
(lib/modeling/backbone.py:30,48 ``self.*_backbone(pixel_values=..., output_hidden_states=True)``;
preprocess/sketch_vit_feature_extractor.py:50-53 for ``last_hidden_state`` vs ``hidden_states[-1]``).  The
algorithm lives in a third-party dependency that is not in /root/reference: ``transformers`` (unpinned in
requirements.txt:12; 5.15.0 in this image), model ``google/vit-base-patch16-224-in21k``: patch embedding =
Conv2d(3, d, kernel 16, stride 16) flattened row-major over the 14x14 grid, [CLS] token prepended, learned position
embeddings added, ``num_hidden_layers`` pre-norm blocks  x += O(softmax(QK^T / sqrt(d_h)) V) with Q,K,V,O linear
maps of LayerNorm_before(x) (eps 1e-12);  x += fc2(gelu_erf(fc1(LayerNorm_after(x)))),  final LayerNorm.
Pinned against ``transformers.ViTModel`` run in this container (tests/golden/make_golden_vit.py).
"""
import math

import torch
import torch.nn.functional as F


def vit_forward(sd, cfg, pixel_values):
    """-> (last_hidden_state [n, 1+P, d], hidden_states[-1] = the same before the final LayerNorm)."""
    d, h = cfg.hidden_size, cfg.num_attention_heads
    dh = d // h
    eps = cfg.layer_norm_eps
    x = F.conv2d(pixel_values, sd['embeddings.patch_embeddings.projection.weight'],
                 sd['embeddings.patch_embeddings.projection.bias'], stride=cfg.patch_size)
    x = x.flatten(2).transpose(1, 2)                                  # [n, P, d], row-major over the patch grid
    n = x.shape[0]
    x = torch.cat([sd['embeddings.cls_token'].expand(n, -1, -1), x], dim=1) + sd['embeddings.position_embeddings']
    for i in range(cfg.num_hidden_layers):
        p = f'layers.{i}.'
        y = F.layer_norm(x, (d,), sd[p + 'layernorm_before.weight'], sd[p + 'layernorm_before.bias'], eps)
        q = F.linear(y, sd[p + 'attention.q_proj.weight'], sd[p + 'attention.q_proj.bias'])
        k = F.linear(y, sd[p + 'attention.k_proj.weight'], sd[p + 'attention.k_proj.bias'])
        v = F.linear(y, sd[p + 'attention.v_proj.weight'], sd[p + 'attention.v_proj.bias'])
        sp = lambda t: t.view(n, -1, h, dh).transpose(1, 2)           # noqa: E731
        a = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(dh), dim=-1) @ sp(v)
        a = a.transpose(1, 2).reshape(n, -1, d)
        x = x + F.linear(a, sd[p + 'attention.o_proj.weight'], sd[p + 'attention.o_proj.bias'])
        y = F.layer_norm(x, (d,), sd[p + 'layernorm_after.weight'], sd[p + 'layernorm_after.bias'], eps)
        y = F.gelu(F.linear(y, sd[p + 'mlp.fc1.weight'], sd[p + 'mlp.fc1.bias']))
        x = x + F.linear(y, sd[p + 'mlp.fc2.weight'], sd[p + 'mlp.fc2.bias'])
    return F.layer_norm(x, (d,), sd['layernorm.weight'], sd['layernorm.bias'], eps), x


def backbone_features(sd_video, sd_sketch, cfg, src_sketch, src_video):
    """sketch -> its [CLS] state after the final norm (backbone.py:31-37 defaults), video frames ->
    ALL patch tokens after the final norm, frames concatenated: ([B,1,d], [B, T*P, d])."""
    B, T = src_video.shape[:2]
    sk, _ = vit_forward(sd_sketch, cfg, src_sketch.reshape(-1, *src_sketch.shape[2:]))
    vd, _ = vit_forward(sd_video, cfg, src_video.reshape(-1, *src_video.shape[2:]))
    return sk[:, :1].reshape(B, 1, -1), vd[:, 1:].reshape(B, -1, vd.shape[-1])

"""Sine positional encoding (reference lib/modeling/position_encoding.py:35-71,101-129).

Only the 'sine' variant is constructible in the reference ('trainable' passes wrong kwargs and
'learned' is a 2-D DETR leftover), so that is what is built; the others raise the
same ValueError family.
"""
from __future__ import annotations

from torch import nn

from .. import ops


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError('normalize should be True if scale is passed')
        if temperature != 10000 or not normalize or scale is not None:
            raise NotImplementedError('the HIP kernel implements the configuration the reference builds: '
                                      'temperature=10000, normalize=True, scale=2*pi')
        self.num_pos_feats = num_pos_feats

    def forward(self, mask_f32, d, dtype):
        """mask_f32 [B,L] (1 = valid) -> [B,L,d] in `dtype` (no gradient)."""
        return ops.posenc_sine(mask_f32, d, dtype)


def build_position_encoding(args):
    n_steps = args.hidden_dim
    out = []
    for kind in (args.sketch_position_embedding, args.video_position_embedding):
        if kind == 'sine':
            out.append(PositionEmbeddingSine(n_steps, normalize=True))
        elif kind in ('trainable', 'learned'):
            raise ValueError(f'position embedding {kind!r} is not constructible on the svanet path '
                             '(see SURVEY.md §2); use "sine"')
        else:
            raise ValueError(f'not supported {kind}')
    return out[0], out[1]

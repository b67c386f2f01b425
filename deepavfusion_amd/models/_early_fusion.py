"""What the image and the video early-fusion encoders share: the learnable fusion tokens, one optional fusion block per
tower layer, the final fusion norm, and the bookkeeping around them.  The arithmetic lives in ``deepavfusion_amd.engine``;
these classes hold parameters under the reference's attribute names / state-dict keys and forward to the single autograd
node of ``autograd_bridge.encoder_apply``."""
from functools import partial

import torch
from torch import nn

from . import fusion_blocks
from .vits import init_linear_and_norm

_BLOCKS = {
    'token': lambda tkns: fusion_blocks.FusionBlock_LocalAVTokens,
    'dense_mmi': lambda tkns: fusion_blocks.FusionBlock_DenseAVInteractions,
    'factorized_mmi': lambda tkns: partial(fusion_blocks.FusionBlock_FactorizedAVInteractions, fusion_tkns=tkns),
}


def parse_fusion_layers(spec, depth):
    """'all' | 'none' | int | 'a-b-c' -> set of layer indices that get a fusion block."""
    if spec == 'all':
        return set(range(depth))
    if spec == 'none':
        return set()
    if isinstance(spec, int):
        return {spec}
    return {int(tok) for tok in str(spec).split('-')}


class EarlyFusionBase(nn.Module):
    """Subclasses create ``self.audio`` and their visual tower (``self.image`` / ``self.video``) first, then call
    ``_build_fusion``."""

    visual_name = 'image'

    @property
    def visual(self):
        return getattr(self, self.visual_name)

    def _build_fusion(self, arch, layers, tkns, mlp_ratio, attn_ratio, heads, drop_path=0., attn_drop=0., drop=0.):
        self.embed_dim = self.visual.embed_dim
        self.num_fusion = tuple(tkns)
        self.fusion_num_heads = heads
        self.fusion_tokens = nn.Parameter(torch.zeros(1, sum(self.num_fusion), self.embed_dim))
        depth = max(len(self.visual.blocks), len(self.audio.blocks))
        wanted = parse_fusion_layers(layers, depth)
        make = _BLOCKS[arch](self.num_fusion) if arch in _BLOCKS else None      # unknown arch: no fusion blocks, as the reference
        slots = []
        for i in range(depth):
            if make is None or i not in wanted:
                slots.append(None)
            else:
                slots.append(make(dim=self.embed_dim, num_heads=heads, attn_ratio=attn_ratio, mlp_ratio=mlp_ratio, qkv_bias=True,
                                  drop=drop, attn_drop=attn_drop, drop_path=drop_path, norm_layer=nn.LayerNorm))
        self.fusion_blocks = nn.ModuleList(slots)
        self.fusion_norm = nn.LayerNorm(self.embed_dim)
        nn.init.normal_(self.fusion_tokens, std=.02)
        self.fusion_blocks.apply(init_linear_and_norm)

    def params_layer_ids(self):
        """(parameter, layer id) pairs for layer-wise lr decay: towers first, fusion tokens at 0, fusion block i at i + 1,
        the fusion norm last."""
        out = list(self.visual.params_layer_ids()) + list(self.audio.params_layer_ids())
        out.append((self.fusion_tokens, 0))
        for i, blk in enumerate(self.fusion_blocks):
            if blk is not None:
                out += [(p, i + 1) for p in blk.parameters()]
        out += [(p, len(self.fusion_blocks) + 1) for p in self.fusion_norm.parameters()]
        return out

    def _encode(self, visual_input, audio, visual_ids_keep, audio_ids_keep, return_embs):
        from ..autograd_bridge import encoder_apply
        return encoder_apply(self, visual_input, audio, visual_ids_keep, audio_ids_keep, return_embs)

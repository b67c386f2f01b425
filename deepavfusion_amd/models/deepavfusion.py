"""DeepAVFusion early-fusion encoder (reference models/deepavfusion.py), parameters + drop-in API."""
from functools import partial

import torch
from torch import nn

from . import fusion_blocks, vits
from .vits import init_linear_and_norm


class DeepAVFusion(nn.Module):
    """Same ctor signature / attributes / state-dict keys as models/deepavfusion.py:6-54.

    ``fusion_arch`` selects the block as models/deepavfusion.py:28-35 does: 'factorized_mmi' (default, every BASELINE
    configuration), 'token' (FusionBlock_LocalAVTokens) or 'dense_mmi' (FusionBlock_DenseAVInteractions).
    """
    def __init__(self, image_arch='vit_base', image_pretrained=True, image_size=(224, 224),
                 audio_arch='vit_base', audio_pretrained=True, audio_size=(128, 192),
                 fusion_arch='factorized_mmi', fusion_layers='all', num_fusion_tkns=(4, 8, 4),
                 fusion_mlp_ratio=1.0, fusion_attn_ratio=0.25, fusion_num_heads=12, drop_path=0., attn_drop=0., drop=0.):
        super().__init__()
        self.image = vits.__dict__[image_arch](pretrained=image_pretrained, input_size=image_size, in_chans=3,
                                               use_cls_token=False, drop_path=drop_path, attn_drop=attn_drop, drop=drop)
        self.audio = vits.__dict__[audio_arch](pretrained=audio_pretrained, input_size=audio_size, in_chans=1,
                                               use_cls_token=False, drop_path=drop_path, attn_drop=attn_drop, drop=drop)
        self.embed_dim = self.image.embed_dim
        self.fusion_arch = fusion_arch
        self.num_fusion = tuple(num_fusion_tkns)
        self.fusion_num_heads = fusion_num_heads
        self.fusion_tokens = nn.Parameter(torch.zeros(1, sum(num_fusion_tkns), self.embed_dim))
        if fusion_arch == 'token':
            make_block = fusion_blocks.FusionBlock_LocalAVTokens
        elif fusion_arch == 'dense_mmi':
            make_block = fusion_blocks.FusionBlock_DenseAVInteractions
        elif fusion_arch == 'factorized_mmi':
            make_block = partial(fusion_blocks.FusionBlock_FactorizedAVInteractions, fusion_tkns=num_fusion_tkns)
        else:
            make_block = None                            # as in the reference: unknown arch -> no fusion blocks at all
        depth = max(len(self.image.blocks), len(self.audio.blocks))
        if fusion_layers == 'all':                       # models/deepavfusion.py:38-45
            layers = set(range(depth))
        elif fusion_layers == 'none':
            layers = set()
        elif isinstance(fusion_layers, int):
            layers = {fusion_layers}
        else:
            layers = {int(l) for l in str(fusion_layers).split('-')}
        self.fusion_blocks = nn.ModuleList([
            make_block(dim=self.embed_dim, num_heads=fusion_num_heads, attn_ratio=fusion_attn_ratio,
                       mlp_ratio=fusion_mlp_ratio, qkv_bias=True, drop=drop, attn_drop=attn_drop, drop_path=drop_path,
                       norm_layer=nn.LayerNorm) if (i in layers and make_block is not None) else None
            for i in range(depth)])
        self.fusion_norm = nn.LayerNorm(self.embed_dim)
        self.initialize_weights()

    def initialize_weights(self):
        nn.init.normal_(self.fusion_tokens, std=.02)
        self.fusion_blocks.apply(init_linear_and_norm)

    def params_layer_ids(self):
        ids = list(self.image.params_layer_ids()) + list(self.audio.params_layer_ids())
        ids.append((self.fusion_tokens, 0))
        for i, blk in enumerate(self.fusion_blocks):
            if blk is not None:
                ids.extend((p, i + 1) for p in blk.parameters())
        ids.extend((p, len(self.fusion_blocks) + 1) for p in self.fusion_norm.parameters())
        return ids

    def load_checkpoint(self, ckpt_fn, prefix):
        ckpt = torch.load(ckpt_fn, map_location='cpu')['state_dict']
        self.load_state_dict({k[len(prefix):]: v for k, v in ckpt.items() if k.startswith(prefix)}, strict=True)
        print(f"Loaded pre-trained checkpoint: {ckpt_fn}")

    def forward(self, image, audio, image_ids_keep=None, audio_ids_keep=None, return_embs=False):
        """models/deepavfusion.py:88-118 -> (x_image, x_audio, x_fusion[, embs]) in fp32."""
        from ..autograd_bridge import encoder_apply
        return encoder_apply(self, image, audio, image_ids_keep, audio_ids_keep, return_embs)

"""Video early-fusion encoder (reference models/video_earlyfusion.py), parameters + drop-in API.

BASELINE.json configs[4]: ``video_efav_base`` on an 8-frame 224x224 clip + a (128,192) spectrogram — 784 + 32
video rows per sample, which is where the attention kernels stream keys through LDS in chunks.
"""
import torch
from torch import nn

from . import video_vits, vits
from .fusion_blocks import FusionBlock_FactorizedAVInteractions
from .vits import init_linear_and_norm


class VideoEarlyFusion(nn.Module):
    """Same ctor signature / attributes / state-dict keys as models/video_earlyfusion.py:9-57."""
    def __init__(self, video_arch='video_vit_base', video_pretrained='', video_size=(24, 224, 224),
                 audio_arch='audio_vit_base', audio_pretrained='', audio_size=(128, 298),
                 fusion_layers='all', num_fusion_tkns=(8, 16, 16), fusion_mlp_ratio=1., fusion_attn_ratio=.25,
                 fusion_num_heads=12, drop_path=0., attn_drop=0., drop=0.):
        super().__init__()
        self.video = video_vits.__dict__[video_arch](pretrained=video_pretrained, input_size=video_size, in_chans=3,
                                                     use_cls_token=False, drop_path=drop_path, attn_drop=attn_drop, drop=drop)
        self.audio = vits.__dict__[audio_arch](pretrained=audio_pretrained, input_size=audio_size, in_chans=1,
                                               use_cls_token=False, drop_path=drop_path, attn_drop=attn_drop, drop=drop)
        self.embed_dim = self.video.embed_dim
        self.num_fusion = tuple(num_fusion_tkns)
        self.fusion_num_heads = fusion_num_heads
        self.fusion_tokens = nn.Parameter(torch.zeros(1, sum(num_fusion_tkns), self.embed_dim))
        depth = max(len(self.video.blocks), len(self.audio.blocks))
        if fusion_layers == 'all':                       # models/video_earlyfusion.py:41-48
            layers = set(range(depth))
        elif fusion_layers == 'none':
            layers = set()
        elif isinstance(fusion_layers, int):
            layers = {fusion_layers}
        else:
            layers = {int(l) for l in str(fusion_layers).split('-')}
        self.fusion_blocks = nn.ModuleList([
            FusionBlock_FactorizedAVInteractions(dim=self.embed_dim, fusion_tkns=num_fusion_tkns, num_heads=fusion_num_heads,
                                                 attn_ratio=fusion_attn_ratio, mlp_ratio=fusion_mlp_ratio, qkv_bias=True,
                                                 drop=drop, attn_drop=attn_drop, drop_path=drop_path) if i in layers else None
            for i in range(depth)])
        self.fusion_norm = nn.LayerNorm(self.embed_dim)
        self.initialize_weights()

    def initialize_weights(self):
        nn.init.normal_(self.fusion_tokens, std=.02)
        self.fusion_blocks.apply(init_linear_and_norm)

    def params_layer_ids(self):
        ids = list(self.video.params_layer_ids()) + list(self.audio.params_layer_ids())
        ids.append((self.fusion_tokens, 0))
        for i, blk in enumerate(self.fusion_blocks):
            if blk is not None:
                ids.extend((p, i + 1) for p in blk.parameters())
        ids.extend((p, len(self.fusion_blocks) + 1) for p in self.fusion_norm.parameters())
        return ids

    def load_checkpoint(self, ckpt_fn, prefix):
        """models/video_earlyfusion.py:83-93: an image DeepAVFusion checkpoint adapted to the clip tower."""
        ckpt = torch.load(ckpt_fn, map_location='cpu')['state_dict']
        ckpt = {k[len(prefix):]: v for k, v in ckpt.items() if k.startswith(prefix)}
        ckpt = {k.replace('image.', 'video.'): v for k, v in ckpt.items()}
        ckpt['video.pos_embed'] = self.video.state_dict()['pos_embed']
        w = ckpt['video.patch_embed.proj.weight']
        if self.video.patch_embed.proj.weight.ndim > w.ndim:
            ckpt['video.patch_embed.proj.weight'] = w.unsqueeze(2).repeat(1, 1, self.video.patch_size[0], 1, 1)
        self.load_state_dict(ckpt, strict=True)
        print(f"Loaded pre-trained checkpoint: {ckpt_fn}")

    def forward(self, video, audio, video_ids_keep=None, audio_ids_keep=None, return_embs=False):
        """models/video_earlyfusion.py:95-131: video (b c t h w), audio (b c n t) ->
        (x_video, x_audio, x_fusion[, embs]) in fp32."""
        from ..autograd_bridge import encoder_apply
        return encoder_apply(self, video, audio, video_ids_keep, audio_ids_keep, return_embs)


def _efav(video_arch, audio_arch, tkns, heads):
    def make(video_pretrained='', audio_pretrained='', **kwargs):
        assert video_pretrained == ''
        assert audio_pretrained == ''
        return VideoEarlyFusion(video_arch=video_arch, video_pretrained=video_pretrained, audio_arch=audio_arch,
                                audio_pretrained=audio_pretrained, fusion_layers='all', num_fusion_tkns=tkns,
                                fusion_num_heads=heads, **kwargs)
    return make


video_efav_micro = _efav('video_vit_micro', 'vit_micro', (4, 3, 2), 2)         # parity-test shape
video_efav_small = _efav('video_vit_small', 'vit_small', (8, 4, 4), 6)         # models/video_earlyfusion.py:134-141
video_efav_base = _efav('video_vit_base', 'vit_base', (16, 8, 8), 12)          # :144-151
video_efav_large = _efav('video_vit_large', 'vit_large', (32, 12, 12), 16)     # :154-161

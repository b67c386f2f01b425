"""Video early-fusion encoder (reference models/video_earlyfusion.py), parameters + drop-in API.

BASELINE.json configs[4]: ``video_efav_base`` on an 8-frame 224x224 clip + a (128,192) spectrogram — 784 + 32
video rows per sample, which is where the attention kernels stream keys through LDS in chunks.
"""
import torch

from . import video_vits, vits
from ._early_fusion import EarlyFusionBase


class VideoEarlyFusion(EarlyFusionBase):
    """Drop-in for models/video_earlyfusion.py:9-131: ``.video`` (clip tower) + ``.audio`` + the factorised fusion blocks."""

    visual_name = 'video'

    def __init__(self, video_arch='video_vit_base', video_pretrained='', video_size=(24, 224, 224),
                 audio_arch='audio_vit_base', audio_pretrained='', audio_size=(128, 298),
                 fusion_layers='all', num_fusion_tkns=(8, 16, 16), fusion_mlp_ratio=1., fusion_attn_ratio=.25,
                 fusion_num_heads=12, drop_path=0., attn_drop=0., drop=0.):
        super().__init__()
        tower = dict(use_cls_token=False, drop_path=drop_path, attn_drop=attn_drop, drop=drop)
        self.video = getattr(video_vits, video_arch)(pretrained=video_pretrained, input_size=video_size, in_chans=3, **tower)
        self.audio = getattr(vits, audio_arch)(pretrained=audio_pretrained, input_size=audio_size, in_chans=1, **tower)
        self._build_fusion('factorized_mmi', fusion_layers, num_fusion_tkns, fusion_mlp_ratio, fusion_attn_ratio,
                           fusion_num_heads, drop_path=drop_path, attn_drop=attn_drop, drop=drop)

    def load_checkpoint(self, ckpt_fn, prefix):
        """models/video_earlyfusion.py:83-93: an image DeepAVFusion checkpoint adapted to the clip tower (image.* -> video.*,
        own 3-D pos table, 2-D patch kernels repeated over the tubelet depth)."""
        state = torch.load(ckpt_fn, map_location='cpu')['state_dict']
        state = {k[len(prefix):].replace('image.', 'video.'): v for k, v in state.items() if k.startswith(prefix)}
        state['video.pos_embed'] = self.video.state_dict()['pos_embed']
        w = state['video.patch_embed.proj.weight']
        if self.video.patch_embed.proj.weight.ndim > w.ndim:
            state['video.patch_embed.proj.weight'] = w.unsqueeze(2).repeat(1, 1, self.video.patch_size[0], 1, 1)
        self.load_state_dict(state, strict=True)
        print(f"Loaded pre-trained checkpoint: {ckpt_fn}")

    def forward(self, video, audio, video_ids_keep=None, audio_ids_keep=None, return_embs=False):
        """models/video_earlyfusion.py:95-131: video (b c t h w), audio (b c n t) -> (x_video, x_audio, x_fusion[, embs])."""
        return self._encode(video, audio, video_ids_keep, audio_ids_keep, return_embs)


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

"""DeepAVFusion image + audio early-fusion encoder: drop-in for the reference's ``models/deepavfusion.py`` (same constructor
arguments, attributes ``.image .audio .embed_dim .fusion_tokens .fusion_blocks .fusion_norm``, state-dict keys and
``forward`` contract — SURVEY.md section 8(b)); everything shared with the video encoder is in ``_early_fusion.py``."""
import torch

from . import vits
from ._early_fusion import EarlyFusionBase


class DeepAVFusion(EarlyFusionBase):
    """``fusion_arch``: 'factorized_mmi' (default, every BASELINE configuration), 'token' or 'dense_mmi'
    (models/deepavfusion.py:28-35)."""

    visual_name = 'image'

    def __init__(self, image_arch='vit_base', image_pretrained=True, image_size=(224, 224),
                 audio_arch='vit_base', audio_pretrained=True, audio_size=(128, 192),
                 fusion_arch='factorized_mmi', fusion_layers='all', num_fusion_tkns=(4, 8, 4),
                 fusion_mlp_ratio=1.0, fusion_attn_ratio=0.25, fusion_num_heads=12, drop_path=0., attn_drop=0., drop=0.):
        super().__init__()
        tower = dict(use_cls_token=False, drop_path=drop_path, attn_drop=attn_drop, drop=drop)
        self.image = getattr(vits, image_arch)(pretrained=image_pretrained, input_size=image_size, in_chans=3, **tower)
        self.audio = getattr(vits, audio_arch)(pretrained=audio_pretrained, input_size=audio_size, in_chans=1, **tower)
        self.fusion_arch = fusion_arch
        self._build_fusion(fusion_arch, fusion_layers, num_fusion_tkns, fusion_mlp_ratio, fusion_attn_ratio, fusion_num_heads,
                           drop_path=drop_path, attn_drop=attn_drop, drop=drop)

    def load_checkpoint(self, ckpt_fn, prefix):
        """Strict load of the ``prefix``-ed part of a pre-training checkpoint (models/deepavfusion.py:81-86)."""
        state = torch.load(ckpt_fn, map_location='cpu')['state_dict']
        self.load_state_dict({k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}, strict=True)
        print(f"Loaded pre-trained checkpoint: {ckpt_fn}")

    def forward(self, image, audio, image_ids_keep=None, audio_ids_keep=None, return_embs=False):
        """models/deepavfusion.py:88-118 -> (x_image, x_audio, x_fusion[, embs]) in fp32."""
        return self._encode(image, audio, image_ids_keep, audio_ids_keep, return_embs)

"""Build AVMAE(DeepAVFusion(...)) from a shape configuration (deepavfusion_amd.configs.PathConfig:
the reference's configs/deepavfusion.yaml model section)."""
from .models.avmae import AVMAE
from .models.deepavfusion import DeepAVFusion
from .models.video_earlyfusion import VideoEarlyFusion

_ARCH = {(128, 2, 2): 'vit_micro', (192, 12, 3): 'vit_tiny', (384, 12, 6): 'vit_small', (768, 12, 12): 'vit_base',
         (1024, 24, 16): 'vit_large'}


def build_avmae(cfg):
    arch = _ARCH[(cfg.embed_dim, cfg.depth, cfg.num_heads)]
    layers = 'all' if tuple(cfg.fusion_layers) == tuple(range(cfg.depth)) else '-'.join(str(l) for l in cfg.fusion_layers)
    enc = DeepAVFusion(image_arch=arch, image_pretrained='', image_size=tuple(cfg.image_size),
                       audio_arch=arch, audio_pretrained='', audio_size=tuple(cfg.audio_size),
                       fusion_arch=getattr(cfg, 'fusion_arch', 'factorized_mmi'), fusion_layers=layers, num_fusion_tkns=tuple(cfg.fusion_tkns),
                       fusion_mlp_ratio=cfg.fusion_mlp_ratio, fusion_attn_ratio=cfg.fusion_attn_ratio,
                       fusion_num_heads=cfg.fusion_num_heads)
    return AVMAE(enc, enc.embed_dim,
                 image_decoder_arch=getattr(cfg, 'decoder_arch', 'plain'), image_decoder_depth=cfg.decoder_depth, image_mask_ratio=cfg.image_mask_ratio,
                 image_norm_loss=cfg.image_norm_loss,
                 audio_decoder_arch=getattr(cfg, 'decoder_arch', 'plain'), audio_decoder_depth=cfg.decoder_depth, audio_mask_ratio=cfg.audio_mask_ratio,
                 audio_norm_loss=cfg.audio_norm_loss,
                 decoder_dim=cfg.decoder_dim, num_heads=cfg.decoder_heads, mlp_ratio=cfg.decoder_mlp_ratio)


def build_video_earlyfusion(cfg):
    """VideoEarlyFusion from a deepavfusion_amd.configs.VideoConfig (what ``video_efav_*`` of
    models/video_earlyfusion.py:134-171 build, with explicit clip / spectrogram sizes)."""
    arch = _ARCH[(cfg.embed_dim, cfg.depth, cfg.num_heads)]
    layers = 'all' if tuple(cfg.fusion_layers) == tuple(range(cfg.depth)) else '-'.join(str(l) for l in cfg.fusion_layers)
    return VideoEarlyFusion(video_arch='video_' + arch, video_pretrained='', video_size=tuple(cfg.video_size),
                            audio_arch=arch, audio_pretrained='', audio_size=tuple(cfg.audio_size),
                            fusion_layers=layers, num_fusion_tkns=tuple(cfg.fusion_tkns),
                            fusion_mlp_ratio=cfg.fusion_mlp_ratio, fusion_attn_ratio=cfg.fusion_attn_ratio,
                            fusion_num_heads=cfg.fusion_num_heads)

"""Shape configurations of the pre-training path (the ``model`` section of the reference's
configs/deepavfusion.yaml + the ViT factory constants of models/vits.py:121-170)."""
from dataclasses import dataclass, field
from typing import Tuple


@dataclass
class PathConfig:
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: float = 4.0
    patch: int = 16
    image_size: Tuple[int, int] = (224, 224)
    audio_size: Tuple[int, int] = (128, 640)            # (n_mels, 64 * audio_dur): 10 s at 16 kHz (train.py:65)
    fusion_tkns: Tuple[int, int, int] = (16, 8, 8)
    fusion_layers: Tuple[int, ...] = field(default_factory=lambda: tuple(range(12)))
    fusion_mlp_ratio: float = 1.0
    fusion_attn_ratio: float = 0.25
    fusion_num_heads: int = 12
    fusion_arch: str = 'factorized_mmi'                 # models/deepavfusion.py:11, 28-35
    decoder_dim: int = 512
    decoder_depth: int = 8
    decoder_heads: int = 16
    decoder_mlp_ratio: float = 4.0
    decoder_arch: str = 'plain'                         # 'plain' | 'swin' (models/avmae.py:37, 67; configs/deepavfusion.yaml: plain)
    image_mask_ratio: float = 0.75
    audio_mask_ratio: float = 0.8
    image_norm_loss: bool = True
    audio_norm_loss: bool = True

    @property
    def image_grid(self):
        return (self.image_size[0] // self.patch, self.image_size[1] // self.patch)

    @property
    def audio_grid(self):
        return (self.audio_size[0] // self.patch, self.audio_size[1] // self.patch)


@dataclass
class VideoConfig:
    """``video_efav_*`` (models/video_earlyfusion.py:134-171): BASELINE.json configs[4] is ``video_efav_base`` on an
    8-frame 224x224 clip with 3 s of audio ((128, 192) log-mel), batch 16."""
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: float = 4.0
    patch: int = 16
    video_size: Tuple[int, int, int] = (8, 224, 224)
    video_patch: Tuple[int, int, int] = (2, 16, 16)
    audio_size: Tuple[int, int] = (128, 192)
    fusion_tkns: Tuple[int, int, int] = (16, 8, 8)
    fusion_layers: Tuple[int, ...] = field(default_factory=lambda: tuple(range(12)))
    fusion_mlp_ratio: float = 1.0
    fusion_attn_ratio: float = 0.25
    fusion_num_heads: int = 12

    @property
    def video_grid(self):
        return tuple(s // p for s, p in zip(self.video_size, self.video_patch))

    @property
    def audio_grid(self):
        return (self.audio_size[0] // self.patch, self.audio_size[1] // self.patch)


CONFIGS = {
    # parity-only micro shape (head widths 64 / 32 / 16 like the real models)
    'micro': PathConfig(embed_dim=128, depth=2, num_heads=2, image_size=(64, 64), audio_size=(32, 112), fusion_tkns=(4, 3, 2),
                        fusion_layers=(0, 1), fusion_num_heads=2, decoder_dim=64, decoder_depth=2, decoder_heads=2),
    'micro_token': PathConfig(embed_dim=128, depth=2, num_heads=2, image_size=(64, 64), audio_size=(32, 112), fusion_tkns=(4, 3, 2),
                              fusion_layers=(0, 1), fusion_num_heads=2, decoder_dim=64, decoder_depth=2, decoder_heads=2, fusion_arch='token'),
    'micro_dense': PathConfig(embed_dim=128, depth=2, num_heads=2, image_size=(64, 64), audio_size=(32, 112), fusion_tkns=(4, 3, 2),
                              fusion_layers=(0, 1), fusion_num_heads=2, decoder_dim=64, decoder_depth=2, decoder_heads=2, fusion_arch='dense_mmi'),
    # Swin decoders: 8 x 8 / 8 x 12 token grids -> 4 / 6 windows of 16 tokens + 9 fusion tokens
    'micro_swin': PathConfig(embed_dim=128, depth=2, num_heads=2, image_size=(128, 128), audio_size=(128, 192), fusion_tkns=(4, 3, 2),
                             fusion_layers=(0, 1), fusion_num_heads=2, decoder_dim=64, decoder_depth=2, decoder_heads=2, decoder_arch='swin'),
    # BASELINE.json configs[0]: ViT-Tiny, 64x64 image + 2 s audio
    'tiny': PathConfig(embed_dim=192, depth=12, num_heads=3, image_size=(64, 64), audio_size=(128, 128), fusion_num_heads=3),
    # configs[1] (bench workload): ViT-B, README VGGSound recipe (attn_ratio 0.25, mlp_ratio 1.0), 10 s audio
    'base': PathConfig(),
    'base_m75': PathConfig(audio_mask_ratio=0.75),
    # ViT-B with the Swin decoders: 256 x 256 frames (16 x 16 tokens; 14 x 14 is not a multiple of the window) -> 16 + 20 windows
    # of 16 tokens + 32 fusion tokens
    'base_swin': PathConfig(image_size=(256, 256), decoder_arch='swin'),
    'base_token': PathConfig(fusion_arch='token'),
    'base_dense': PathConfig(fusion_arch='dense_mmi'),      # 63 x 49 = 3087 (audio, image) pairs per sample
    # configs[2]: AudioSet-style fusion widths
    'base_as': PathConfig(fusion_mlp_ratio=4.0, fusion_attn_ratio=1.0),
    # configs[3]: ViT-L
    'large': PathConfig(embed_dim=1024, depth=24, num_heads=16, fusion_layers=tuple(range(24)), fusion_num_heads=16),
    # configs[4]: video early fusion (temporal-token stress: 784 + 32 rows per clip)
    'video_micro': VideoConfig(embed_dim=128, depth=2, num_heads=2, video_size=(4, 48, 32), audio_size=(32, 48),
                               fusion_tkns=(4, 3, 2), fusion_layers=(0, 1), fusion_num_heads=2),
    'video_base': VideoConfig(),
}

"""Named path configurations used by fixtures, tests and the CPU baseline.

TEST INFRASTRUCTURE (see oracle/__init__.py).  ``micro`` is a parity-only shape
whose head widths (64 encoder / 32 decoder / 16 fusion q-k) match what the HIP
attention kernels are specialised for; ``tiny`` is BASELINE.json configs[0];
``base`` is configs[1] (the bench workload); ``base_as`` configs[2]; ``large``
configs[3]; ``video_base`` configs[4] (``video_efav_base`` on an 8-frame clip + 3 s audio),
``video_micro`` its parity-only miniature.
"""
from .avmae_oracle import PathConfig, VideoConfig

CONFIGS = {
    'micro': PathConfig(embed_dim=128, depth=2, num_heads=2, image_size=(64, 64), audio_size=(32, 112),
                        fusion_tkns=(4, 3, 2), fusion_layers=(0, 1), fusion_mlp_ratio=1.0, fusion_attn_ratio=0.25,
                        fusion_num_heads=2, decoder_dim=64, decoder_depth=2, decoder_heads=2),
    # the two non-default fusion archs at the micro shape
    'micro_token': PathConfig(embed_dim=128, depth=2, num_heads=2, image_size=(64, 64), audio_size=(32, 112),
                              fusion_tkns=(4, 3, 2), fusion_layers=(0, 1), fusion_num_heads=2, decoder_dim=64, decoder_depth=2,
                              decoder_heads=2, fusion_arch='token'),
    'micro_dense': PathConfig(embed_dim=128, depth=2, num_heads=2, image_size=(64, 64), audio_size=(32, 112),
                              fusion_tkns=(4, 3, 2), fusion_layers=(0, 1), fusion_num_heads=2, decoder_dim=64, decoder_depth=2,
                              decoder_heads=2, fusion_arch='dense_mmi'),
    # Swin decoders: 8 x 8 image grid -> 4 windows, 8 x 12 audio grid -> 6; block 1 is shifted by 2;
    # 16 window tokens + 9 fusion tokens = 25-row attention sequences at head width 32
    'micro_swin': PathConfig(embed_dim=128, depth=2, num_heads=2, image_size=(128, 128), audio_size=(128, 192),
                             fusion_tkns=(4, 3, 2), fusion_layers=(0, 1), fusion_num_heads=2, decoder_dim=64, decoder_depth=2,
                             decoder_heads=2, decoder_arch='swin'),
    'tiny': PathConfig(embed_dim=192, depth=12, num_heads=3, image_size=(64, 64), audio_size=(128, 128),
                       fusion_tkns=(16, 8, 8), fusion_layers=tuple(range(12)), fusion_mlp_ratio=1.0,
                       fusion_attn_ratio=0.25, fusion_num_heads=3),
    'base': PathConfig(),
    'base_m75': PathConfig(audio_mask_ratio=0.75),
    'base_swin': PathConfig(image_size=(256, 256), decoder_arch='swin'),      # 16 x 16 image tokens: a multiple of the 4 x 4 window
    'base_token': PathConfig(fusion_arch='token'),
    'base_dense': PathConfig(fusion_arch='dense_mmi'),      # 63 x 49 = 3087 (audio, image) pairs per sample
    'base_as': PathConfig(fusion_mlp_ratio=4.0, fusion_attn_ratio=1.0),
    'large': PathConfig(embed_dim=1024, depth=24, num_heads=16, fusion_layers=tuple(range(24)), fusion_num_heads=16),
    'video_micro': VideoConfig(embed_dim=128, depth=2, num_heads=2, video_size=(4, 48, 32), audio_size=(32, 48),
                               fusion_tkns=(4, 3, 2), fusion_layers=(0, 1), fusion_num_heads=2),
    'video_base': VideoConfig(),
}

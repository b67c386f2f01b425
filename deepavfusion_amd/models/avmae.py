"""Audio-visual masked auto-encoder (reference models/avmae.py), parameters + drop-in API."""
import types

import torch
from torch import nn

from ..util.pos_embed import get_2d_sincos_pos_embed
from .swin import SwinTransformerBlock
from .vits import Block, init_linear_and_norm


class AVMAE(nn.Module):
    """Same ctor / attribute names / state-dict keys as models/avmae.py:9-90 (decoder_arch 'plain' or 'swin')."""
    def __init__(self, encoder, encoder_dim,
                 image_decoder_arch='plain', image_decoder_depth=8, image_mask_ratio=0.75, image_norm_loss=False,
                 audio_decoder_arch='plain', audio_decoder_depth=8, audio_mask_ratio=0.8, audio_norm_loss=False,
                 decoder_dim=512, num_heads=16, mlp_ratio=4., norm_layer=nn.LayerNorm):
        super().__init__()
        for a in (image_decoder_arch, audio_decoder_arch):
            if a not in ('plain', 'swin'):
                raise ValueError(f"decoder_arch {a!r}: 'plain' | 'swin' (models/avmae.py:37, 67)")
        self.image_mask_ratio, self.image_norm_loss = image_mask_ratio, image_norm_loss
        self.audio_mask_ratio, self.audio_norm_loss = audio_mask_ratio, audio_norm_loss
        self.decoder_dim, self.decoder_heads = decoder_dim, num_heads
        self.image_decoder_arch, self.audio_decoder_arch = image_decoder_arch, audio_decoder_arch
        self.encoder = encoder
        self.image_gs, self.audio_gs = encoder.image.patch_embed.grid_size, encoder.audio.patch_embed.grid_size
        self.image_ps, self.audio_ps = encoder.image.patch_embed.patch_size, encoder.audio.patch_embed.patch_size
        for mod, gs, ps, cin, depth, arch in (('audio', self.audio_gs, self.audio_ps, 1, audio_decoder_depth, audio_decoder_arch),
                                              ('image', self.image_gs, self.image_ps, 3, image_decoder_depth, image_decoder_arch)):
            setattr(self, f'{mod}_decoder_embed', nn.Linear(encoder_dim, decoder_dim, bias=True))
            setattr(self, f'{mod}_decoder_mask_token', nn.Parameter(torch.zeros(1, 1, decoder_dim)))
            # trainable although initialised from the sin-cos table (no requires_grad=False in models/avmae.py:34,64)
            setattr(self, f'{mod}_decoder_pos_embed', nn.Parameter(torch.zeros(1, gs[0] * gs[1], decoder_dim)))
            if arch == 'swin':          # models/avmae.py:37-51, 67-81: window 4, odd blocks shifted by 2
                blocks = [SwinTransformerBlock(dim=decoder_dim, input_resolution=gs, window_size=4, shift_size=(index % 2) * 2,
                                               num_heads=num_heads, mlp_ratio=mlp_ratio, drop=0.0, attn_drop=0.0, drop_path=0.0,
                                               norm_layer=norm_layer) for index in range(depth)]
            else:
                blocks = [Block(decoder_dim, num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer) for _ in range(depth)]
            setattr(self, f'{mod}_decoder_blocks', nn.ModuleList(blocks))
            setattr(self, f'{mod}_decoder_norm', norm_layer(decoder_dim))
            setattr(self, f'{mod}_decoder_pred', nn.Linear(decoder_dim, ps[0] * ps[1] * cin, bias=True))
        self.initialize_weights()

    def initialize_weights(self):
        for mod, gs in (('image', self.image_gs), ('audio', self.audio_gs)):
            pe = get_2d_sincos_pos_embed(self.decoder_dim, gs)
            getattr(self, f'{mod}_decoder_pos_embed').data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
            nn.init.normal_(getattr(self, f'{mod}_decoder_mask_token'), std=.02)
        for n, m in self.named_modules():                 # the encoder keeps its own init (models/avmae.py:105-107)
            if not n.startswith('encoder'):
                init_linear_and_norm(m)

    def decoder(self, modality):
        g = lambda s: getattr(self, f'{modality}_decoder_{s}')
        return types.SimpleNamespace(embed=g('embed'), mask_token=g('mask_token'), pos_embed=g('pos_embed'), blocks=g('blocks'),
                                     norm=g('norm'), pred=g('pred'), heads=self.decoder_heads,
                                     arch=getattr(self, f'{modality}_decoder_arch'))

    def random_masking(self, N, L, mask_ratio, device, noise=None):
        """models/avmae.py:120-142 -> (ids_keep, mask, ids_restore); ``noise`` may be injected for tests."""
        from .. import ops
        if noise is None:
            noise = torch.rand(N, L, device=device)
        len_keep = int(L * (1 - mask_ratio))
        ids_keep, mask, ids_restore, _, _ = ops.mask_build(noise.to(device=device, dtype=torch.float32), len_keep)
        return ids_keep, mask, ids_restore

    def forward_encoder(self, image, audio):
        return self.encoder(image, audio)

    @staticmethod
    def patchify(x, patch_size):
        """models/avmae.py:200-214 (pure data movement; the loss kernel reads NCHW directly instead)."""
        bs, c = x.shape[:2]
        pH, pW = patch_size
        gH, gW = x.shape[2] // pH, x.shape[3] // pW
        return x.reshape(bs, c, gH, pH, gW, pW).permute(0, 2, 4, 3, 5, 1).reshape(bs, gH * gW, pH * pW * c)

    def forward(self, image, audio, noise_image=None, noise_audio=None):
        """models/avmae.py:216-236 -> (loss_image, loss_audio, pred_image, pred_audio)."""
        from ..autograd_bridge import avmae_apply
        return avmae_apply(self, image, audio, noise_image, noise_audio)

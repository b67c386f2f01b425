"""Video ViT tower holding the parameters of the clip modality (reference models/video_vits.py).

Same constructor arguments, attribute names and state-dict keys as the reference's ``VideoViTEncoder``
(``patch_embed.proj`` is a Conv3d over (2,16,16) tubelets, ``pos_embed`` the fixed 3-D sin-cos table,
``blocks.N.*`` timm-style pre-LN blocks, ``norm``); the arithmetic runs on the HIP kernels via
``deepavfusion_amd.engine``.  Only ``attention_type='joint_all'`` (what ``video_efav_*`` builds,
models/video_earlyfusion.py:134-171) is on the MI355X path; the TimeSformer ``divided_space_time``
variants are not.
"""
from functools import partial

import torch
import torch.nn as nn

from ..util.pos_embed import get_3d_sincos_pos_embed
from .vits import Block, init_linear_and_norm


class PatchEmbed3D(nn.Module):
    """Parameter holder of util/pos_embed.py:123-146 (Conv3d, kernel = stride = patch)."""
    def __init__(self, input_size=(16, 224, 224), patch_size=(2, 16, 16), in_chans=3, embed_dim=768, stride=(2, 16, 16)):
        super().__init__()
        self.input_size, self.patch_size, self.in_chans = tuple(input_size), tuple(patch_size), in_chans
        if tuple(stride) != self.patch_size or self.patch_size[1:] != (16, 16):
            raise NotImplementedError('the gfx950 patch kernels are specialised for non-overlapping (pt,16,16) tubelets')
        if any(s % p for s, p in zip(self.input_size, self.patch_size)):
            raise ValueError(f'input_size {self.input_size} is not a multiple of patch_size {self.patch_size}')
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.patch_thw = tuple(s // p for s, p in zip(self.input_size, self.patch_size))
        self.num_patches = self.patch_thw[0] * self.patch_thw[1] * self.patch_thw[2]


class VideoViTEncoder(nn.Module):
    """models/video_vits.py:107-239."""
    def __init__(self, input_size=(16, 224, 224), patch_size=(2, 16, 16), stride=None, in_chans=3, embed_dim=1024, depth=24,
                 num_heads=16, mlp_ratio=4., norm_layer='layer_norm', norm_eps=1e-6, use_cls_token=False, pos_trainable=False,
                 drop_path=0., attn_drop=0., drop=0., attention_type='joint_all'):
        super().__init__()
        if norm_layer != 'layer_norm':
            raise Exception()
        if attention_type != 'joint_all':
            raise NotImplementedError("attention_type='divided_space_time' (TimeSformer) is outside the gfx950 path")
        if use_cls_token:
            raise NotImplementedError('VideoEarlyFusion builds its towers with use_cls_token=False (models/video_earlyfusion.py:32)')
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.input_size, self.patch_size = tuple(input_size), tuple(patch_size)
        self.stride = self.patch_size if stride is None else tuple(stride)
        self.use_cls_token, self.attention_type, self.encoder_depth = use_cls_token, attention_type, depth
        norm = partial(nn.LayerNorm, eps=norm_eps)
        self.patch_embed = PatchEmbed3D(self.input_size, self.patch_size, in_chans, embed_dim, self.stride)
        self.cls_token = None
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim), requires_grad=pos_trainable)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm, drop_path=drop_path,
                                           attn_drop=attn_drop, proj_drop=drop) for _ in range(depth)])
        self.norm = norm(embed_dim)
        self.initialize_weights()

    def initialize_weights(self):
        pe = get_3d_sincos_pos_embed(self.pos_embed.shape[-1], self.patch_embed.patch_thw, cls_token=self.use_cls_token)
        self.pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        nn.init.xavier_uniform_(w.view(w.shape[0], -1))       # conv initialised like a Linear (models/video_vits.py:188-190)
        self.apply(init_linear_and_norm)

    def load_checkpoint(self, ckpt_fn, prefix='', skip_keys_prefix=('decoder', 'mask_token')):
        """models/video_vits.py:165-183: image MAE weights are inflated along time (repeat over the tubelet depth)."""
        ckpt = torch.load(ckpt_fn, map_location='cpu')
        ckpt = ckpt.get('state_dict', ckpt.get('model', ckpt))
        ckpt = {k[len(prefix):]: v for k, v in ckpt.items() if k.startswith(prefix)}
        ckpt = {k: v for k, v in ckpt.items() if not k.startswith(skip_keys_prefix)}
        ckpt.pop('cls_token', None)
        ckpt['pos_embed'] = self.state_dict()['pos_embed']
        w = ckpt['patch_embed.proj.weight']
        if self.patch_embed.proj.weight.ndim > w.ndim:
            ckpt['patch_embed.proj.weight'] = w.unsqueeze(2).repeat(1, 1, self.patch_size[0], 1, 1)
        self.load_state_dict(ckpt, strict=True)

    def params_layer_ids(self):
        ids = [(p, 0) for p in self.patch_embed.parameters()]
        ids.append((self.cls_token, 0))
        for i, blk in enumerate(self.blocks):
            ids.extend((p, i + 1) for p in blk.parameters())
        ids.extend((p, len(self.blocks) + 1) for p in self.norm.parameters())
        return ids

    def prepare_patch_tokens(self, x, ids_keep=None):
        """tubelet embed -> gather -> + pos_embed (models/video_vits.py:218-239, no cls token) -> fp32 [B, n, D]."""
        from ..autograd_bridge import patch_tokens
        return patch_tokens(self, x, ids_keep)


def _factory(embed_dim, depth, num_heads, patch=(2, 16, 16)):
    def make(pretrained=None, **kwargs):
        if pretrained not in (None, False, ''):
            raise NotImplementedError('pre-trained weights are loaded with load_checkpoint(path)')
        return VideoViTEncoder(patch_size=patch, embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=4, **kwargs)
    return make


video_vit_micro = _factory(128, 2, 2)          # parity-test shape
video_vit_small = _factory(384, 12, 6)
video_vit_base = _factory(768, 12, 12)
video_vit_large = _factory(1024, 24, 16)
video_vit_huge = _factory(1280, 32, 16, patch=(2, 14, 14))

"""ViT backbone holding the parameters of one modality tower (reference models/vits.py).

Same constructor arguments, attribute names and state-dict keys as the reference's ``ViT``
(``patch_embed.proj``, ``pos_embed``, ``blocks.N.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}``,
``norm``); the arithmetic runs on the HIP kernels via ``deepavfusion_amd.engine``.
"""
from functools import partial

import torch
import torch.nn as nn

from ..util.pos_embed import get_2d_sincos_pos_embed


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def init_linear_and_norm(m):
    """xavier-uniform Linear weights / zero biases, unit LayerNorm (models/vits.py:54-62)."""
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)


class PatchEmbed(nn.Module):
    """Parameter holder of timm's PatchEmbed (Conv2d kernel = stride = patch)."""
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size, self.patch_size = _pair(img_size), _pair(patch_size)
        if self.patch_size != (16, 16):
            raise NotImplementedError('the gfx950 patch kernels are specialised for 16x16 patches')
        self.grid_size = (self.img_size[0] // 16, self.img_size[1] // 16)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size, bias=True)


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias=True):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


def _prob(p):
    """A dropout probability as nn.Dropout accepts it, short of 1 (the kept values are scaled by 1 / (1 - p))."""
    p = float(p)
    if not 0.0 <= p < 1.0:
        raise ValueError(f'dropout probability has to be in [0, 1), got {p}')
    return p


class Block(nn.Module):
    """Parameter holder of timm's pre-LN Block (no LayerScale; DropPath on both residual branches when drop_path > 0)."""
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=True, norm_layer=nn.LayerNorm, drop_path=0., attn_drop=0.,
                 proj_drop=0.):
        super().__init__()
        # timm Block: Attention(attn_drop, proj_drop) and Mlp(drop=proj_drop) — nn.Dropout on the softmax probabilities, behind
        # proj, behind GELU and behind fc2; active in training mode only (every pre-training config has 0)
        self.attn_drop_prob, self.proj_drop_prob = _prob(attn_drop), _prob(proj_drop)
        self.num_heads = num_heads
        self.drop_path_prob = float(drop_path)        # timm DropPath on both residual branches (training mode only)
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads, qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class ViT(nn.Module):
    """models/vits.py:16-119."""
    def __init__(self, input_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4.,
                 norm_layer=nn.LayerNorm, use_cls_token=False, drop_path=0., attn_drop=0., drop=0.):
        super().__init__()
        if use_cls_token:
            raise NotImplementedError('DeepAVFusion builds its towers with use_cls_token=False (models/deepavfusion.py:20-21)')
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.patch_embed = PatchEmbed(input_size, patch_size, in_chans, embed_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim), requires_grad=False)
        self.cls_token = None
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer,
                                           drop_path=drop_path, attn_drop=attn_drop, proj_drop=drop) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.initialize_weights()

    def initialize_weights(self):
        pe = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], self.patch_embed.grid_size)
        self.pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        nn.init.xavier_uniform_(w.view(w.shape[0], -1))       # conv initialised like a Linear (models/vits.py:43-45)
        self.apply(init_linear_and_norm)

    def load_checkpoint(self, ckpt_fn, prefix='', skip_keys_prefix=('decoder', 'mask_token')):
        """models/vits.py:64-80: strict load of an MAE-style checkpoint into this tower."""
        ckpt = torch.load(ckpt_fn, map_location='cpu')
        ckpt = ckpt.get('state_dict', ckpt.get('model', ckpt))
        ckpt = {k[len(prefix):]: v for k, v in ckpt.items() if k.startswith(prefix)}
        ckpt = {k: v for k, v in ckpt.items() if not k.startswith(skip_keys_prefix)}
        ckpt.pop('cls_token', None)
        ckpt['pos_embed'] = self.state_dict()['pos_embed']
        self.load_state_dict(ckpt, strict=True)

    def params_layer_ids(self):
        ids = [(p, 0) for p in self.patch_embed.parameters()]
        ids.append((self.cls_token, 0))
        for i, blk in enumerate(self.blocks):
            ids.extend((p, i + 1) for p in blk.parameters())
        ids.extend((p, len(self.blocks) + 1) for p in self.norm.parameters())
        return ids

    def prepare_patch_tokens(self, x, ids_keep=None):
        """patch-embed + pos_embed + gather of the kept patches (models/vits.py:91-107) -> fp32 [B, n, D]."""
        from ..autograd_bridge import patch_tokens
        return patch_tokens(self, x, ids_keep)


def _factory(embed_dim, depth, num_heads):
    def make(pretrained=False, **kwargs):
        if pretrained not in (None, False, ''):
            raise NotImplementedError('pre-trained MAE/AudioMAE weights are loaded with load_checkpoint(path)')
        return ViT(patch_size=16, embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=4,
                   norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    return make


vit_micro = _factory(128, 2, 2)        # parity-test shape
vit_tiny = _factory(192, 12, 3)        # BASELINE.json configs[0] (timm-tiny widths; the reference has no vit_tiny)
vit_small = vit_small_patch16 = _factory(384, 12, 6)
vit_base = vit_base_patch16 = _factory(768, 12, 12)
vit_large = vit_large_patch16 = _factory(1024, 24, 16)

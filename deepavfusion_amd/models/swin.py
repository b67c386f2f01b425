"""Swin decoder blocks (reference models/swin.py), parameters + window geometry.

Same constructor arguments, attribute names and state-dict keys as the reference's ``WindowAttention`` /
``SwinTransformerBlock`` (``norm1``, ``attn.relative_position_bias_table``, ``attn.relative_position_index`` (buffer),
``attn.qkv``, ``attn.proj``, ``norm2``, ``mlp.fc1``, ``mlp.fc2``, ``attn_mask`` (buffer of shifted blocks)); the arithmetic
runs on the HIP kernels via ``deepavfusion_amd.engine.swin_block_fwd / swin_block_bwd``.  The cyclic shift, timm's
``window_partition`` / ``window_reverse`` and the per-window repeat of the fusion tokens (models/swin.py:172-197) are folded
into ONE row map per block, computed here once.
"""
import torch
import torch.nn as nn

from .vits import Mlp, _pair


def relative_position_index(win_h, win_w):
    """timm 0.9.2 get_relative_position_index (models/swin.py:39): [A, A] table index of each (query, key) slot pair."""
    coords = torch.stack(torch.meshgrid(torch.arange(win_h), torch.arange(win_w), indexing='ij')).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += win_h - 1
    rel[:, :, 1] += win_w - 1
    rel[:, :, 0] *= 2 * win_w - 1
    return rel.sum(-1)


def _windows(x2d, ws):
    """timm window_partition on an [H, W] index / label map -> [nW, ws*ws]."""
    H, W = x2d.shape
    return x2d.view(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)


class WindowAttention(nn.Module):
    """models/swin.py:10-53 (parameter holder)."""
    def __init__(self, dim, num_heads, head_dim=None, window_size=7, qkv_bias=True, attn_drop=0., proj_drop=0.):
        super().__init__()
        if attn_drop or proj_drop:
            raise NotImplementedError('attention / projection dropout is not on the gfx950 path')
        self.dim, self.window_size, self.num_heads = dim, _pair(window_size), num_heads
        win_h, win_w = self.window_size
        self.window_area = win_h * win_w
        head_dim = head_dim or dim // num_heads
        attn_dim = head_dim * num_heads
        self.scale = head_dim ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * win_h - 1) * (2 * win_w - 1), num_heads))
        self.register_buffer('relative_position_index', relative_position_index(win_h, win_w))
        self.qkv = nn.Linear(dim, attn_dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(attn_dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)


class SwinTransformerBlock(nn.Module):
    """models/swin.py:90-158 (parameter holder + the block's row maps)."""
    def __init__(self, dim, input_resolution, num_heads=4, head_dim=None, window_size=7, shift_size=0, mlp_ratio=4.,
                 qkv_bias=True, drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        if drop or attn_drop or drop_path:
            raise NotImplementedError('the decoders are built with drop = attn_drop = drop_path = 0 (models/avmae.py:46-48)')
        self.dim, self.input_resolution = dim, tuple(input_resolution)
        self.window_size, self.shift_size, self.mlp_ratio = window_size, shift_size, mlp_ratio
        if min(self.input_resolution) <= self.window_size:      # one unshifted window (models/swin.py:121-124)
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size, 'shift_size must in 0-window_size'
        H, W = self.input_resolution
        ws = self.window_size
        if H % ws or W % ws:
            raise ValueError(f'token grid {H}x{W} is not a multiple of the window {ws} (timm window_partition would fail too)')
        self.num_heads = num_heads
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, num_heads=num_heads, head_dim=head_dim, window_size=_pair(ws), qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        if self.shift_size > 0:                                  # models/swin.py:136-156
            img_mask = torch.zeros(H, W)
            cnt = 0
            for h in (slice(0, -ws), slice(-ws, -self.shift_size), slice(-self.shift_size, None)):
                for w in (slice(0, -ws), slice(-ws, -self.shift_size), slice(-self.shift_size, None)):
                    img_mask[h, w] = cnt
                    cnt += 1
            mw = _windows(img_mask, ws)
            am = mw.unsqueeze(1) - mw.unsqueeze(2)
            attn_mask = am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))
        else:
            attn_mask = None
        self.register_buffer('attn_mask', attn_mask)
        # row maps of the kernels (not part of the state dict): token at slot i of window w after roll(-shift), its inverse,
        # and the int32 copy of the table index
        ids = torch.arange(H * W).view(H, W)
        if self.shift_size > 0:
            ids = torch.roll(ids, shifts=(-self.shift_size, -self.shift_size), dims=(0, 1))
        rows = _windows(ids, ws).reshape(-1)
        inv = torch.empty_like(rows)
        inv[rows] = torch.arange(rows.numel())
        self.register_buffer('rows32', rows.to(torch.int32), persistent=False)
        self.register_buffer('inv32', inv.to(torch.int32), persistent=False)
        self.register_buffer('index32', self.attn.relative_position_index.reshape(-1).to(torch.int32), persistent=False)
        self.num_windows = (H // ws) * (W // ws)

    def forward(self, x, x_fusion=None):
        """models/swin.py:160-209 -> (x, x_fusion).  Standalone use of one block; AVMAE drives the decoder through the engine."""
        from ..autograd_bridge import swin_block_apply
        if x_fusion is None:
            raise NotImplementedError('the decoder always passes the fusion tokens (models/avmae.py:176)')
        return swin_block_apply(self, x, x_fusion)

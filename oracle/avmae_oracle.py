"""Functional fp32 CPU restatement of the AVMAE / DeepAVFusion pre-training path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Pure ``torch`` fp32 on the
host for the floating-point work, numpy for the integer/index work.  Every
function takes an explicit ``sd`` (a ``state_dict``-style mapping using the
reference's parameter names) so the same weights can be fed to the reference,
to this oracle and to the HIP path.  Gradients of the oracle are obtained by
autograd over this restatement.

Each function cites the reference ``file:line`` it follows (paths are relative
to ``/root/reference``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- #
# configuration
# --------------------------------------------------------------------------- #
@dataclass
class PathConfig:
    """Shapes/hyper-parameters of one AVMAE(DeepAVFusion(...)) instance.

    Mirrors the ctor arguments of ``models/deepavfusion.py:7-16`` and
    ``models/avmae.py:10-15`` plus the ViT factory constants of
    ``models/vits.py:121-170``.
    """
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: float = 4.0
    patch: int = 16
    image_size: Tuple[int, int] = (224, 224)
    audio_size: Tuple[int, int] = (128, 640)
    fusion_tkns: Tuple[int, int, int] = (16, 8, 8)      # (nmm, nv, na)
    fusion_layers: Tuple[int, ...] = field(default_factory=lambda: tuple(range(12)))
    fusion_mlp_ratio: float = 1.0
    fusion_attn_ratio: float = 0.25
    fusion_num_heads: int = 12
    fusion_arch: str = 'factorized_mmi'     # 'token' | 'dense_mmi' | 'factorized_mmi' (models/deepavfusion.py:28-35)
    decoder_dim: int = 512
    decoder_depth: int = 8
    decoder_heads: int = 16
    decoder_mlp_ratio: float = 4.0
    decoder_arch: str = 'plain'             # 'plain' | 'swin' for both decoders (models/avmae.py:37-51, 67-81)
    image_mask_ratio: float = 0.75
    audio_mask_ratio: float = 0.8
    image_norm_loss: bool = True
    audio_norm_loss: bool = True
    enc_eps: float = 1e-6      # models/vits.py:125,133,147,161
    fus_eps: float = 1e-5      # nn.LayerNorm default: models/deepavfusion.py:50,52
    dec_eps: float = 1e-5      # models/avmae.py:14

    @property
    def image_grid(self):
        return (self.image_size[0] // self.patch, self.image_size[1] // self.patch)

    @property
    def audio_grid(self):
        return (self.audio_size[0] // self.patch, self.audio_size[1] // self.patch)


@dataclass
class VideoConfig:
    """Shapes of one ``VideoEarlyFusion`` instance (BASELINE.json configs[4]): ctor arguments of
    ``models/video_earlyfusion.py:10-27`` + the factory constants of ``models/video_vits.py:246-345``
    (patch (2,16,16), norm eps 1e-6) and ``models/video_earlyfusion.py:134-171``."""
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
    enc_eps: float = 1e-6      # models/video_vits.py:135 / models/vits.py:125
    fus_eps: float = 1e-5      # nn.LayerNorm default: models/video_earlyfusion.py:50,54

    @property
    def video_grid(self):
        return tuple(s // p for s, p in zip(self.video_size, self.video_patch))

    @property
    def audio_grid(self):
        return (self.audio_size[0] // self.patch, self.audio_size[1] // self.patch)


# --------------------------------------------------------------------------- #
# util/pos_embed.py
# --------------------------------------------------------------------------- #
def sincos_1d(dim: int, pos: np.ndarray) -> np.ndarray:
    """util/pos_embed.py:72-90 — [sin | cos] of pos * 10000^(-i/(dim/2))."""
    assert dim % 2 == 0
    omega = np.arange(dim // 2, dtype=np.float32)
    omega /= dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum('m,d->md', pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(dim: int, grid_size: Tuple[int, int]) -> np.ndarray:
    """util/pos_embed.py:42-69.  Note ``meshgrid(grid_w, grid_h)`` (w first,
    :51): grid[0] varies along the W axis and is encoded into the first half
    of the channels."""
    gh, gw = grid_size
    ys = np.arange(gh, dtype=np.float32)
    xs = np.arange(gw, dtype=np.float32)
    g0, g1 = np.meshgrid(xs, ys)           # both [gh, gw]; g0 = x (w), g1 = y (h)
    emb_a = sincos_1d(dim // 2, g0)
    emb_b = sincos_1d(dim // 2, g1)
    return np.concatenate([emb_a, emb_b], axis=1)


def sincos_3d(dim: int, grid_size: Tuple[int, int, int], thw_props=(2, 1, 1)) -> np.ndarray:
    """util/pos_embed.py:16-40 (meshgrid(t, w, h, indexing='ij') quirk kept)."""
    h_dim = int(dim * (thw_props[1] / float(sum(thw_props))))
    w_dim = int(dim * (thw_props[2] / float(sum(thw_props))))
    t_dim = dim - h_dim - w_dim
    gt = np.arange(grid_size[0], dtype=np.float32)
    gh = np.arange(grid_size[1], dtype=np.float32)
    gw = np.arange(grid_size[2], dtype=np.float32)
    grid = np.stack(np.meshgrid(gt, gw, gh, indexing='ij'), axis=0)
    grid = grid.reshape([3, 1, grid_size[0], grid_size[1], grid_size[2]])
    return np.concatenate([sincos_1d(t_dim, grid[0]), sincos_1d(h_dim, grid[1]),
                           sincos_1d(w_dim, grid[2])], axis=1)


# --------------------------------------------------------------------------- #
# integer / index work (numpy): models/avmae.py:120-142
# --------------------------------------------------------------------------- #
def len_keep_of(L: int, mask_ratio: float) -> int:
    """models/avmae.py:132 — Python double arithmetic then ``int`` truncation
    (320 @ 0.8 -> 63, not 64)."""
    return int(L * (1 - mask_ratio))


def random_masking_from_noise(noise: np.ndarray, mask_ratio: float):
    """models/avmae.py:120-142 given the noise it would have drawn at :127.

    Returns (ids_keep int64 [N,len_keep], mask float32 [N,L], ids_restore int64
    [N,L]).  ``kind='stable'`` only matters for ties, which the fixtures avoid
    (torch.argsort is not stable).
    """
    N, L = noise.shape
    ids_shuffle = np.argsort(noise, axis=1, kind='stable')
    ids_restore = np.argsort(ids_shuffle, axis=1, kind='stable')
    lk = len_keep_of(L, mask_ratio)
    ids_keep = ids_shuffle[:, :lk]
    mask = np.ones((N, L), dtype=np.float32)
    mask[:, :lk] = 0
    mask = np.take_along_axis(mask, ids_restore, axis=1)
    return ids_keep.astype(np.int64), mask, ids_restore.astype(np.int64)


# --------------------------------------------------------------------------- #
# timm 0.9.2 pieces (restated from published semantics)
# --------------------------------------------------------------------------- #
def linear(x: Tensor, sd: Dict[str, Tensor], name: str) -> Tensor:
    return F.linear(x, sd[name + '.weight'], sd.get(name + '.bias'))


def layer_norm(x: Tensor, sd: Dict[str, Tensor], name: str, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[name + '.weight'], sd[name + '.bias'], eps)


def dropout_keep_mask(seed: int, name: str, shape, p: float) -> np.ndarray:
    """Closed-form 0 / 1 keep mask of the nn.Dropout called ``name`` (fixtures and tests regenerate it instead of storing it)."""
    import zlib
    rs = np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) & 0x7fffffff)
    return (rs.uniform(size=tuple(shape)) >= p).astype(np.float32)


def _do(dm, site: str, x: Tensor) -> Tensor:
    """nn.Dropout in training mode with the draw made elsewhere: ``dm(site, x)`` returns keep_mask * x / (1 - p) for the Dropout
    module called ``site`` of the current block ('attn', 'proj', 'fc1', 'fc2', 'attn_v.attn', ...); None = eval mode / p = 0."""
    return x if dm is None else dm(site, x)


def softmax_attention(q: Tensor, k: Tensor, v: Tensor, scale: float, dm=None, site: str = 'attn') -> Tensor:
    """q [B,H,Nq,dqk], k [B,H,Nk,dqk], v [B,H,Nk,dv] -> [B,H,Nq,dv].
    Explicit form of F.scaled_dot_product_attention / the reference's
    ``(q @ k.T) * scale -> softmax -> attn_drop -> @ v`` (models/fusion_blocks.py:52-56)."""
    attn = (q @ k.transpose(-2, -1)) * scale
    attn = _do(dm, site, attn.softmax(dim=-1))
    return attn @ v


def patch_embed(x: Tensor, sd: Dict[str, Tensor], name: str, patch: int) -> Tensor:
    """timm PatchEmbed: Conv2d(kernel=stride=patch) -> flatten(2).transpose(1,2).
    Restated as an unfold + matmul so the patch ordering is explicit:
    token t = gy*gW + gx, feature f = c*p*p + py*p + px."""
    B, C, H, W = x.shape
    gh, gw = H // patch, W // patch
    w = sd[name + '.proj.weight'].reshape(-1, C * patch * patch)
    cols = x.reshape(B, C, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * patch * patch)
    return cols @ w.t() + sd[name + '.proj.bias']


def timm_attention(x: Tensor, sd, name: str, heads: int, dm=None) -> Tensor:
    """timm 0.9.2 Attention.forward (attn_drop on the probabilities, proj_drop behind proj)."""
    B, N, C = x.shape
    hd = C // heads
    qkv = linear(x, sd, name + '.qkv').reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    o = softmax_attention(qkv[0], qkv[1], qkv[2], hd ** -0.5, dm)
    o = o.transpose(1, 2).reshape(B, N, C)
    return _do(dm, 'proj', linear(o, sd, name + '.proj'))


def timm_mlp(x: Tensor, sd, name: str, dm=None) -> Tensor:
    """timm Mlp: fc1 -> GELU -> drop1 -> fc2 -> drop2."""
    return _do(dm, 'fc2', linear(_do(dm, 'fc1', F.gelu(linear(x, sd, name + '.fc1'))), sd, name + '.fc2'))   # exact erf GELU


def _dp(branch: Tensor, s: Optional[Tensor]) -> Tensor:
    """timm DropPath with the per-sample scale s[b] in {0, 1/keep} already drawn (None = inactive)."""
    return branch if s is None else branch * s.view(-1, *([1] * (branch.ndim - 1)))


def timm_block(x: Tensor, sd, name: str, heads: int, eps: float, dp=None, dm=None) -> Tensor:
    """Pre-LN block, no LayerScale; ``dp`` = (s_attn, s_mlp) DropPath scales of drop_path1 / drop_path2 (fine-tuning);
    ``dm``: the block's dropout, see _do."""
    x = x + _dp(timm_attention(layer_norm(x, sd, name + '.norm1', eps), sd, name + '.attn', heads, dm), None if dp is None else dp[0])
    x = x + _dp(timm_mlp(layer_norm(x, sd, name + '.norm2', eps), sd, name + '.mlp', dm), None if dp is None else dp[1])
    return x


# --------------------------------------------------------------------------- #
# models/vits.py
# --------------------------------------------------------------------------- #
def prepare_patch_tokens(x: Tensor, sd, name: str, patch: int, ids_keep: Optional[Tensor]) -> Tensor:
    """models/vits.py:91-107 with use_cls_token=False (models/deepavfusion.py:20-21)."""
    t = patch_embed(x, sd, name + '.patch_embed', patch) + sd[name + '.pos_embed']
    if ids_keep is not None:
        t = t.gather(1, ids_keep.unsqueeze(-1).expand(-1, -1, t.shape[-1]))
    return t


def patch_embed3d(x: Tensor, sd: Dict[str, Tensor], name: str, patch: Tuple[int, int, int]) -> Tensor:
    """util/pos_embed.py:123-146 — Conv3d(kernel=stride=patch) -> flatten(2).transpose(1,2), restated as
    unfold + matmul: token = (gt*gH + gy)*gW + gx, feature = ((c*pt + dt)*ph + py)*pw + px."""
    B, C, T, H, W = x.shape
    pt, ph, pw = patch
    gt, gh, gw = T // pt, H // ph, W // pw
    w = sd[name + '.proj.weight'].reshape(-1, C * pt * ph * pw)
    cols = x.reshape(B, C, gt, pt, gh, ph, gw, pw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, gt * gh * gw, C * pt * ph * pw)
    return cols @ w.t() + sd[name + '.proj.bias']


def video_prepare_patch_tokens(x: Tensor, sd, name: str, patch: Tuple[int, int, int], ids_keep: Optional[Tensor]) -> Tensor:
    """models/video_vits.py:218-239 with use_cls_token=False: the gather comes BEFORE the pos-embed add
    (:229-232), so kept-token subsets only broadcast when len(ids_keep) == num_patches."""
    t = patch_embed3d(x, sd, name + '.patch_embed', patch)
    if ids_keep is not None:
        t = t.gather(1, ids_keep.unsqueeze(-1).expand(-1, -1, t.shape[-1]))
    return t + sd[name + '.pos_embed']


# --------------------------------------------------------------------------- #
# models/fusion_blocks.py
# --------------------------------------------------------------------------- #
def cross_attention(x1: Tensor, x2: Tensor, sd, name: str, heads: int, dm=None, site: str = '') -> Tensor:
    """models/fusion_blocks.py:46-59."""
    B, N1, C = x1.shape
    N2 = x2.shape[1]
    hd = C // heads
    q = linear(x1, sd, name + '.q').reshape(B, N1, heads, hd).permute(0, 2, 1, 3)
    kv = linear(x2, sd, name + '.kv').reshape(B, N2, 2, heads, hd).permute(2, 0, 3, 1, 4)
    o = softmax_attention(q, kv[0], kv[1], hd ** -0.5, dm, site + 'attn')
    return _do(dm, site + 'proj', linear(o.transpose(1, 2).reshape(B, N1, C), sd, name + '.proj'))


def factorized_attention(xmm: Tensor, xv: Tensor, xa: Tensor, sd, name: str, heads: int,
                         tkns: Sequence[int], dm=None) -> Tensor:
    """models/fusion_blocks.py:235-263.

    Pair p = i*na + j carries features [xmm_v[i] || xmm_a[j]] (:245-248); the
    softmax scale is (dim/heads)^-0.5 regardless of the q/k width (:220-222).
    """
    B, _, C = xmm.shape
    nmm, nv, na = tkns
    x2, x_v, x_a = xmm.split((nmm, nv, na), dim=1)
    x_v = cross_attention(x_v, xv, sd, name + '.attn_v', heads, dm, 'attn_v.')
    x_a = cross_attention(x_a, xa, sd, name + '.attn_a', heads, dm, 'attn_a.')
    pairs = torch.cat((x_v[:, :, None, :].expand(B, nv, na, C),
                       x_a[:, None, :, :].expand(B, nv, na, C)), dim=3).reshape(B, nv * na, 2 * C)
    q = linear(x2, sd, name + '.q').reshape(B, nmm, heads, -1).permute(0, 2, 1, 3)
    k = linear(pairs, sd, name + '.k').reshape(B, nv * na, heads, -1).permute(0, 2, 1, 3)
    v = linear(pairs, sd, name + '.v').reshape(B, nv * na, heads, -1).permute(0, 2, 1, 3)
    o = softmax_attention(q, k, v, (C // heads) ** -0.5, dm)
    x2 = _do(dm, 'proj', linear(o.transpose(1, 2).flatten(2), sd, name + '.proj'))
    return torch.cat((x2, x_v, x_a), dim=1)


def fusion_block_factorized(xmm: Tensor, xv: Tensor, xa: Tensor, sd, name: str, heads: int,
                            tkns: Sequence[int], eps: float, dp=None, dm=None) -> Tensor:
    """models/fusion_blocks.py:280-289 — norm-THEN-residual: the residual base
    is the normed xmm (:281-283)."""
    xmm = layer_norm(xmm, sd, name + '.norm1_mm', eps)
    xv = layer_norm(xv, sd, name + '.norm1_img', eps)
    xa = layer_norm(xa, sd, name + '.norm1_aud', eps)
    xmm = xmm + _dp(factorized_attention(xmm, xv, xa, sd, name + '.attn', heads, tkns, dm), None if dp is None else dp[0])
    xmm = xmm + _dp(timm_mlp(layer_norm(xmm, sd, name + '.norm2', eps), sd, name + '.mlp', dm), None if dp is None else dp[1])
    return xmm


def local_av_attention(xmm: Tensor, xv: Tensor, xa: Tensor, sd, name: str, heads: int, dm=None) -> Tensor:
    """CrossAttention_LocalAVTokens.forward, models/fusion_blocks.py:103-117: fusion tokens attend to
    cat(xv, xa); q/k/v all of width Da = dim*dim_ratio, scale (Da/heads)^-0.5 (:93-95)."""
    B, nmm, _ = xmm.shape
    Da = sd[name + '.q.weight'].shape[0]
    hd = Da // heads
    src = torch.cat((xv, xa), dim=1)
    q = linear(xmm, sd, name + '.q').reshape(B, nmm, heads, hd).permute(0, 2, 1, 3)
    kv = linear(src, sd, name + '.kv').reshape(B, src.shape[1], 2, heads, hd).permute(2, 0, 3, 1, 4)
    o = softmax_attention(q, kv[0], kv[1], hd ** -0.5, dm)
    return _do(dm, 'proj', linear(o.transpose(1, 2).reshape(B, nmm, Da), sd, name + '.proj'))


def fusion_block_token(x_f: Tensor, x_image: Tensor, x_audio: Tensor, sd, name: str, heads: int, eps: float, dp=None, dm=None) -> Tensor:
    """FusionBlock_LocalAVTokens, models/fusion_blocks.py:120-145, as CALLED by models/deepavfusion.py:106
    ``blk_fusion(x_fusion, x_image, x_audio)`` against the signature ``forward(self, xmm, xa, xv)`` (:135):
    xa := x_image, xv := x_audio, so norm1_img normalises the AUDIO tokens and norm1_aud the IMAGE tokens (:136),
    and the key/value sequence is cat(normed audio, normed image) (:106 of fusion_blocks)."""
    xmm = layer_norm(x_f, sd, name + '.norm1_mm', eps)
    xv = layer_norm(x_audio, sd, name + '.norm1_img', eps)
    xa = layer_norm(x_image, sd, name + '.norm1_aud', eps)
    xmm = xmm + _dp(local_av_attention(xmm, xv, xa, sd, name + '.attn', heads, dm), None if dp is None else dp[0])
    return xmm + _dp(timm_mlp(layer_norm(xmm, sd, name + '.norm2', eps), sd, name + '.mlp', dm), None if dp is None else dp[1])


def dense_av_attention(xmm: Tensor, first: Tensor, second: Tensor, sd, name: str, heads: int, dm=None) -> Tensor:
    """CrossAttention_DenseAVInteractions.forward(self, xmm, xa, xv), models/fusion_blocks.py:168-188, with
    ``first`` bound to its ``xa`` and ``second`` to its ``xv``: pairs p = i*na + j carry [xv[i] || xa[j]] =
    [second[i] || first[j]] (:171-174); k/v width Da, scale (dim/heads)^-0.5 from the FULL dim (:157-158)."""
    B, nmm, C = xmm.shape
    Da = sd[name + '.q.weight'].shape[0]
    hd = Da // heads
    nv, na = second.shape[1], first.shape[1]
    pairs = torch.cat((second[:, :, None, :].expand(B, nv, na, C), first[:, None, :, :].expand(B, nv, na, C)), dim=3)
    pairs = pairs.reshape(B, nv * na, 2 * C)
    q = linear(xmm, sd, name + '.q').reshape(B, nmm, heads, hd).permute(0, 2, 1, 3)
    kv = linear(pairs, sd, name + '.kv').reshape(B, nv * na, 2, heads, hd).permute(2, 0, 3, 1, 4)
    o = softmax_attention(q, kv[0], kv[1], (C // heads) ** -0.5, dm)
    return _do(dm, 'proj', linear(o.transpose(1, 2).reshape(B, nmm, Da), sd, name + '.proj'))


def fusion_block_dense(x_f: Tensor, x_image: Tensor, x_audio: Tensor, sd, name: str, heads: int, eps: float, dp=None, dm=None) -> Tensor:
    """FusionBlock_DenseAVInteractions, models/fusion_blocks.py:191-213: forward(xmm, xv, xa) takes the call of
    models/deepavfusion.py:106 in order, but passes ``self.attn(xmm, xv, xa)`` (:206) to a forward declared
    ``(xmm, xa, xv)`` (:168) — inside the attention the image tokens play "xa" and the audio tokens "xv", i.e.
    pairs are (audio_i, image_j)."""
    xmm = layer_norm(x_f, sd, name + '.norm1_mm', eps)
    xv = layer_norm(x_image, sd, name + '.norm1_img', eps)
    xa = layer_norm(x_audio, sd, name + '.norm1_aud', eps)
    xmm = xmm + _dp(dense_av_attention(xmm, xv, xa, sd, name + '.attn', heads, dm), None if dp is None else dp[0])
    return xmm + _dp(timm_mlp(layer_norm(xmm, sd, name + '.norm2', eps), sd, name + '.mlp', dm), None if dp is None else dp[1])


# --------------------------------------------------------------------------- #
# models/deepavfusion.py
# --------------------------------------------------------------------------- #
def _early_fusion_layers(sd, cfg, x_v: Tensor, x_a: Tensor, prefix: str, vis: str, return_embs: bool, drop=None, dropout=None):
    """The layer loop shared by models/deepavfusion.py:96-118 and models/video_earlyfusion.py:107-131
    (``vis`` = 'image' / 'video'; the video Block in 'joint_all' mode is the timm pre-LN block,
    models/video_vits.py:46-47,94)."""
    B = x_v.shape[0]
    x_f = sd[prefix + 'fusion_tokens'].expand(B, -1, -1)
    nF = x_f.shape[1]
    embs = []
    dpo = (lambda tag: None) if drop is None else (lambda tag: drop.get(tag))     # drop: {'visual.l' | 'audio.l' | 'fusion.l': (s_attn, s_mlp)}
    # dropout(name, x) -> keep * x / (1 - p) for the nn.Dropout called name = '<visual|audio|fusion>.<l>.<site>' (training mode, attn_drop / drop > 0)
    dmo = (lambda tag: None) if dropout is None else (lambda tag: (lambda site, x: dropout(f'{tag}.{site}', x)))
    for l in range(cfg.depth):
        if l not in cfg.fusion_layers:
            x_v = timm_block(x_v, sd, f'{prefix}{vis}.blocks.{l}', cfg.num_heads, cfg.enc_eps, dpo(f'visual.{l}'), dmo(f'visual.{l}'))
            x_a = timm_block(x_a, sd, f'{prefix}audio.blocks.{l}', cfg.num_heads, cfg.enc_eps, dpo(f'audio.{l}'), dmo(f'audio.{l}'))
        else:
            # fusion tokens are context rows whose own outputs are dropped (:104-105);
            # the fusion block reads the layer's INPUT x_v / x_a (:106-107)
            n_v = timm_block(torch.cat((x_f, x_v), 1), sd, f'{prefix}{vis}.blocks.{l}', cfg.num_heads, cfg.enc_eps, dpo(f'visual.{l}'), dmo(f'visual.{l}'))[:, nF:]
            n_a = timm_block(torch.cat((x_f, x_a), 1), sd, f'{prefix}audio.blocks.{l}', cfg.num_heads, cfg.enc_eps, dpo(f'audio.{l}'), dmo(f'audio.{l}'))[:, nF:]
            arch = getattr(cfg, 'fusion_arch', 'factorized_mmi')
            if arch == 'token':
                x_f = fusion_block_token(x_f, x_v, x_a, sd, f'{prefix}fusion_blocks.{l}', cfg.fusion_num_heads, cfg.fus_eps, dpo(f'fusion.{l}'), dmo(f'fusion.{l}'))
            elif arch == 'dense_mmi':
                x_f = fusion_block_dense(x_f, x_v, x_a, sd, f'{prefix}fusion_blocks.{l}', cfg.fusion_num_heads, cfg.fus_eps, dpo(f'fusion.{l}'), dmo(f'fusion.{l}'))
            else:
                x_f = fusion_block_factorized(x_f, x_v, x_a, sd, f'{prefix}fusion_blocks.{l}', cfg.fusion_num_heads,
                                              cfg.fusion_tkns, cfg.fus_eps, dpo(f'fusion.{l}'), dmo(f'fusion.{l}'))
            x_v, x_a = n_v, n_a
        if return_embs:
            embs.append((x_v, x_a, x_f))
    x_v = layer_norm(x_v, sd, f'{prefix}{vis}.norm', cfg.enc_eps)
    x_a = layer_norm(x_a, sd, prefix + 'audio.norm', cfg.enc_eps)
    x_f = layer_norm(x_f, sd, prefix + 'fusion_norm', cfg.fus_eps)
    if return_embs:
        return x_v, x_a, x_f, embs
    return x_v, x_a, x_f


def deepavfusion_forward(sd, cfg: PathConfig, image: Tensor, audio: Tensor,
                         image_ids_keep: Optional[Tensor] = None, audio_ids_keep: Optional[Tensor] = None,
                         prefix: str = '', return_embs: bool = False, drop=None, dropout=None):
    """models/deepavfusion.py:88-118.  ``drop``: DropPath scales per block (training with drop_path > 0), ``dropout``: the
    nn.Dropout modules' draws (training with attn_drop / drop > 0), see _early_fusion_layers."""
    x_i = prepare_patch_tokens(image, sd, prefix + 'image', cfg.patch, image_ids_keep)
    x_a = prepare_patch_tokens(audio, sd, prefix + 'audio', cfg.patch, audio_ids_keep)
    return _early_fusion_layers(sd, cfg, x_i, x_a, prefix, 'image', return_embs, drop, dropout)


def video_earlyfusion_forward(sd, cfg: VideoConfig, video: Tensor, audio: Tensor,
                              video_ids_keep: Optional[Tensor] = None, audio_ids_keep: Optional[Tensor] = None,
                              prefix: str = '', return_embs: bool = False):
    """models/video_earlyfusion.py:95-131: video [B,3,T,H,W], audio [B,1,n_mels,frames]."""
    x_v = video_prepare_patch_tokens(video, sd, prefix + 'video', cfg.video_patch, video_ids_keep)
    x_a = prepare_patch_tokens(audio, sd, prefix + 'audio', cfg.patch, audio_ids_keep)
    return _early_fusion_layers(sd, cfg, x_v, x_a, prefix, 'video', return_embs)


# --------------------------------------------------------------------------- #
# models/avmae.py
# --------------------------------------------------------------------------- #
def patchify(x: Tensor, patch: Tuple[int, int]) -> Tensor:
    """models/avmae.py:200-214 — 'nchpwq->nhwpqc': channel is the fastest axis
    of the patch vector."""
    B, C, H, W = x.shape
    pH, pW = patch
    gH, gW = H // pH, W // pW
    return x.reshape(B, C, gH, pH, gW, pW).permute(0, 2, 4, 3, 5, 1).reshape(B, gH * gW, pH * pW * C)


def forward_loss(target: Tensor, pred: Tensor, mask: Tensor, norm_pix_loss: bool = True) -> Tensor:
    """models/avmae.py:182-198 — unbiased variance (:191), eps 1e-6 inside the sqrt."""
    if norm_pix_loss:
        mean = target.mean(dim=-1, keepdim=True)
        var = target.var(dim=-1, keepdim=True, unbiased=True)
        target = (target - mean) / (var + 1.e-6) ** .5
    loss = ((pred - target) ** 2).mean(dim=-1)
    return (loss * mask).sum() / mask.sum()


# --------------------------------------------------------------------------- #
# Swin decoder blocks (models/swin.py; decoder_arch == 'swin', models/avmae.py:37-51, 174-176)
# --------------------------------------------------------------------------- #
SWIN_WINDOW = 4          # models/avmae.py:42, 72


def swin_geometry(res: Tuple[int, int], index: int):
    """(window, shift) of decoder block ``index`` on a ``res`` token grid: models/avmae.py:42-43 (window 4, shift 2 on odd
    blocks) after the clamp of models/swin.py:121-124 (a grid no larger than the window is ONE unshifted window)."""
    window, shift = SWIN_WINDOW, (index % 2) * 2
    if min(res) <= window:
        window, shift = min(res), 0
    return window, shift


def relative_position_index(win: int) -> Tensor:
    """timm 0.9.2 ``get_relative_position_index(win, win)`` (not vendored in the reference; published algorithm: pairwise
    coordinate differences of the window's tokens, shifted to start at 0, row difference scaled by 2*win - 1)."""
    coords = torch.stack(torch.meshgrid(torch.arange(win), torch.arange(win), indexing='ij')).flatten(1)      # [2, A]
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0) + (win - 1)
    return rel[..., 0] * (2 * win - 1) + rel[..., 1]                                                          # [A, A]


def window_rows(res: Tuple[int, int], window: int, shift: int) -> Tensor:
    """Token index (row of the [H*W] sequence) of every window slot after the cyclic shift: ``out[w, i]`` is the token
    that ``window_partition(torch.roll(x, (-shift, -shift), (1, 2)), window)`` puts at slot i of window w
    (models/swin.py:172-179; timm window_partition: windows row-major over the grid, slots row-major inside a window)."""
    H, W = res
    ids = torch.arange(H * W).view(H, W)
    if shift > 0:
        ids = torch.roll(ids, shifts=(-shift, -shift), dims=(0, 1))
    return ids.view(H // window, window, W // window, window).permute(0, 2, 1, 3).reshape(-1, window * window)


def shifted_window_mask(res: Tuple[int, int], window: int, shift: int) -> Optional[Tensor]:
    """models/swin.py:136-156: region ids of the 3 x 3 slices of the (shifted) grid; pairs of window slots from different
    regions get -100.  None for unshifted blocks."""
    if shift == 0:
        return None
    H, W = res
    img = torch.zeros(H, W)
    cnt = 0
    for h in (slice(0, -window), slice(-window, -shift), slice(-shift, None)):
        for w in (slice(0, -window), slice(-window, -shift), slice(-shift, None)):
            img[h, w] = cnt
            cnt += 1
    mw = img.view(H // window, window, W // window, window).permute(0, 2, 1, 3).reshape(-1, window * window)
    diff = mw[:, None, :] - mw[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))                    # [nW, A, A]


def window_attention(x: Tensor, sd, name: str, heads: int, win: int, mask: Optional[Tensor]) -> Tensor:
    """models/swin.py:55-87 on [nW*B, N, C] sequences of A = win*win window tokens followed by N - A fusion tokens: the
    relative-position bias (and the shift mask) cover the A x A corner, zero elsewhere (:69-72, :75-78)."""
    B_, N, C = x.shape
    A = win * win
    qkv = linear(x, sd, name + '.qkv').reshape(B_, N, 3, heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    scale = q.shape[-1] ** -0.5
    attn = (q * scale) @ k.transpose(-2, -1)
    idx = sd.get(name + '.relative_position_index')
    idx = relative_position_index(win) if idx is None else idx.long()
    bias = sd[name + '.relative_position_bias_table'][idx.reshape(-1)].view(A, A, -1).permute(2, 0, 1)      # [heads, A, A]
    attn = attn + torch.nn.functional.pad(bias, (0, N - A, 0, N - A))[None]
    if mask is not None:
        nW = mask.shape[0]
        m = torch.nn.functional.pad(mask, (0, N - A, 0, N - A))
        attn = (attn.view(B_ // nW, nW, heads, N, N) + m[None, :, None]).view(-1, heads, N, N)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B_, N, -1)
    return linear(x, sd, name + '.proj')


def swin_block(x: Tensor, x_fusion: Tensor, sd, name: str, heads: int, res: Tuple[int, int], index: int, eps: float):
    """models/swin.py:160-209 with x_fusion given (the decoder always passes it, models/avmae.py:176): every window attends
    over its 16 tokens + ALL fusion tokens; the fusion tokens' outputs are averaged over the windows (:199)."""
    B, L, C = x.shape
    Lf = x_fusion.shape[1]
    win, shift = swin_geometry(res, index)
    rows = window_rows(res, win, shift)                            # [nW, A]
    nW, A = rows.shape
    xn = layer_norm(x, sd, name + '.norm1', eps)
    fn = layer_norm(x_fusion, sd, name + '.norm1', eps)
    xw = xn[:, rows.reshape(-1)].reshape(B * nW, A, C)             # window_partition(roll(.)) as one gather
    seq = torch.cat([xw, fn[:, None].expand(B, nW, Lf, C).reshape(B * nW, Lf, C)], dim=1)
    mask = sd.get(name + '.attn_mask')
    if mask is None:
        mask = shifted_window_mask(res, win, shift)
    y = window_attention(seq, sd, name + '.attn', heads, win, mask)
    yw, yf = y[:, :A], y[:, A:]
    merged = torch.zeros_like(x)
    merged[:, rows.reshape(-1)] = yw.reshape(B, nW * A, C)         # window_reverse + roll back = the inverse scatter
    yf = yf.reshape(B, nW, Lf, C).mean(1)
    z = torch.cat([x, x_fusion], dim=1) + torch.cat([merged, yf], dim=1)
    z = z + timm_mlp(layer_norm(z, sd, name + '.norm2', eps), sd, name + '.mlp')
    return z[:, :L], z[:, L:]


def forward_decoder(x: Tensor, x_fusion: Tensor, ids_restore: Tensor, sd, cfg: PathConfig, modality: str) -> Tensor:
    """models/avmae.py:147-180, decoder_arch 'plain' or 'swin'.  ``embed`` is shared by
    modality tokens and fusion tokens (:158)."""
    B, nF = x.shape[0], x_fusion.shape[1]
    L = ids_restore.shape[1]
    p = f'{modality}_decoder_'
    x = linear(x, sd, p + 'embed')
    x_fusion = linear(x_fusion, sd, p + 'embed')
    n_mask = L - x.shape[1]
    x = torch.cat([x, sd[p + 'mask_token'].expand(B, n_mask, -1)], dim=1)
    x = x.gather(1, ids_restore.unsqueeze(-1).expand(-1, -1, x.shape[2]))
    x = x + sd[p + 'pos_embed']
    if cfg.decoder_arch == 'swin':                                 # models/avmae.py:174-176
        res = cfg.image_grid if modality == 'image' else cfg.audio_grid
        for l in range(cfg.decoder_depth):
            x, x_fusion = swin_block(x, x_fusion, sd, f'{p}blocks.{l}', cfg.decoder_heads, res, l, cfg.dec_eps)
        return linear(layer_norm(x, sd, p + 'norm', cfg.dec_eps), sd, p + 'pred')
    x = torch.cat([x_fusion, x], dim=1)
    for l in range(cfg.decoder_depth):
        x = timm_block(x, sd, f'{p}blocks.{l}', cfg.decoder_heads, cfg.dec_eps)
    x = x[:, nF:]
    return linear(layer_norm(x, sd, p + 'norm', cfg.dec_eps), sd, p + 'pred')


def avmae_forward(sd, cfg: PathConfig, image: Tensor, audio: Tensor,
                  noise_image: np.ndarray, noise_audio: np.ndarray):
    """models/avmae.py:216-236 with the masking noise injected.

    Returns (loss_image, loss_audio, pred_image, pred_audio, aux) where ``aux``
    holds the index tensors and encoder outputs.
    """
    ik, im, ir = random_masking_from_noise(noise_image, cfg.image_mask_ratio)
    ak, am, ar = random_masking_from_noise(noise_audio, cfg.audio_mask_ratio)
    dev = image.device          # (host by default; the GPU suite also runs this restatement on the device through stock PyTorch fp32 kernels
    #                              for the bench-size checks, where the host would need minutes)
    ik_t, ir_t, im_t = torch.from_numpy(ik).to(dev), torch.from_numpy(ir).to(dev), torch.from_numpy(im).to(dev)
    ak_t, ar_t, am_t = torch.from_numpy(ak).to(dev), torch.from_numpy(ar).to(dev), torch.from_numpy(am).to(dev)
    x_i, x_a, x_f = deepavfusion_forward(sd, cfg, image, audio, ik_t, ak_t, prefix='encoder.')
    pred_i = forward_decoder(x_i, x_f, ir_t, sd, cfg, 'image')
    loss_i = forward_loss(patchify(image, (cfg.patch, cfg.patch)), pred_i, im_t, cfg.image_norm_loss)
    pred_a = forward_decoder(x_a, x_f, ar_t, sd, cfg, 'audio')
    loss_a = forward_loss(patchify(audio, (cfg.patch, cfg.patch)), pred_a, am_t, cfg.audio_norm_loss)
    aux = dict(image_ids_keep=ik, image_mask=im, image_ids_restore=ir,
               audio_ids_keep=ak, audio_mask=am, audio_ids_restore=ar,
               x_image=x_i, x_audio=x_a, x_fusion=x_f)
    return loss_i, loss_a, pred_i, pred_a, aux


# --------------------------------------------------------------------------- #
# parameter construction (shapes / names of the state-dict contract)
# --------------------------------------------------------------------------- #
def _block_shapes(prefix: str, dim: int, hidden: int):
    return {
        f'{prefix}.norm1.weight': (dim,), f'{prefix}.norm1.bias': (dim,),
        f'{prefix}.attn.qkv.weight': (3 * dim, dim), f'{prefix}.attn.qkv.bias': (3 * dim,),
        f'{prefix}.attn.proj.weight': (dim, dim), f'{prefix}.attn.proj.bias': (dim,),
        f'{prefix}.norm2.weight': (dim,), f'{prefix}.norm2.bias': (dim,),
        f'{prefix}.mlp.fc1.weight': (hidden, dim), f'{prefix}.mlp.fc1.bias': (hidden,),
        f'{prefix}.mlp.fc2.weight': (dim, hidden), f'{prefix}.mlp.fc2.bias': (dim,),
    }


def _tower_shapes(s, pre, n_patches, conv_shape, D, depth, mlp_ratio):
    s[f'{pre}.pos_embed'] = (1, n_patches, D)
    s[f'{pre}.patch_embed.proj.weight'] = conv_shape
    s[f'{pre}.patch_embed.proj.bias'] = (D,)
    for l in range(depth):
        s.update(_block_shapes(f'{pre}.blocks.{l}', D, int(D * mlp_ratio)))
    s[f'{pre}.norm.weight'] = (D,)
    s[f'{pre}.norm.bias'] = (D,)


def _fusion_shapes(s, enc, cfg):
    D = cfg.embed_dim
    s[f'{enc}fusion_tokens'] = (1, sum(cfg.fusion_tkns), D)
    Da = int(D * cfg.fusion_attn_ratio)
    Hf = int(D * cfg.fusion_mlp_ratio)
    for l in cfg.fusion_layers:
        pre = f'{enc}fusion_blocks.{l}'
        for n in ('norm1_mm', 'norm1_aud', 'norm1_img', 'norm2'):
            s[f'{pre}.{n}.weight'] = (D,)
            s[f'{pre}.{n}.bias'] = (D,)
        arch = getattr(cfg, 'fusion_arch', 'factorized_mmi')
        if arch in ('token', 'dense_mmi'):          # models/fusion_blocks.py:97-101 / :160-164
            kin = D if arch == 'token' else 2 * D
            s[f'{pre}.attn.q.weight'] = (Da, D); s[f'{pre}.attn.q.bias'] = (Da,)
            s[f'{pre}.attn.kv.weight'] = (2 * Da, kin); s[f'{pre}.attn.kv.bias'] = (2 * Da,)
            s[f'{pre}.attn.proj.weight'] = (D, Da); s[f'{pre}.attn.proj.bias'] = (D,)
        else:
            for ca in ('attn_v', 'attn_a'):
                s[f'{pre}.attn.{ca}.q.weight'] = (D, D); s[f'{pre}.attn.{ca}.q.bias'] = (D,)
                s[f'{pre}.attn.{ca}.kv.weight'] = (2 * D, D); s[f'{pre}.attn.{ca}.kv.bias'] = (2 * D,)
                s[f'{pre}.attn.{ca}.proj.weight'] = (D, D); s[f'{pre}.attn.{ca}.proj.bias'] = (D,)
            s[f'{pre}.attn.q.weight'] = (Da, D); s[f'{pre}.attn.q.bias'] = (Da,)
            s[f'{pre}.attn.k.weight'] = (Da, 2 * D); s[f'{pre}.attn.k.bias'] = (Da,)
            s[f'{pre}.attn.v.weight'] = (D, 2 * D); s[f'{pre}.attn.v.bias'] = (D,)
            s[f'{pre}.attn.proj.weight'] = (D, D); s[f'{pre}.attn.proj.bias'] = (D,)
        s[f'{pre}.mlp.fc1.weight'] = (Hf, D); s[f'{pre}.mlp.fc1.bias'] = (Hf,)
        s[f'{pre}.mlp.fc2.weight'] = (D, Hf); s[f'{pre}.mlp.fc2.bias'] = (D,)
    s[f'{enc}fusion_norm.weight'] = (D,)
    s[f'{enc}fusion_norm.bias'] = (D,)


def video_state_shapes(cfg: VideoConfig) -> Dict[str, Tuple[int, ...]]:
    """Names and shapes of ``VideoEarlyFusion(...).state_dict()`` (models/video_earlyfusion.py:29-54)."""
    D, p = cfg.embed_dim, cfg.patch
    s: Dict[str, Tuple[int, ...]] = {}
    gv, ga = cfg.video_grid, cfg.audio_grid
    _tower_shapes(s, 'video', gv[0] * gv[1] * gv[2], (D, 3) + tuple(cfg.video_patch), D, cfg.depth, cfg.mlp_ratio)
    _tower_shapes(s, 'audio', ga[0] * ga[1], (D, 1, p, p), D, cfg.depth, cfg.mlp_ratio)
    _fusion_shapes(s, '', cfg)
    return s


def state_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    """Names and shapes of ``AVMAE(DeepAVFusion(...)).state_dict()``; for a VideoConfig those
    of ``VideoEarlyFusion``."""
    if isinstance(cfg, VideoConfig):
        return video_state_shapes(cfg)
    D, p = cfg.embed_dim, cfg.patch
    s: Dict[str, Tuple[int, ...]] = {}
    for mod, grid, cin in (('image', cfg.image_grid, 3), ('audio', cfg.audio_grid, 1)):
        _tower_shapes(s, f'encoder.{mod}', grid[0] * grid[1], (D, cin, p, p), D, cfg.depth, cfg.mlp_ratio)
    _fusion_shapes(s, 'encoder.', cfg)
    Dd = cfg.decoder_dim
    for mod, grid, cin in (('image', cfg.image_grid, 3), ('audio', cfg.audio_grid, 1)):
        pre = f'{mod}_decoder_'
        s[pre + 'embed.weight'] = (Dd, D); s[pre + 'embed.bias'] = (Dd,)
        s[pre + 'mask_token'] = (1, 1, Dd)
        s[pre + 'pos_embed'] = (1, grid[0] * grid[1], Dd)
        for l in range(cfg.decoder_depth):
            s.update(_block_shapes(f'{pre}blocks.{l}', Dd, int(Dd * cfg.decoder_mlp_ratio)))
            if cfg.decoder_arch == 'swin':          # models/swin.py:36-39, 158: bias table + two registered buffers
                win, shift = swin_geometry(grid, l)
                s[f'{pre}blocks.{l}.attn.relative_position_bias_table'] = ((2 * win - 1) ** 2, cfg.decoder_heads)
                s[f'{pre}blocks.{l}.attn.relative_position_index'] = (win * win, win * win)
                if shift > 0:
                    s[f'{pre}blocks.{l}.attn_mask'] = ((grid[0] // win) * (grid[1] // win), win * win, win * win)
        s[pre + 'norm.weight'] = (Dd,); s[pre + 'norm.bias'] = (Dd,)
        s[pre + 'pred.weight'] = (p * p * cin, Dd); s[pre + 'pred.bias'] = (p * p * cin,)
    return s


FROZEN = ('encoder.image.pos_embed', 'encoder.audio.pos_embed',   # models/vits.py:29
          'video.pos_embed', 'audio.pos_embed')                    # models/video_vits.py:148 (pos_trainable=False)


def is_buffer(name: str) -> bool:
    """State-dict entries that are registered buffers, not parameters (models/swin.py:39, 158)."""
    return name.endswith('.relative_position_index') or name.endswith('.attn_mask')


def closed_form_state(cfg: PathConfig, seed: int = 0) -> Dict[str, Tensor]:
    """Deterministic, platform-independent weights for fixtures: every tensor is
    drawn from a numpy MT19937 stream keyed by (seed, crc32(name)); encoder
    pos-embeds are the real sin-cos tables.  Scales keep activations O(1)."""
    import zlib
    sd: Dict[str, Tensor] = {}
    for name, shape in state_shapes(cfg).items():
        rs = np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        if is_buffer(name):
            mod, l = name.split('_decoder_')[0], int(name.split('blocks.')[1].split('.')[0])
            grid = cfg.image_grid if mod == 'image' else cfg.audio_grid
            win, shift = swin_geometry(grid, l)
            sd[name] = relative_position_index(win) if name.endswith('index') else shifted_window_mask(grid, win, shift)
            continue
        if name == 'video.pos_embed':
            arr = sincos_3d(shape[-1], cfg.video_grid)[None]
        elif name in FROZEN:
            grid = cfg.image_grid if '.image.' in name else cfg.audio_grid
            arr = sincos_2d(shape[-1], grid)[None]
        elif name.endswith('decoder_pos_embed'):
            grid = cfg.image_grid if name.startswith('image') else cfg.audio_grid
            arr = sincos_2d(shape[-1], grid)[None] + 0.02 * rs.standard_normal(shape)
        elif 'norm' in name and name.endswith('.weight'):
            arr = 1.0 + 0.1 * rs.standard_normal(shape)
        elif name.endswith('.bias'):
            arr = 0.05 * rs.standard_normal(shape)
        elif name.endswith('tokens') or name.endswith('mask_token'):
            arr = 0.5 * rs.standard_normal(shape)
        elif name.endswith('relative_position_bias_table'):
            arr = 0.5 * rs.standard_normal(shape)       # O(1) logits: the bias must matter in the fixtures
        else:
            fan_in = int(np.prod(shape[1:]))
            arr = rs.standard_normal(shape) / math.sqrt(fan_in)
        sd[name] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
    return sd


def synthetic_batch(cfg: PathConfig, B: int, seed: int = 1234):
    """Tensor contract of SURVEY §2 row 19 / §8(d): ImageNet-normalised frames
    ~ N(0,1); log10-mel in about [-7, 4]."""
    rs = np.random.RandomState(seed)
    image = rs.standard_normal((B, 3) + tuple(cfg.image_size)).astype(np.float32)
    audio = np.clip(rs.standard_normal((B, 1) + tuple(cfg.audio_size)) * 2.0 - 3.0, -7, 4).astype(np.float32)
    noise_i = rs.permutation(B * cfg.image_grid[0] * cfg.image_grid[1]).reshape(B, -1).astype(np.float32)
    noise_a = rs.permutation(B * cfg.audio_grid[0] * cfg.audio_grid[1]).reshape(B, -1).astype(np.float32)
    # tie-free noise in [0,1)
    noise_i = (noise_i + 0.5) / noise_i.size
    noise_a = (noise_a + 0.5) / noise_a.size
    return torch.from_numpy(image), torch.from_numpy(audio), noise_i, noise_a


def synthetic_video_batch(cfg: VideoConfig, B: int, seed: int = 1234):
    """Clip [B,3,T,H,W] ~ N(0,1) + log-mel [B,1,n_mels,frames] in about [-7, 4] (the ``__main__`` probe of
    models/video_earlyfusion.py:174-186 uses randn for both)."""
    rs = np.random.RandomState(seed)
    video = rs.standard_normal((B, 3) + tuple(cfg.video_size)).astype(np.float32)
    audio = np.clip(rs.standard_normal((B, 1) + tuple(cfg.audio_size)) * 2.0 - 3.0, -7, 4).astype(np.float32)
    return torch.from_numpy(video), torch.from_numpy(audio)


def structured_batch(cfg: PathConfig, B: int, seed: int):
    """Learnable synthetic AV pairs for loss-curve comparisons: each sample is a
    sum of a few low-frequency plane waves (shared phase family between the
    image and the spectrogram so the fusion path carries signal) plus small
    noise — masked patches are predictable from visible ones, so the MAE loss
    actually falls during the 1k-step curve."""
    rs = np.random.RandomState(seed)

    def waves(C, H, W, k, freqs, phase):
        yy, xx = np.meshgrid(np.arange(H, dtype=np.float32) / H, np.arange(W, dtype=np.float32) / H, indexing='ij')
        out = np.zeros((C, H, W), dtype=np.float32)
        for c in range(C):
            for i in range(k):
                fy, fx = freqs[i]
                out[c] += np.sin(2 * np.pi * (fy * yy + fx * xx) + phase[i] + 0.7 * c).astype(np.float32)
        return out / np.sqrt(k)
    image = np.zeros((B, 3) + tuple(cfg.image_size), dtype=np.float32)
    audio = np.zeros((B, 1) + tuple(cfg.audio_size), dtype=np.float32)
    for b in range(B):
        k = 3
        freqs = rs.uniform(0.5, 3.0, size=(k, 2)) * rs.choice([-1, 1], size=(k, 2))
        phase = rs.uniform(0, 2 * np.pi, size=k)
        image[b] = waves(3, *cfg.image_size, k, freqs, phase) + 0.05 * rs.standard_normal((3,) + tuple(cfg.image_size))
        audio[b] = np.clip(2.0 * waves(1, *cfg.audio_size, k, freqs, phase) - 3.0
                           + 0.1 * rs.standard_normal((1,) + tuple(cfg.audio_size)), -7, 4)
    Li = cfg.image_grid[0] * cfg.image_grid[1]
    La = cfg.audio_grid[0] * cfg.audio_grid[1]
    noise_i = ((rs.permutation(B * Li).reshape(B, Li) + 0.5) / (B * Li)).astype(np.float32)
    noise_a = ((rs.permutation(B * La).reshape(B, La) + 0.5) / (B * La)).astype(np.float32)
    return torch.from_numpy(image), torch.from_numpy(audio), noise_i, noise_a

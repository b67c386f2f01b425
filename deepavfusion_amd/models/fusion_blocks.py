"""Parameter holders for the audio-visual fusion blocks (reference models/fusion_blocks.py): the factorised block
(:216-289, default) and the two alternatives behind ``fusion_arch`` — local AV tokens (:89-145) and dense AV
interactions (:154-213)."""
import torch.nn as nn

from .vits import Mlp, _prob


class CrossAttention(nn.Module):
    """models/fusion_blocks.py:33-59."""
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.attn_drop_prob, self.proj_drop_prob = _prob(attn_drop), _prob(proj_drop)     # nn.Dropout at :42,44 (training mode only)

    def forward(self, x1, x2):
        """-> (x1', attn) as models/fusion_blocks.py:46-59; attn [B, heads, N1, N2] is returned detached."""
        from ..autograd_bridge import cross_attention
        return cross_attention(self, x1, x2)


class CrossAttention_FactorizedAVInteractions(nn.Module):
    """models/fusion_blocks.py:216-263.  q/k are dim*dim_ratio wide, v and proj stay full width, and the
    softmax scale is (dim/num_heads)^-0.5 whatever dim_ratio is (:220-222)."""
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0., dim_ratio=1., fusion_tkns=(8, 4, 4)):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.dim = int(dim * dim_ratio)
        self.fusion_tkns = tuple(fusion_tkns)
        self.attn_v = CrossAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=proj_drop)
        self.attn_a = CrossAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=proj_drop)
        self.q = nn.Linear(dim, self.dim, bias=qkv_bias)
        self.k = nn.Linear(dim * 2, self.dim, bias=qkv_bias)
        self.v = nn.Linear(dim * 2, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class FusionBlock_FactorizedAVInteractions(nn.Module):
    """models/fusion_blocks.py:266-289."""
    arch = 'factorized_mmi'
    def __init__(self, dim, num_heads, attn_ratio=0.25, mlp_ratio=4., qkv_bias=False, fusion_tkns=(8, 4, 4), drop=0.,
                 attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        # nn.Dropout(attn_drop) on every softmax of the block, nn.Dropout(drop) behind every proj and inside the Mlp (:225-233, :278)
        self.attn_drop_prob, self.proj_drop_prob = _prob(attn_drop), _prob(drop)
        self.drop_path_prob = float(drop_path)        # one DropPath module called on both branches (models/fusion_blocks.py:278)
        self.num_heads, self.fusion_tkns = num_heads, tuple(fusion_tkns)
        self.norm1_mm = norm_layer(dim)
        self.norm1_aud = norm_layer(dim)
        self.norm1_img = norm_layer(dim)
        self.attn = CrossAttention_FactorizedAVInteractions(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop,
                                                            proj_drop=drop, dim_ratio=attn_ratio, fusion_tkns=fusion_tkns)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, xmm, xv, xa):
        from ..autograd_bridge import fusion_block
        return fusion_block(self, xmm, xv, xa)


class CrossAttention_LocalAVTokens(nn.Module):
    """models/fusion_blocks.py:89-117: fusion tokens attend to cat(xv, xa); q/k/v width dim*dim_ratio,
    scale from THAT width (:93-95)."""
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0., dim_ratio=1.):
        super().__init__()
        self.num_heads = num_heads
        self.dim = int(dim * dim_ratio)
        self.scale = (self.dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, self.dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, self.dim * 2, bias=qkv_bias)
        self.proj = nn.Linear(self.dim, dim)


class CrossAttention_DenseAVInteractions(nn.Module):
    """models/fusion_blocks.py:154-188: keys/values are projections of ALL (v, a) pairs [x_i || x_j]; scale from the
    full dim (:157-158)."""
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0., dim_ratio=1.):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.dim = int(dim * dim_ratio)
        self.q = nn.Linear(dim, self.dim, bias=qkv_bias)
        self.kv = nn.Linear(dim * 2, self.dim * 2, bias=qkv_bias)
        self.proj = nn.Linear(self.dim, dim)


class _FusionBlockAlt(nn.Module):
    arch = None
    attn_cls = None

    def __init__(self, dim, num_heads, attn_ratio=0.25, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.attn_drop_prob, self.proj_drop_prob = _prob(attn_drop), _prob(drop)      # models/fusion_blocks.py:99,101,133 / :164,166,202
        self.drop_path_prob = float(drop_path)
        self.num_heads = num_heads
        self.norm1_mm = norm_layer(dim)
        self.norm1_aud = norm_layer(dim)
        self.norm1_img = norm_layer(dim)
        self.attn = self.attn_cls(dim, num_heads=num_heads, qkv_bias=qkv_bias, dim_ratio=attn_ratio)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class FusionBlock_LocalAVTokens(_FusionBlockAlt):
    """models/fusion_blocks.py:120-145.  ``forward(xmm, xa, xv)`` keeps the reference's parameter ORDER: the encoder
    calls it positionally with (x_fusion, x_image, x_audio), so the second argument (named xa) receives the image
    tokens."""
    arch = 'token'
    attn_cls = CrossAttention_LocalAVTokens

    def forward(self, xmm, xa, xv):
        from ..autograd_bridge import fusion_block
        return fusion_block(self, xmm, xa, xv)


class FusionBlock_DenseAVInteractions(_FusionBlockAlt):
    """models/fusion_blocks.py:191-213."""
    arch = 'dense_mmi'
    attn_cls = CrossAttention_DenseAVInteractions

    def forward(self, xmm, xv, xa):
        from ..autograd_bridge import fusion_block
        return fusion_block(self, xmm, xv, xa)

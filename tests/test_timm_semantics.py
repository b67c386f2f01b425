"""CPU: third-party anchor for the timm 0.9.2 half of the arithmetic (DESIGN §6 "unpinned part").

timm is not in the image and the reference vendors none of it, so `Block / Attention / Mlp / PatchEmbed` and the Swin
helpers are restated twice here: the stand-in the fixtures were generated with (tests/golden/ref_shim/timm) and the oracle
(oracle/avmae_oracle.py `timm_*`).  A misreading shared by both would pass every fixture test.  The HuggingFace `transformers`
package IS in the image and carries independent ports of the same published models (ViT-MAE, Swin); its MAE conversion
convention (fused `qkv` rows = [q; k; v]) is how timm-layout checkpoints — the ones models/vits.py:80 strict-loads — are
read.  These tests load identical weights into that implementation and require the same numbers from the stand-in and from
the oracle, forward and backward.  Nothing here touches the GPU or /root/reference."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import avmae_oracle as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_shim'))
timm_vt = pytest.importorskip('timm.models.vision_transformer')
timm_layers = pytest.importorskip('timm.models.layers')
timm_swin = pytest.importorskip('timm.models.swin_transformer')
sys.path.pop(0)
if 'ref_shim' not in (getattr(timm_vt, '__file__', '') or ''):
    pytest.skip('a real timm is installed: the stand-in is not what the fixtures used', allow_module_level=True)
hf_mae = pytest.importorskip('transformers.models.vit_mae.modeling_vit_mae')
hf_swin = pytest.importorskip('transformers.models.swin.modeling_swin')
from transformers import ViTMAEConfig  # noqa: E402

D, H, HID, EPS = 128, 2, 512, 1e-6


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _hf_layer_with(block):
    """A transformers ViTMAELayer holding ``block``'s weights (fused qkv split the way the MAE conversion does)."""
    cfg = ViTMAEConfig(hidden_size=D, num_attention_heads=H, intermediate_size=HID, num_hidden_layers=1, layer_norm_eps=EPS,
                       hidden_act='gelu', qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                       attn_implementation='eager')
    layer = hf_mae.ViTMAELayer(cfg).eval()
    sd = block.state_dict()
    qw, kw, vw = sd['attn.qkv.weight'].split(D, 0)
    qb, kb, vb = sd['attn.qkv.bias'].split(D, 0)
    layer.load_state_dict({
        'attention.q_proj.weight': qw, 'attention.q_proj.bias': qb, 'attention.k_proj.weight': kw, 'attention.k_proj.bias': kb,
        'attention.v_proj.weight': vw, 'attention.v_proj.bias': vb,
        'attention.o_proj.weight': sd['attn.proj.weight'], 'attention.o_proj.bias': sd['attn.proj.bias'],
        'layernorm_before.weight': sd['norm1.weight'], 'layernorm_before.bias': sd['norm1.bias'],
        'layernorm_after.weight': sd['norm2.weight'], 'layernorm_after.bias': sd['norm2.bias'],
        'mlp.fc1.weight': sd['mlp.fc1.weight'], 'mlp.fc1.bias': sd['mlp.fc1.bias'],
        'mlp.fc2.weight': sd['mlp.fc2.weight'], 'mlp.fc2.bias': sd['mlp.fc2.bias']}, strict=True)
    return layer


def _random_block(seed):
    torch.manual_seed(seed)
    from functools import partial
    blk = timm_vt.Block(D, H, 4.0, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=EPS)).eval()
    with torch.no_grad():
        for p in blk.parameters():                       # biases and LN affine away from their 0 / 1 defaults
            p.copy_(torch.randn_like(p) * (0.5 if p.ndim == 1 else 0.08))
    return blk


@pytest.mark.parametrize('tokens', [49, 63, 5])
def test_block_matches_the_transformers_vit_mae_layer(tokens):
    """Block = x + attn(LN(x)); x + mlp(LN(x)); fused qkv reshaped [B,N,3,H,hd]; scale hd^-0.5; erf GELU; no LayerScale."""
    blk = _random_block(tokens)
    layer = _hf_layer_with(blk)
    sd = {'b.' + k: v for k, v in blk.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    x0 = torch.randn(3, tokens, D, generator=g)
    w = torch.randn(3, tokens, D, generator=g)
    outs, grads = [], []
    for f in (lambda x: layer(x), lambda x: blk(x), lambda x: O.timm_block(x, sd, 'b', H, EPS)):
        x = x0.clone().requires_grad_(True)
        y = f(x)
        y = y[0] if isinstance(y, tuple) else y
        (y * w).sum().backward()
        outs.append(y.detach()), grads.append(x.grad.clone())
    for i, who in ((1, 'stand-in'), (2, 'oracle')):
        assert _rel(outs[i], outs[0]) < 1e-5, who
        assert _rel(grads[i], grads[0]) < 1e-5, who


def test_fused_qkv_row_order_is_q_then_k_then_v():
    """Permuting the three row blocks of the fused weight must change the result (the comparison above is sensitive to the
    layout it claims to pin), and heads are the SLOWER index inside each block (row = which*D + head*hd + d)."""
    blk = _random_block(1)
    layer = _hf_layer_with(blk)
    x = torch.randn(2, 9, D, generator=torch.Generator().manual_seed(3))
    base = layer(x)
    base = base[0] if isinstance(base, tuple) else base
    with torch.no_grad():
        w = blk.attn.qkv.weight
        q, k, v = w.split(D, 0)
        blk.attn.qkv.weight.copy_(torch.cat([k, q, v], 0))
        b = blk.attn.qkv.bias
        qb, kb, vb = b.split(D, 0)
        blk.attn.qkv.bias.copy_(torch.cat([kb, qb, vb], 0))
    assert _rel(blk(x), base) > 1e-3
    # head-major inside a block: oracle attention on one head's slice equals the fused result's slice
    blk = _random_block(2)
    sd = {'a.' + k[5:]: v for k, v in blk.state_dict().items() if k.startswith('attn.')}
    hd = D // H
    full = O.timm_attention(x, sd, 'a', H)
    qkv = torch.nn.functional.linear(x, sd['a.qkv.weight'], sd['a.qkv.bias'])
    heads = []
    for h in range(H):
        q, k, v = (qkv[..., i * D + h * hd: i * D + (h + 1) * hd] for i in range(3))
        heads.append(torch.softmax(q @ k.transpose(1, 2) * hd ** -0.5, -1) @ v)
    ref = torch.nn.functional.linear(torch.cat(heads, -1), sd['a.proj.weight'], sd['a.proj.bias'])
    assert _rel(full, ref) < 1e-5
    assert _rel(blk.attn(x), ref) < 1e-5


def test_patch_embed_token_and_feature_order():
    """Conv2d(kernel = stride = patch) -> flatten(2).transpose(1,2): token = gy*gW + gx, feature = (c, py, px)."""
    cfg = ViTMAEConfig(hidden_size=D, image_size=(32, 48), patch_size=16, num_channels=3)
    hf = hf_mae.ViTMAEPatchEmbeddings(cfg).eval()
    pe = timm_layers.PatchEmbed((32, 48), 16, 3, D).eval()
    torch.manual_seed(0)
    with torch.no_grad():
        for p in hf.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    pe.load_state_dict({'proj.weight': hf.projection.weight, 'proj.bias': hf.projection.bias})
    x = torch.randn(2, 3, 32, 48)
    want = hf(x)
    assert pe.grid_size == (2, 3) and pe.num_patches == 6 and want.shape == (2, 6, D)
    assert _rel(pe(x), want) < 1e-5
    sd = {'p.proj.weight': hf.projection.weight.detach(), 'p.proj.bias': hf.projection.bias.detach()}
    assert _rel(O.patch_embed(x, sd, 'p', 16), want) < 1e-5
    with pytest.raises(AssertionError):
        pe(torch.randn(2, 3, 32, 32))                        # exact input size is asserted


def test_mlp_uses_the_erf_gelu():
    import math
    m = timm_layers.Mlp(8, 16).eval()
    x = torch.linspace(-4, 4, 64).reshape(8, 8)
    h = torch.nn.functional.linear(x, m.fc1.weight, m.fc1.bias)
    erf = torch.tensor([[0.5 * v * (1 + math.erf(v / math.sqrt(2))) for v in row] for row in h.tolist()])
    want = torch.nn.functional.linear(erf, m.fc2.weight, m.fc2.bias)
    assert _rel(m(x), want) < 1e-6
    sd = {'m.' + k: v.detach() for k, v in m.state_dict().items()}
    assert _rel(O.timm_mlp(x, sd, 'm'), want) < 1e-6
    tanh = torch.nn.functional.gelu(h, approximate='tanh')
    assert _rel(torch.nn.functional.linear(tanh, m.fc2.weight, m.fc2.bias), want) > 1e-5     # and the test can tell them apart


@pytest.mark.parametrize('win', [4, 7])
def test_swin_helpers_match_the_transformers_port(win):
    """window_partition / window_reverse / relative position index of timm.models.swin_transformer (used by the reference's
    models/swin.py:8) against transformers' Swin, and against the oracle's closed forms."""
    x = torch.randn(2, 2 * win, 3 * win, 5)
    a, b = timm_swin.window_partition(x, win), hf_swin.window_partition(x, win)
    assert torch.equal(a, b)
    assert torch.equal(timm_swin.window_reverse(a, win, 2 * win, 3 * win), x)
    assert torch.equal(hf_swin.window_reverse(b, win, 2 * win, 3 * win), x)
    idx = timm_swin.get_relative_position_index(win, win)
    hf_idx = hf_swin.SwinRelativePositionBias(2, (win, win)).relative_position_index.view(win * win, win * win)
    assert torch.equal(idx, hf_idx)
    assert torch.equal(O.relative_position_index(win).long(), hf_idx.long())
    # the oracle's row map of an un-shifted grid is the window partition of the token ids
    ids = torch.arange(2 * win * 3 * win).view(1, 2 * win, 3 * win, 1)
    rows = O.window_rows((2 * win, 3 * win), win, 0)
    assert torch.equal(rows.reshape(-1).long(), hf_swin.window_partition(ids, win).reshape(-1))


def test_drop_path_and_param_groups_follow_appendix_b():
    dp = timm_layers.DropPath(0.25).train()
    torch.manual_seed(0)
    x = torch.ones(4000, 3, 2)
    y = dp(x)
    per_sample = y[:, 0, 0]
    assert set(np.round(per_sample.unique().tolist(), 5)) == {0.0, round(1 / 0.75, 5)}      # keep -> 1/keep_prob, drop -> 0
    assert torch.equal(y, per_sample.view(-1, 1, 1).expand_as(y))                            # one draw per sample
    assert abs(float((per_sample > 0).float().mean()) - 0.75) < 0.03
    assert torch.equal(dp.eval()(x), x)
    from timm.optim.optim_factory import param_groups_weight_decay
    m = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.LayerNorm(4))
    m.register_parameter('pos_embed', torch.nn.Parameter(torch.zeros(1, 3, 4)))
    m.register_parameter('frozen', torch.nn.Parameter(torch.zeros(1, 3, 4), requires_grad=False))
    no_decay, decay = param_groups_weight_decay(m, 0.05, ('pos_embed',))
    assert no_decay['weight_decay'] == 0. and decay['weight_decay'] == 0.05
    assert [tuple(p.shape) for p in decay['params']] == [(4, 4)]
    assert sorted(tuple(p.shape) for p in no_decay['params']) == [(1, 3, 4), (4,), (4,), (4,)]

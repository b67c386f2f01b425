"""CPU: pin the oracle (oracle/) against fixtures produced by the reference itself
(tests/golden/gen_golden.py imported /root/reference in the build container)."""
import numpy as np
import pytest
import torch

from oracle import avmae_oracle as O
from oracle import train_oracle as T
from oracle.configs import CONFIGS

RTOL = 1e-5   # fp32 oracle vs fp32 reference (fused-SDPA vs explicit softmax floor ~2e-6)


def close(a, b, rtol=RTOL, atol=None):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, f'rel-max err {err:.3e} > {rtol}'


def test_random_masking_bit_exact(golden):
    g = golden('masking')
    tags = sorted({k.split('.')[0] for k in g.files})
    assert 'aud320' in tags
    for t in tags:
        ids_keep, mask, ids_restore = O.random_masking_from_noise(g[f'{t}.noise'], float(g[f'{t}.ratio']))
        assert np.array_equal(ids_keep, g[f'{t}.ids_keep']), t
        assert np.array_equal(ids_restore, g[f'{t}.ids_restore']), t
        assert np.array_equal(mask, g[f'{t}.mask']), t
    assert g['aud320.ids_keep'].shape[1] == 63      # int(320*(1-0.8)) quirk
    assert g['aud320_75.ids_keep'].shape[1] == 80


def test_sincos_tables(golden):
    g = golden('posembed')
    for k in g.files:
        kind, dim, grid = k.split('_')[:3]
        grid = tuple(int(x) for x in grid.split('x'))
        full = (O.sincos_2d if kind == '2d' else O.sincos_3d)(int(dim), grid).astype(np.float32)
        if k.endswith('_sub'):
            step = (3, 5) if kind == '2d' else (7, 5)
            assert np.allclose(full[::step[0], ::step[1]], g[k], atol=1e-6)
        elif k.endswith('_sum'):
            assert abs(full.astype(np.float64).sum() - float(g[k])) < 1e-3
        else:
            assert np.allclose(full, g[k], atol=1e-6)


def _micro():
    cfg = CONFIGS['micro']
    return cfg, O.closed_form_state(cfg, 0)


def _grads(y, gout, inputs, params):
    (y * torch.from_numpy(gout)).sum().backward()
    return [i.grad for i in inputs], {k: p.grad for k, p in params.items()}


@pytest.mark.parametrize('tag', ['cross_attention', 'factorized_attention', 'fusion_block', 'timm_block', 'decoder_block'])
def test_module_fixtures(golden, tag):
    g = golden('ops_micro')
    cfg, sd0 = _micro()
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    ins = [torch.from_numpy(g[f'{tag}.in{i}']).requires_grad_(True) for i in range(3) if f'{tag}.in{i}' in g.files]
    H = cfg.fusion_num_heads
    if tag == 'cross_attention':
        pre = 'encoder.fusion_blocks.0.attn.attn_v'
        y = O.cross_attention(ins[0], ins[1], sd, pre, H)
    elif tag == 'factorized_attention':
        pre = 'encoder.fusion_blocks.0.attn'
        y = O.factorized_attention(*ins, sd, pre, H, cfg.fusion_tkns)
    elif tag == 'fusion_block':
        pre = 'encoder.fusion_blocks.0'
        y = O.fusion_block_factorized(*ins, sd, pre, H, cfg.fusion_tkns, cfg.fus_eps)
    elif tag == 'timm_block':
        pre = 'encoder.image.blocks.1'
        y = O.timm_block(ins[0], sd, pre, cfg.num_heads, cfg.enc_eps)
    else:
        pre = 'image_decoder_blocks.0'
        y = O.timm_block(ins[0], sd, pre, cfg.decoder_heads, cfg.dec_eps)
    close(y.detach().numpy(), g[f'{tag}.out'])
    (y * torch.from_numpy(g[f'{tag}.gout'])).sum().backward()
    for i, x in enumerate(ins):
        close(x.grad.numpy(), g[f'{tag}.gin{i}'], rtol=5e-5)
    n_w = 0
    for k in g.files:
        if k.startswith(f'{tag}.gw.'):
            close(sd[f'{pre}.{k[len(tag) + 4:]}'].grad.numpy(), g[k], rtol=5e-5)
            n_w += 1
    assert n_w >= 6


def test_prepare_patch_tokens_and_encoder(golden):
    g = golden('ops_micro')
    cfg, sd0 = _micro()
    sd = {k: v.clone().requires_grad_(k not in O.FROZEN) for k, v in sd0.items()}
    img = torch.from_numpy(g['prepare.image'])
    ids = torch.from_numpy(g['prepare.ids_keep'])
    tok = O.prepare_patch_tokens(img, sd, 'encoder.image', cfg.patch, ids)
    close(tok.detach().numpy(), g['prepare.out'])
    (tok * torch.from_numpy(g['prepare.gout'])).sum().backward()
    close(sd['encoder.image.patch_embed.proj.weight'].grad.numpy(), g['prepare.gw.weight'], rtol=5e-5)
    close(sd['encoder.image.patch_embed.proj.bias'].grad.numpy(), g['prepare.gw.bias'], rtol=5e-5)
    aud = torch.from_numpy(g['encoder.audio'])
    ids_a = torch.from_numpy(g['encoder.ids_keep_audio'])
    with torch.no_grad():
        xi, xa, xf, embs = O.deepavfusion_forward(sd0, cfg, img, aud, ids, ids_a, prefix='encoder.', return_embs=True)
        close(xi.numpy(), g['encoder.x_image'], rtol=5e-5)
        close(xa.numpy(), g['encoder.x_audio'], rtol=5e-5)
        close(xf.numpy(), g['encoder.x_fusion'], rtol=5e-5)
        close(embs[0][2].numpy(), g['encoder.emb0_fusion'], rtol=5e-5)
        close(embs[0][0].numpy(), g['encoder.emb0_image'], rtol=5e-5)
        xi, xa, xf = O.deepavfusion_forward(sd0, cfg, img, aud, prefix='encoder.')
        close(xi.numpy(), g['encoder_full.x_image'], rtol=5e-5)
        close(xa.numpy(), g['encoder_full.x_audio'], rtol=5e-5)
        close(xf.numpy(), g['encoder_full.x_fusion'], rtol=5e-5)


def test_forward_decoder(golden):
    g = golden('ops_micro')
    cfg, sd0 = _micro()
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    x = torch.from_numpy(g['decoder.x']).requires_grad_(True)
    xf = torch.from_numpy(g['decoder.x_fusion']).requires_grad_(True)
    pred = O.forward_decoder(x, xf, torch.from_numpy(g['decoder.ids_restore']), sd, cfg, 'image')
    close(pred.detach().numpy(), g['decoder.out'])
    (pred * torch.from_numpy(g['decoder.gout'])).sum().backward()
    close(x.grad.numpy(), g['decoder.gx'], rtol=5e-5)
    close(xf.grad.numpy(), g['decoder.gx_fusion'], rtol=5e-5)
    close(sd['image_decoder_mask_token'].grad.numpy(), g['decoder.gw.mask_token'], rtol=5e-5)
    close(sd['image_decoder_pos_embed'].grad.numpy(), g['decoder.gw.pos_embed'], rtol=5e-5)
    close(sd['image_decoder_embed.weight'].grad.numpy(), g['decoder.gw.embed.weight'], rtol=5e-5)
    close(sd['image_decoder_pred.bias'].grad.numpy(), g['decoder.gw.pred.bias'], rtol=5e-5)


def test_patchify_and_loss(golden):
    g = golden('ops_micro')
    for mod, key in (('image', 'prepare.image'), ('audio', 'encoder.audio')):
        x = torch.from_numpy(g[key])
        target = O.patchify(x, (16, 16))
        assert np.array_equal(target.numpy(), g[f'patchify.{mod}'])       # pure data movement: exact
        for norm in (1, 0):
            p = torch.from_numpy(g[f'loss.{mod}.{norm}.pred']).requires_grad_(True)
            loss = O.forward_loss(target, p, torch.from_numpy(g[f'loss.{mod}.{norm}.mask']), bool(norm))
            loss.backward()
            close(loss.detach().numpy(), g[f'loss.{mod}.{norm}.loss'])
            close(p.grad.numpy(), g[f'loss.{mod}.{norm}.gpred'])


@pytest.mark.parametrize('name', ['micro', 'tiny', 'micro_token', 'micro_dense', 'micro_swin'])
def test_end_to_end(golden, name):
    g = golden(f'e2e_{name}')
    cfg = CONFIGS[name]
    sd = {k: v.clone().requires_grad_(k not in O.FROZEN and not O.is_buffer(k)) for k, v in O.closed_form_state(cfg, 0).items()}
    image, audio, ni, na = O.synthetic_batch(cfg, int(g['B']), seed=int(g['seed']))
    li, la, pi, pa, aux = O.avmae_forward(sd, cfg, image, audio, ni, na)
    close(li.detach().numpy(), g['loss_image'])
    close(la.detach().numpy(), g['loss_audio'])
    if 'pred_image' in g.files:
        close(pi.detach().numpy(), g['pred_image'], rtol=5e-5)
        close(pa.detach().numpy(), g['pred_audio'], rtol=5e-5)
    else:
        close(pi.detach().numpy()[:, ::3, ::7], g['pred_image_sub'], rtol=5e-5)
        close(pa.detach().numpy()[:, ::3, ::7], g['pred_audio_sub'], rtol=5e-5)
    (li + la).backward()
    norms = dict(zip(g['grad_names'].tolist(), g['grad_norms'].tolist()))
    assert set(norms) == {k for k, v in sd.items() if v.requires_grad}       # every trainable param gets a grad (DDP find_unused=False)
    for k, ref in norms.items():
        got = float(sd[k].grad.double().norm())
        assert abs(got - ref) <= 2e-4 * max(ref, 1e-6) + 1e-7, (k, got, ref)
    for k in g.files:
        if k.startswith('grad.'):
            close(sd[k[5:]].grad.numpy(), g[k], rtol=1e-4)
    total = T.global_grad_norm([v.grad for v in sd.values() if v.grad is not None])
    assert abs(total - float(g['grad_norm_total'])) < 1e-4 * float(g['grad_norm_total'])


def _sampled(gradient, g):
    """the compact form of tests/golden/gen_golden.py::_compact_grads: small gradients in full, a strided sample of large ones"""
    flat = gradient.detach().reshape(-1)
    return (flat if flat.numel() <= int(g['grad_full_max']) else flat[::int(g['grad_stride'])]).numpy()


@pytest.mark.parametrize('name', ['base', 'base_as', 'large'])
def test_published_configs_at_full_width(golden, name):
    """BASELINE.json configs[1] / [2] / [3] at their published widths and depths (ViT-B, ViT-B with the AudioSet fusion widths, ViT-L),
    batch 2 / 2 / 1: the oracle against what the imported reference computed (losses, prediction samples, every gradient's norm,
    sampled gradients) — the pin of the restatement at the sizes the GPU numbers are published for."""
    g = golden(f'e2e_{name}')
    cfg = CONFIGS[name]
    sd = {k: v.clone().requires_grad_(k not in O.FROZEN and not O.is_buffer(k)) for k, v in O.closed_form_state(cfg, 0).items()}
    image, audio, ni, na = O.synthetic_batch(cfg, int(g['B']), seed=int(g['seed']))
    li, la, pi, pa, aux = O.avmae_forward(sd, cfg, image, audio, ni, na)
    close(li.detach().numpy(), g['loss_image'])
    close(la.detach().numpy(), g['loss_audio'])
    close(pi.detach().numpy()[:, ::5, ::11], g['pred_image_sub'], rtol=1e-4)
    close(pa.detach().numpy()[:, ::5, ::11], g['pred_audio_sub'], rtol=1e-4)
    (li + la).backward()
    norms = dict(zip(g['grad_names'].tolist(), g['grad_norms'].tolist()))
    assert set(norms) == {k for k, v in sd.items() if v.requires_grad}
    for k, ref in norms.items():
        got = float(sd[k].grad.double().norm())
        assert abs(got - ref) <= 5e-4 * max(ref, 1e-6) + 1e-6, (k, got, ref)
        a, b = _sampled(sd[k].grad, g).astype(np.float64), g['grad.' + k].astype(np.float64)
        # (exactly-zero gradients — the key biases, softmax shift invariance — are rounding noise on both sides: absolute floor)
        assert np.abs(a - b).max() <= 5e-4 * np.abs(b).max() + 1e-7 * float(g['grad_norm_total']), (k, np.abs(a - b).max(), np.abs(b).max())
    total = T.global_grad_norm([v.grad for v in sd.values() if v.grad is not None])
    assert abs(total - float(g['grad_norm_total'])) < 1e-4 * float(g['grad_norm_total'])


def test_video_base_at_full_width(golden):
    """BASELINE.json configs[4] (ViT-B video early fusion, 8 x 224 x 224 clip + 3 s of audio) at B = 1 through the reference."""
    g = golden('e2e_video_base')
    cfg = CONFIGS['video_base']
    sd = {k: v.clone().requires_grad_(k not in O.FROZEN) for k, v in O.closed_form_state(cfg, 0).items()}
    video, audio = O.synthetic_video_batch(cfg, int(g['B']), seed=int(g['seed']))
    xv, xa, xf = O.video_earlyfusion_forward(sd, cfg, video, audio)
    for got, key in ((xv, 'x_video_sub'), (xa, 'x_audio_sub'), (xf, 'x_fusion_sub')):
        close(got.detach().numpy()[:, ::3, ::7], g[key], rtol=1e-4)
    w = probe_weights([xv.shape, xa.shape, xf.shape], int(g['seed']) + 1)
    loss = (xv * w[0]).sum() + (xa * w[1]).sum() + (xf * w[2]).sum()
    assert abs(float(loss) - float(g['loss_probe'])) < 2e-4 * abs(float(g['loss_probe']))
    loss.backward()
    norms = dict(zip(g['grad_names'].tolist(), g['grad_norms'].tolist()))
    assert set(norms) == {k for k, v in sd.items() if v.requires_grad}
    gtot = float(np.sqrt(sum(v * v for v in norms.values())))
    for k, ref in norms.items():
        got = float(sd[k].grad.double().norm())
        assert abs(got - ref) <= 5e-4 * max(ref, 1e-6) + 1e-5, (k, got, ref)
        a, b = _sampled(sd[k].grad, g).astype(np.float64), g['grad.' + k].astype(np.float64)
        assert np.abs(a - b).max() <= 5e-4 * np.abs(b).max() + 1e-7 * gtot, (k, np.abs(a - b).max(), np.abs(b).max())


def probe_weights(shapes, seed):
    rs = np.random.RandomState(seed)
    return [torch.from_numpy(rs.standard_normal(tuple(s)).astype(np.float32)) for s in shapes]


def test_video_earlyfusion(golden):
    """BASELINE configs[4] family: models/video_earlyfusion.py:95-131 through the reference."""
    g = golden('e2e_video_micro')
    cfg = CONFIGS['video_micro']
    assert np.allclose(O.sincos_3d(cfg.embed_dim, cfg.video_grid), g['video_pos_embed_init'][0], atol=1e-6)   # H != W grid
    sd = {k: v.clone().requires_grad_(k not in O.FROZEN) for k, v in O.closed_form_state(cfg, 0).items()}
    video, audio = O.synthetic_video_batch(cfg, int(g['B']), seed=int(g['seed']))
    xv, xa, xf, embs = O.video_earlyfusion_forward(sd, cfg, video, audio, return_embs=True)
    close(xv.detach().numpy(), g['x_video'], rtol=5e-5)
    close(xa.detach().numpy(), g['x_audio'], rtol=5e-5)
    close(xf.detach().numpy(), g['x_fusion'], rtol=5e-5)
    close(embs[-1][0].detach().numpy()[:, ::2, ::5], g['emb_last_video_sub'], rtol=5e-5)
    close(embs[0][2].detach().numpy(), g['emb_first_fusion'], rtol=5e-5)
    assert abs(float(xv.sum() + xa.sum() + xf.sum()) - float(g['loss_sum'])) < 1e-3
    w = probe_weights([xv.shape, xa.shape, xf.shape], int(g['seed']) + 1)
    loss = (xv * w[0]).sum() + (xa * w[1]).sum() + (xf * w[2]).sum()
    assert abs(float(loss) - float(g['loss_probe'])) < 1e-4 * abs(float(g['loss_probe']))
    loss.backward()
    norms = dict(zip(g['grad_names'].tolist(), g['grad_norms'].tolist()))
    assert set(norms) == {k for k, v in sd.items() if v.requires_grad}
    for k, ref in norms.items():
        got = float(sd[k].grad.double().norm())
        assert abs(got - ref) <= 2e-4 * max(ref, 1e-6) + 1e-6, (k, got, ref)
    for k in g.files:
        if k.startswith('grad.'):
            close(sd[k[5:]].grad.numpy(), g[k], rtol=1e-4)


def drop_scales_from_fixture(g, depth):
    """masks are stored in the reference's call order: per layer visual (attn, mlp), audio (attn, mlp), fusion (attn, mlp)."""
    keep = 1.0 - float(g['p'])
    m = torch.from_numpy(g['masks']) / keep
    drop = {}
    for l in range(depth):
        for j, tag in enumerate(('visual', 'audio', 'fusion')):
            drop[f'{tag}.{l}'] = (m[6 * l + 2 * j], m[6 * l + 2 * j + 1])
    return drop


def test_drop_path_training_mode(golden):
    """DeepAVFusion(drop_path=0.25).train() through the reference with injected DropPath masks (fine-tuning setting)."""
    g = golden('droppath_micro')
    cfg = CONFIGS['micro']
    full = O.closed_form_state(cfg, 0)
    sd = {k[len('encoder.'):]: v.clone().requires_grad_(k not in O.FROZEN) for k, v in full.items() if k.startswith('encoder.')}
    image, audio, ni, na = O.synthetic_batch(cfg, int(g['B']), seed=int(g['seed']))
    ik = torch.from_numpy(O.random_masking_from_noise(ni, cfg.image_mask_ratio)[0])
    ak = torch.from_numpy(O.random_masking_from_noise(na, cfg.audio_mask_ratio)[0])
    xi, xa, xf = O.deepavfusion_forward(sd, cfg, image, audio, ik, ak, drop=drop_scales_from_fixture(g, cfg.depth))
    close(xi.detach().numpy(), g['x_image'], rtol=5e-5)
    close(xa.detach().numpy(), g['x_audio'], rtol=5e-5)
    close(xf.detach().numpy(), g['x_fusion'], rtol=5e-5)
    w = probe_weights([xi.shape, xa.shape, xf.shape], int(g['seed']) + 1)
    loss = (xi * w[0]).sum() + (xa * w[1]).sum() + (xf * w[2]).sum()
    assert abs(float(loss) - float(g['loss_probe'])) < 1e-4 * abs(float(g['loss_probe']))
    loss.backward()
    norms = dict(zip(g['grad_names'].tolist(), g['grad_norms'].tolist()))
    for k, ref in norms.items():
        got = float(sd[k].grad.double().norm())
        assert abs(got - ref) <= 2e-4 * max(ref, 1e-6) + 1e-6, (k, got, ref)
    for k in g.files:
        if k.startswith('grad.'):
            close(sd[k[5:]].grad.numpy(), g[k], rtol=1e-4)
    # without the scales the result is a different one (the masks do drop samples)
    xi0 = O.deepavfusion_forward(sd, cfg, image, audio, ik, ak)[0]
    assert float((xi0 - xi).abs().max()) > 1e-2


def dropout_from_fixture(g, calls=None):
    """oracle ``dropout(name, x)``: the closed-form masks the fixture generator injected into the reference's nn.Dropout calls."""
    seed, pa, pp = int(g['seed']), float(g['p_attn']), float(g['p_proj'])

    def dropout(name, x):
        p = pa if name.endswith('attn') else pp
        if calls is not None:
            calls.append((name, tuple(x.shape)))
        m = torch.from_numpy(O.dropout_keep_mask(seed, name, x.shape, p)).to(device=x.device, dtype=x.dtype)
        return x * m / (1.0 - p)
    return dropout


@pytest.mark.parametrize('name', ['micro', 'micro_token', 'micro_dense'])
def test_dropout_training_mode(golden, name):
    """DeepAVFusion(attn_drop=0.2, drop=0.1).train() through the reference with every nn.Dropout draw injected (the constructor
    surface of eval_finetune.py:170-171): attention dropout on the probabilities, dropout behind every proj and inside the Mlps, for
    the three fusion-block architectures; the oracle makes the reference's Dropout calls, with the same shapes, in the same order."""
    g = golden(f'dropout_{name}')
    cfg = CONFIGS[name]
    full = O.closed_form_state(cfg, 0)
    sd = {k[len('encoder.'):]: v.clone().requires_grad_(k not in O.FROZEN) for k, v in full.items() if k.startswith('encoder.')}
    image, audio, ni, na = O.synthetic_batch(cfg, int(g['B']), seed=int(g['seed']))
    ik = torch.from_numpy(O.random_masking_from_noise(ni, cfg.image_mask_ratio)[0])
    ak = torch.from_numpy(O.random_masking_from_noise(na, cfg.audio_mask_ratio)[0])
    calls = []
    xi, xa, xf = O.deepavfusion_forward(sd, cfg, image, audio, ik, ak, dropout=dropout_from_fixture(g, calls))
    assert [c[0] for c in calls] == g['site_names'].tolist()
    assert [list(c[1]) for c in calls] == [[d for d in row if d] for row in g['site_shapes'].tolist()]
    close(xi.detach().numpy(), g['x_image'], rtol=5e-5)
    close(xa.detach().numpy(), g['x_audio'], rtol=5e-5)
    close(xf.detach().numpy(), g['x_fusion'], rtol=5e-5)
    w = probe_weights([xi.shape, xa.shape, xf.shape], int(g['seed']) + 1)
    loss = (xi * w[0]).sum() + (xa * w[1]).sum() + (xf * w[2]).sum()
    assert abs(float(loss) - float(g['loss_probe'])) < 1e-4 * abs(float(g['loss_probe']))
    loss.backward()
    norms = dict(zip(g['grad_names'].tolist(), g['grad_norms'].tolist()))
    for k, ref in norms.items():
        got = float(sd[k].grad.double().norm())
        assert abs(got - ref) <= 2e-4 * max(ref, 1e-6) + 1e-6, (k, got, ref)
    for k in g.files:
        if k.startswith('grad.'):
            close(sd[k[5:]].grad.numpy(), g[k], rtol=1e-4)
    xi0 = O.deepavfusion_forward(sd, cfg, image, audio, ik, ak)[0]          # eval mode: a different result
    assert float((xi0 - xi).abs().max()) > 1e-2


def test_lr_schedule_and_param_groups(golden):
    g = golden('lr_groups')
    cfg = CONFIGS['micro']
    shapes = O.state_shapes(cfg)
    groups = T.param_groups(shapes, 0.05, image_pt='', audio_pt='')
    assert len(groups) == int(g['n_groups']) == 6
    for gi, grp in enumerate(groups):
        assert sorted(grp['names']) == sorted(g[f'group{gi}.names'].tolist()), gi
        assert grp['weight_decay'] == float(g[f'group{gi}.weight_decay'])
        assert grp['pretrained'] == bool(g[f'group{gi}.pretrained'])
    for e, row in zip(g['epochs'], g['lr_table']):
        cur, cur_pt = T.lr_at(float(e), 1e-3, 2, 10, pt_warmup_epochs=10 / 2)
        assert abs(cur - row[0]) < 1e-12
        for gi, grp in enumerate(groups):
            assert abs((cur_pt if grp['pretrained'] else cur) - row[1 + gi]) < 1e-12


def test_trainer_step_semantics(golden):
    g = golden('trainer_steps')
    cfg = CONFIGS['micro']
    tr = T.OracleTrainer(cfg, O.closed_form_state(cfg, 0), lr=1e-3, accum_iter=2)
    for step in range(6):
        if step % 2 == 0:
            lr = tr.set_lr(step / 6 * 4, 1e-3, 1, 4, pt_warmup_epochs=4 / 2)
            assert abs(lr - g['lr'][step // 2]) < 1e-12
        image, audio, ni, na = O.synthetic_batch(cfg, 2, seed=300 + step)
        li, la = tr.forward(image, audio, ni, na)[:2]
        gn = tr.step(li + la)
        assert abs(float(li + la) - g['loss'][step]) < 2e-5 * g['loss'][step], step
        assert abs(gn - g['grad_norm'][step]) < 2e-4 * g['grad_norm'][step], step
    assert tr.n_steps == int(g['n_steps']) == 3
    sums = dict(zip(g['param_names'].tolist(), g['param_sums'].tolist()))
    for k, ref in sums.items():
        # key biases have an exactly-zero true gradient (softmax shift invariance): Adam turns their
        # rounding noise into +-lr steps, so their checksums are not reproducible between two fp32 runs
        if k.endswith(('qkv.bias', 'kv.bias', '.k.bias')):
            continue
        got = float(tr.sd[k].detach().double().sum())
        assert abs(got - ref) < 1e-4 * max(abs(ref), 1.0), k


def test_audio_frontend_oracle_stft_pinned_and_filterbank_properties():
    """oracle/audio_oracle.py: the STFT half equals the defining DFT sum (torch.stft's centre / reflect / periodic-Hann /
    one-sided conventions), the HTK filterbank is non-negative, triangular, peaks at 1 between mel-equidistant edges, and
    the output has the reference's shape (128, 64 * dur) after the [:, :, :-1] of datasets.py:242."""
    import torch

    from oracle import audio_oracle as A
    torch.manual_seed(0)
    w = torch.randn(2, 5000, dtype=torch.float64).clamp(-1, 1)
    n_fft, hop = 800, 250
    win = torch.hann_window(n_fft, periodic=True, dtype=torch.float64)
    spec = torch.stft(w, n_fft, hop_length=hop, win_length=n_fft, window=win, center=True, pad_mode='reflect',
                      return_complex=True).abs() ** 2
    xp = torch.nn.functional.pad(w[:, None], (n_fft // 2, n_fft // 2), mode='reflect')[:, 0]
    for (b, t) in ((0, 0), (1, 7), (0, spec.shape[-1] - 1)):
        d = A.dft_power_direct((xp[b, t * hop:t * hop + n_fft] * win).numpy())
        assert np.abs(spec[b, :, t].numpy() - d).max() <= 1e-12 * d.max()
    fb = A.melscale_fbanks(401, 0.0, 8000.0, 128, 16000)
    assert fb.shape == (401, 128) and fb.min() >= 0 and fb.max() <= 1.0 + 1e-12
    nz = fb > 0
    assert all(np.all(np.diff(np.flatnonzero(nz[:, m])) == 1) for m in range(128) if nz[:, m].any())      # one contiguous band each
    centres = np.array([fb[:, m].argmax() for m in range(128)])
    assert np.all(np.diff(centres) >= 0) and centres[-1] < 400
    out = A.log_mel(torch.zeros(1, 160000))
    assert out.shape == (1, 1, 128, 640) and torch.allclose(out, torch.full_like(out, -7.0))               # log10(0 + 1e-7)
    assert A.pad(torch.arange(5.)[None], 0.5, 16)[0].tolist() == [0., 1., 2., 3., 4., 4., 3., 2.]            # mirror extension

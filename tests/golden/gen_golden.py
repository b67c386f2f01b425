#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE.

Runs only in the build container (needs /root/reference).  The reference's own
Python files are imported unmodified from /root/reference; the two missing
third-party imports (timm==0.9.2, wandb) are satisfied by the stand-ins in
tests/golden/ref_shim (see its README).  Fixtures are DATA only: inputs are produced
by seeded closed-form generators that live in oracle/avmae_oracle.py, expected
outputs are what the reference computed.

    python tests/golden/gen_golden.py [--curve]      # --curve adds the 1k-step loss curve (~5 min)
"""
import argparse
import os
import sys
from functools import partial

import numpy as np
import torch
import torch.nn.functional as F_torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, 'ref_shim'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, ROOT)

from models import vits, fusion_blocks                      # noqa: E402  (reference)
from models.deepavfusion import DeepAVFusion                 # noqa: E402
from models.avmae import AVMAE                               # noqa: E402
from util import pos_embed as ref_pos_embed                  # noqa: E402
from util import lr_sched as ref_lr_sched                    # noqa: E402
from util import misc as ref_misc                            # noqa: E402

from oracle import avmae_oracle as O                         # noqa: E402
from oracle.configs import CONFIGS                           # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
torch.set_num_threads(8)


def register_archs():
    def mk(dim, depth, heads):
        return lambda pretrained=False, **kw: vits.ViT(patch_size=16, embed_dim=dim, depth=depth, num_heads=heads,
                                                       mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kw)
    vits.vit_micro = mk(128, 2, 2)
    vits.vit_tiny = mk(192, 12, 3)


ARCH_OF = {'micro': 'vit_micro', 'micro_token': 'vit_micro', 'micro_dense': 'vit_micro', 'micro_swin': 'vit_micro', 'tiny': 'vit_tiny',
           'base': 'vit_base', 'base_as': 'vit_base', 'large': 'vit_large'}


def build_reference(name):
    cfg = CONFIGS[name]
    pt = None if ARCH_OF[name] == 'vit_large' else ''       # models/vits.py:150-151: vit_large asserts on anything but None
    enc = DeepAVFusion(
        image_arch=ARCH_OF[name], image_pretrained=pt, image_size=cfg.image_size,
        audio_arch=ARCH_OF[name], audio_pretrained=pt, audio_size=cfg.audio_size,
        fusion_arch=cfg.fusion_arch, fusion_layers='all', num_fusion_tkns=cfg.fusion_tkns,
        fusion_mlp_ratio=cfg.fusion_mlp_ratio, fusion_attn_ratio=cfg.fusion_attn_ratio,
        fusion_num_heads=cfg.fusion_num_heads)
    model = AVMAE(enc, enc.embed_dim,
                  image_decoder_arch=cfg.decoder_arch, image_decoder_depth=cfg.decoder_depth,
                  image_mask_ratio=cfg.image_mask_ratio, image_norm_loss=cfg.image_norm_loss,
                  audio_decoder_arch=cfg.decoder_arch, audio_decoder_depth=cfg.decoder_depth,
                  audio_mask_ratio=cfg.audio_mask_ratio, audio_norm_loss=cfg.audio_norm_loss,
                  decoder_dim=cfg.decoder_dim, num_heads=cfg.decoder_heads, mlp_ratio=cfg.decoder_mlp_ratio)
    sd = O.closed_form_state(cfg, seed=0)
    for k, v in model.state_dict().items():         # the reference's own registered buffers (models/swin.py:39, 158) pin the
        if O.is_buffer(k):                          # oracle's restatement of the window index / shift mask BEFORE they are overwritten
            assert torch.equal(v.float(), sd[k].float()), k
    model.load_state_dict(sd, strict=True)          # also pins the state-dict contract (names + shapes)
    return cfg, model, sd


class InjectNoise:
    """Make AVMAE.random_masking (models/avmae.py:127) draw the given noise."""
    def __init__(self, noises):
        self.noises = list(noises)

    def __enter__(self):
        self._orig = torch.rand
        it = iter(self.noises)

        def fake_rand(N, L, device=None):
            n = next(it)
            assert n.shape == (N, L)
            return torch.from_numpy(n)
        torch.rand = fake_rand
        return self

    def __exit__(self, *a):
        torch.rand = self._orig


def gen_masking():
    _, model, _ = build_reference('micro')
    out = {}
    rs = np.random.RandomState(11)
    for tag, (N, L, r) in {'img196': (4, 196, 0.75), 'aud320': (4, 320, 0.8), 'aud320_75': (3, 320, 0.75),
                           'aud96': (2, 96, 0.8), 'aud64': (2, 64, 0.8), 'odd15': (3, 15, 0.8), 'one': (1, 7, 0.5)}.items():
        noise = ((rs.permutation(N * L).reshape(N, L) + 0.5) / (N * L)).astype(np.float32)
        with InjectNoise([noise]):
            ids_keep, mask, ids_restore = model.random_masking(N, L, r, device='cpu')
        out[f'{tag}.noise'] = noise
        out[f'{tag}.ratio'] = np.float64(r)
        out[f'{tag}.ids_keep'] = ids_keep.numpy()
        out[f'{tag}.mask'] = mask.numpy()
        out[f'{tag}.ids_restore'] = ids_restore.numpy()
    np.savez_compressed(os.path.join(OUT, 'masking.npz'), **out)


def gen_posembed():
    out = {}
    for dim, grid in [(192, (4, 4)), (192, (8, 8)), (128, (2, 7)), (64, (4, 4))]:
        out[f'2d_{dim}_{grid[0]}x{grid[1]}'] = ref_pos_embed.get_2d_sincos_pos_embed(dim, grid).astype(np.float32)
    for dim, grid in [(768, (14, 14)), (768, (8, 40)), (768, (8, 12)), (512, (14, 14)), (512, (8, 40))]:
        full = ref_pos_embed.get_2d_sincos_pos_embed(dim, grid).astype(np.float32)
        out[f'2d_{dim}_{grid[0]}x{grid[1]}_sub'] = full[::3, ::5].copy()
        out[f'2d_{dim}_{grid[0]}x{grid[1]}_sum'] = np.float64(full.astype(np.float64).sum())
    full = ref_pos_embed.get_3d_sincos_pos_embed(768, (4, 14, 14)).astype(np.float32)
    out['3d_768_4x14x14_sub'] = full[::7, ::5].copy()
    out['3d_768_4x14x14_sum'] = np.float64(full.astype(np.float64).sum())
    np.savez_compressed(os.path.join(OUT, 'posembed.npz'), **out)


def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def gen_ops():
    """Per-module outputs and gradients at micro shapes with odd row counts."""
    cfg, model, sd = build_reference('micro')
    D, H = cfg.embed_dim, cfg.fusion_num_heads
    rs = np.random.RandomState(5)
    B, nI, nA = 3, 5, 7
    nF = sum(cfg.fusion_tkns)
    t = lambda *s: torch.from_numpy(rs.standard_normal(s).astype(np.float32))
    out = {}

    def record(tag, module, inputs, fwd):
        inputs = [i.clone().requires_grad_(True) for i in inputs]
        module.zero_grad()
        y = fwd(*inputs)
        g = torch.from_numpy(np.random.RandomState(99).standard_normal(tuple(y.shape)).astype(np.float32))
        (y * g).sum().backward()
        out[f'{tag}.out'] = y.detach().numpy()
        out[f'{tag}.gout'] = g.numpy()
        for i, x in enumerate(inputs):
            out[f'{tag}.in{i}'] = x.detach().numpy()
            out[f'{tag}.gin{i}'] = x.grad.numpy()
        for n, p in module.named_parameters():
            if p.grad is not None:
                out[f'{tag}.gw.{n}'] = p.grad.numpy().copy()

    fb = model.encoder.fusion_blocks[0]
    record('cross_attention', fb.attn.attn_v, [t(B, cfg.fusion_tkns[1], D), t(B, nI, D)], lambda a, b: fb.attn.attn_v(a, b)[0])
    record('factorized_attention', fb.attn, [t(B, nF, D), t(B, nI, D), t(B, nA, D)], lambda a, b, c: fb.attn(a, b, c)[0])
    record('fusion_block', fb, [t(B, nF, D), t(B, nI, D), t(B, nA, D)], lambda a, b, c: fb(a, b, c))
    blk = model.encoder.image.blocks[1]
    record('timm_block', blk, [t(B, nF + nI, D)], lambda a: blk(a))          # stand-in arithmetic: unpinned
    dblk = model.image_decoder_blocks[0]
    record('decoder_block', dblk, [t(B, 11, cfg.decoder_dim)], lambda a: dblk(a))

    # prepare_patch_tokens (models/vits.py:91-107)
    img = t(B, 3, *cfg.image_size)
    L = cfg.image_grid[0] * cfg.image_grid[1]
    ids = torch.from_numpy(np.stack([np.random.RandomState(i).permutation(L)[:nI] for i in range(B)]).astype(np.int64))
    vit = model.encoder.image
    vit.zero_grad()
    tok = vit.prepare_patch_tokens(img, ids)
    g = t(*tok.shape)
    (tok * g).sum().backward()
    out.update({'prepare.image': img.numpy(), 'prepare.ids_keep': ids.numpy(), 'prepare.out': tok.detach().numpy(),
                'prepare.gout': g.numpy(), 'prepare.gw.weight': vit.patch_embed.proj.weight.grad.numpy().copy(),
                'prepare.gw.bias': vit.patch_embed.proj.bias.grad.numpy().copy()})

    # whole encoder (models/deepavfusion.py:88-118), return_embs
    aud = t(B, 1, *cfg.audio_size)
    La = cfg.audio_grid[0] * cfg.audio_grid[1]
    ids_a = torch.from_numpy(np.stack([np.random.RandomState(50 + i).permutation(La)[:nA] for i in range(B)]).astype(np.int64))
    xi, xa, xf, embs = model.encoder(img, aud, ids, ids_a, return_embs=True)
    out.update({'encoder.audio': aud.numpy(), 'encoder.ids_keep_audio': ids_a.numpy(),
                'encoder.x_image': xi.detach().numpy(), 'encoder.x_audio': xa.detach().numpy(),
                'encoder.x_fusion': xf.detach().numpy(),
                'encoder.emb0_fusion': embs[0][2].detach().numpy(), 'encoder.emb0_image': embs[0][0].detach().numpy()})
    # un-masked forward (forward_encoder path used by the kNN probe, models/avmae.py:144-145)
    xi, xa, xf = model.forward_encoder(img, aud)
    out.update({'encoder_full.x_image': xi.detach().numpy(), 'encoder_full.x_audio': xa.detach().numpy(),
                'encoder_full.x_fusion': xf.detach().numpy()})

    # forward_decoder (models/avmae.py:147-180)
    x = t(B, nI, D).requires_grad_(True)
    xfus = t(B, nF, D).requires_grad_(True)
    restore = torch.from_numpy(np.stack([np.random.RandomState(70 + i).permutation(L) for i in range(B)]).astype(np.int64))
    model.zero_grad()
    pred = model.forward_decoder(x, xfus, restore, modality='image')
    g = t(*pred.shape)
    (pred * g).sum().backward()
    out.update({'decoder.x': x.detach().numpy(), 'decoder.x_fusion': xfus.detach().numpy(), 'decoder.ids_restore': restore.numpy(),
                'decoder.out': pred.detach().numpy(), 'decoder.gout': g.numpy(),
                'decoder.gx': x.grad.numpy(), 'decoder.gx_fusion': xfus.grad.numpy(),
                'decoder.gw.mask_token': model.image_decoder_mask_token.grad.numpy().copy(),
                'decoder.gw.pos_embed': model.image_decoder_pos_embed.grad.numpy().copy(),
                'decoder.gw.embed.weight': model.image_decoder_embed.weight.grad.numpy().copy(),
                'decoder.gw.pred.bias': model.image_decoder_pred.bias.grad.numpy().copy()})

    # patchify + forward_loss (models/avmae.py:182-214), both norm_pix settings, both modalities
    for mod, x_in, Lm in (('image', img, L), ('audio', aud, La)):
        target = AVMAE.patchify(x_in, (16, 16))
        out[f'patchify.{mod}'] = target.numpy()
        mask = torch.from_numpy((np.random.RandomState(3).rand(B, Lm) > 0.3).astype(np.float32))
        for norm in (True, False):
            p = t(*target.shape).requires_grad_(True)
            loss = AVMAE.forward_loss(target, p, mask, norm_pix_loss=norm)
            loss.backward()
            out[f'loss.{mod}.{int(norm)}.pred'] = p.detach().numpy()
            out[f'loss.{mod}.{int(norm)}.mask'] = mask.numpy()
            out[f'loss.{mod}.{int(norm)}.loss'] = loss.detach().numpy()
            out[f'loss.{mod}.{int(norm)}.gpred'] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'ops_micro.npz'), **out)


def gen_e2e(name, B, seed, keep_preds):
    cfg, model, sd = build_reference(name)
    image, audio, ni, na = O.synthetic_batch(cfg, B, seed=seed)
    model.zero_grad()
    with InjectNoise([ni, na]):
        li, la, pi, pa = model(image, audio)
    (li + la).backward()
    out = {'loss_image': li.detach().numpy(), 'loss_audio': la.detach().numpy(),
           'B': np.int64(B), 'seed': np.int64(seed)}
    if keep_preds:
        out['pred_image'] = pi.detach().numpy()
        out['pred_audio'] = pa.detach().numpy()
    else:
        out['pred_image_sub'] = pi.detach().numpy()[:, ::3, ::7].copy()
        out['pred_audio_sub'] = pa.detach().numpy()[:, ::3, ::7].copy()
    names, norms = [], []
    for n, p in model.named_parameters():
        if p.grad is not None:
            names.append(n)
            norms.append(float(p.grad.double().norm()))
    out['grad_names'] = np.array(names)
    out['grad_norms'] = np.array(norms, dtype=np.float64)
    out['grad_norm_total'] = np.float64(ref_misc.get_grad_norm_(model.parameters()).item())   # util/misc.py:151-163
    # a few full gradients
    fk = 'encoder.fusion_blocks.0.attn.k.weight' if cfg.fusion_arch == 'factorized_mmi' else 'encoder.fusion_blocks.0.attn.kv.weight'
    extra = (('image_decoder_blocks.1.attn.relative_position_bias_table', 'audio_decoder_blocks.0.attn.relative_position_bias_table',
              'audio_decoder_blocks.1.attn.qkv.weight') if cfg.decoder_arch == 'swin' else ())
    for n in ('encoder.fusion_tokens', 'image_decoder_mask_token', 'encoder.image.patch_embed.proj.bias',
              fk, 'encoder.audio.blocks.0.attn.qkv.bias') + extra:
        out['grad.' + n] = dict(model.named_parameters())[n].grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, f'e2e_{name}.npz'), **out)
    print(f'e2e[{name}] loss_image={float(li):.6f} loss_audio={float(la):.6f} gnorm={float(out["grad_norm_total"]):.6f}')


def _compact_grads(model, out, full_max=256, budget=150000):
    """Every parameter's gradient norm, the total norm, every SMALL gradient in full and a strided sample of every large one
    (the stride — odd, so that it walks across rows — keeps the samples of all large gradients together near ``budget`` floats)."""
    total = sum(p.numel() for p in model.parameters() if p.grad is not None and p.numel() > full_max)
    stride = max(97, total // budget) | 1
    names, norms = [], []
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        names.append(n)
        norms.append(float(p.grad.double().norm()))
        g = p.grad.detach().reshape(-1)
        out['grad.' + n] = (g if g.numel() <= full_max else g[::stride]).numpy().astype(np.float32).copy()
    out['grad_names'] = np.array(names)
    out['grad_norms'] = np.array(norms, dtype=np.float64)
    out['grad_stride'] = np.int64(stride)
    out['grad_full_max'] = np.int64(full_max)


def gen_e2e_full(name, B, seed):
    """The BASELINE.json configurations at their published widths (ViT-B cfg 2 / cfg 3, ViT-L cfg 4), small batch: what the imported
    reference computes — losses, strided prediction samples, every gradient's norm, small gradients in full and a strided sample of
    the large ones (the fixture stays near 1 MB)."""
    cfg, model, sd = build_reference(name)
    image, audio, ni, na = O.synthetic_batch(cfg, B, seed=seed)
    model.zero_grad()
    with InjectNoise([ni, na]):
        li, la, pi, pa = model(image, audio)
    (li + la).backward()
    out = {'loss_image': li.detach().numpy(), 'loss_audio': la.detach().numpy(), 'B': np.int64(B), 'seed': np.int64(seed),
           'pred_image_sub': pi.detach().numpy()[:, ::5, ::11].copy(), 'pred_audio_sub': pa.detach().numpy()[:, ::5, ::11].copy(),
           'grad_norm_total': np.float64(ref_misc.get_grad_norm_(model.parameters()).item())}
    _compact_grads(model, out)
    np.savez_compressed(os.path.join(OUT, f'e2e_{name}.npz'), **out)
    print(f'e2e_full[{name}] B={B} loss_image={float(li):.6f} loss_audio={float(la):.6f} gnorm={float(out["grad_norm_total"]):.6f}')


class _NS(dict):
    """attribute + .get() access, enough for util/lr_sched.py's ``args.opt``."""
    __getattr__ = dict.__getitem__


def train_args(epochs, warmup, lr):
    return _NS(opt=_NS(lr=lr, warmup_epochs=warmup, epochs=epochs, pt_warmup_epochs=f'{epochs}/2',
                       pt_lr_mult_start=0, pt_lr_mult_end=1))


def gen_lr_and_groups():
    cfg, model, _ = build_reference('micro')
    nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]            # train.py:89
    groups = ref_lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
    name_of = {id(p): n for n, p in model.named_parameters()}
    out = {}
    for gi, g in enumerate(groups):
        out[f'group{gi}.names'] = np.array([name_of[id(p)] for p in g['params']])
        out[f'group{gi}.weight_decay'] = np.float64(g['weight_decay'])
        out[f'group{gi}.pretrained'] = np.bool_(g.get('pretrained', False))
    out['n_groups'] = np.int64(len(groups))
    opt = torch.optim.AdamW(groups, lr=1e-3, betas=(0.9, 0.95))
    args = train_args(epochs=10, warmup=2, lr=1e-3)
    eps = np.linspace(0, 10, 41)[:-1]
    table = []
    for e in eps:
        lr = ref_lr_sched.adjust_learning_rate(opt, float(e), args)
        table.append([lr] + [g['lr'] for g in opt.param_groups])
    out['epochs'] = eps
    out['lr_table'] = np.array(table, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'lr_groups.npz'), **out)


def make_optimizer(model, lr, wd=0.05):
    nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
    groups = ref_lr_sched.param_groups_pretrained(model, wd, no_weight_decay_list=nd, image_pt='', audio_pt='')
    return torch.optim.AdamW(groups, lr=lr, betas=(0.9, 0.95))


def gen_trainer_steps():
    """util/misc.py Trainer.step semantics: accumulation, grad-norm scaling, AdamW update."""
    cfg, model, _ = build_reference('micro')
    opt = make_optimizer(model, lr=1e-3)
    trainer = ref_misc.Trainer(model, optimizer=opt, accum_iter=2, use_amp=False, distributed=False)
    args = train_args(epochs=4, warmup=1, lr=1e-3)
    rec = {'grad_norm': [], 'loss': [], 'lr': []}
    for step in range(6):
        if step % 2 == 0:
            rec['lr'].append(ref_lr_sched.adjust_learning_rate(opt, step / 6 * 4, args))
        image, audio, ni, na = O.synthetic_batch(cfg, 2, seed=300 + step)
        with InjectNoise([ni, na]):
            li, la = trainer.model(image, audio)[:2]
        loss = li + la
        gn, scale = trainer.step(loss)
        rec['grad_norm'].append(gn)
        rec['loss'].append(float(loss))
    out = {k: np.array(v, dtype=np.float64) for k, v in rec.items()}
    out['n_steps'] = np.int64(int(trainer.n_steps))
    names, sums = [], []
    for n, p in model.named_parameters():
        names.append(n)
        sums.append(float(p.detach().double().sum()))
    out['param_names'] = np.array(names)
    out['param_sums'] = np.array(sums, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'trainer_steps.npz'), **out)
    print('trainer steps: grad_norm', out['grad_norm'], 'loss', out['loss'])


def gen_curve(steps=1000, B=2):
    """1k-step ViT-Tiny loss curve (train.py:151-180 loop on synthetic data, fp32 CPU)."""
    cfg, model, _ = build_reference('tiny')
    eff = B
    lr = 1.5e-4 * eff / 256 * 64       # blr scaled so 1k steps move the loss visibly (recorded below)
    opt = make_optimizer(model, lr=lr)
    trainer = ref_misc.Trainer(model, optimizer=opt, accum_iter=1, use_amp=False, distributed=False)
    steps_per_epoch = 100
    args = train_args(epochs=steps // steps_per_epoch, warmup=1, lr=lr)
    li_c, la_c, gn_c = [], [], []
    for s in range(steps):
        ref_lr_sched.adjust_learning_rate(opt, s / steps_per_epoch, args)
        image, audio, ni, na = O.structured_batch(cfg, B, seed=10_000 + s)
        with InjectNoise([ni, na]):
            li, la = trainer.model(image, audio)[:2]
        gn, _ = trainer.step(li + la)
        li_c.append(float(li)); la_c.append(float(la)); gn_c.append(gn)
        if s % 50 == 0:
            print(f'curve step {s}: {li_c[-1]:.4f} {la_c[-1]:.4f} gn {gn:.3f}', flush=True)
    np.savez_compressed(os.path.join(OUT, 'curve_tiny.npz'), loss_image=np.array(li_c), loss_audio=np.array(la_c),
                        grad_norm=np.array(gn_c), lr=np.float64(lr), B=np.int64(B),
                        steps_per_epoch=np.int64(steps_per_epoch), warmup_epochs=np.int64(1))


def probe_weights(shapes, seed):
    """Fixed output weights so that the probe loss sum_i <w_i, out_i> has well-conditioned gradients
    (the plain sum of LayerNorm outputs used by models/video_earlyfusion.py:185 is nearly gradient-free)."""
    rs = np.random.RandomState(seed)
    return [torch.from_numpy(rs.standard_normal(s).astype(np.float32)) for s in shapes]


def gen_video(name, B, seed):
    """BASELINE configs[4] family: VideoEarlyFusion forward + backward (models/video_earlyfusion.py:95-131)."""
    from models import video_vits                                 # reference
    from models.video_earlyfusion import VideoEarlyFusion         # reference
    cfg = CONFIGS[name]
    video_vits.video_vit_micro = lambda pretrained=None, **kw: video_vits.VideoViTEncoder(
        patch_size=(2, 16, 16), embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, **kw)
    arch = {'video_micro': ('video_vit_micro', 'vit_micro'), 'video_base': ('video_vit_base', 'vit_base')}[name]
    model = VideoEarlyFusion(video_arch=arch[0], video_pretrained='', video_size=cfg.video_size,
                             audio_arch=arch[1], audio_pretrained='', audio_size=cfg.audio_size,
                             fusion_layers='all', num_fusion_tkns=cfg.fusion_tkns, fusion_mlp_ratio=cfg.fusion_mlp_ratio,
                             fusion_attn_ratio=cfg.fusion_attn_ratio, fusion_num_heads=cfg.fusion_num_heads)
    out = {'video_pos_embed_init': model.video.pos_embed.detach().numpy().copy(),      # the reference's own 3-D table
           'B': np.int64(B), 'seed': np.int64(seed)}
    sd = O.closed_form_state(cfg, seed=0)
    model.load_state_dict(sd, strict=True)          # pins the state-dict contract (names + shapes)
    video, audio = O.synthetic_video_batch(cfg, B, seed=seed)
    xv, xa, xf = model(video, audio)
    w = probe_weights([xv.shape, xa.shape, xf.shape], seed + 1)
    loss = (xv * w[0]).sum() + (xa * w[1]).sum() + (xf * w[2]).sum()
    loss.backward()
    out.update(x_video=xv.detach().numpy(), x_audio=xa.detach().numpy(), x_fusion=xf.detach().numpy(),
               loss_sum=np.float64((xv.sum() + xa.sum() + xf.sum()).item()),        # the reference probe's loss (:185)
               loss_probe=np.float64(loss.item()))
    if name == 'video_base':                                      # full width: compact form (see gen_e2e_full)
        _compact_grads(model, out)
        for k in ('x_video', 'x_audio', 'x_fusion'):
            out[k + '_sub'] = out.pop(k)[:, ::3, ::7].copy()
    else:
        names, norms = [], []
        for n, p in model.named_parameters():
            if p.grad is not None:
                names.append(n)
                norms.append(float(p.grad.double().norm()))
        out['grad_names'] = np.array(names)
        out['grad_norms'] = np.array(norms, dtype=np.float64)
        for n in ('fusion_tokens', 'video.patch_embed.proj.weight', 'video.blocks.0.attn.qkv.bias', 'audio.patch_embed.proj.bias',
                  'fusion_blocks.1.attn.k.weight', 'video.norm.weight'):
            out['grad.' + n] = dict(model.named_parameters())[n].grad.numpy().copy()
    # embeddings per layer (return_embs=True) of a second call pin that surface too
    with torch.no_grad():
        embs = model(video, audio, return_embs=True)[3]
    out['emb_last_video_sub'] = embs[-1][0].numpy()[:, ::2, ::5].copy()
    out['emb_first_fusion'] = embs[0][2].numpy().copy()
    np.savez_compressed(os.path.join(OUT, f'e2e_{name}.npz'), **out)
    print(f'video[{name}] loss_sum={float(out["loss_sum"]):.6f} loss_probe={float(out["loss_probe"]):.6f}')


def gen_droppath(B=4, seed=41, p=0.25):
    """DeepAVFusion in TRAINING mode with drop_path > 0 (the fine-tuning setting, configs/finetune.yaml:47): the
    Bernoulli draws of every DropPath are replaced, in call order, by masks stored in the fixture."""
    cfg = CONFIGS['micro']
    enc = DeepAVFusion(image_arch='vit_micro', image_pretrained='', image_size=cfg.image_size,
                       audio_arch='vit_micro', audio_pretrained='', audio_size=cfg.audio_size,
                       fusion_arch='factorized_mmi', fusion_layers='all', num_fusion_tkns=cfg.fusion_tkns,
                       fusion_mlp_ratio=cfg.fusion_mlp_ratio, fusion_attn_ratio=cfg.fusion_attn_ratio,
                       fusion_num_heads=cfg.fusion_num_heads, drop_path=p)
    sd = {k[len('encoder.'):]: v for k, v in O.closed_form_state(cfg, seed=0).items() if k.startswith('encoder.')}
    enc.load_state_dict(sd, strict=True)
    enc.train()
    image, audio, ni, na = O.synthetic_batch(cfg, B, seed=seed)
    ik = torch.from_numpy(O.random_masking_from_noise(ni, cfg.image_mask_ratio)[0])
    ak = torch.from_numpy(O.random_masking_from_noise(na, cfg.audio_mask_ratio)[0])
    rs = np.random.RandomState(seed + 2)
    masks = (rs.uniform(size=(cfg.depth * 6, B)) < (1 - p)).astype(np.float32)      # call order: per layer visual(2), audio(2), fusion(2)
    masks[0, :] = [1, 0, 1, 1][:B]                                                   # make sure both outcomes occur early
    calls = [0]
    real = torch.Tensor.bernoulli_

    def fake(self, prob=0.5, *a, **k):
        m = torch.from_numpy(masks[calls[0]]).view(self.shape)
        calls[0] += 1
        return self.copy_(m)
    torch.Tensor.bernoulli_ = fake
    try:
        xi, xa, xf = enc(image, audio, ik, ak)
    finally:
        torch.Tensor.bernoulli_ = real
    assert calls[0] == cfg.depth * 6, calls[0]
    w = probe_weights([xi.shape, xa.shape, xf.shape], seed + 1)
    loss = (xi * w[0]).sum() + (xa * w[1]).sum() + (xf * w[2]).sum()
    loss.backward()
    out = {'B': np.int64(B), 'seed': np.int64(seed), 'p': np.float64(p), 'masks': masks, 'x_image': xi.detach().numpy(),
           'x_audio': xa.detach().numpy(), 'x_fusion': xf.detach().numpy(), 'loss_probe': np.float64(loss.item())}
    names, norms = [], []
    for n, q in enc.named_parameters():
        if q.grad is not None:
            names.append(n)
            norms.append(float(q.grad.double().norm()))
    out['grad_names'] = np.array(names)
    out['grad_norms'] = np.array(norms, dtype=np.float64)
    for n in ('fusion_tokens', 'image.blocks.0.mlp.fc2.weight', 'audio.blocks.1.attn.proj.bias', 'fusion_blocks.0.attn.attn_v.q.weight'):
        out['grad.' + n] = dict(enc.named_parameters())[n].grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'droppath_micro.npz'), **out)
    print(f'droppath loss_probe={float(loss):.6f}, kept {masks.mean():.2f}')


def dropout_site_names(cfg, arch):
    """The nn.Dropout modules of a DeepAVFusion forward in CALL order (models/deepavfusion.py:96-107: image block, audio block,
    fusion block per layer), named as the oracle / the engine name them."""
    names = []
    for l in range(cfg.depth):
        for t in ('visual', 'audio'):
            names += [f'{t}.{l}.{s}' for s in ('attn', 'proj', 'fc1', 'fc2')]
        if arch == 'factorized_mmi':
            names += [f'fusion.{l}.{s}' for s in ('attn_v.attn', 'attn_v.proj', 'attn_a.attn', 'attn_a.proj', 'attn', 'proj', 'fc1', 'fc2')]
        else:
            names += [f'fusion.{l}.{s}' for s in ('attn', 'proj', 'fc1', 'fc2')]
    return names


def gen_dropout(name='micro', B=3, seed=61, p_attn=0.2, p_proj=0.1, p_path=0.0):
    """DeepAVFusion in TRAINING mode with attn_drop / drop > 0 (the constructor surface of eval_finetune.py:170-171; every
    shipped config sets 0): every nn.Dropout draw is replaced, in call order, by the closed-form mask
    oracle.dropout_keep_mask(seed, name, shape, p) — the fixture stores seed and probabilities, not the masks."""
    from timm.models.vision_transformer import Attention as TimmAttention
    cfg = CONFIGS[name]
    arch = cfg.fusion_arch
    TimmAttention.fused_attn = False
    try:
        enc = DeepAVFusion(image_arch=ARCH_OF[name], image_pretrained='', image_size=cfg.image_size,
                           audio_arch=ARCH_OF[name], audio_pretrained='', audio_size=cfg.audio_size,
                           fusion_arch=arch, fusion_layers='all', num_fusion_tkns=cfg.fusion_tkns,
                           fusion_mlp_ratio=cfg.fusion_mlp_ratio, fusion_attn_ratio=cfg.fusion_attn_ratio,
                           fusion_num_heads=cfg.fusion_num_heads, attn_drop=p_attn, drop=p_proj, drop_path=p_path)
        sd = {k[len('encoder.'):]: v for k, v in O.closed_form_state(cfg, seed=0).items() if k.startswith('encoder.')}
        enc.load_state_dict(sd, strict=True)
        enc.train()
        image, audio, ni, na = O.synthetic_batch(cfg, B, seed=seed)
        ik = torch.from_numpy(O.random_masking_from_noise(ni, cfg.image_mask_ratio)[0])
        ak = torch.from_numpy(O.random_masking_from_noise(na, cfg.audio_mask_ratio)[0])
        names = dropout_site_names(cfg, arch)
        calls, shapes = [0], []
        real = F_torch.dropout

        def fake(x, p=0.5, training=True, inplace=False):
            assert training and p in (p_attn, p_proj), (p, training)
            nm = names[calls[0]]
            assert p == (p_attn if nm.endswith('attn') else p_proj), (nm, p)
            calls[0] += 1
            shapes.append(list(x.shape))
            m = torch.from_numpy(O.dropout_keep_mask(seed, nm, x.shape, p))
            return x * m / (1.0 - p)
        F_torch.dropout = fake
        try:
            xi, xa, xf = enc(image, audio, ik, ak)
        finally:
            F_torch.dropout = real
    finally:
        TimmAttention.fused_attn = True
    assert calls[0] == len(names), (calls[0], len(names))
    w = probe_weights([xi.shape, xa.shape, xf.shape], seed + 1)
    loss = (xi * w[0]).sum() + (xa * w[1]).sum() + (xf * w[2]).sum()
    loss.backward()
    out = {'B': np.int64(B), 'seed': np.int64(seed), 'p_attn': np.float64(p_attn), 'p_proj': np.float64(p_proj),
           'site_names': np.array(names), 'site_shapes': np.array([s + [0] * (4 - len(s)) for s in shapes], dtype=np.int64),
           'x_image': xi.detach().numpy(), 'x_audio': xa.detach().numpy(), 'x_fusion': xf.detach().numpy(),
           'loss_probe': np.float64(loss.item())}
    gn, norms = [], []
    for n, q in enc.named_parameters():
        if q.grad is not None:
            gn.append(n)
            norms.append(float(q.grad.double().norm()))
    out['grad_names'] = np.array(gn)
    out['grad_norms'] = np.array(norms, dtype=np.float64)
    for n in ('fusion_tokens', 'image.blocks.0.mlp.fc1.weight', 'audio.blocks.1.attn.qkv.weight', 'fusion_blocks.0.attn.q.weight',
              'fusion_blocks.1.mlp.fc2.bias'):
        out['grad.' + n] = dict(enc.named_parameters())[n].grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, f'dropout_{name}.npz'), **out)
    print(f'dropout[{name}] loss_probe={float(loss):.6f} over {len(names)} Dropout calls')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--curve', action='store_true')
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    register_archs()
    torch.manual_seed(0)
    jobs = {'masking': gen_masking, 'posembed': gen_posembed, 'ops': gen_ops,
            'e2e_micro': lambda: gen_e2e('micro', 3, 21, True), 'e2e_tiny': lambda: gen_e2e('tiny', 2, 22, False),
            'lr': gen_lr_and_groups, 'trainer': gen_trainer_steps,
            'video_micro': lambda: gen_video('video_micro', 2, 31),
            'droppath': gen_droppath,
            'dropout_micro': lambda: gen_dropout('micro'), 'dropout_micro_token': lambda: gen_dropout('micro_token', seed=62),
            'dropout_micro_dense': lambda: gen_dropout('micro_dense', seed=63),
            'e2e_micro_token': lambda: gen_e2e('micro_token', 3, 23, False),
            'e2e_micro_dense': lambda: gen_e2e('micro_dense', 3, 24, False),
            'e2e_micro_swin': lambda: gen_e2e('micro_swin', 2, 25, True),
            # the published configurations at full width (BASELINE.json configs[1] .. [4]), small batches
            'e2e_base': lambda: gen_e2e_full('base', 2, 51), 'e2e_base_as': lambda: gen_e2e_full('base_as', 2, 52),
            'e2e_large': lambda: gen_e2e_full('large', 1, 53), 'video_base': lambda: gen_video('video_base', 1, 54)}
    if a.curve:
        jobs = {'curve': gen_curve}
    for k, f in jobs.items():
        if a.only and k != a.only:
            continue
        f()
        print('generated', k)

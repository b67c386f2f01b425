#!/usr/bin/env python3
"""Pre-training worker: drop-in for the reference's train.py (main_worker :20-137, train_one_epoch :140-187)
on the MI355X path.  Same sequence per iteration: lr schedule -> forward under autocast/autosync -> loss sum ->
non-finite guard -> Trainer.step -> metrics {loss, loss_image, loss_audio, grad_norm, amp_scale, lr}.

    python train.py [key=value ...]                       # single GPU, configs/deepavfusion.yaml
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py env.ngpu=8

Hydra/omegaconf are not available in this environment: the YAML files under configs/ (same keys as the
reference) are composed by a small loader with ``${a.b}`` interpolation and ``a.b=value`` overrides.
Only ``data.dataset=synthetic`` is implemented (tensor contract of SURVEY.md section 2 row 19)."""
import math
import os
import re
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


class Cfg(dict):
    """dict with attribute access (what the reference gets from omegaconf)."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    return Cfg({k: _wrap(v) for k, v in x.items()}) if isinstance(x, dict) else x


def load_config(name='deepavfusion', overrides=()):
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'configs', f'{name}.yaml')))
    for d in cfg.pop('defaults', []):
        (group, choice), = d.items()
        cfg[group] = yaml.safe_load(open(os.path.join(ROOT, 'configs', group, f'{choice}.yaml')))
    cfg = _wrap(cfg)
    for ov in overrides:
        key, _, val = ov.partition('=')
        node = cfg
        *path, leaf = key.split('.')
        for p in path:
            node = node[p]
        node[leaf] = yaml.safe_load(val)

    def lookup(path):
        node = cfg
        for p in path.split('.'):
            node = node[p]
        return node

    def resolve(node):
        for k, v in node.items():
            if isinstance(v, dict):
                resolve(v)
            elif isinstance(v, str) and '${' in v:
                node[k] = re.sub(r'\$\{([^}]+)\}', lambda m: str(lookup(m.group(1))), v)
    resolve(cfg)
    return cfg


class SyntheticAV(torch.utils.data.Dataset):
    """ImageNet-normalised frames ~ N(0,1); log10-mel spectrograms in about [-7, 4] (train.py:50-54, datasets.py:242) — or,
    with ``wave_samples`` set (data.audio_frontend=gpu), raw waveforms in [-1, 1] that the device-side front-end
    (deepavfusion_amd.util.audio_transforms.LogMelSpectrogram) turns into the same log-mel tensor."""
    def __init__(self, n, image_size, audio_size, seed=0, wave_samples=0):
        self.n, self.image_size, self.audio_size, self.seed, self.wave_samples = n, image_size, audio_size, seed, wave_samples

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + i)
        image = torch.randn(3, *self.image_size, generator=g)
        if self.wave_samples:
            audio = (torch.randn(self.wave_samples, generator=g) * 0.1).clamp(-1, 1)
            return image, audio, i
        audio = (torch.randn(1, *self.audio_size, generator=g) * 2.0 - 3.0).clamp(-7, 4)
        return image, audio, i


def main_worker(local_rank, args):
    from deepavfusion_amd.models.avmae import AVMAE
    from deepavfusion_amd.models.deepavfusion import DeepAVFusion
    from deepavfusion_amd.util import distributed as dist_utils
    from deepavfusion_amd.util import lr_sched, misc as misc_utils
    from deepavfusion_amd.util.flat import FlatAdamW

    job_dir = f'{args.output_dir}/{args.job_name}'
    os.makedirs(job_dir, exist_ok=True)
    dist_utils.init_distributed_mode(local_rank, args, log_fn=f'{job_dir}/train.log')
    device = torch.device('cuda', torch.cuda.current_device())
    print(f'job dir: {job_dir}')
    num_tasks = dist_utils.get_world_size()
    eff_batch_size = args.opt.batch_size * args.opt.accum_iter * num_tasks
    if args.opt.lr is None:
        args.opt.lr = args.opt.blr * eff_batch_size / 256              # train.py:32-34
    print('base lr: %.2e' % args.opt.blr)
    print('actual lr: %.2e' % args.opt.lr)
    print('effective batch size: %d' % eff_batch_size)

    image_size = (args.data.image_size, args.data.image_size)
    audio_size = (args.data.audio_mels, int(args.data.audio_dur * 64))  # train.py:65
    if args.data.dataset != 'synthetic':
        raise NotImplementedError('only data.dataset=synthetic is on the MI355X path (the reference datasets need PyAV/torchaudio)')
    gpu_frontend = args.data.get('audio_frontend', 'loader') == 'gpu'     # waveforms in, log-mel computed on the device
    dataset = SyntheticAV(args.data.steps_per_epoch * eff_batch_size, image_size, audio_size, seed=args.env.seed or 0,
                          wave_samples=int(args.data.audio_dur * args.data.audio_rate) if gpu_frontend else 0)
    sampler = torch.utils.data.DistributedSampler(dataset, shuffle=True) if num_tasks > 1 else torch.utils.data.RandomSampler(dataset)
    loader = torch.utils.data.DataLoader(dataset, batch_size=args.opt.batch_size, sampler=sampler, num_workers=args.env.workers,
                                         pin_memory=True, drop_last=True)

    m = args.model
    encoder = DeepAVFusion(
        image_arch=m.image.backbone, image_pretrained=m.image.pretrained, image_size=image_size,
        audio_arch=m.audio.backbone, audio_pretrained=m.audio.pretrained, audio_size=audio_size,
        fusion_arch=m.fusion.arch, fusion_layers=m.fusion.layers,
        num_fusion_tkns=(m.fusion.num_fusion_tkns, m.fusion.num_aggr_image_tkns, m.fusion.num_aggr_audio_tkns),
        fusion_mlp_ratio=m.fusion.mlp_ratio, fusion_attn_ratio=m.fusion.attn_ratio, fusion_num_heads=m.fusion.num_heads)
    model = AVMAE(encoder, encoder.embed_dim,
                  image_decoder_arch=m.image.decoder_arch, image_decoder_depth=m.image.decoder_depth,
                  image_mask_ratio=m.image.mask_ratio, image_norm_loss=m.image.norm_loss,
                  audio_decoder_arch=m.audio.decoder_arch, audio_decoder_depth=m.audio.decoder_depth,
                  audio_mask_ratio=m.audio.mask_ratio, audio_norm_loss=m.audio.norm_loss)
    model.to(device)

    no_wd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]            # train.py:89
    groups = lr_sched.param_groups_pretrained(model, args.opt.weight_decay, no_weight_decay_list=no_wd,
                                              image_pt=m.image.pretrained, audio_pt=m.audio.pretrained)
    optimizer = FlatAdamW(groups, lr=args.opt.lr, betas=(0.9, 0.95), model=model)             # train.py:93
    trainer = misc_utils.Trainer(model, optimizer=optimizer, use_amp=args.opt.use_amp, accum_iter=args.opt.accum_iter,
                                 distributed=num_tasks > 1)
    ckpt = misc_utils.CheckpointManager(trainer.module_dict(), ckpt_dir=f'{job_dir}/checkpoints', epochs=args.opt.epochs,
                                        save_freq=args.log.save_freq)
    start_epoch = ckpt.resume()[0] if args.opt.resume else 0
    graphed = None
    if args.opt.get('graph', False) and args.opt.accum_iter == 1:
        B = args.opt.batch_size
        graphed = misc_utils.GraphedStep(trainer, (B, 3, *image_size), (B, 1, *audio_size), clip_grad=args.opt.clip_grad)

    frontend = None
    if gpu_frontend:
        from deepavfusion_amd.util.audio_transforms import LogMelSpectrogram
        frontend = LogMelSpectrogram(sample_rate=args.data.audio_rate, n_mels=args.data.audio_mels).to(device)
    print(f'Start training for {args.opt.epochs} epochs')
    for epoch in range(start_epoch, args.opt.epochs):
        if num_tasks > 1:
            loader.sampler.set_epoch(epoch)
        train_one_epoch(loader, trainer, epoch, device, args, graphed, frontend)
        if graphed is not None:
            graphed.check()          # never write a checkpoint behind a skipped (non-finite) captured step: raise like train.py:166-167
        ckpt.checkpoint(epoch + 1, {'epoch': epoch + 1})


def train_one_epoch(loader, trainer, epoch, device, args, graphed=None, frontend=None):
    from deepavfusion_amd.util import lr_sched
    trainer.model.train(True)
    trainer.zero_grad()
    t0, seen = time.time(), 0
    for step, (image, audio, _) in enumerate(loader):
        if step % args.opt.accum_iter == 0:
            lr = lr_sched.adjust_learning_rate(trainer.optimizer, epoch + step / len(loader), args)
        image = image.to(device, non_blocking=True).float()
        audio = audio.to(device, non_blocking=True).float()
        if frontend is not None:
            audio = frontend(audio)                 # [B, samples] -> [B, 1, n_mels, 64 * dur] on the device
        if graphed is not None:
            loss_image, loss_audio, grad_norm = graphed(image, audio)
            loss = loss_image + loss_audio
            amp_scale = 1.0
        else:
            with trainer.autocast(), trainer.autosync():
                loss_image, loss_audio = trainer.model(image, audio)[:2]
                loss = loss_image + loss_audio
            if not math.isfinite(loss.item()):
                raise RuntimeError(f'Loss is {loss.item()}, stopping training')
            grad_norm, amp_scale = trainer.step(loss, clip_grad=args.opt.clip_grad)
        seen += image.shape[0]
        if step % args.log.print_freq == 0 and trainer.accums == 0:
            if graphed is not None:
                graphed.check()                     # a captured step with a non-finite loss skipped its update on the device
            if not math.isfinite(float(loss)):
                raise RuntimeError(f'Loss is {float(loss)}, stopping training')
            print(f'[Train][Ep-{epoch}/{args.opt.epochs}] step {step}/{len(loader)}  loss {float(loss):.4f}  '
                  f'loss_image {float(loss_image):.4f}  loss_audio {float(loss_audio):.4f}  grad_norm {float(grad_norm):.3f}  '
                  f'amp_scale {amp_scale}  lr {lr:.3e}  {seen / (time.time() - t0):.1f} pairs/s/GPU')
        if args.debug and step == 100:
            break
    trainer.zero_grad()


def _spawned_worker(local_rank, cfg_dict):
    main_worker(local_rank, _wrap(cfg_dict))


def main(argv):
    """launcher.py:63-72 of the reference: one worker per GPU.  Under torch.distributed.run the ranks already exist;
    a plain ``python train.py env.ngpu=N`` (N > 1) spawns them itself; N == 1 runs in this process, no process group."""
    cfg = load_config('deepavfusion', [a for a in argv if '=' in a])
    under_torchrun = int(os.environ.get('WORLD_SIZE', '1')) > 1
    ngpu = int(cfg.env.ngpu or 1)
    if under_torchrun or ngpu <= 1:
        return main_worker(int(os.environ.get('LOCAL_RANK', '0')), cfg)
    cfg.env.spawned = True
    # launcher.py:80-86 of the reference: every job gets its own rendezvous (file:// under the job's output directory) so that two
    # jobs on one host do not meet on a fixed port; a dist_url given explicitly (multi-node: the same one on every node) is kept
    if not cfg.env.get('dist_url') or str(cfg.env.dist_url) == 'tcp://127.0.0.1:50000':
        jd = os.path.join(str(cfg.get('output_dir', 'checkpoints')), str(cfg.get('job_name', 'job')))
        os.makedirs(jd, exist_ok=True)
        url_file = os.path.join(os.path.abspath(jd), f'.dist_{os.getpid()}')
        if os.path.exists(url_file):
            os.remove(url_file)
        cfg.env.dist_url = 'file://' + url_file
    import torch.multiprocessing as mp
    mp.spawn(_spawned_worker, args=(_to_plain(cfg),), nprocs=ngpu)       # children start before anything touches the GPU here


def _to_plain(x):
    return {k: _to_plain(v) for k, v in x.items()} if isinstance(x, dict) else x


if __name__ == '__main__':
    main(sys.argv[1:])

"""GPU, two PROCESSES sharing the one MI355X, gradients exchanged over gloo (RCCL refuses two ranks on one device):
the whole data-parallel step — DataParallel wrapper, per-parameter "gradient final" hooks of the engine, bucketed
reduction, AdamW — in its eager form against a single process on the concatenated batch, and in its segmented-hipGraph
form (collectives between graph segments) for rank-identical results.  SURVEY.md section 8(e)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B_PER, STEPS = 8, 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(distributed, perturb=0.0):
    from deepavfusion_amd.build_model import build_avmae
    from deepavfusion_amd.configs import CONFIGS
    from deepavfusion_amd.util import lr_sched
    from deepavfusion_amd.util.flat import FlatAdamW
    from deepavfusion_amd.util.misc import Trainer
    from oracle import avmae_oracle as O
    from oracle.configs import CONFIGS as OC
    model = build_avmae(CONFIGS['micro']).cuda()
    model.load_state_dict(O.closed_form_state(OC['micro'], 0), strict=True)
    if perturb:                                  # a rank that was initialised differently: the wrapper must fix that
        with torch.no_grad():
            for q in model.parameters():
                if q.requires_grad:
                    q.add_(perturb * torch.randn_like(q))
    nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
    groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
    opt = FlatAdamW(groups, lr=1e-3, betas=(0.9, 0.95), model=model)
    tr = Trainer(model, optimizer=opt, accum_iter=1, distributed=distributed, bucket_mb=0.5, first_bucket_mb=0.25)
    return model, opt, tr, OC['micro'], O


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from deepavfusion_amd.util.misc import GraphedStep
    # ---- eager data-parallel steps on this rank's shard of the batch (masking noise injected) -----------------------
    model, opt, tr, cfg, O = _setup(True, perturb=0.05 * rank)        # rank 1 starts from different weights
    assert len(tr.model.reducer.buckets) >= 3            # (AVG natively or SUM + scale, whichever this backend offers)
    image, audio, ni, na = O.synthetic_batch(cfg, world * B_PER, seed=11)
    sl = slice(rank * B_PER, (rank + 1) * B_PER)
    im, au = image[sl].cuda(), audio[sl].cuda()
    n_i, n_a = torch.from_numpy(ni[sl]).cuda(), torch.from_numpy(na[sl]).cuda()
    losses = []
    for _ in range(STEPS):
        li, la = tr.model(im, au, n_i, n_a)[:2]
        tr.step(li + la)
        losses.append(float(li) + float(la))
        assert sorted(tr.model.reducer.launch_order) == list(range(len(tr.model.reducer.buckets)))   # every bucket, once
    torch.save({'p': opt.flat.flat_p.cpu(), 'losses': losses, 'order': list(tr.model.reducer.launch_order)},
               os.path.join(outdir, f'eager_{rank}.pt'))
    # ---- the segmented hipGraph form: collectives run between graph segments -----------------------------------------
    from deepavfusion_amd import engine
    engine.set_grad_ready_hook(None)
    model, opt, tr, cfg, O = _setup(True)
    image, audio, _, _ = O.structured_batch(cfg, world * 64, seed=5)
    im, au = image[rank * 64:(rank + 1) * 64].cuda(), audio[rank * 64:(rank + 1) * 64].cuda()
    gs = GraphedStep(tr, im.shape, au.shape)
    assert gs.dist_active and gs.n_seg >= 2
    sched = [bi for seg in gs.bucket_sched for bi in seg]
    assert sorted(sched) == list(range(len(gs.reducer.buckets)))
    p0 = opt.flat.flat_p.clone()
    run = []
    for s in range(4):
        torch.manual_seed(900 + 10 * s + rank)                      # per-rank masking noise, as in training
        li, la, gn = gs(im, au)
        run.append((float(li), float(la), float(gn)))
    torch.cuda.synchronize()
    assert float((opt.flat.flat_p - p0).abs().max()) > 0
    torch.save({'p': opt.flat.flat_p.cpu(), 'run': run, 'sched': gs.bucket_sched}, os.path.join(outdir, f'graph_{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_data_parallel_step(tmp_path):
    import torch.multiprocessing as mp
    world, port, outdir = 2, _free_port(), str(tmp_path)
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker, args=(r, world, port, outdir)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    e0, e1 = (torch.load(os.path.join(outdir, f'eager_{r}.pt')) for r in range(2))
    g0, g1 = (torch.load(os.path.join(outdir, f'graph_{r}.pt')) for r in range(2))
    # replicas stay bit-identical: same averaged gradients, same optimizer step on every rank
    assert torch.equal(e0['p'], e1['p']) and e0['order'] == e1['order']
    assert torch.equal(g0['p'], g1['p']) and g0['sched'] == g1['sched']
    assert g0['run'][-1][0] + g0['run'][-1][1] < g0['run'][0][0] + g0['run'][0][1]                    # it trains
    assert all(abs(a[2] - b[2]) < 1e-5 * a[2] for a, b in zip(g0['run'], g1['run']))                  # same reduced grad norm

    # single process, concatenated batch, same injected noise: the mean loss of the two shards and the same update
    model, opt, tr, cfg, O = _setup(False)
    p_init = opt.flat.flat_p.clone().cpu()
    image, audio, ni, na = O.synthetic_batch(cfg, world * B_PER, seed=11)
    ref_losses = []
    for _ in range(STEPS):
        li, la = tr.model(image.cuda(), audio.cuda(), torch.from_numpy(ni).cuda(), torch.from_numpy(na).cuda())[:2]
        tr.step(li + la)
        ref_losses.append(float(li) + float(la))
    p_ref = opt.flat.flat_p.cpu()
    for s in range(STEPS):
        mean_dp = 0.5 * (e0['losses'][s] + e1['losses'][s])      # equal mask counts per sample -> mean of shard losses
        assert abs(mean_dp - ref_losses[s]) < 2e-3 * ref_losses[s], (s, mean_dp, ref_losses[s])
    upd = float((p_ref - p_init).norm())
    assert float((e0['p'] - p_ref).norm()) < 0.05 * upd, (float((e0['p'] - p_ref).norm()), upd)

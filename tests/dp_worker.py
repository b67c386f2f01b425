"""Worker process of tests/test_dp_rccl_gpu.py (not collected by pytest).  Started as a FRESH python process
(subprocess, its own session) so that nothing here shares GPU state with the pytest parent, writes one line per
phase to <outdir>/rank<r>.log, arms faulthandler so a hang leaves a traceback, and always exits.

    python tests/dp_worker.py rccl1 <outdir> <port>                 # 1-rank RCCL group on cuda:0 (DAV_FORCE_DIST=1)
    python tests/dp_worker.py gloo2 <outdir> <port> <rank>          # opt-in: two processes on one GPU over gloo

Both drive the data-parallel form of the pre-training step the way train.py does (reference util/misc.py:32-34,
train.py:151-180): DataParallel wrapper + bucketed reducer + AdamW eagerly, then the segmented-hipGraph step with the
collectives between the graph segments."""
import datetime
import faulthandler
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODE, OUTDIR, PORT = sys.argv[1], sys.argv[2], sys.argv[3]
RANK = int(sys.argv[4]) if len(sys.argv) > 4 else 0
WORLD = 2 if MODE == 'gloo2' else 1
_LOG = open(os.path.join(OUTDIR, f'rank{RANK}.log'), 'w', buffering=1)
_TRACE = open(os.path.join(OUTDIR, f'rank{RANK}.trace'), 'w', buffering=1)
faulthandler.enable(file=_TRACE)
faulthandler.dump_traceback_later(int(os.environ.get('DAV_WORKER_DUMP_S', '150')), repeat=False, file=_TRACE, exit=True)
_T0 = time.time()


def phase(msg):
    _LOG.write(f'[{time.time() - _T0:7.2f}s] {msg}\n')


os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=PORT, RANK=str(RANK), WORLD_SIZE=str(WORLD),
                  HSA_ENABLE_IPC_MODE_LEGACY='0')
if MODE == 'rccl1':
    os.environ['DAV_FORCE_DIST'] = '1'

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

B_PER, STEPS = 8, 2


def setup(distributed, perturb=0.0, accum_iter=1):
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
    tr = Trainer(model, optimizer=opt, accum_iter=accum_iter, distributed=distributed, bucket_mb=0.5, first_bucket_mb=0.25)
    return model, opt, tr, OC['micro'], O


def eager_steps(tr, batch, steps):
    im, au, n_i, n_a = batch
    losses = []
    for _ in range(steps):
        li, la = tr.model(im, au, n_i, n_a)[:2]
        tr.step(li + la)
        losses.append(float(li.detach()) + float(la.detach()))
    return losses


def main_rccl1():
    from deepavfusion_amd import engine
    from deepavfusion_amd.util.misc import GraphedStep
    out = {}
    phase('init_process_group nccl')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, timeout=datetime.timedelta(seconds=60))
    phase('process group up')
    # ---- (1) eager: DataParallel + reducer with real RCCL all-reduces on the comm stream == the plain step --------
    res = []
    for distributed in (False, True):
        engine.set_grad_ready_hook(None)
        model, opt, tr, cfg, O = setup(distributed)
        image, audio, ni, na = O.synthetic_batch(cfg, B_PER, seed=11)
        batch = (image.cuda(), audio.cuda(), torch.from_numpy(ni).cuda(), torch.from_numpy(na).cuda())
        phase(f'eager steps distributed={distributed}')
        losses = eager_steps(tr, batch, STEPS)
        torch.cuda.synchronize()
        if distributed:
            red = tr.model.reducer
            assert red.force and red.world == 1
            out['n_buckets'] = len(red.buckets)
            out['launch_order'] = list(red.launch_order)
            out['use_avg'] = bool(red.use_avg)
        res.append((losses, opt.flat.flat_p.detach().clone()))
    (l0, p0), (l1, p1) = res
    out['eager_losses'] = [l0, l1]
    out['eager_param_rel'] = float((p1 - p0).norm() / p0.norm())
    out['eager_bit_equal'] = bool(torch.equal(p0, p1))
    phase(f"eager done rel {out['eager_param_rel']:.3e}")
    # ---- (2) gradient accumulation, the train.py pattern: forward inside autosync(), backward in Trainer.step ------
    engine.set_grad_ready_hook(None)
    model, opt, tr, cfg, O = setup(True, accum_iter=2)
    image, audio, ni, na = O.synthetic_batch(cfg, B_PER, seed=12)
    batch = (image.cuda(), audio.cuda(), torch.from_numpy(ni).cuda(), torch.from_numpy(na).cuda())
    launches = []
    for micro in range(2):
        with tr.autocast(), tr.autosync():
            li, la = tr.model(*batch)[:2]
        tr.step(li + la)
        launches.append(len(tr.model.reducer.launch_order))
    torch.cuda.synchronize()
    out['accum_launches'] = launches
    out['accum_n_steps'] = int(tr.n_steps)
    phase(f'accumulation done {launches}')
    # ---- (3) the segmented hipGraph step, collectives between the segments, vs the single graph -------------------
    finals, runs = [], []
    for distributed in (False, True):
        engine.set_grad_ready_hook(None)
        model, opt, tr, cfg, O = setup(distributed)
        image, audio, _, _ = O.structured_batch(cfg, 64, seed=5)
        im, au = image.cuda(), audio.cuda()
        phase(f'capture distributed={distributed}')
        gs = GraphedStep(tr, im.shape, au.shape)
        if distributed:
            assert gs.dist_active and gs.n_seg >= 2
            sched = [bi for seg in gs.bucket_sched for bi in seg]
            out['graph_segments'] = gs.n_seg
            out['graph_sched_complete'] = sorted(sched) == list(range(len(gs.reducer.buckets)))
        else:
            assert gs.n_seg == 1
        run = []
        for s in range(4):
            torch.manual_seed(900 + 10 * s)
            li, la, gn = gs(im, au)
            run.append((float(li), float(la), float(gn)))
        torch.cuda.synchronize()
        phase(f'replays done distributed={distributed}')
        runs.append(run)
        finals.append(opt.flat.flat_p.detach().clone())
    out['graph_runs'] = runs
    out['graph_param_rel'] = float((finals[1] - finals[0]).norm() / finals[0].norm())
    with open(os.path.join(OUTDIR, 'result0.json'), 'w') as f:
        json.dump(out, f)
    phase('result written')
    dist.destroy_process_group()
    phase('done')


def main_gloo2():
    from deepavfusion_amd import engine
    from deepavfusion_amd.util.misc import GraphedStep
    phase('init_process_group gloo')
    dist.init_process_group('gloo', rank=RANK, world_size=WORLD, timeout=datetime.timedelta(seconds=60))
    torch.cuda.set_device(0)
    phase('process group up')
    model, opt, tr, cfg, O = setup(True, perturb=0.05 * RANK)        # rank 1 starts from different weights
    assert len(tr.model.reducer.buckets) >= 3
    image, audio, ni, na = O.synthetic_batch(cfg, WORLD * B_PER, seed=11)
    sl = slice(RANK * B_PER, (RANK + 1) * B_PER)
    batch = (image[sl].cuda(), audio[sl].cuda(), torch.from_numpy(ni[sl]).cuda(), torch.from_numpy(na[sl]).cuda())
    phase('eager steps')
    losses = eager_steps(tr, batch, STEPS)
    torch.cuda.synchronize()
    torch.save({'p': opt.flat.flat_p.cpu(), 'losses': losses, 'order': list(tr.model.reducer.launch_order)},
               os.path.join(OUTDIR, f'eager_{RANK}.pt'))
    phase('eager done')
    engine.set_grad_ready_hook(None)
    model, opt, tr, cfg, O = setup(True)
    image, audio, _, _ = O.structured_batch(cfg, WORLD * 64, seed=5)
    im, au = image[RANK * 64:(RANK + 1) * 64].cuda(), audio[RANK * 64:(RANK + 1) * 64].cuda()
    phase('capture')
    gs = GraphedStep(tr, im.shape, au.shape)
    assert gs.dist_active and gs.n_seg >= 2
    sched = [bi for seg in gs.bucket_sched for bi in seg]
    run = []
    for s in range(4):
        torch.manual_seed(900 + 10 * s + RANK)
        li, la, gn = gs(im, au)
        phase(f'replay {s} enqueued')
        run.append((float(li), float(la), float(gn)))
    torch.cuda.synchronize()
    torch.save({'p': opt.flat.flat_p.cpu(), 'run': run, 'sched': sched}, os.path.join(OUTDIR, f'graph_{RANK}.pt'))
    phase('graph done')
    dist.barrier()
    dist.destroy_process_group()
    phase('done')


if __name__ == '__main__':
    try:
        (main_rccl1 if MODE == 'rccl1' else main_gloo2)()
    except BaseException as e:                                   # noqa: BLE001
        import traceback
        phase('FAILED ' + repr(e))
        traceback.print_exc(file=_TRACE)
        _TRACE.flush()
        os._exit(1)
    faulthandler.cancel_dump_traceback_later()
    os._exit(0)          # skip interpreter teardown: nothing after the results may hang this process

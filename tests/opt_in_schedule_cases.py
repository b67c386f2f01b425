"""Cases of tests/test_hip_parity.py::test_opt_in_schedules_in_a_process_of_their_own (collected only when named explicitly: the file
name carries no test_ prefix).  See that test for why they have a process to themselves."""
import numpy as np
import pytest
import torch

from test_hip_parity import _build, rel          # noqa: E402  (tests/ is on sys.path: conftest.py)

pytestmark = pytest.mark.gpu


def test_early_adamw_ranges_equal_the_single_pass(monkeypatch):
    """DAV_EARLY_ADAMW=1 (opt-in, measured slower in the step: DESIGN.md section 4): the captured step runs AdamW on the ranges of
    the flat buffer whose gradients are final at a few points of the backward, on a side stream, and on the rest at the end.
    The update is element-wise: the same seeds must give the same parameters, moments, losses and gradient norm as the single
    pass at the end — and every parameter must be covered exactly once."""
    from deepavfusion_amd.util import lr_sched
    from deepavfusion_amd.util.flat import FlatAdamW
    from deepavfusion_amd.util.misc import GraphedStep, Trainer
    finals, moments, losses, norms = [], [], [], []
    for mode in ('0', '1'):
        monkeypatch.setenv('DAV_EARLY_ADAMW', mode)
        monkeypatch.setenv('DAV_EARLY_ADAMW_CUTS', '2,1')         # micro is 2 layers deep: after the decoders (2) and after layer 1
        model, sd, cfg, O = _build('micro')
        nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
        groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
        opt = FlatAdamW(groups, lr=1e-3, betas=(0.9, 0.95), model=model)
        tr = Trainer(model, optimizer=opt, accum_iter=1)
        image, audio, _, _ = O.structured_batch(cfg, 64, seed=3)
        image, audio = image.cuda(), audio.cuda()
        torch.manual_seed(77)
        gs = GraphedStep(tr, image.shape, audio.shape)
        assert gs.early == (mode == '1')
        if gs.early:
            covered = sorted(i for rs in list(gs.early_ranges.values()) + [gs.late_ranges] for r in rs for i in range(r['first'], r['first'] + r['n']))
            assert covered == list(range(len(opt.flat.params)))
            assert sum(len(rs) for rs in gs.early_ranges.values()) >= 2 and sum(r['n'] for rs in gs.early_ranges.values() for r in rs) > len(covered) // 2
        run = []
        for s in range(5):
            torch.manual_seed(500 + s)
            li, la, gn = gs(image, audio)
            run.append(float(li) + float(la))
        torch.cuda.synchronize()
        gs.check()
        finals.append(opt.flat.flat_p.clone()); moments.append(opt.exp_avg_sq.clone()); losses.append(run); norms.append(float(gn))
    assert all(np.isfinite(losses[1])) and losses[1][-1] < losses[1][0]
    for a, b in zip(*losses):
        assert abs(a - b) <= 5e-5 * abs(b), losses          # (two samples of the bias gradients' atomic-add noise, amplified by AdamW)
    assert abs(norms[0] - norms[1]) <= 1e-4 * norms[0], norms
    assert rel(finals[1], finals[0]) < 2e-4 and rel(moments[1], moments[0]) < 2e-3      # (fp32 atomics in the bias gradients: not bit-equal)


def test_deferred_adamw_equals_the_plain_schedule(monkeypatch):
    """DAV_DEFER_ADAMW=1 / GraphedStep(defer=True): the update with step i's gradients is issued at the top of replay i + 1, layer by
    layer on a side stream under that replay's forward (each forward stage gated on its own parameters' chunk).  The update is
    element-wise and every forward sees the same parameters as in the plain schedule: the same seeds — with a learning rate that
    changes every step, and a ``flush()`` in the middle of the run — must give the same losses, and after the final ``flush()`` the
    same parameters and moments; the reported gradient norm is the previous step's; no replay counts as skipped."""
    from deepavfusion_amd.util import lr_sched
    from deepavfusion_amd.util.flat import FlatAdamW
    from deepavfusion_amd.util.misc import GraphedStep, Trainer
    finals, moments, losses, norms, mirrors = [], [], [], [], []
    for defer in (False, True):
        model, sd, cfg, O = _build('micro')
        nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
        groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
        opt = FlatAdamW(groups, lr=1e-3, betas=(0.9, 0.95), model=model)
        tr = Trainer(model, optimizer=opt, accum_iter=1)
        image, audio, _, _ = O.structured_batch(cfg, 64, seed=3)
        image, audio = image.cuda(), audio.cuda()
        torch.manual_seed(77)
        gs = GraphedStep(tr, image.shape, audio.shape, defer=defer)
        assert gs.defer == defer
        if defer:
            covered = sorted(i for rs in gs.chunks for r in rs for i in range(r['first'], r['first'] + r['n']))
            assert covered == list(range(len(opt.flat.params)))                       # every parameter in exactly one chunk
            assert len([rs for rs in gs.chunks if rs]) >= 3                           # embeddings + layer 0 | layer 1 | decoders
            ks = [gs.stage_chunk[st] for st in sorted(gs.stage_chunk)]
            assert ks == sorted(ks) and ks[0] >= 0, gs.stage_chunk                    # the forward meets the chunks in update order
        run, gns = [], []
        for s in range(6):
            torch.manual_seed(500 + s)
            for g in opt.param_groups:
                g['lr'] = 1e-3 * (1.0 + 0.5 * s)                                       # (a schedule: the update must use ITS step's value)
            if s == 4:                                                                # an EAGER step in between: the pending update must be applied first
                li, la = tr.model(image, audio)[:2]
                tr.step(li + la)
                gn = torch.zeros(())
            else:
                li, la, gn = gs(image, audio)
            run.append(float(li) + float(la)); gns.append(float(gn))
            if s == 2:
                gs.flush()                                                            # e.g. a checkpoint in the middle of an epoch
                gs.flush()                                                            # (idempotent)
        gs.flush()
        torch.cuda.synchronize()
        gs.check()
        assert int(opt.step_count) == 6
        finals.append(opt.flat.flat_p.clone()); moments.append(opt.exp_avg_sq.clone()); losses.append(run); norms.append(gns)
        mirrors.append(opt.flat_bf16.clone())
    assert all(np.isfinite(losses[1])) and losses[1][-1] < losses[1][0]
    for a, b in zip(*losses):
        assert abs(a - b) <= 5e-5 * abs(b), losses          # (two samples of the bias gradients' atomic-add noise, amplified by AdamW)
    for s in (1, 2):                                                                  # (call 3 follows a flush, 4 is eager, 5 follows it: nothing pending there)
        assert abs(norms[1][s] - norms[0][s - 1]) <= 1e-4 * norms[0][s - 1], norms
    assert rel(finals[1], finals[0]) < 2e-4 and rel(moments[1], moments[0]) < 2e-3      # (fp32 atomics in the bias gradients: not bit-equal)
    assert rel(mirrors[1].float(), mirrors[0].float()) < 2e-3


def test_wgrad_side_stream_equals_the_serial_placement(monkeypatch):
    """DAV_WGRAD_SIDE=1: inside a captured step a layer's grouped weight-gradient launch is a parallel branch of the graph (side
    stream, joined before the optimizer and at every graph cut) instead of a node between two layers of the input-gradient chain.
    Same arithmetic, other placement: same seeds must give the same losses, gradient norm, parameters and moments — also with the
    step cut into segments (where the join has to come before the cut)."""
    from deepavfusion_amd.util import lr_sched
    from deepavfusion_amd.util.flat import FlatAdamW
    from deepavfusion_amd.util.misc import GraphedStep, Trainer
    finals, moments, losses, norms = [], [], [], []
    for mode, segments in (('0', 1), ('1', 1), ('1', 3)):
        monkeypatch.setenv('DAV_WGRAD_SIDE', mode)
        model, sd, cfg, O = _build('micro')
        nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
        groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
        opt = FlatAdamW(groups, lr=1e-3, betas=(0.9, 0.95), model=model)
        tr = Trainer(model, optimizer=opt, accum_iter=1)
        image, audio, _, _ = O.structured_batch(cfg, 64, seed=3)
        image, audio = image.cuda(), audio.cuda()
        torch.manual_seed(77)
        gs = GraphedStep(tr, image.shape, audio.shape, segments=segments)
        run = []
        for s in range(5):
            torch.manual_seed(500 + s)
            li, la, gn = gs(image, audio)
            run.append(float(li) + float(la))
        torch.cuda.synchronize()
        gs.check()
        finals.append(opt.flat.flat_p.clone()); moments.append(opt.exp_avg_sq.clone()); losses.append(run); norms.append(float(gn))
    assert all(np.isfinite(losses[1])) and losses[1][-1] < losses[1][0]
    for k in (1, 2):
        for a, b in zip(losses[0], losses[k]):
            assert abs(a - b) <= 5e-5 * abs(a), losses      # (two samples of the bias gradients' atomic-add noise, amplified by AdamW)
        assert abs(norms[0] - norms[k]) <= 1e-4 * norms[0], norms
        assert rel(finals[k], finals[0]) < 2e-4 and rel(moments[k], moments[0]) < 2e-3      # (fp32 atomics in the bias gradients: not bit-equal)




def test_fused_adamw_equals_the_optimizer_kernel(monkeypatch):
    """GraphedStep(fuse=True) / DAV_FUSED_ADAMW=1: a Linear weight whose one weight-gradient problem of the step is a written tile set gets
    its AdamW update from the workgroups that own those tiles (dav_gemm_tn_grouped_adamw_bf16: the gradient is never stored, the optimizer
    kernel skips the weight).  Same arithmetic in the same order: same seeds (with a learning rate that changes every step) must give the
    same losses, gradient norm, parameters, moments and bf16 mirror as the optimizer kernel alone; most Linear weights must really have
    taken the fused route, a weight with two contributions per step (decoder_embed) must not; a replay with a non-finite loss leaves
    everything untouched here too, and ``check()`` raises."""
    from deepavfusion_amd.util import lr_sched
    from deepavfusion_amd.util.flat import FlatAdamW
    from deepavfusion_amd.util.misc import GraphedStep, Trainer
    finals, moments, losses, norms, mirrors = [], [], [], [], []
    for fuse in (False, True):
        model, sd, cfg, O = _build('micro')
        nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
        groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
        opt = FlatAdamW(groups, lr=1e-3, betas=(0.9, 0.95), model=model)
        tr = Trainer(model, optimizer=opt, accum_iter=1)
        image, audio, _, _ = O.structured_batch(cfg, 64, seed=3)
        image, audio = image.cuda(), audio.cuda()
        torch.manual_seed(77)
        gs = GraphedStep(tr, image.shape, audio.shape, fuse=fuse)
        assert gs.fuse == fuse
        if fuse:
            n_linear = sum(1 for n, p in model.named_parameters() if p.ndim == 2 and p.requires_grad and 'embed' not in n and 'token' not in n)
            assert gs.fused_params > n_linear // 2, (gs.fused_params, n_linear)
            names = {id(p): n for n, p in model.named_parameters()}
            kg = gs.keep_grad.cpu().tolist()
            fused_names = [names[id(p)] for p, k in zip(opt.flat.params, kg) if k & 2]
            assert not any('decoder_embed' in n for n in fused_names), fused_names      # two contributions per step: stays with the kernel
            assert all(k in (0, 1, 3) for k in kg)
        run, gns = [], []
        for s in range(6):
            torch.manual_seed(500 + s)
            for g in opt.param_groups:
                g['lr'] = 1e-3 * (1.0 + 0.5 * s)
            li, la, gn = gs(image, audio)
            run.append(float(li) + float(la)); gns.append(float(gn))
        torch.cuda.synchronize()
        gs.check()
        finals.append(opt.flat.flat_p.clone()); moments.append((opt.exp_avg.clone(), opt.exp_avg_sq.clone())); losses.append(run); norms.append(gns)
        mirrors.append(opt.flat_bf16.clone())
        if fuse:                                       # the guard: a NaN input -> nothing moves, the host check raises
            before = (opt.flat.flat_p.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.flat_bf16.clone())
            bad = image.clone(); bad[0, 0, 0, 0] = float('nan')
            gs(bad, audio)
            torch.cuda.synchronize()
            after = (opt.flat.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.flat_bf16)
            assert all(torch.equal(a, b) for a, b in zip(before, after))
            with pytest.raises(RuntimeError):
                gs.check()
    assert all(np.isfinite(losses[1])) and losses[1][-1] < losses[1][0]
    for a, b in zip(*losses):
        assert abs(a - b) <= 5e-5 * abs(b), losses
    for a, b in zip(*norms):
        assert abs(a - b) <= 1e-4 * abs(b), norms
    assert rel(finals[1], finals[0]) < 2e-4 and rel(moments[1][0], moments[0][0]) < 2e-3 and rel(moments[1][1], moments[0][1]) < 2e-3
    assert rel(mirrors[1].float(), mirrors[0].float()) < 2e-3

"""Per-iteration lr schedule and optimizer param groups (reference util/lr_sched.py:4-24, 77-93; train.py:89-93)."""
import math


def _get(opt, key, default):
    return opt.get(key, default) if hasattr(opt, 'get') else getattr(opt, key, default)


def adjust_learning_rate(optimizer, epoch, args):
    """Linear warm-up then half-cosine; groups tagged ``pretrained`` get an extra 0->1 cosine ramp over
    ``pt_warmup_epochs`` (a string such as "300/2" is evaluated arithmetically, util/lr_sched.py:12)."""
    o = args.opt
    wu = _get(o, 'warmup_epochs', 0)
    if epoch < wu:
        lr = o.lr * epoch / wu
    else:
        lr = o.lr * 0.5 * (1. + math.cos(math.pi * (epoch - wu) / (o.epochs - wu)))
    ptw = _get(o, 'pt_warmup_epochs', -1)
    if isinstance(ptw, str):
        num, _, den = ptw.partition('/')
        ptw = float(num) / float(den) if den else float(num)
    end = _get(o, 'pt_lr_mult_end', 1.)
    if epoch < ptw:
        start = _get(o, 'pt_lr_mult_start', 0.)
        pt_scale = (0.5 - 0.5 * math.cos(math.pi * epoch / ptw)) * (end - start) + start
    else:
        pt_scale = end
    for g in optimizer.param_groups:
        scale = g.get('lr_scale', 1.)
        g['lr'] = lr * scale * (pt_scale if g.get('pretrained', False) else 1.)
    return lr


def _wd_groups(named_params, weight_decay, no_decay_names):
    """timm optim_factory.param_groups_weight_decay: no decay for ndim<=1, '*.bias' and listed names."""
    decay, no_decay = [], []
    for name, p in named_params:
        if not p.requires_grad:
            continue
        (no_decay if (p.ndim <= 1 or name.endswith('.bias') or name in no_decay_names) else decay).append(p)
    return [{'params': no_decay, 'weight_decay': 0.}, {'params': decay, 'weight_decay': weight_decay}]


def param_groups_pretrained(model, weight_decay=0.05, no_weight_decay_list=(), image_pt=None, audio_pt=None):
    """util/lr_sched.py:77-93.  A tower is tagged pretrained whenever its ``pretrained`` setting is not None —
    the empty string included."""
    nd = set(no_weight_decay_list)
    groups = _wd_groups(model.named_parameters(), weight_decay, nd)
    pt = []
    if image_pt is not None:
        pt += _wd_groups(model.encoder.image.named_parameters(), weight_decay, nd)
    if audio_pt is not None:
        pt += _wd_groups(model.encoder.audio.named_parameters(), weight_decay, nd)
    for g in pt:
        g['pretrained'] = True
    taken = {id(p) for g in pt for p in g['params']}
    for g in groups:
        g['params'] = [p for p in g['params'] if id(p) not in taken]
    return groups + pt

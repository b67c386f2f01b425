"""CPU, world_size 2 over gloo: the bucketed gradient reducer (the DDP replacement of util/misc.py:32-34)
averages the flat gradient buffer, launches buckets in the same order on every rank, honours no_sync()."""
import datetime
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from deepavfusion_amd.util.distributed import DataParallel, get_rank, get_world_size
    from deepavfusion_amd.util.flat import FlatParams
    torch.manual_seed(100 + rank)                     # different init per rank: the wrapper must broadcast rank 0's
    model = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.LayerNorm(300), torch.nn.Linear(300, 300), torch.nn.Linear(300, 7))
    flat = FlatParams(reversed(list(model.parameters())))
    dp = DataParallel(model, flat, bucket_mb=0.02, first_bucket_mb=0.005)
    dp.reducer.comm_stream = None
    assert get_world_size() == world and get_rank() == rank
    ref = [p.detach().clone() for p in flat.params]
    gathered = [torch.zeros_like(flat.flat_p) for _ in range(world)]
    dist.all_gather(gathered, flat.flat_p)
    assert all(torch.equal(g, gathered[0]) for g in gathered), 'parameters not broadcast'
    assert len(dp.reducer.buckets) >= 3

    def fake_backward(scale):
        dp.reducer.begin_backward()
        for i, p in enumerate(flat.params):           # the engine's order == flat order
            p.grad.add_(torch.full_like(p, scale * (rank + 1) * (i + 1)))
            dp.reducer.grad_ready(p)
        dp.reducer.finish()
    fake_backward(1.0)
    mean_rank = sum(r + 1 for r in range(world)) / world
    for i, p in enumerate(flat.params):
        assert torch.allclose(p.grad, torch.full_like(p, mean_rank * (i + 1))), i
    order1 = list(dp.reducer.launch_order)
    assert order1 == sorted(order1) and len(order1) == len(dp.reducer.buckets)
    # gradient accumulation: first micro-step under no_sync() stays local, second one reduces the sum
    flat.zero_grad()
    with dp.no_sync():
        fake_backward(1.0)
    for i, p in enumerate(flat.params):
        assert torch.allclose(p.grad, torch.full_like(p, float((rank + 1) * (i + 1))))
    fake_backward(2.0)
    for i, p in enumerate(flat.params):
        assert torch.allclose(p.grad, torch.full_like(p, 3.0 * mean_rank * (i + 1)))
    # the train.py pattern (train.py:166-170 of the reference): only the FORWARD sits inside no_sync(), the backward runs
    # later in Trainer.step — the sync decision must have been latched at forward time, as DistributedDataParallel does
    flat.zero_grad()
    with dp.no_sync():
        dp.reducer.begin_backward()                   # = DataParallel.forward
    for i, p in enumerate(flat.params):
        p.grad.add_(torch.full_like(p, float((rank + 1) * (i + 1))))
        dp.reducer.grad_ready(p)
    dp.reducer.finish()
    assert dp.reducer.launch_order == []              # nothing was reduced
    for i, p in enumerate(flat.params):
        assert torch.allclose(p.grad, torch.full_like(p, float((rank + 1) * (i + 1))))
    dp.reducer.begin_backward()                       # next micro-step, forward outside no_sync(): reduces the sum
    for i, p in enumerate(flat.params):
        p.grad.add_(torch.full_like(p, 2.0 * (rank + 1) * (i + 1)))
        dp.reducer.grad_ready(p)
    dp.reducer.finish()
    for i, p in enumerate(flat.params):
        assert torch.allclose(p.grad, torch.full_like(p, 3.0 * mean_rank * (i + 1)))
    # N-rank averaged gradients == single-process gradient of the concatenated batch (mean loss)
    flat.zero_grad()
    torch.manual_seed(7)
    xs = torch.randn(world * 4, 40)
    x = xs[rank * 4:(rank + 1) * 4]
    dp.reducer.begin_backward()
    model(x).pow(2).mean().backward()
    for p in flat.params:
        dp.reducer.grad_ready(p)
    dp.reducer.finish()
    g_dp = flat.flat_g.clone()
    flat.zero_grad()
    model(xs).pow(2).mean().backward()
    assert torch.allclose(g_dp, flat.flat_g, rtol=1e-4, atol=1e-6)
    q.put((rank, order1))
    dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.daemon = True
        p.start()
    try:
        for p in procs:
            p.join(timeout=180)
            assert p.exitcode == 0
    finally:
        for p in procs:                               # never leave a worker behind: pytest must always be able to exit
            if p.is_alive():
                p.terminate()
                p.join(5)
                if p.is_alive():
                    p.kill()
    res = dict(q.get(timeout=5) for _ in range(world))
    assert res[0] == res[1]          # identical collective order on every rank


def test_segment_cuts_are_pure_and_decoders_first():
    from deepavfusion_amd.util.misc import segment_cuts
    assert segment_cuts(12, 1) == []
    assert segment_cuts(12, 2) == [12]
    assert segment_cuts(12, 5) == [12, 7, 3, 1]                 # [decoders] | 11..7 | 6..3 | 2..1 | 0
    for depth in (2, 12, 24):
        for seg in range(2, depth + 3):
            c = segment_cuts(depth, seg)
            assert c[0] == depth and c == sorted(set(c), reverse=True) and all(0 < x <= depth for x in c) and len(c) <= seg - 1
    for depth, seg in ((12, 5), (24, 6), (12, 4)):              # ever fewer layers left behind a cut (up to rounding at many cuts)
        c = segment_cuts(depth, seg)
        gaps = [a - b for a, b in zip(c, c[1:] + [0])][1:]
        assert gaps == sorted(gaps, reverse=True), (depth, seg, c)


def _sched_worker(rank, world, port, q):
    """The segmented step's collective schedule without a GPU: each 'graph segment' finishes a contiguous range of
    parameters (the flat buffer is in backward order), then its completed buckets are all-reduced — exactly what
    GraphedStep.__call__ does between replayed segments (launch_buckets per segment, finish() before the optimizer)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from deepavfusion_amd.util.distributed import DataParallel
    from deepavfusion_amd.util.flat import FlatParams
    from deepavfusion_amd.util.misc import segment_cuts
    torch.manual_seed(5)
    depth = 6
    layers = [torch.nn.Linear(48, 48) for _ in range(depth)]
    dec = torch.nn.Linear(48, 200)
    model = torch.nn.Sequential(*layers, dec)
    flat = FlatParams(reversed(list(model.parameters())))       # decoder first: the order the backward finishes gradients
    dp = DataParallel(model, flat, bucket_mb=0.012, first_bucket_mb=0.004)
    dp.reducer.comm_stream = None
    red = dp.reducer
    cuts = segment_cuts(depth, 4)
    # capture-time bookkeeping: which buckets does each segment complete?
    pending = [len(b[2]) for b in red.buckets]
    sched, seg = [[] for _ in range(len(cuts) + 1)], 0
    order = [(depth, list(dec.parameters()))] + [(l, list(layers[l].parameters())) for l in reversed(range(depth))]
    for l, params in order:
        for p in reversed(params):
            bi = red._bucket_of[id(p)]
            pending[bi] -= 1
            if pending[bi] == 0:
                sched[seg].append(bi)
        if l in cuts:
            seg += 1
    assert sorted(b for s in sched for b in s) == list(range(len(red.buckets))) and sched[0] and 0 in sched[0]
    # replay: gradients appear segment by segment, buckets are reduced between segments
    red.begin_backward()
    seg = 0
    for l, params in order:
        for i, p in enumerate(params):
            p.grad.add_(torch.full_like(p, float((rank + 1) * (l + 1) + i)))
        if l in cuts:
            red.launch_buckets(sched[seg])
            seg += 1
    red.launch_buckets(sched[seg])
    red.finish()
    mean_rank = sum(r + 1 for r in range(world)) / world
    for l, params in order:
        for i, p in enumerate(params):
            assert torch.allclose(p.grad, torch.full_like(p, mean_rank * (l + 1) + i)), (l, i)
    q.put((rank, list(red.launch_order), sched))
    dist.destroy_process_group()


def test_segment_schedule_world4_gloo():
    world, port = 4, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_sched_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.daemon = True
        p.start()
    try:
        for p in procs:
            p.join(timeout=180)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(5)
                if p.is_alive():
                    p.kill()
    res = [q.get(timeout=5) for _ in range(world)]
    assert all(r[1] == res[0][1] and r[2] == res[0][2] for r in res)      # same collective order and schedule on every rank
    assert res[0][1] == [b for s in res[0][2] for b in s]                 # buckets go out segment by segment


def _avmae_equiv_worker(rank, world, port, q):
    """SURVEY.md section 8(e) (reference util/misc.py:32-34, 144-148): N ranks, each with its own slice of a global batch and the same
    injected masking noise, through the real DataParallel wrapper + GradReducer + the captured step's segment schedule
    (gradients appear segment by segment — decoders, then encoder layers last to first — and the buckets a segment completes are
    all-reduced between segments)  ==  one rank on the concatenated batch.  The AVMAE arithmetic on the CPU is the oracle's (the
    product model has no CPU path); parameters, flat buffers, buckets and collectives are the product's."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    import re
    from deepavfusion_amd.build_model import build_avmae
    from deepavfusion_amd.configs import CONFIGS
    from deepavfusion_amd.util.distributed import DataParallel
    from deepavfusion_amd.util.flat import FlatParams
    from deepavfusion_amd.util.misc import segment_cuts
    from oracle import avmae_oracle as O
    from oracle.configs import CONFIGS as OC
    cfg = OC['micro']
    model = build_avmae(CONFIGS['micro'])
    sd0 = O.closed_form_state(cfg, 0)
    if rank:                                          # a rank that was initialised differently: the wrapper broadcasts rank 0's
        sd0 = {k: (v + 0.01 if v.is_floating_point() and k not in O.FROZEN else v) for k, v in sd0.items()}
    model.load_state_dict(sd0, strict=True)
    flat = FlatParams(reversed(list(model.parameters())))
    dp = DataParallel(model, flat, bucket_mb=0.05, first_bucket_mb=0.01)
    dp.reducer.comm_stream = None
    red = dp.reducer
    names = {id(p): n for n, p in model.named_parameters()}
    depth = len(model.encoder.image.blocks)
    cuts = segment_cuts(depth, depth + 1)             # a cut behind the decoders and behind every layer but the last

    def layer_of(n):                                  # where the backward finishes a gradient: decoders first, layer 0 and the embeddings last
        if not n.startswith('encoder.'):
            return depth
        m = re.search(r'blocks\.(\d+)\.', n)
        return int(m.group(1)) if m else (depth - 1 if 'norm' in n else 0)
    B_per = 3
    image, audio, ni, na = O.synthetic_batch(cfg, B_per * world, seed=91)

    def oracle_grads(sl):
        sd = {n: p.detach().clone().requires_grad_(p.requires_grad) for n, p in model.named_parameters()}
        sd.update({k: v for k, v in model.state_dict().items() if k not in sd})
        li, la, _, _, _ = O.avmae_forward(sd, cfg, image[sl], audio[sl], ni[sl], na[sl])
        (li + la).backward()
        return {n: v.grad for n, v in sd.items() if getattr(v, 'grad', None) is not None}, float(li + la)
    # capture-time bookkeeping of GraphedStep: which buckets does each segment complete?
    order = sorted(flat.params, key=lambda p: -layer_of(names[id(p)]))
    pending = [len(b[2]) for b in red.buckets]
    sched, seg, last = [[] for _ in range(len(cuts) + 1)], 0, depth
    for p in order:
        l = layer_of(names[id(p)])
        if l != last and last in cuts:
            seg += 1
        last = l
        bi = red._bucket_of[id(p)]
        pending[bi] -= 1
        if pending[bi] == 0:
            sched[seg].append(bi)
    assert sorted(b for s_ in sched for b in s_) == list(range(len(red.buckets)))
    # the data-parallel step on this rank's slice
    g_rank, loss_rank = oracle_grads(slice(rank * B_per, (rank + 1) * B_per))
    flat.zero_grad()
    red.begin_backward()
    seg, last = 0, depth
    for p in order:
        l = layer_of(names[id(p)])
        if l != last and last in cuts:
            red.launch_buckets(sched[seg])
            seg += 1
        last = l
        p.grad.add_(g_rank[names[id(p)]])
    red.launch_buckets(sched[seg])
    red.finish()
    g_dp = flat.flat_g.clone()
    t = torch.tensor([loss_rank])
    dist.all_reduce(t)
    # one rank, concatenated batch (every sample masks the same number of patches: the global masked mean is the mean of the ranks')
    g_all, loss_all = oracle_grads(slice(0, B_per * world))
    flat.zero_grad()
    for p in flat.params:
        p.grad.add_(g_all[names[id(p)]])
    assert abs(float(t) / world - loss_all) < 1e-5 * abs(loss_all)
    assert torch.allclose(g_dp, flat.flat_g, rtol=2e-4, atol=1e-6 * float(flat.flat_g.abs().max())), float((g_dp - flat.flat_g).abs().max())
    q.put((rank, list(red.launch_order), sched))
    dist.destroy_process_group()


def test_avmae_dp_equivalence_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_avmae_equiv_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.daemon = True
        p.start()
    try:
        for p in procs:
            p.join(timeout=240)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(5)
                if p.is_alive():
                    p.kill()
    res = [q.get(timeout=5) for _ in range(world)]
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]              # same collective order and schedule on every rank
    assert res[0][1] == [b for s in res[0][2] for b in s]                 # buckets go out segment by segment


def _switch_worker(rank, world, port, q, algo, bf16):
    """DAV_DP_ALGO / DAV_DP_BF16 / DAV_DP_BUCKET_MB (README "Data-parallel switches"): every combination must produce the
    group average in every gradient element — bucket lengths that are not multiples of the world size included (rs_ag reduces
    the leftover elements with a plain all-reduce)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DAV_DP_ALGO=algo, DAV_DP_BF16='1' if bf16 else '0',
                      DAV_DP_BUCKET_MB='0.011', DAV_DP_FIRST_BUCKET_MB='0.003')
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from deepavfusion_amd.util.distributed import DataParallel
    from deepavfusion_amd.util.flat import FlatParams
    torch.manual_seed(3)
    model = torch.nn.Sequential(torch.nn.Linear(41, 53), torch.nn.LayerNorm(53), torch.nn.Linear(53, 61), torch.nn.Linear(61, 47), torch.nn.Linear(47, 59),
                                torch.nn.Linear(59, 43), torch.nn.Linear(43, 5))
    flat = FlatParams(reversed(list(model.parameters())))
    dp = DataParallel(model, flat)                    # sizes from the environment
    dp.reducer.comm_stream = None
    red = dp.reducer
    assert red.algo == algo and red.bf16_wire == bool(bf16) and len(red.buckets) >= 3
    if world == 3:      # (flat segments are padded to even lengths: only an odd world size meets a ragged bucket)
        assert any((hi - lo) % world for lo, hi, _ in red.buckets)
    red.begin_backward()
    for i, p in enumerate(flat.params):                # small integers: exact in bf16 as well
        p.grad.add_(torch.full_like(p, float((rank + 1) * (i % 7 + 1))))
        red.grad_ready(p)
    red.finish()
    mean_rank = sum(r + 1 for r in range(world)) / world
    for i, p in enumerate(flat.params):
        assert torch.allclose(p.grad, torch.full_like(p, mean_rank * (i % 7 + 1)), rtol=1e-6 if not bf16 else 4e-3), (algo, bf16, i)
    # skip_collectives (bench.py's no-communication replay): the schedule runs, the gradients stay local
    flat.zero_grad()
    red.skip_collectives = True
    red.begin_backward()
    for i, p in enumerate(flat.params):
        p.grad.add_(torch.full_like(p, float(rank + 1)))
        red.grad_ready(p)
    red.finish()
    assert all(torch.equal(p.grad, torch.full_like(p, float(rank + 1))) for p in flat.params)
    assert sorted(red.launch_order) == list(range(len(red.buckets)))
    q.put((rank, list(red.launch_order)))
    dist.destroy_process_group()


def _run_switch(world, algo, bf16):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_switch_worker, args=(r, world, port, q, algo, bf16)) for r in range(world)]
    for p in procs:
        p.daemon = True
        p.start()
    try:
        for p in procs:
            p.join(timeout=180)
            assert p.exitcode == 0, (world, algo, bf16)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(5)
                if p.is_alive():
                    p.kill()
    res = [q.get(timeout=5) for _ in range(world)]
    assert all(r[1] == res[0][1] for r in res)


def test_dp_switches_world2_and_world4_gloo():
    for world, algo, bf16 in ((2, 'rs_ag', False), (2, 'allreduce', True), (4, 'rs_ag', True), (3, 'rs_ag', False)):
        _run_switch(world, algo, bf16)

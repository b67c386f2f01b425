#!/usr/bin/env python3
"""Benchmark of the DeepAVFusion/AVMAE pre-training step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--config base|base_m75|base_as|large]

One "step" = bf16 weight refresh + forward + backward + (DP gradient all-reduce) + global grad norm + AdamW
on one synthetic batch (B AV pairs per GPU) resident in HBM.  Prints ONE JSON line on rank 0.
For N > 1 the driver launches this file with torch.distributed.run (one rank per GPU, RCCL).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
BIG_LAUNCH_TILES = 400             # launches of >= 400 128x128-tile equivalents: the 8-wave big-tile regime of gemm_nt2_grouped_kernel (csrc/gemm.hip nt_auto_config_tiles)
HBM_PEAK_GBS = 8000.0
VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 2      # wave64 VALU instructions per second: 4 SIMD-32 per CU, 2 cycles per instruction (MI355X_MICROARCH.md "Wave scheduling")
# (plain VALU, transcendental) instructions per 16 x 32 score tile and wave on the HOT path of the key loops (the rescale branch and the
# padded-key masking excluded), counted in the ISA of this tree: tools/isa_valu_count.py + profiles/r06_attn_valu_isa.txt
ATTN_VALU_PER_TILE = {(32, 32): {'fwd': (38, 8), 'dq': (28, 8), 'dkv': (44, 8)},
                      (64, 64): {'fwd': (44, 8), 'dq': (36, 8), 'dkv': (56, 8)},
                      (16, 64): {'fwd': (44, 8), 'dq': (36, 8), 'dkv': (56, 8)}}


def lin(t, i, o):
    return 2.0 * t * i * o


def attn(nq, nk, dqk, dv):
    return 2.0 * nq * nk * (dqk + dv)


def necessary_fwd_flops_per_pair(cfg):
    """SURVEY.md section 8(d) work model: forward FLOPs the path cannot avoid (fusion-row outputs of the modality
    blocks, un-kept patches and redundant (v,a)-pair rows are NOT counted).  fwd+bwd = 3x."""
    D, c = cfg.embed_dim, sum(cfg.fusion_tkns)
    nmm, nv, na = cfg.fusion_tkns
    Li, La = cfg.image_grid[0] * cfg.image_grid[1], cfg.audio_grid[0] * cfg.audio_grid[1]
    nI, nA = int(Li * (1 - cfg.image_mask_ratio)), int(La * (1 - cfg.audio_mask_ratio))
    Hm = int(D * cfg.mlp_ratio)
    f = lin(nI, 3 * 256, D) + lin(nA, 256, D)
    for n in (nI, nA):
        blk = lin(n, D, D) + lin(n + c, D, 2 * D) + attn(n, n + c, D, D) + lin(n, D, D) + 2 * lin(n, D, Hm)
        f += cfg.depth * blk
    Da, Hf = int(D * cfg.fusion_attn_ratio), int(D * cfg.fusion_mlp_ratio)
    fus = 0.0
    for nq, nk in ((nv, nI), (na, nA)):
        fus += lin(nq, D, D) + lin(nk, D, 2 * D) + attn(nq, nk, D, D) + lin(nq, D, D)
    fus += lin(nv + na, D, Da) + lin(nv + na, D, D)                  # factorised k / v projections
    fus += lin(nmm, D, Da) + attn(nmm, nv * na, Da, D) + lin(nmm, D, D)
    fus += 2 * lin(c, D, Hf)
    f += len(cfg.fusion_layers) * fus
    Dd, Hd = cfg.decoder_dim, int(cfg.decoder_dim * cfg.decoder_mlp_ratio)
    for n, L, P in ((nI, Li, 768), (nA, La, 256)):
        t = c + L
        f += lin(n + c, D, Dd) + cfg.decoder_depth * (lin(t, Dd, 3 * Dd) + attn(t, t, Dd, Dd) + lin(t, Dd, Dd) + 2 * lin(t, Dd, Hd))
        f += lin(L, Dd, P)
    return f


def _spawn_ranks(n):
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is read by a thread while ALL children are polled: a rank that dies before the rendezvous would leave the
    # others (and a blocking communicate() on rank 0) waiting forever — kill the rest as soon as any rank exits non-zero
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    p.kill()
                    rcs[i] = p.wait()
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    sys.stdout.write(b''.join(c for c in chunks if c).decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=0, help='AV pairs per GPU (default: 64 ViT-B, 32 ViT-L)')
    ap.add_argument('--config', default='base')
    ap.add_argument('--no-graph', action='store_true', help='eager kernel launches instead of hipGraph replay')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-threads', type=int, default=0,
                    help='host threads for the cpu_baseline leg (0 = min(16, cores): more threads oversubscribe these matmuls — '
                         '128 threads are 7x slower than 16 on the MI355X host; 1 = scalar figure)')
    ap.add_argument('--no-roofline', action='store_true', help='skip the dominant-kernel replay (clean per-step profiles)')
    ap.add_argument('--roofline-only', action='store_true',
                    help='one eager step (to record the launch mix) + 20 replays of the dominant kernel: the run profiled for profiles/*roofline*')
    a = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('DAV_BENCH_SPAWN_DRY') == '1' and world > 1:      # CPU test hook of the rank spawner below: no GPU call, report and leave
        if rank == 0:
            print(json.dumps({'dry': True, 'world': world, 'rank': rank, 'master': os.environ.get('MASTER_ADDR'), 'port': int(os.environ.get('MASTER_PORT', '0')), 'argv_gpus': a.gpus}))
        return 0
    if a.gpus > 1 and world == 1:
        # Plain `python bench.py --gpus N` (no torch.distributed.run around it): this process starts the N ranks itself, as
        # fresh child processes, BEFORE it has made any GPU call (it never does), waits for them and forwards rank 0's
        # one JSON line.  Never re-exec a process that has initialised the GPU.
        return _spawn_ranks(a.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    force_dist = os.environ.get('DAV_FORCE_DIST', '0') == '1'      # test hook: 1-rank RCCL group on a single GPU
    # stdout carries exactly ONE line, the result: anything a library prints there meanwhile (RCCL's version banner at
    # communicator creation does) is sent to stderr instead; the JSON goes to the saved descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        torch.distributed.init_process_group(backend='nccl', init_method='env://', world_size=world, rank=rank)

    from deepavfusion_amd import ops
    from deepavfusion_amd.build_model import build_avmae
    from deepavfusion_amd.configs import CONFIGS
    from deepavfusion_amd.util import lr_sched
    from deepavfusion_amd.util.flat import FlatAdamW
    from deepavfusion_amd.util.misc import GraphedStep, Trainer

    for kv in os.environ.get('DAV_TUNE', '').split(','):      # e.g. DAV_TUNE=3:0,1:8 (launch-geometry experiments)
        if ':' in kv:
            from deepavfusion_amd import _lib
            _lib.load().dav_tune(int(kv.split(':')[0]), int(kv.split(':')[1]))
    cfg = CONFIGS[a.config]
    B = a.batch or (32 if cfg.embed_dim >= 1024 else 64)
    torch.manual_seed(0)                                   # identical init on every rank (then broadcast anyway)
    model = build_avmae(cfg).to(dev)
    nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]                 # train.py:89
    groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
    lr = 1.5e-4 * B * world / 256                                                              # train.py:32-34
    opt = FlatAdamW(groups, lr=lr, betas=(0.9, 0.95), model=model)
    trainer = Trainer(model, optimizer=opt, accum_iter=1, use_amp=True, distributed=world > 1 or force_dist)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    image = torch.randn(B, 3, *cfg.image_size, device=dev, generator=g)
    audio = (torch.randn(B, 1, *cfg.audio_size, device=dev, generator=g) * 2.0 - 3.0).clamp(-7, 4)
    torch.manual_seed(0 + rank)                            # masking noise stream, per rank (util/distributed.py:90-94)

    # ---- record the launch mix of ONE step (rank 0): NT launches as the library really issues them (grouped or single,
    # from the library's own issue log), the grouped weight-gradient launches and the attention calls ------------------
    tn_log, attn_log = [], []
    if rank == 0:
        orig_tn, orig_tg, orig_af, orig_ab = ops.gemm_tn_grouped, ops.gemm_tn_gang, ops.attn_fwd, ops.attn_bwd

        def log_tn(problems):
            tn_log.append(('grouped', [(d['Mc'], d['N'], d['K']) for d in problems]))
            return orig_tn(problems)

        def log_tg(problems):
            tn_log.append(('gang', [(d['Mc'], d['N'], d['K']) for d in problems]))
            return orig_tg(problems)

        def log_af(q, k, v, O, LSE, B_, H, Nq, Nk, dqk, dv, *rest):
            attn_log.append(('fwd', B_, H, Nq, Nk, dqk, dv))
            return orig_af(q, k, v, O, LSE, B_, H, Nq, Nk, dqk, dv, *rest)

        def log_ab(q, k, v, O, dO, LSE, Delta, dq, dk, dv_, B_, H, Nq, Nk, dqk, dv, *rest, **kw):
            if kw.get('part', 3) & 1:
                attn_log.append(('bwd', B_, H, Nq, Nk, dqk, dv))
            return orig_ab(q, k, v, O, dO, LSE, Delta, dq, dk, dv_, B_, H, Nq, Nk, dqk, dv, *rest, **kw)
        ops.gemm_tn_grouped, ops.gemm_tn_gang, ops.attn_fwd, ops.attn_bwd = log_tn, log_tg, log_af, log_ab
        ops.nt_issue_log(True)

    def eager_step():
        li, la = trainer.model(image, audio)[:2]
        trainer.step(li + la)
        return li, la
    eager_step()                          # every rank (collectives inside); also the first warm-up pass
    nt_log = []
    if rank == 0:
        nt_log = ops.nt_issue_log()
        ops.nt_issue_log(False)
        ops.gemm_tn_grouped, ops.gemm_tn_gang, ops.attn_fwd, ops.attn_bwd = orig_tn, orig_tg, orig_af, orig_ab
        if os.environ.get('DAV_DUMP_MIX'):            # the step's NT launches: [tile configuration, b_kn, [(M, N, K) ...], [epilogue flags ...]]
            with open(os.environ['DAV_DUMP_MIX'], 'w') as f:
                json.dump({'nt': ops.nt_issue_log(with_flags=True), 'tn': tn_log, 'attn': attn_log}, f)
    if a.no_graph or a.roofline_only:
        step = eager_step
    else:
        gs = GraphedStep(trainer, image.shape, audio.shape)
        step = lambda: gs(image, audio)

    if a.roofline_only:
        a.warmup, a.steps, a.no_cpu_baseline = 0, 1, True
    for _ in range(a.warmup):
        step()

    def sync():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            torch.distributed.barrier()
            torch.cuda.synchronize()
    sync()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(a.steps):
        out = step()
        ev[i + 1].record()
    sync()
    dt = time.perf_counter() - t0
    step_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps))       # device-event time per step
    median_ms = step_ms[len(step_ms) // 2]
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    loss = float(out[0]) + float(out[1])
    ms = dt / a.steps * 1e3
    pairs_per_s = B * world * a.steps / dt
    # ---- data-parallel runs: what the collectives cost the step.  The same captured step is replayed with the reducer's whole
    # schedule intact but NO collective issued (GradReducer.skip_collectives): comm_ms_exposed = step time - that.  After the
    # timed region; ranks train on un-averaged gradients meanwhile, which nothing reads any more. -------------------------------
    comm = None
    if (world > 1 or force_dist) and not (a.no_graph or a.roofline_only):
        red = trainer.model.reducer
        red.skip_collectives = True
        for _ in range(2):
            step()
        sync()
        n_nc = max(5, a.steps // 2)
        t1 = time.perf_counter()
        for _ in range(n_nc):
            step()
        sync()
        t_nc = torch.tensor([(time.perf_counter() - t1) / n_nc * 1e3], device=dev, dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(t_nc, op=torch.distributed.ReduceOp.MAX)
        red.skip_collectives = False
        # what the data-parallel MACHINERY costs a rank, collectives aside: the segmented replay without collectives against ONE graph of
        # the same step (segments = 1, also without collectives) — graph cuts, per-segment weight-gradient launches, bucket bookkeeping.
        # (a diagnostic behind the timed region: no collective is issued by any rank; a failure here must not cost the bench line)
        t_one = None
        try:
            gs1 = GraphedStep(trainer, image.shape, audio.shape, segments=1)
            red.skip_collectives = True
            for _ in range(3):
                gs1(image, audio)
            torch.cuda.synchronize()          # (rank-local on purpose: no barrier / collective inside this leg — a rank that fails here cannot hang the others)
            t2 = time.perf_counter()
            for _ in range(n_nc):
                gs1(image, audio)
            torch.cuda.synchronize()
            t_one = (time.perf_counter() - t2) / n_nc * 1e3          # this rank's figure (rank 0 reports)
            del gs1
        except Exception as e:          # noqa: BLE001
            print(f'[bench] single-rank overhead leg skipped: {type(e).__name__}: {e}', file=sys.stderr)
            t_one = None
        finally:
            red.skip_collectives = False
        comm = {'ms_per_step_no_collectives': round(float(t_nc), 3), 'comm_ms_exposed': round(ms - float(t_nc), 3),
                'ms_per_step_one_graph_no_collectives': None if t_one is None else round(t_one, 3),
                'single_rank_overhead_ms': None if t_one is None else round(float(t_nc) - t_one, 3),
                'algo': red.algo, 'bf16_wire': red.bf16_wire, 'buckets': len(red.buckets), 'segments': getattr(gs, 'n_seg', None),
                'grad_bytes': int(red.flat.flat_g.numel() * 4), 'rccl_ranks': torch.distributed.get_world_size(),
                'backend': torch.distributed.get_backend()}
    if rank != 0:
        return 0

    flops_pair = 3.0 * necessary_fwd_flops_per_pair(cfg)
    result = {
        'metric': 'AV-pairs/sec, DeepAVFusion ViT-B pre-training step (224px + 10s audio, image mask 0.75)',
        'value': round(pairs_per_s, 2), 'unit': 'AV-pairs/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
        'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16',
        'data': 'synthetic',
        'config': {'workload': f'AVMAE(DeepAVFusion {a.config}) fwd+bwd+allreduce+AdamW, B={B}/GPU, 224x224 RGB + (128,640) log-mel, '
                               f'masks {cfg.image_mask_ratio}/{cfg.audio_mask_ratio}, fusion attn_ratio {cfg.fusion_attn_ratio} mlp_ratio {cfg.fusion_mlp_ratio}',
                   'global_batch': B * world, 'parallelism': f'dp{world}', 'graph': not (a.no_graph or a.roofline_only)},
        'pairs_per_s_per_gpu': round(pairs_per_s / world, 2),
        'median_ms_per_step_device_events': round(median_ms, 3),
        'ms_per_step_quantiles_device_events': {k: round(step_ms[min(len(step_ms) - 1, int(f * (len(step_ms) - 1) + 0.5))], 3)
                                                for k, f in (('min', 0.0), ('p10', 0.1), ('p50', 0.5), ('p90', 0.9), ('max', 1.0))},
        'loss': round(loss, 5),
        'step_necessary_gflop_per_pair': round(flops_pair / 1e9, 1),
        'step_mfma_frac': round(flops_pair * pairs_per_s / world / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
    }

    if comm is not None:
        result['dp'] = comm
    if a.roofline_only:
        result['note'] = 'roofline-only run: value/ms_per_step are ONE eager step, not the metric; see roofline'

    # ---- BASELINE.md section 4 asks for the audio-mask-0.75 variant beside the default 0.8 (80 instead of 63 kept audio
    # tokens): same model and optimizer, a second captured step, a short timed run; labelled secondary -----------------
    if world == 1 and not (a.no_graph or a.roofline_only) and abs(cfg.audio_mask_ratio - 0.8) < 1e-9:
        model.audio_mask_ratio = 0.75
        gs75 = GraphedStep(trainer, image.shape, audio.shape)
        for _ in range(3):
            gs75(image, audio)
        torch.cuda.synchronize()
        n75 = 10
        t75 = time.perf_counter()
        for _ in range(n75):
            gs75(image, audio)
        torch.cuda.synchronize()
        t75 = time.perf_counter() - t75
        result['secondary'] = {'what': 'same workload with audio mask ratio 0.75 (BASELINE.md section 4)', 'value': round(B * n75 / t75, 2),
                               'unit': 'AV-pairs/s', 'ms_per_step': round(t75 / n75 * 1e3, 3), 'steps': n75}
        model.audio_mask_ratio = cfg.audio_mask_ratio
        del gs75

    from deepavfusion_amd import engine as E

    def time_replay(fn, reps):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()                     # torch's current stream == the stream ops.* launch on
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def time_graph(fn, reps):
        """Device time per call of ``fn`` from a replayed hipGraph of ``reps`` calls: the python + ctypes launch path takes ~10 us
        per call, more than the small attention kernels run — timed eagerly, those would report the host, not the kernel."""
        fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(reps):
                fn()
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    reps = 20 if a.roofline_only else 5
    # ---- roofline of the dominant kernel: gemm_nt2_grouped_kernel — the BIG grouped launches (image tower + audio tower +
    # fusion block problems of one step of a layer, the two decoders: >= 400 tiles of 128x128, the same 138 launches per step
    # as in round 1 / 2) exactly as the library issued them in the recorded step, each with the tile configuration it ran with
    # there (rule-based or from the tuned table: 3 = 128x128 on a 64-deep 2-stage ring, 43-46 = 256x128 / 128x256 on 32-deep
    # rings, 8 = 128x64; forward NT and b_kn input-gradient forms).  The set is defined by SIZE, not by configuration:
    # dropping the launches that moved to another tile shape would flatter the figure ------------------------------------
    def t128(probs):
        return sum(((M + 127) // 128) * ((N + 127) // 128) for (M, N, K) in probs)
    big = [(cfg_id, bt, probs) for (cfg_id, bt, probs) in nt_log if t128(probs) >= BIG_LAUNCH_TILES]
    if big and not a.no_roofline:
        bufs = {}
        for _c, bt, probs in big:
            for (M, N, K) in probs:
                if (M, N, K, bt) not in bufs:
                    Bm = (torch.randn(K, N, device=dev) * 0.05).bfloat16() if bt else (torch.randn(N, K, device=dev) * 0.05).bfloat16()
                    bufs[(M, N, K, bt)] = (torch.randn(M, K, device=dev).bfloat16(), Bm, torch.empty(M, N, device=dev, dtype=torch.bfloat16))

        def replay_nt():
            for c, bt, probs in big:
                with E.batch(auto_lanes=True):           # ONE grouped launch with the step's tile configuration
                    for (M, N, K) in probs:
                        A, Bm, C = bufs[(M, N, K, bt)]
                        ops.gemm_nt(A, Bm, M, N, K, ldb=N if bt else K, C_out=C, c_bf16=True, variant=(c << 4) | (bt << 12))
        total_ms = time_replay(replay_nt, reps)
        fl = sum(2.0 * M * N * K for _c, _b, probs in big for (M, N, K) in probs)
        ach = fl / (total_ms * 1e-3) / 1e12
        # per launch class: the K <= 512 launches (decoder qkv / fc1 / proj and their input gradients) sit at the MFMA / HBM ridge
        # (2K flop per output byte; fc1: 23 MB in, 2 x 92 MB out for 47 GFLOP), so their rate is reported against BOTH roofs
        classes = []
        for cname, pick in (('K <= 512 (decoders)', lambda probs: max(K for (_M, _N, K) in probs) <= 512),
                            ('K >= 768 (towers, fusion, wide-K decoder fc2)', lambda probs: max(K for (_M, _N, K) in probs) > 512)):
            sub = [(c, bt, probs) for (c, bt, probs) in big if pick(probs)]
            if not sub:
                continue

            def replay_sub(sub=sub):
                for c, bt, probs in sub:
                    with E.batch(auto_lanes=True):
                        for (M, N, K) in probs:
                            A, Bm, C = bufs[(M, N, K, bt)]
                            ops.gemm_nt(A, Bm, M, N, K, ldb=N if bt else K, C_out=C, c_bf16=True, variant=(c << 4) | (bt << 12))
            ms_c = time_replay(replay_sub, reps)
            fl_c = sum(2.0 * M * N * K for _c, _b, probs in sub for (M, N, K) in probs)
            by_c = sum(2.0 * (M * K + N * K + M * N) for _c, _b, probs in sub for (M, N, K) in probs)
            tf_c, gbs_c = fl_c / (ms_c * 1e-3) / 1e12, by_c / (ms_c * 1e-3) / 1e9
            classes.append({'class': cname, 'launches': len(sub), 'tflops': round(tf_c, 1), 'mfma_frac': round(tf_c / MFMA_BF16_PEAK_TFLOPS, 4),
                            'algorithmic_GBps': round(gbs_c, 0), 'hbm_frac': round(gbs_c / HBM_PEAK_GBS, 4),
                            'bound': 'mfma/hbm ridge' if gbs_c / HBM_PEAK_GBS > 0.5 * tf_c / MFMA_BF16_PEAK_TFLOPS else 'mfma',
                            'avg_launch_us': round(ms_c * 1e3 / len(sub), 2)})
        # ---- the same set IN THE STEP: one eager step (default stream schedule) with a HIP event pair around every big NT launch, on
        # the stream it is launched on — begin -> end of the kernel while the other streams' kernels run beside it (what a
        # rocprofv3 kernel trace of the step shows per kernel: profiles/r04_instep_kernel_stats.csv is the offline twin) ----------
        in_step = None
        if not a.roofline_only and world == 1:      # (an eager step joins the gradient all-reduce: rank 0 alone would wait for the other ranks for ever)
            orig_nt = ops.gemm_nt
            pairs = []

            def timed_nt(A_, B_m, M, N, K, **kw):
                if A_.dtype != torch.bfloat16 or ((M + 127) // 128) * ((N + 127) // 128) < BIG_LAUNCH_TILES or E.batch.current() is not None:
                    return orig_nt(A_, B_m, M, N, K, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                orig_nt(A_, B_m, M, N, K, **kw)
                e1.record()
                pairs.append((e0, e1, 2.0 * M * N * K))
            eager_step()
            torch.cuda.synchronize()
            ops.gemm_nt = timed_nt
            try:
                eager_step()
                torch.cuda.synchronize()
            finally:
                ops.gemm_nt = orig_nt
            if pairs:
                us = sum(e0.elapsed_time(e1) for e0, e1, _f in pairs) * 1e3
                fl_s = sum(f for _e0, _e1, f in pairs)
                in_step = {'what': 'the same launches timed inside one eager step of the default stream schedule (HIP event pair per launch on its own stream; other streams\' kernels run beside it)',
                           'launches': len(pairs), 'avg_launch_us': round(us / len(pairs), 2), 'achieved': round(fl_s / (us * 1e-6) / 1e12, 1),
                           'frac': round(fl_s / (us * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), 'ms_per_step': round(us * 1e-3, 3)}
        traffic = traffic_source = None          # HBM bytes per launch: offline rocprofv3 PMC passes over THIS replay (profiles/), only for the profiled workload
        tf = os.path.join(ROOT, 'profiles', 'dominant_kernel_traffic.json')
        from deepavfusion_amd._lib import kernel_source_hash
        src_hash = kernel_source_hash()
        traffic_note = ('offline rocprofv3 --pmc FETCH_SIZE (x2, gfx950 note) + WRITE_SIZE over this replay (the committed files named in traffic_source: a constant '
                        'of the tree, not of this run — reported only while the kernel sources and the tuned table still hash to what was profiled); null unless profiled for this workload')
        if os.path.exists(tf):
            rec = json.load(open(tf)).get(f'{a.config}_b{B}', {})
            traffic_source = rec.get('source')
            if traffic_source and rec.get('measured_on'):
                traffic_source += f" [measured {rec['measured_on']}]"
            if rec.get('kernel_source_hash') == src_hash:
                traffic = rec.get('hbm_bytes_per_launch')
            elif rec:
                traffic_note = f'STALE, not reported: the kernel sources / tuned table changed since the PMC passes ({rec.get("kernel_source_hash")} -> {src_hash}); re-run tools/collect_r05.sh'
        result['roofline'] = {'bound': 'mfma', 'kernel': 'gemm_nt2_kernel / gemm_nt2_grouped_kernel, the launches of >= 400 tile equivalents exactly as the step issues them (default stream schedule: one launch per tower GEMM), each with the tile configuration it runs with in the step (launches_by_config: 3 = 128x128 on a 64-deep two-stage ring, 44 / 46 = 128x256 on three / two 32-deep stages — the majority of the launches —, 8 = 128x64; rules + tuned table): forward + b_kn dgrad',
                              'launches_by_config': {str(c): sum(1 for cc, _b, _p in big if cc == c) for c in sorted({cc for cc, _b, _p in big})},
                              'achieved': round(ach, 1), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                              'frac': round(ach / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': traffic,
                              'traffic_source': traffic_source,
                              'traffic_note': traffic_note,
                              'algorithmic_bytes_per_launch': int(sum(2.0 * (M * K + N * K + M * N) for _c, _b, probs in big for (M, N, K) in probs) / len(big)),
                              'launches_per_step': len(big), 'problems_per_step': sum(len(pr) for _c, _b, pr in big),
                              'avg_launch_us': round(total_ms * 1e3 / len(big), 2),
                              'avg_gflop_per_launch': round(fl / len(big) / 1e9, 3),
                              'classes': classes, 'in_step': in_step}
        st = os.path.join(ROOT, 'profiles', 'step_traffic.json')        # offline rocprofv3 FETCH_SIZE / WRITE_SIZE passes over the whole step
        if os.path.exists(st):
            rec = json.load(open(st)).get(f'{a.config}_b{B}')
            if rec and rec.get('kernel_source_hash') == src_hash:      # (a constant of the tree like roofline.traffic: dropped when stale)
                result['step_fabric'] = {'GB_per_step': rec['GB_per_step'], 'GBps': round(rec['GB_per_step'] / (ms * 1e-3), 0),
                                         'hbm_frac_of_8TBps': round(rec['GB_per_step'] / (ms * 1e-3) / HBM_PEAK_GBS, 3), 'source': rec.get('source'),
                                         'measured_on': rec.get('measured_on')}
        # the same kernel in the LANES schedule (engine.BATCH_POLICY 'on', three lanes on one queue: the towers' and the fusion
        # block's equal-rank GEMMs merged into one grid): what the kernel reaches with 1400-2400 tiles per launch — reported
        # beside the as-issued figure because the default stream schedule (faster end to end) launches per tower.
        if world == 1 and E.BATCH_POLICY != 'on' and not a.roofline_only:
            prev_policy, prev_fs = E.BATCH_POLICY, E.FUSION_ON_STREAM
            try:
                E.set_batch_policy('on')
                E.FUSION_ON_STREAM = False
                ops.nt_issue_log(True)
                eager_step()
                lane_log = ops.nt_issue_log()
                ops.nt_issue_log(False)
            finally:
                E.set_batch_policy(prev_policy)
                E.FUSION_ON_STREAM = prev_fs
            big2 = [(c, bt, probs) for (c, bt, probs) in lane_log if t128(probs) >= BIG_LAUNCH_TILES]
            for _c, bt, probs in big2:
                for (M, N, K) in probs:
                    if (M, N, K, bt) not in bufs:
                        Bm = (torch.randn(K, N, device=dev) * 0.05).bfloat16() if bt else (torch.randn(N, K, device=dev) * 0.05).bfloat16()
                        bufs[(M, N, K, bt)] = (torch.randn(M, K, device=dev).bfloat16(), Bm, torch.empty(M, N, device=dev, dtype=torch.bfloat16))

            def replay_lanes():
                for c, bt, probs in big2:
                    with E.batch(auto_lanes=True):
                        for (M, N, K) in probs:
                            A, Bm, C = bufs[(M, N, K, bt)]
                            ops.gemm_nt(A, Bm, M, N, K, ldb=N if bt else K, C_out=C, c_bf16=True, variant=(c << 4) | (bt << 12))
            ms2 = time_replay(replay_lanes, reps)
            fl2 = sum(2.0 * M * N * K for _c, _b, probs in big2 for (M, N, K) in probs)
            result['roofline']['lanes_schedule'] = {'what': "the same GEMMs as the launches of DAV_BATCH=1 DAV_FUSION_STREAM=0 (towers + fusion block merged per step)",
                                                    'achieved': round(fl2 / (ms2 * 1e-3) / 1e12, 1),
                                                    'frac': round(fl2 / (ms2 * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                                    'launches_per_step': len(big2), 'avg_launch_us': round(ms2 * 1e3 / len(big2), 2)}
        del bufs
    # ---- weight-gradient kernels: the step's launches as the engine issued them — the gang-scheduled 256 x 256 launch(es)
    # (gemm_tn_gang_kernel: both decoders, then all encoder layers of a captured segment) and whatever small flushes stayed on the
    # 128 x 128 grouped kernel.  Operands are shared per shape (read only), every problem has its own gradient buffer.
    if tn_log and not a.no_roofline:
        tbufs, launches = {}, []
        for kind, probs in tn_log:
            plist = []
            for (Mc, N, K) in probs:
                if (Mc, N, K) not in tbufs:
                    tbufs[(Mc, N, K)] = (torch.randn(Mc, N, device=dev).bfloat16(), torch.randn(Mc, K, device=dev).bfloat16())
                A_, B_ = tbufs[(Mc, N, K)]
                plist.append(dict(A=A_, B=B_, Mc=Mc, N=N, K=K, C=torch.zeros(N, K, device=dev), lda=N, ldb=K, ldc=K,
                                  bias_grad=torch.zeros(N, device=dev), overwrite=True))
            launches.append((kind, plist))

        def replay_tn():
            for kind, plist in launches:
                if kind == 'gang':
                    ops.gemm_tn_gang(plist)
                else:
                    n_l = (len(plist) + ops.TN_GROUP_MAX - 1) // ops.TN_GROUP_MAX
                    for i in range(n_l):
                        ops.gemm_tn_grouped(plist[i::n_l])
        ms_tn = time_replay(replay_tn, reps)
        fl_tn = sum(2.0 * Mc * N * K for _k, probs in tn_log for (Mc, N, K) in probs)
        n_gang = sum(1 for k, _p in tn_log if k == 'gang')
        wtraffic = wsrc = None                  # HBM bytes per gang launch: offline PMC passes over tools/tn_gang_bench.py (profiles/wgrad_traffic.json), hash-guarded like roofline.traffic
        wt = os.path.join(ROOT, 'profiles', 'wgrad_traffic.json')
        if os.path.exists(wt):
            from deepavfusion_amd._lib import kernel_source_hash as _ksh
            rec = json.load(open(wt)).get(f'{a.config}_b{B}', {})
            if rec.get('kernel_source_hash') == _ksh():
                wtraffic = rec.get('hbm_bytes_per_launch')
                wsrc = f"{rec.get('source')} [measured {rec.get('measured_on')}]; algorithmic bytes per launch {rec.get('algorithmic_bytes_per_launch')}, L2 hit {rec.get('l2_hit_pct')} %"
        result['roofline_wgrad'] = {'bound': 'mfma', 'kernel': 'gemm_tn_gang_kernel (256 x 256 tiles, per-XCD ticket queues: both decoders / all encoder layers of a segment per launch)'
                                    + ('' if n_gang == len(tn_log) else ' + gemm_tn_grouped_kernel<128,4,2> for the small flushes'),
                                    'achieved': round(fl_tn / (ms_tn * 1e-3) / 1e12, 1), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                    'frac': round(fl_tn / (ms_tn * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': wtraffic, 'traffic_source': wsrc,
                                    'launches_per_step': len(tn_log), 'gang_launches': n_gang, 'problems_per_step': sum(len(p) for _k, p in tn_log),
                                    'ms_per_step': round(ms_tn, 3),
                                    'note': 'isolated replay of the recorded launches on fresh operands, written (not accumulated) tiles as in the captured step'}
        del tbufs, launches
    # ---- attention family (isolated replays of the step's shapes; in the step equal-rank calls of both towers are one grid)
    if attn_log and not a.no_roofline:
        classes = {}
        for (kind, B_, H, Nq, Nk, dqk, dv) in attn_log:
            classes.setdefault((B_, H, Nq, Nk, dqk, dv), {'fwd': 0, 'bwd': 0})[kind] += 1
        names = {}
        nI, nA = int(cfg.image_grid[0] * cfg.image_grid[1] * (1 - cfg.image_mask_ratio)), int(cfg.audio_grid[0] * cfg.audio_grid[1] * (1 - cfg.audio_mask_ratio))
        nF = sum(cfg.fusion_tkns)
        for key in classes:
            B_, H, Nq, Nk, dqk, dv = key
            if dqk == 16: names[key] = 'K6 factorised pair attention'
            elif dqk == 32: names[key] = f'K4 decoder self-attention ({Nq} rows, d 32)'
            elif Nq in (nI, nA) and Nk in (nI + nF, nA + nF): names[key] = f'K4 tower self-attention ({Nq} x {Nk}, fusion rows as context)'
            else: names[key] = f'K5 aggregation cross-attention ({Nq} x {Nk})'
        entries = []
        for key, cnt in classes.items():
            B_, H, Nq, Nk, dqk, dv = key
            q = torch.randn(B_, Nq, H, dqk, device=dev).bfloat16()
            k = torch.randn(B_, Nk, H, dqk, device=dev).bfloat16()
            v = torch.randn(B_, Nk, H, dv, device=dev).bfloat16()
            O = torch.empty(B_, Nq, H, dv, device=dev, dtype=torch.bfloat16)
            dO = torch.randn(B_, Nq, H, dv, device=dev).bfloat16()
            LSE, Dl = torch.empty(B_, H, Nq, device=dev), torch.empty(B_, H, Nq, device=dev)
            dq, dk, dvv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            sc = float(dqk) ** -0.5
            fwd = lambda: ops.attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), O, LSE, B_, H, Nq, Nk, dqk, dv, Nq * H * dqk, H * dqk,
                                       Nk * H * dqk, H * dqk, Nk * H * dv, H * dv, Nq * H * dv, H * dv, sc)
            bwd = lambda: ops.attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), O, dO, LSE, Dl, dq.data_ptr(), dk.data_ptr(), dvv.data_ptr(),
                                       B_, H, Nq, Nk, dqk, dv, Nq * H * dqk, H * dqk, Nk * H * dqk, H * dqk, Nk * H * dv, H * dv,
                                       Nq * H * dv, H * dv, Nq * H * dv, H * dv, Nq * H * dqk, H * dqk, Nk * H * dqk, H * dqk, Nk * H * dv, H * dv, sc)
            ms_f, ms_b = time_graph(fwd, 20), time_graph(bwd, 20)
            ff = attn(Nq, Nk, dqk, dv) * B_ * H
            # VALU roofline of the same launches: the key loops' vector instructions per 16 x 32 score tile counted in the ISA
            # (tools/isa_valu_count.py, hot path: profiles/r06_attn_valu_isa.txt), a transcendental priced at 5/3 of a plain
            # instruction (MI355X_MICROARCH.md), against 4 SIMD-32 per CU x 256 CUs x 2.4 GHz / 2 cycles per wave instruction
            tiles = B_ * H * ((Nq + 15) // 16) * ((Nk + 31) // 32)
            vi = ATTN_VALU_PER_TILE.get((dqk, dv), ATTN_VALU_PER_TILE[(64, 64)])
            valu_f = tiles * (vi['fwd'][0] + vi['fwd'][1] * 5.0 / 3.0)
            valu_b = tiles * (vi['dq'][0] + vi['dkv'][0] + (vi['dq'][1] + vi['dkv'][1]) * 5.0 / 3.0)
            entries.append({'what': names[key], 'B': B_, 'heads': H, 'Nq': Nq, 'Nk': Nk, 'dqk': dqk, 'dv': dv,
                            'calls_per_step': cnt, 'fwd_us': round(ms_f * 1e3, 1), 'bwd_us': round(ms_b * 1e3, 1),
                            'fwd_tflops': round(ff / (ms_f * 1e-3) / 1e12, 1), 'bwd_tflops': round(2.5 * ff / (ms_b * 1e-3) / 1e12, 1),
                            'fwd_frac': round(ff / (ms_f * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                            'fwd_valu_frac': round(valu_f / (ms_f * 1e-3) / VALU_PEAK_WAVE_INSTR, 4),
                            'bwd_valu_frac': round(valu_b / (ms_b * 1e-3) / VALU_PEAK_WAVE_INSTR, 4),
                            'fwd_mfma_floor_us': round(ff / (MFMA_BF16_PEAK_TFLOPS * 1e12) * 1e6, 1),
                            'fwd_valu_floor_us': round(valu_f / VALU_PEAK_WAVE_INSTR * 1e6, 1)})
        entries.sort(key=lambda e: -(e['fwd_us'] + e['bwd_us']) * e['calls_per_step']['fwd'])
        result['roofline_attention'] = {'bound': 'neither roof: fwd_frac = fraction of the bf16 MFMA peak, fwd_valu_frac / bwd_valu_frac = fraction of the VALU issue peak '
                                                 '(plain-equivalent wave instructions of the key loops per second / 1.2288e12); at d = 32 the VALU floor is 1.6 x the MFMA floor and both are '
                                                 'a quarter of the measured time — the kernels are bound by the dependent chain of a wave (MFMA -> max -> cross-lane -> exp -> cvt -> MFMA per 16 x 32 tile) '
                                                 'at 6 waves per SIMD and by whole-workgroup rounds (1024 (batch, head) workgroups on 768 resident slots at 352 keys): DESIGN.md section 3',
                                        'valu_peak_wave_instr_per_s': VALU_PEAK_WAVE_INSTR,
                                        'valu_instr_per_16x32_tile': {f'{k[0]}x{k[1]}': v for k, v in ATTN_VALU_PER_TILE.items()},
                                        'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'kernels': 'attn_fwd / attn_bwd_dq / attn_bwd_dkv (isolated, back to back inside a replayed hipGraph: device time, no host launch path)',
                                        'shapes': entries}

    # ---- CPU baseline: the fp32 oracle (a port of the reference's path) on this box's host cores ---------------
    if world == 1 and not a.no_cpu_baseline:
        from oracle import avmae_oracle as O
        from oracle.configs import CONFIGS as OC
        ocfg = OC[a.config]
        sd = {k: v.detach().float().cpu().clone().requires_grad_(k not in O.FROZEN) for k, v in model.state_dict().items()}
        optc = torch.optim.AdamW([p for p in sd.values() if p.requires_grad], lr=1e-4, betas=(0.9, 0.95))
        torch.set_num_threads(a.cpu_threads if a.cpu_threads > 0 else min(16, os.cpu_count() or 16))
        cores = torch.get_num_threads()
        Bc = 8 if cores > 1 else 1
        im, au, ni, na = O.synthetic_batch(ocfg, Bc, seed=7)
        n_done, t_start = 0, time.perf_counter()
        while n_done < 6 and (time.perf_counter() - t_start < 15.0 or n_done < 1):
            li, la = O.avmae_forward(sd, ocfg, im, au, ni, na)[:2]
            (li + la).backward()
            optc.step()
            optc.zero_grad()
            n_done += 1
        tc = time.perf_counter() - t_start
        result['cpu_baseline'] = {'value': round(Bc * n_done / tc, 3), 'unit': 'AV-pairs/s', 'cores': cores, 'kind': 'port',
                                  'sample': f'{n_done} fp32 oracle steps (fwd+bwd+AdamW) at B={Bc}, same shapes'}
        if cores > 1:                       # "also report a 1-thread number" — one step of one pair
            torch.set_num_threads(1)
            im1, au1, ni1, na1 = O.synthetic_batch(ocfg, 1, seed=8)
            t1 = time.perf_counter()
            li, la = O.avmae_forward(sd, ocfg, im1, au1, ni1, na1)[:2]
            (li + la).backward()
            optc.step()
            optc.zero_grad()
            result['cpu_baseline']['one_thread'] = {'value': round(1.0 / (time.perf_counter() - t1), 3), 'unit': 'AV-pairs/s', 'cores': 1,
                                                    'sample': '1 fp32 oracle step (fwd+bwd+AdamW) at B=1'}
            torch.set_num_threads(cores)
    try:                                    # RCCL's version banner sits in C stdio's buffer: push it out BEFORE the JSON line
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stderr.flush()
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(result) + '\n').encode())
    return 0


if __name__ == '__main__':
    rc = main()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    sys.exit(rc)

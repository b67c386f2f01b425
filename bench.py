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
HBM_PEAK_GBS = 8000.0


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
    if a.gpus > 1 and world == 1:
        print('bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)', file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    force_dist = os.environ.get('DAV_FORCE_DIST', '0') == '1'      # test hook: 1-rank RCCL group on a single GPU
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

    gemm_log = []
    if rank == 0:
        orig = ops.gemm_nt

        def logged(A, Bm, M, N, K, **kw):
            gemm_log.append((M, N, K, (kw.get('variant', 0) >> 12) & 1))
            return orig(A, Bm, M, N, K, **kw)
        ops.gemm_nt = logged

    if a.no_graph or a.roofline_only:
        def step():
            li, la = trainer.model(image, audio)[:2]
            trainer.step(li + la)
            return li, la
        step()
    else:
        gs = GraphedStep(trainer, image.shape, audio.shape)
        step = lambda: gs(image, audio)
    if rank == 0:
        from deepavfusion_amd import ops as _o
        _o.gemm_nt = orig
        n_per_pass = len(gemm_log) // (1 if (a.no_graph or a.roofline_only) else 3)     # GraphedStep: 2 warm-up passes + 1 capture pass
        gemm_log = gemm_log[-n_per_pass:]

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
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    loss = float(out[0]) + float(out[1])
    ms = dt / a.steps * 1e3
    pairs_per_s = B * world * a.steps / dt
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
        'loss': round(loss, 5),
        'step_necessary_gflop_per_pair': round(flops_pair / 1e9, 1),
        'step_mfma_frac': round(flops_pair * pairs_per_s / world / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
    }

    if a.roofline_only:
        result['note'] = 'roofline-only run: value/ms_per_step are ONE eager step, not the metric; see roofline'

    # ---- roofline of the dominant kernel (gemm_nt_kernel<128,128>): replay this step's launch mix --------------
    # the launches the library's dispatch (gemm.hip: nt_auto_config) sends to the 128x128 / 8-wave kernel
    big = [(M, N, K, kn) for (M, N, K, kn) in gemm_log if N > 64 and N % 4 == 0 and K % 64 == 0 and ((M + 127) // 128) * ((N + 127) // 128) >= 400]
    if big and not a.no_roofline:
        bufs = {}
        for (M, N, K, kn) in set(big):
            Bm = (torch.randn(K, N, device=dev) * 0.05).bfloat16() if kn else (torch.randn(N, K, device=dev) * 0.05).bfloat16()
            bufs[(M, N, K, kn)] = (torch.randn(M, K, device=dev).bfloat16(), Bm, torch.empty(M, N, device=dev, dtype=torch.bfloat16))

        def replay():
            for key in big:
                M, N, K, kn = key
                A, Bm, C = bufs[key]
                ops.gemm_nt(A, Bm, M, N, K, ldb=N if kn else K, C_out=C, c_bf16=True, variant=kn << 12)
        replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        reps = 20 if a.roofline_only else 5
        e0.record()                     # torch's current stream == the stream ops.* launch on
        for _ in range(reps):
            replay()
        e1.record()
        torch.cuda.synchronize()
        total_ms = e0.elapsed_time(e1) / reps
        fl = sum(2.0 * M * N * K for (M, N, K, _) in big)
        ach = fl / (total_ms * 1e-3) / 1e12
        traffic = None          # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, KiB), see profiles/
        tf = os.path.join(ROOT, 'profiles', 'dominant_kernel_traffic.json')
        if os.path.exists(tf):
            traffic = json.load(open(tf)).get(a.config, {}).get('hbm_bytes_per_launch')
        result['roofline'] = {'bound': 'mfma', 'kernel': 'gemm_nt2_kernel<128,128,2,4,2> (fwd + b_kn dgrad)', 'achieved': round(ach, 1),
                              'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                              'traffic': traffic, 'launches_per_step': len(big),
                              'avg_launch_us': round(total_ms * 1e3 / len(big), 2),
                              'avg_gflop_per_launch': round(fl / len(big) / 1e9, 3)}

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
    try:                                    # RCCL's version banner sits in C stdio's buffer: push it out BEFORE the JSON line
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stderr.flush()
    print(json.dumps(result), flush=True)
    return 0


if __name__ == '__main__':
    rc = main()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    sys.exit(rc)

#!/usr/bin/env python3
"""What is the captured step sensitive to: memory-side bandwidth or CU time?   (round-4 byte audit; measurement only)

A throttled background kernel (tools/native/hog.hip, compiled here with hipcc — not part of the product library) runs on a side
stream while the step graph replays:
  bw<N>    N workgroups, each streaming its own 64 MB window from HBM over and over  (~N x 15 GB/s of extra memory traffic)
  cu<N>    the same N workgroups / instruction stream over a 32 KB window each (cache resident): the same CU occupancy, no traffic
The step's slow-down under bw<N> beyond that under cu<N> is what the extra BYTES cost; the hog's own rate (alone and beside the
step) is printed with it.

    python tools/step_sensitivity.py [--config base] [--batch 64] [--loads bw32,cu32,bw64,cu64,...]
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_hog():
    src = os.path.join(ROOT, 'tools', 'native', 'hog.hip')
    out = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'libhog.so')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', src, '-o', out])
    lib = ctypes.CDLL(out)
    lib.hog_stream.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.hog_stream.restype = ctypes.c_int
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='base')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--loads', default='none,bw32,cu32,bw64,cu64,bw128,cu128,bw256,cu256,bw512,cu512,none')
    a = ap.parse_args()
    lib = build_hog()

    from deepavfusion_amd.build_model import build_avmae
    from deepavfusion_amd.configs import CONFIGS
    from deepavfusion_amd.util import lr_sched
    from deepavfusion_amd.util.flat import FlatAdamW
    from deepavfusion_amd.util.misc import GraphedStep, Trainer

    dev = torch.device('cuda', 0)
    cfg = CONFIGS[a.config]
    B = a.batch
    torch.manual_seed(0)
    model = build_avmae(cfg).to(dev)
    nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
    groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
    opt = FlatAdamW(groups, lr=1.5e-4 * B / 256, betas=(0.9, 0.95), model=model)
    trainer = Trainer(model, optimizer=opt, accum_iter=1, use_amp=True, distributed=False)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    image = torch.randn(B, 3, *cfg.image_size, device=dev, generator=g)
    audio = (torch.randn(B, 1, *cfg.audio_size, device=dev, generator=g) * 2.0 - 3.0).clamp(-7, 4)
    gs = GraphedStep(trainer, image.shape, audio.shape)
    for _ in range(5):
        gs(image, audio)
    torch.cuda.synchronize()

    WIN_BW, WIN_CU = 64 << 20, 32 << 10
    buf = torch.zeros(512 * WIN_BW // 4, dtype=torch.float32, device=dev)         # 32 GB: 512 windows of 64 MB
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    side = torch.cuda.Stream()

    def hog(kind, n, passes):
        win = WIN_BW if kind == 'bw' else WIN_CU
        rc = lib.hog_stream(buf.data_ptr(), win, n, passes, sink.data_ptr(), side.cuda_stream)
        assert rc == 0, rc
        return n * win * passes

    def hog_rate(kind, n):                       # bytes per ms of the hog running ALONE, and the passes that last ~`target` ms
        passes = 2 if kind == 'bw' else 4096
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            e0.record(side)
            nbytes = hog(kind, n, passes)
            e1.record(side)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        return nbytes / ms, passes / ms

    def time_step(load):
        torch.cuda.synchronize()
        hog_bytes, hog_ms = 0, 0.0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0, h1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if load != 'none':
            kind, n = load[:2], int(load[2:])
            _, ppm = hog_rate(kind, n)
            passes = max(1, int(ppm * 27.0 * (a.reps + 3) * 2.0))          # outlasts the timed replays (the step may slow it down)
            with torch.cuda.stream(side):
                h0.record(side)
                hog_bytes = hog(kind, n, passes)
                h1.record(side)
        gs(image, audio)                             # (one replay for the hog to have started)
        e0.record()
        for _ in range(a.reps):
            gs(image, audio)
        e1.record()
        torch.cuda.synchronize()
        if load != 'none':
            hog_ms = h0.elapsed_time(h1)
        return e0.elapsed_time(e1) / a.reps, hog_bytes, hog_ms

    print(f'# step sensitivity, config {a.config} B={B}: background load on a side stream while the step graph replays ({a.reps} replays timed)')
    print('# load     step ms   hog alone GB/s   hog beside the step GB/s (whole run, incl. its tail alone)')
    res = {}
    for load in a.loads.split(','):
        alone = 0.0
        if load != 'none':
            alone = hog_rate(load[:2], int(load[2:]))[0] / 1e6
        ms, hb, hms = time_step(load)
        beside = hb / hms / 1e6 if hms else 0.0
        res.setdefault(load, []).append(round(ms, 3))
        tag = '(cache-resident window: its rate is L1/L2 traffic, not memory-side)' if load.startswith('cu') else ''
        print(f'{load:8s} {ms:8.3f}   {alone:10.0f}       {beside:10.0f}   {tag}')
    print('JSON ' + json.dumps(res))


if __name__ == '__main__':
    main()

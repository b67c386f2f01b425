#!/usr/bin/env python3
"""The step's weight-gradient launches (profiles/r02_step_launch_mix.json 'tn') timed alone, two ways: the 128 x 128 grouped
kernel, one launch per layer (the round-4 product path), and the gang-scheduled 256 x 256 kernel (dav_gemm_tn_gang_bf16) per
layer, merged over all encoder layers and merged over the whole step.  First a correctness pass against torch fp32.
   python tools/tn_gang_bench.py            (env DAV_TN_GANG_DEBUG=2 no epilogue, 4 no MFMAs: timing ablations)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import _libsel  # noqa: E402,F401
from deepavfusion_amd import ops  # noqa: E402

dev = torch.device('cuda')
mix = json.load(open(os.path.join(ROOT, 'profiles', 'r02_step_launch_mix.json')))['tn']
launches = [(l[0] if isinstance(l[0][0], list) else l) for l in mix]
REPS = int(os.environ.get('TN_BENCH_REPS', '5'))


def make(shapes, overwrite=False, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    probs = []
    for (Mc, N, K) in shapes:
        A = torch.randn(Mc, N, device=dev, generator=g).bfloat16()
        B = torch.randn(Mc, K, device=dev, generator=g).bfloat16()
        C = torch.zeros(N, K, device=dev)
        d = dict(A=A, B=B, Mc=Mc, N=N, K=K, C=C, lda=N, ldb=K, ldc=K, a_rowmap=None, b_rowmap=None, bias_grad=None)
        if overwrite:
            d['overwrite'] = True
        probs.append(d)
    return probs


def check():
    shapes = [(512, 768, 768), (3136, 192, 768), (4032, 2304, 768), (3136, 768, 3072), (640, 8, 264), (14592, 512, 1536), (2048, 520, 776)]
    for ow in (False, True):
        probs = make(shapes, overwrite=ow, seed=3)
        for d in probs:
            d['bias_grad'] = torch.full((d['N'],), 0.5, device=dev)
            d['C'].fill_(1.0)
        ops.gemm_tn_gang(probs)
        torch.cuda.synchronize()
        worst = 0.0
        for d in probs:
            ref = d['A'].float().t() @ d['B'].float() + (0.0 if ow else 1.0)
            err = float((d['C'] - ref).abs().max() / ref.abs().max())
            bref = d['A'].float().sum(0) + 0.5
            berr = float((d['bias_grad'] - bref).abs().max() / bref.abs().max())
            worst = max(worst, err, berr)
            assert err < 2e-3 and berr < 2e-3, (d['Mc'], d['N'], d['K'], ow, err, berr)
        print(f'check overwrite={ow}: worst relative error {worst:.2e} over {len(shapes)} problems', flush=True)
    # row maps: contraction rows taken from [B, rows, D] buffers (rpb rows per batch element at offset off, batch stride bs)
    Bn, rows, off, rpb, N, K = 8, 40, 8, 32, 768, 512
    g = torch.Generator(device='cuda').manual_seed(5)
    Af = torch.randn(Bn * rows, N, device=dev, generator=g).bfloat16()
    Bf = torch.randn(Bn * rpb, K, device=dev, generator=g).bfloat16()
    C = torch.zeros(N, K, device=dev)
    ops.gemm_tn_gang([dict(A=Af, B=Bf, Mc=Bn * rpb, N=N, K=K, C=C, lda=N, ldb=K, ldc=K, a_rowmap=(rpb, rows, off), b_rowmap=None,
                           bias_grad=None, overwrite=True)])
    sel = Af.view(Bn, rows, N)[:, off:off + rpb].reshape(-1, N).float()
    ref = sel.t() @ Bf.float()
    err = float((C - ref).abs().max() / ref.abs().max())
    assert err < 2e-3, err
    print(f'check row map: {err:.2e}', flush=True)


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


def old(probs):
    n_l = (len(probs) + 39) // 40
    for i in range(n_l):
        ops.gemm_tn_grouped(probs[i::n_l])


def report(name, shapes_list, us):
    fl = sum(2.0 * a * b * c for s in shapes_list for (a, b, c) in s)
    print(f'{name:58s} {fl / 1e9:8.1f} GFLOP {us:9.1f} us {fl / us / 1e6:7.0f} TF', flush=True)


if __name__ == '__main__':
    if os.environ.get('TN_BENCH_CHECK', '1') == '1':
        check()
    ow = os.environ.get('TN_BENCH_OVERWRITE', '1') == '1'
    which = sys.argv[1:] or ['dec', 'layer', 'enc', 'all', 'big']
    if 'dec' in which:
        p = make(launches[0], ow)
        report('decoders (68 problems): 128 x 128 grouped', [launches[0]], timeit(lambda: old(p)))
        report('decoders: gang', [launches[0]], timeit(lambda: ops.gemm_tn_gang(p)))
        del p
    if 'layer' in which:
        p = make(launches[2], ow)
        report('one encoder layer (22 problems): 128 x 128 grouped', [launches[2]], timeit(lambda: old(p)))
        report('one encoder layer: gang', [launches[2]], timeit(lambda: ops.gemm_tn_gang(p)))
        del p
    if 'enc' in which or 'all' in which:
        enc = [make(l, ow, seed=i) for i, l in enumerate(launches[2:])]
        flat = [d for l in enc for d in l]
        if 'enc' in which:
            report('12 encoder layers: 128 x 128 grouped, 12 launches', launches[2:], timeit(lambda: [old(l) for l in enc]))
            report('12 encoder layers: gang, 12 launches', launches[2:], timeit(lambda: [ops.gemm_tn_gang(l) for l in enc]))
            report('12 encoder layers: gang, ONE launch', launches[2:], timeit(lambda: ops.gemm_tn_gang(flat)))
            for n in (2, 3, 4, 6):
                groups = [[d for l in enc[i:i + n] for d in l] for i in range(0, 12, n)]
                report(f'12 encoder layers: gang, {12 // n} launches of {n} layers', launches[2:], timeit(lambda: [ops.gemm_tn_gang(x) for x in groups]))
        if 'all' in which:
            dec = make(launches[0], ow, seed=77)
            small = make(launches[1], ow, seed=78)
            report('whole step: 128 x 128 grouped, 14 launches', launches, timeit(lambda: [old(dec), old(small)] + [old(l) for l in enc]))
            report('whole step: gang, decoders + 12 layers in ONE launch', launches, timeit(lambda: [ops.gemm_tn_gang(dec + flat), old(small)]))
            report('whole step: gang, decoders | 12 layers (2 launches)', launches, timeit(lambda: [ops.gemm_tn_gang(dec), old(small), ops.gemm_tn_gang(flat)]))
            del dec, small
        del enc, flat
    if 'pmc' in which:          # one pass each, for rocprofv3 --pmc (averages per kernel name)
        enc = [make(l, ow, seed=i) for i, l in enumerate(launches[2:])]
        flat = [d for l in enc for d in l]
        for l in enc:
            old(l)
        ops.gemm_tn_gang(flat)
        torch.cuda.synchronize()
        del enc, flat
    if 'pmc_step' in which:     # the step's two gang launches (decoders, then all encoder layers), once each: what bench.py's roofline_wgrad replays
        dec = make(launches[0], ow, seed=77)
        enc = [make(l, ow, seed=i) for i, l in enumerate(launches[2:])]
        flat = [d for l in enc for d in l]
        ops.gemm_tn_gang(dec)
        ops.gemm_tn_gang(flat)
        torch.cuda.synchronize()
        alg = [sum(2.0 * (a * b + a * c) + 4.0 * b * c for (a, b, c) in l) for l in (launches[0], [s for l in launches[2:] for s in l])]
        print('algorithmic bytes per gang launch (bf16 operands once + fp32 gradient written):', [int(x) for x in alg], flush=True)
        del dec, enc, flat
    if 'big' in which:
        for (Mc, N, K) in ((4096, 4096, 4096), (8192, 4096, 4096)):
            p = make([(Mc, N, K)], True)
            report(f'single problem {Mc} x {N} x {K}: 128 x 128 grouped', [[(Mc, N, K)]], timeit(lambda: old(p)))
            report(f'single problem {Mc} x {N} x {K}: gang', [[(Mc, N, K)]], timeit(lambda: ops.gemm_tn_gang(p)))
            del p

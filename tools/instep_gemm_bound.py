#!/usr/bin/env python3
"""How many ms per step would a vendor-library-class GEMM buy IN THE STEP?  (round-4 review item 1, measurement only.)

NOT on the product path: this tool patches ``ops.gemm_nt`` in its own process and is imported by nothing.  hipBLASLt is
reached through ``torch.mm`` / ``torch.addmm`` and serves as a COMPARATOR for the big NT launches of the captured step
(>= 400 tile equivalents of 128 x 128: the set bench.py's ``roofline`` is defined on).

Captured steps compared in ONE process, replays interleaved round-robin (same box, same clocks):
  base        the product step
  dup_ours    every big NT launch is followed by a second copy of itself into scratch outputs (same operands, same epilogue)
  dup_blas    every big NT launch is followed by hipBLASLt's GEMM of the same operands into a bf16 scratch output
                -> (dup_ours - base) = in-situ cost O of our big NT set, (dup_blas - base) = in-situ cost H of the library's;
                   O - H bounds what a library-class kernel could buy (the library's side has NO epilogue: an upper bound)
  sub_blas    DIRECT substitution where it is exact: big launches whose epilogue is none / bias with one bf16 output
              (qkv and kv forward, the plain input-gradient GEMMs) run on hipBLASLt instead of our kernel; results stay
              valid, the loss is printed as the check
  dupS_ours / dupS_blas   the duplicate experiment restricted to that same exact subset (calibrates the duplicate method
              against the direct one: dupS_ours - dupS_blas should equal base - sub_blas)

    python tools/instep_gemm_bound.py [--config base] [--batch 64] [--rounds 6] [--reps 10] [--modes base,dup_ours,...]
Environment switches of the package (DAV_STREAMS=0 DAV_BATCH=0 for the serial schedule) apply as usual.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BIG_TILES = 400


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='base')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--rounds', type=int, default=6)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--modes', default='base,dup_ours,dup_blas,sub_blas,dupS_ours,dupS_blas')
    a = ap.parse_args()

    from deepavfusion_amd import ops
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
    torch.manual_seed(0)

    orig = ops.gemm_nt
    scratch = {}
    bias_bf16 = {}
    counts = {}

    def scr(t, tag):
        key = (tag, t.numel(), t.dtype)
        if key not in scratch:
            scratch[key] = torch.empty(t.numel(), dtype=t.dtype, device=t.device)
        return scratch[key].view(t.shape)

    def is_big(A, M, N):
        return A.dtype == torch.bfloat16 and ((M + 127) // 128) * ((N + 127) // 128) >= BIG_TILES

    def plain2d(A, Bm, M, N, K, kw):
        """operands as plain 2-D matrices for torch.mm, or None when the call reads through row maps / column offsets"""
        if kw.get('a_rowmap') is not None or (kw.get('lda') or K) != K or A.numel() < M * K:
            return None
        b_kn = (kw.get('variant', 0) >> 12) & 1
        if b_kn:                       # B is W [K(contraction), N(out)] row-major with ldb = N
            if Bm.dim() != 2 or tuple(Bm.shape) != (K, N) or (kw.get('ldb') or N) != N:
                return None
            return A.reshape(-1)[:M * K].view(M, K), Bm
        if Bm.dim() != 2 or tuple(Bm.shape) != (N, K) or (kw.get('ldb') or K) != K:
            return None
        return A.reshape(-1)[:M * K].view(M, K), Bm.t()

    def exact_subset(M, N, kw):
        C = kw.get('C_out')
        return (kw.get('act', 0) == 0 and kw.get('aux') is None and kw.get('res') is None and kw.get('C2') is None
                and kw.get('beta', 0) == 0 and kw.get('alpha', 1.0) == 1.0 and kw.get('c_rowmap') is None
                and kw.get('c_bf16') and C is not None and (kw.get('ldc') or N) == N and C.numel() == M * N)

    def blas(A2, B2, bias, out):
        if bias is not None:
            k = bias.data_ptr()
            if k not in bias_bf16:
                bias_bf16[k] = bias.detach().to(torch.bfloat16)
            torch.addmm(bias_bf16[k], A2, B2, out=out)
        else:
            torch.mm(A2, B2, out=out)

    def make_patch(mode):
        def patched(A, Bm, M, N, K, **kw):
            if not is_big(A, M, N):
                return orig(A, Bm, M, N, K, **kw)
            ops2 = plain2d(A, Bm, M, N, K, kw)
            sub = exact_subset(M, N, kw) and ops2 is not None
            counts.setdefault(mode, [0, 0])
            counts[mode][0] += 1
            counts[mode][1] += int(sub)
            if mode == 'sub_blas' and sub:
                blas(ops2[0], ops2[1], kw.get('bias'), kw['C_out'].view(M, N))
                return
            flt = mode.split(':')[1] if ':' in mode else ''
            b_kn = (kw.get('variant', 0) >> 12) & 1
            ok = {'': True, 'fwd': not b_kn, 'bkn': bool(b_kn), 'k768': K >= 768, 'k512': K <= 512, 'dec': N in (512, 1536, 2048) or K in (512, 2048) and N in (512, 1536, 2048),
                  'enc': K in (768, 2304, 3072) and N in (768, 2304, 3072)}.get(flt, True)
            if mode.startswith('cfg') and ok and (sub or mode.split(':')[0].endswith('all')):
                # our own tile configuration <n> (e.g. 60 = the 256 x 256 body) forced on the same launches hipBLASLt was substituted for
                # (cfg<n>) or on every big launch (cfg<n>all), optionally filtered (cfg60:fwd / :bkn / :k768 / :k512 / :dec / :enc);
                # launches the configuration cannot take fall through to the default
                n = int(mode.split(':')[0][3:].replace('all', ''))
                kw2 = dict(kw)
                kw2['variant'] = (n << 4) | (kw.get('variant', 0) & (1 << 12))
                try:
                    orig(A, Bm, M, N, K, **kw2)
                    counts[mode][1] += 1000            # thousands digit: launches that really ran the forced configuration
                    return
                except RuntimeError:
                    pass
            orig(A, Bm, M, N, K, **kw)
            if mode in ('dup_ours', 'dupS_ours') and (mode == 'dup_ours' or sub):
                kw2 = dict(kw)
                kw2['C_out'] = scr(kw['C_out'], 'c')
                if kw.get('C2') is not None:
                    kw2['C2'] = scr(kw['C2'], 'c2')
                kw2['beta'] = 0
                orig(A, Bm, M, N, K, **kw2)
            elif mode in ('dup_blas', 'dupS_blas') and (mode == 'dup_blas' or sub) and ops2 is not None:
                key = ('blas', M * N)
                if key not in scratch:
                    scratch[key] = torch.empty(M * N, dtype=torch.bfloat16, device=A.device)
                out = scratch[key].view(M, N)
                blas(ops2[0], ops2[1], kw.get('bias') if exact_subset(M, N, kw) else None, out)
        return patched

    modes = [m for m in a.modes.split(',') if m]
    steps = {}
    for m in modes:
        ops.gemm_nt = orig if m.startswith('base') else make_patch(m)      # (baseB, baseC ...: further unpatched captures — the tool's own noise floor)
        counts.pop(m, None)
        # one eager pass first: creates scratch / bias copies OUTSIDE the capture
        li, la = trainer.model(image, audio)[:2]
        trainer.step(li + la)
        counts.pop(m, None)
        gs = GraphedStep(trainer, image.shape, audio.shape)
        steps[m] = gs
        ops.gemm_nt = orig
    # GraphedStep's warm-up passes + the capture itself went through the patch: counts = (2 warm-ups + 1 capture) x per-step
    per_step = {m: (c[0] // 3, (c[1] % 1000) // 3, (c[1] // 1000) // 3) for m, c in counts.items()}

    for m in modes:
        for _ in range(3):
            steps[m](image, audio)
    torch.cuda.synchronize()
    times = {m: [] for m in modes}
    losses = {}
    for r in range(a.rounds):
        for m in modes:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                out = steps[m](image, audio)
            torch.cuda.synchronize()
            times[m].append((time.perf_counter() - t0) / a.reps * 1e3)
            losses[m] = float(out[0]) + float(out[1])
    med = {m: sorted(v)[len(v) // 2] for m, v in times.items()}
    mn = {m: min(v) for m, v in times.items()}
    sched = f"DAV_STREAMS={os.environ.get('DAV_STREAMS', '1')} DAV_BATCH={os.environ.get('DAV_BATCH', 'auto')}"
    print(f'# in-step GEMM bound, config {a.config} B={B}, schedule [{sched}], {a.rounds} interleaved rounds x {a.reps} replays')
    print(f'# big NT launches per step (>= {BIG_TILES} tiles) / of which in the exact subset: {per_step}')
    for m in modes:
        print(f'{m:10s} median {med[m]:7.3f} ms  min {mn[m]:7.3f} ms  loss {losses[m]:.5f}   rounds ' + ' '.join(f'{t:.2f}' for t in times[m]))
    res = {'schedule': sched, 'config': a.config, 'B': B, 'median_ms': med, 'min_ms': mn, 'per_step_launches': per_step}
    if all(k in med for k in ('base', 'dup_ours', 'dup_blas')):
        O, H = med['dup_ours'] - med['base'], med['dup_blas'] - med['base']
        res['all_big'] = {'ours_in_situ_ms': round(O, 3), 'blas_in_situ_ms': round(H, 3), 'prize_upper_bound_ms': round(O - H, 3)}
        print(f'ALL big NT: in-situ cost ours {O:.3f} ms, hipBLASLt (no epilogue) {H:.3f} ms -> upper bound of the prize {O - H:.3f} ms/step')
    if all(k in med for k in ('base', 'sub_blas')):
        res['subset_direct_ms'] = round(med['base'] - med['sub_blas'], 3)
        print(f'exact subset, DIRECT substitution: step {med["base"]:.3f} -> {med["sub_blas"]:.3f} ms  (gain {med["base"] - med["sub_blas"]:.3f} ms)')
    if all(k in med for k in ('base', 'dupS_ours', 'dupS_blas')):
        O, H = med['dupS_ours'] - med['base'], med['dupS_blas'] - med['base']
        res['subset_dup'] = {'ours_in_situ_ms': round(O, 3), 'blas_in_situ_ms': round(H, 3), 'diff_ms': round(O - H, 3)}
        print(f'exact subset, duplicate method: ours {O:.3f} ms, hipBLASLt {H:.3f} ms -> {O - H:.3f} ms (compare with the direct gain)')
    print('JSON ' + json.dumps(res))


if __name__ == '__main__':
    main()

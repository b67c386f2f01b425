#!/usr/bin/env python3
"""Which tile configuration is fastest for each grouped NT launch of the step?  Takes the launch mix bench.py dumps
(DAV_DUMP_MIX=file: tile configuration, b_kn, problems, epilogue flags per launch), rebuilds every distinct group with
equivalent epilogues (GELU + twin, aux multiply, fp32 residual, plain bf16) and times it as ONE grouped launch per candidate
configuration (hipGraph replay, rotating operand sets).
Usage: mix_sweep.py mix.json [--write table.json] [cfg ...]        default candidates: 3 8 5 43 44 45 46
--write merges the winners into a tuning table (deepavfusion_amd/tuning/nt_gfx950.json is the one the package loads): a group
gets an entry only where the best configuration beats the rule-based choice (the mix must have been dumped with
DAV_NT_TUNE=0) by at least 3 %."""
import json
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepavfusion_amd import engine as E  # noqa: E402
from deepavfusion_amd import ops  # noqa: E402

dev, bf = 'cuda', torch.bfloat16
REPS, NSETS = 10, 3


def timed(fn):
    fn()
    torch.cuda.synchronize()
    g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g.capture_begin()
        for _ in range(REPS):
            fn()
        g.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / REPS * 1e3)
    return best


def build(M, N, K, bt, fl):
    act, c_bf16, has_res, c2_mode, beta = fl & 3, bool(fl & 4), bool(fl & 8), (fl >> 4) & 15, bool(fl & 256)
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(K, N, device=dev) * 0.05).to(bf) if bt else (torch.randn(N, K, device=dev) * 0.05).to(bf)
    kw = dict(ldb=N if bt else K)
    C = torch.empty(M, N, device=dev, dtype=bf if c_bf16 else torch.float32)
    kw.update(C_out=C, c_bf16=c_bf16, beta=1 if beta else 0)
    if has_res:
        kw.update(res=torch.randn(M, N, device=dev), ldres=N)
    if act == 1:
        kw.update(act=1, bias=torch.randn(N, device=dev))
    elif act in (2, 3):
        kw.update(act=act, aux=torch.randn(M, N, device=dev).to(bf), ldaux=N)
    if c2_mode:
        kw.update(C2=torch.empty(M, N, device=dev, dtype=bf), ldc2=N, c2_mode=c2_mode)
    return (A, W, M, N, K), kw


def main():
    mix = json.load(open(sys.argv[1]))['nt']
    args = sys.argv[2:]
    table_path = None
    if '--write' in args:
        i = args.index('--write')
        table_path = args[i + 1]
        del args[i:i + 2]
    cands = [int(x) for x in args] or [3, 8, 5, 43, 44, 45, 46]
    winners = []
    groups = Counter((c, bt, tuple(map(tuple, probs)), tuple(flags)) for c, bt, probs, flags in mix)
    total = {c: 0.0 for c in cands}
    tot_now = tot_best = 0.0
    for (cur, bt, probs, flags), cnt in sorted(groups.items(), key=lambda kv: -kv[1]):
        gf = sum(2.0 * M * N * K for M, N, K in probs) / 1e9
        if gf < 5.0:
            continue
        sets = [[build(M, N, K, bt, fl) for (M, N, K), fl in zip(probs, flags)] for _ in range(NSETS)]
        rot = [0]
        cells = {}
        for c in dict.fromkeys([cur] + cands):
            def fn():
                rot[0] += 1
                with E.batch(auto_lanes=True):
                    for (A, W, M, N, K), kw in sets[rot[0] % NSETS]:
                        ops.gemm_nt(A, W, M, N, K, variant=(c << 4) | (bt << 12), **kw)
            try:
                cells[c] = timed(fn)
            except Exception as e:      # a configuration the shape does not admit
                cells[c] = float('nan')
        best = min((v, k) for k, v in cells.items() if v == v)
        tot_now += cnt * cells[cur]
        tot_best += cnt * best[0]
        if best[1] != cur and cells[cur] / best[0] >= 1.03:
            winners.append({'cfg': best[1], 'b_kn': bt, 'problems': [[M, N, K, fl] for (M, N, K), fl in zip(probs, flags)],
                            'rule_cfg': cur, 'us_rule': round(cells[cur], 1), 'us': round(best[0], 1)})
        shp = ' '.join(f'{M}x{N}x{K}' for M, N, K in probs[:3]) + (' ...' if len(probs) > 3 else '')
        print(f'{cnt:3d}x bt{bt} fl{flags[0]:4d} {shp:58s} {gf:6.1f} GF | now cfg{cur}: {cells[cur]:7.1f} us {gf / cells[cur] * 1e3:5.0f} TF | '
              + ' '.join(f'{k}:{v:6.1f}' for k, v in cells.items() if k != cur) + f' | best {best[1]} ({cells[cur] / best[0]:.2f}x)', flush=True)
        del sets
    print(f'per step: now {tot_now / 1e3:.2f} ms, best-per-group {tot_best / 1e3:.2f} ms')
    if table_path:
        table = {'device': 'gfx950 (MI355X)', 'entries': []}
        if os.path.exists(table_path):
            table = json.load(open(table_path))

        def key(e):
            return (e['b_kn'], tuple(sorted(map(tuple, e['problems']))))
        held = {key(e): e for e in table['entries']}
        for w in winners:
            held[key(w)] = w
        table['entries'] = list(held.values())
        table['note'] = 'written by tools/mix_sweep.py --write; cfg = NT tile configuration (csrc/gemm.hip nt2_issue_auto), problems = [M, N, K, epilogue flags]'
        with open(table_path, 'w') as f:
            json.dump(table, f, indent=1)
        print(f'{table_path}: {len(winners)} entries written / updated, {len(table["entries"])} held')


if __name__ == '__main__':
    main()

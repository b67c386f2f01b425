#!/usr/bin/env python3
"""In-step tile-configuration sweep: unlike tools/mix_sweep.py (isolated launches) this times the WHOLE step with ONE launch
signature moved to another configuration at a time — what counts once launches of several streams share the GPU.
For every distinct NT launch signature of the recorded mix (>= 5 GFLOP x count) and every candidate configuration it writes a
one-entry tuning table, runs `bench.py --no-roofline` with it (DAV_NT_TUNE_FILE) and logs ms per step next to the baseline.
Usage (on the GPU box): instep_sweep.py mix.json out.json [cfg ...]"""
import json
import os
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mix = json.load(open(sys.argv[1]))['nt']
out_path = sys.argv[2]
cands = [int(x) for x in sys.argv[3:]] or [3, 8, 43, 44, 46]
tmp = os.path.join(ROOT, 'gpurun_out', 'instep_table.json')


def bench(table_entries):
    json.dump({'entries': table_entries}, open(tmp, 'w'))
    env = dict(os.environ, DAV_NT_TUNE_FILE=tmp)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--no-roofline', '--steps', '30', '--warmup', '10'],
                       env=env, capture_output=True, text=True, timeout=600)
    try:
        return json.loads(r.stdout.strip().splitlines()[-1])['ms_per_step']
    except Exception:
        return float('nan')


groups = Counter((c, bt, tuple(map(tuple, probs)), tuple(flags)) for c, bt, probs, flags in mix)
base = [bench([]) for _ in range(2)]
print('baseline (rules only):', base, flush=True)
results = {'baseline': base, 'trials': []}
for (cur, bt, probs, flags), cnt in sorted(groups.items(), key=lambda kv: -kv[1] * sum(2.0 * M * N * K for M, N, K in kv[0][2])):
    gf = sum(2.0 * M * N * K for M, N, K in probs) / 1e9
    if gf * cnt < 60.0:
        continue
    for c in cands:
        if c == cur:
            continue
        entry = {'cfg': c, 'b_kn': bt, 'problems': [[M, N, K, fl] for (M, N, K), fl in zip(probs, flags)]}
        ms = bench([entry])
        results['trials'].append({'entry': entry, 'rule_cfg': cur, 'count': cnt, 'ms': ms})
        print(f'{cnt:3d}x bt{bt} fl{flags[0]:4d} {probs[0]} ... rule cfg{cur} -> cfg{c}: {ms:.3f} ms (baseline {min(base):.3f})', flush=True)
        json.dump(results, open(out_path, 'w'), indent=1)
base2 = bench([])
results['baseline'].append(base2)
json.dump(results, open(out_path, 'w'), indent=1)
print('baseline again:', base2)

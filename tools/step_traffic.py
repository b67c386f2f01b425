#!/usr/bin/env python3
"""Where the bytes of one training step go: per kernel family, launches per step, time, memory-side fetch and write bytes (L2
misses: Infinity-Cache hits are included, MI355X_MICROARCH.md "HBM") and the rate they imply — from two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE; kernels run one at a time under counter collection) over `bench.py --no-graph --no-roofline`.
FETCH_SIZE is doubled (gfx950 note of the guide); both are KiB.  Steps = launches of adamw_flat_kernel.
Usage: step_traffic.py fetch_counter_collection.csv write_counter_collection.csv [out.json key [source text]]"""
import csv
import re
import sys
from collections import defaultdict


def fam(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*', '', n)[:66]


def load(path, counter):
    val, dur, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = fam(r['Kernel_Name'])
        val[k] += float(r['Counter_Value'])
        dur[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        cnt[k] += 1
    return val, dur, cnt


F, dF, cF = load(sys.argv[1], 'FETCH_SIZE')
W, dW, cW = load(sys.argv[2], 'WRITE_SIZE')
steps = max(cF.get('adamw_flat_kernel', 1), 1)
rows = []
for k in F:
    f = 2 * F[k] * 1024 / steps                       # bytes per step
    w = W.get(k, 0.0) * 1024 / max(cW.get('adamw_flat_kernel', steps), 1)
    t = dF[k] / steps                                 # ns per step (serialised launches)
    rows.append((t, k, cF[k] / steps, f, w))
rows.sort(reverse=True)
T = sum(r[0] for r in rows); FF = sum(r[3] for r in rows); WW = sum(r[4] for r in rows)
print(f'{steps} steps; per step: {T / 1e6:.2f} ms of kernel time (serialised), {FF / 1e9:.2f} GB fetched, {WW / 1e9:.2f} GB written '
      f'-> {(FF + WW) / T:.0f} GB/s averaged over the kernels\' own time')
print(f'{"kernel family":66s} {"n/step":>7s} {"ms/step":>8s} {"us each":>8s} {"fetch MB":>9s} {"write MB":>9s} {"GB/s":>6s}')
for t, k, n, f, w in rows[:45]:
    print(f'{k:66s} {n:7.1f} {t / 1e6:8.3f} {t / 1e3 / max(n, 1e-9):8.1f} {f / 1e6:9.1f} {w / 1e6:9.1f} {(f + w) / max(t, 1):6.0f}')

if len(sys.argv) > 4:                                # step_traffic.py fetch.csv write.csv <out.json> <key>: totals for bench.py's step_fabric field
    import json
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepavfusion_amd._lib import kernel_source_hash
    path, key = sys.argv[3], sys.argv[4]
    j = json.load(open(path)) if os.path.exists(path) else {'_comment': 'memory-side traffic of one whole training step (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, separate passes over bench.py --no-graph; Infinity-Cache hits included); tools/step_traffic.sh; key = <config>_b<batch>'}
    j[key] = {'GB_per_step': round((FF + WW) / 1e9, 2), 'fetched_GB': round(FF / 1e9, 2), 'written_GB': round(WW / 1e9, 2),
              'serialised_kernel_ms': round(T / 1e6, 2),
              'source': sys.argv[5] if len(sys.argv) > 5 else 'profiles/r05_step_traffic.txt (tools/collect_r05.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py --no-graph)',
              'kernel_source_hash': kernel_source_hash(),
              'measured_on': os.environ.get('DAV_MEASURED_ON') or __import__('datetime').date.today().isoformat()}
    json.dump(j, open(path, 'w'), indent=1)

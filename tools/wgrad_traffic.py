#!/usr/bin/env python3
"""profiles/wgrad_traffic.json: HBM-side bytes per gang weight-gradient launch (the step's two: decoders, all encoder layers), from
rocprofv3 --pmc passes over `tools/tn_gang_bench.py pmc_step` (FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, KiB;
WRITE_SIZE KiB; TCC_HIT_sum / TCC_MISS_sum for the L2 hit rate).  bench.py reports it as roofline_wgrad.traffic while the kernel sources
still hash to what was profiled.
Usage: wgrad_traffic.py fetch_counter.csv write_counter.csv tcc_counter.csv bench_log.txt key source"""
import csv
import datetime
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd._lib import kernel_source_hash  # noqa: E402


def per_launch(path, counter):
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter and 'gemm_tn_gang_kernel' in r['Kernel_Name']]
    return v


fetch, write = per_launch(sys.argv[1], 'FETCH_SIZE'), per_launch(sys.argv[2], 'WRITE_SIZE')
hit, miss = per_launch(sys.argv[3], 'TCC_HIT_sum'), per_launch(sys.argv[3], 'TCC_MISS_sum')
alg = [int(x) for x in re.search(r'algorithmic bytes per gang launch.*?\[(.*?)\]', open(sys.argv[4]).read()).group(1).split(',')]
key, source = sys.argv[5], sys.argv[6]
n = min(len(fetch), len(write))
hbm = [int((2 * fetch[i] + write[i]) * 1024) for i in range(n)]
rec = {'launches': n, 'hbm_bytes_by_launch': hbm, 'hbm_bytes_per_launch': int(sum(hbm) / max(n, 1)),
       'fetched_bytes_by_launch': [int(2 * f * 1024) for f in fetch[:n]], 'written_bytes_by_launch': [int(w * 1024) for w in write[:n]],
       'algorithmic_bytes_by_launch': alg, 'algorithmic_bytes_per_launch': int(sum(alg) / max(len(alg), 1)),
       'over_fetch': round(sum(2 * f * 1024 for f in fetch[:n]) / max(sum(alg) - sum(w * 1024 for w in write[:n]), 1), 3),
       'l2_hit_pct': round(100.0 * sum(hit) / max(sum(hit) + sum(miss), 1), 1),
       'source': source, 'measured_on': os.environ.get('DAV_MEASURED_ON') or datetime.date.today().isoformat(),
       'kernel_source_hash': kernel_source_hash()}
path = os.path.join(ROOT, 'profiles', 'wgrad_traffic.json')
j = json.load(open(path)) if os.path.exists(path) else {'_comment': 'HBM-side traffic per gang weight-gradient launch (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, separate passes over tools/tn_gang_bench.py pmc_step); tools/wgrad_traffic.py; key = <config>_b<batch>'}
j[key] = rec
json.dump(j, open(path, 'w'), indent=1)
print(key, rec)

#!/usr/bin/env python3
"""profiles/dominant_kernel_traffic.json from the PMC passes of tools/collect_profiles.sh: per-launch HBM traffic of the big
NT GEMM launches = launch-count-weighted mean over the kernel instantiations of the 8-wave big-tile configurations
(FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, KiB -> bytes).
Usage: traffic_json.py <dir with r02_pmc_FETCH_SIZE.txt / r02_pmc_WRITE_SIZE.txt / r02_roofline_kernel_stats.csv> <key> [prefix]"""
import csv
import json
import os
import re
import sys

d, key = sys.argv[1], sys.argv[2]
pre = sys.argv[3] if len(sys.argv) > 3 else 'r02'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd._lib import kernel_source_hash  # noqa: E402
import datetime  # noqa: E402
MEASURED_ON = os.environ.get('DAV_MEASURED_ON') or datetime.date.today().isoformat()      # "<date>, <device / box>" of the PMC passes (tools/collect_r06.sh sets it on the GPU box)


def fam(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    return re.sub(r'\(.*', '', n)[:60]


def load(p):
    out = {}
    for line in open(p).read().splitlines()[1:]:
        parts = line.rsplit(None, 1)
        if len(parts) == 2:
            try:
                out[parts[0].strip()] = float(parts[1])
            except ValueError:
                pass
    return out


calls = {}
for r in csv.DictReader(open(os.path.join(d, f'{pre}_roofline_kernel_stats.csv'))):
    calls[fam(r['Name'])] = calls.get(fam(r['Name']), 0) + int(r['Calls'])
F, W = load(os.path.join(d, f'{pre}_pmc_FETCH_SIZE.txt')), load(os.path.join(d, f'{pre}_pmc_WRITE_SIZE.txt'))
big = re.compile(r'^void gemm_nt2(_grouped)?_kernel<(128, 128, 4, 2|128, 128, 2, 4|128, 256, 2, 4|256, 128, 4, 2)')
fams = [k for k in F if big.match(k) and k in W and calls.get(k, 0) > 0]
tot = sum(calls[k] for k in fams)
fetch = sum(calls[k] * F[k] for k in fams) / tot
write = sum(calls[k] * W[k] for k in fams) / tot
path = os.path.join(ROOT, 'profiles', 'dominant_kernel_traffic.json')
j = json.load(open(path))
j[key] = {'fetch_kib_per_launch_raw': {k: F[k] for k in fams}, 'write_kib_per_launch': {k: W[k] for k in fams},
          'launches_profiled': {k: calls[k] for k in fams}, 'fetch_kib_per_launch_raw_weighted': round(fetch, 1),
          'write_kib_per_launch_weighted': round(write, 1), 'hbm_bytes_per_launch': int((2 * fetch + write) * 1024),
          'source': f'profiles/{pre}_pmc_FETCH_SIZE.txt, profiles/{pre}_pmc_WRITE_SIZE.txt, profiles/{pre}_roofline_kernel_stats.csv (launch counts); tools/traffic_json.py',
          'kernel_source_hash': kernel_source_hash(), 'measured_on': MEASURED_ON}
json.dump(j, open(path, 'w'), indent=1)
print(key, tot, 'launches:', round(fetch), 'KiB fetched (raw),', round(write), 'KiB written ->', j[key]['hbm_bytes_per_launch'], 'bytes per launch')

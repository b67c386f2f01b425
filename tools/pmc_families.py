#!/usr/bin/env python3
"""Per-kernel-family averages of rocprofv3 --pmc counters. Usage: pmc_families.py counter_collection.csv"""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = re.sub(r'\(.*', '', n)[:60]
    acc[n][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[n][r['Counter_Name']] += 1
names = sorted({c for v in acc.values() for c in v})
print(f'{"kernel":60s} ' + ' '.join(f'{c[-16:]:>16s}' for c in names))
for n, v in sorted(acc.items(), key=lambda x: -x[1].get('SQ_WAVE_CYCLES', 0)):
    print(f'{n:60s} ' + ' '.join(f'{v.get(c, 0) / max(cnt[n].get(c, 1), 1):16.4g}' for c in names))

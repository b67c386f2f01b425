#!/usr/bin/env python3
"""Effective shader clock per kernel family: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, from a rocprofv3
--kernel-trace --pmc GRBM_GUI_ACTIVE run (counter_collection.csv + kernel_trace.csv in one directory).
Usage: clock_in_step.py <dir>"""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1]
cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
acc = defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc)):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE':
        continue
    t = dur.get(r['Dispatch_Id'])
    if not t:
        continue
    name = re.sub(r'\(anonymous namespace\)::', '', t[1])
    name = re.sub(r'\(.*', '', name)[:70]
    a = acc[name]
    a[0] += float(r['Counter_Value']); a[1] += t[0]; a[2] += 1
tot_c = sum(a[0] for a in acc.values()); tot_t = sum(a[1] for a in acc.values())
print(f'all kernels: {tot_c / 8 / tot_t:.3f} GHz effective over {tot_t / 1e6:.1f} ms of kernel time')
for n, a in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'{a[0] / 8 / a[1]:6.3f} GHz  {a[1] / 1e6:8.2f} ms  {a[2]:6d} calls  {n}')

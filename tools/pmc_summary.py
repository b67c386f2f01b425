#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per launch for kernels whose name contains a pattern.
Usage: pmc_summary.py counter_collection.csv COUNTER name_substring"""
import csv
import sys

path, counter, pat = sys.argv[1:4]
tot, n = 0.0, 0
for r in csv.DictReader(open(path)):
    if r['Counter_Name'] == counter and pat in r['Kernel_Name']:
        tot += float(r['Counter_Value'])
        n += 1
print(f'{counter} {pat!r}: launches {n} avg {tot / max(n, 1):.1f}')

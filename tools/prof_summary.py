#!/usr/bin/env python3
"""Per-step summary of a rocprofv3 --stats kernel_stats.csv. Usage: prof_summary.py file.csv passes [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
passes = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r['TotalDurationNs']) for r in rows) / passes / 1e6
for r in rows[:top]:
    per = float(r['TotalDurationNs']) / passes / 1e6
    print(f"{per:7.2f} ms/step {100 * per / tot:5.1f}%  calls/step {int(r['Calls']) / passes:6.1f}  avg {float(r['AverageNs']) / 1e3:7.1f}us  {r['Name'][:100]}")
print(f'total kernel time per step: {tot:.2f} ms')

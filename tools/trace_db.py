#!/usr/bin/env python3
"""Timeline of ONE replayed step out of a rocprofv3 --kernel-trace result (rocpd sqlite .db or kernel_trace.csv):
step span, concurrency profile, per-family time, and (with --dump A:B) the kernels between two offsets in µs.
Usage: trace_db.py results.db [--step -2] [--dump 1500:2000]"""
import csv
import re
import sqlite3
import sys
from collections import Counter, defaultdict


def load(path):
    if path.endswith('.db'):
        db = sqlite3.connect(path)
        rows = db.execute('select start, end, name, queue_id, stream_id, grid_x, workgroup_x from kernels').fetchall()
        return sorted((int(s), int(e), n, str(q), str(st), int(gx) // max(int(wx), 1)) for s, e, n, q, st, gx, wx in rows)
    rows = list(csv.DictReader(open(path)))
    return sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id'], r.get('Stream_Id', '0'),
                   int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)) for r in rows)


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*', '', n)[:56]


def main():
    ev = load(sys.argv[1])
    step = int(sys.argv[sys.argv.index('--step') + 1]) if '--step' in sys.argv else -2
    ad = [e for e in ev if 'adamw' in e[2]]
    a0, a1 = ad[step - 1][1], ad[step][1]
    win = [e for e in ev if a0 < e[0] <= a1]
    print(f'step span {(a1 - a0) / 1e6:.3f} ms, {len(win)} kernels, queues {dict(Counter(e[3] for e in win))}')
    pts = []
    for s, e, *_ in win:
        pts += [(s, 1), (e, -1)]
    pts.sort()
    cur, last, prof = 0, pts[0][0], Counter()
    for t, d in pts:
        prof[cur] += t - last
        cur += d
        last = t
    print('time by #concurrent kernels (ms):', {k: round(v / 1e6, 2) for k, v in sorted(prof.items())})
    fam = defaultdict(lambda: [0, 0])
    for s, e, n, *_ in win:
        fam[short(n)][0] += e - s
        fam[short(n)][1] += 1
    tot = sum(v[0] for v in fam.values())
    print(f'sum of kernel durations {tot / 1e6:.2f} ms')
    for n, (d, c) in sorted(fam.items(), key=lambda x: -x[1][0])[:28]:
        print(f'  {d / 1e6:7.2f} ms {c:5d} x {d / c / 1e3:7.1f} us  {n}')
    if '--dump' in sys.argv:
        lo, hi = (float(x) for x in sys.argv[sys.argv.index('--dump') + 1].split(':'))
        for s, e, n, q, st, g in win:
            if lo <= (s - a0) / 1e3 <= hi:
                print(f'{(s - a0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{q} s{st} g{g:5d} {short(n)}')


if __name__ == '__main__':
    main()

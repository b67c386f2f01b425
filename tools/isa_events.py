#!/usr/bin/env python3
"""Compact event trace of one kernel in a hipcc -S dump (M mfma, D LDS-DMA x4, p LDS-DMA dword, r ds_read tr, R ds_read b128,
| barrier, [vN] / [lN] waits, other memory ops by name): isa_events.py file.s name_substring [--mfma-only-region]"""
import re
import sys

src = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2]
for i, l in enumerate(src):
    if re.match(r'^\S*' + re.escape(pat) + r'\S*:', l):
        j = i
        while not src[j].startswith('.Lfunc_end'):
            j += 1
        ev = []
        for t in (x.strip() for x in src[i:j]):
            if t.startswith('v_mfma'): ev.append('M')
            elif 'global_load_lds_dwordx4' in t: ev.append('D')
            elif re.match(r'global_load_lds_dword\s', t): ev.append('p')
            elif 'ds_read_b64_tr' in t: ev.append('t')
            elif t.startswith('ds_read_b128'): ev.append('T')
            elif t.startswith('s_barrier'): ev.append('|')
            elif t.startswith('s_waitcnt'):
                a = re.search(r'vmcnt\((\d+)\)', t); b = re.search(r'lgkmcnt\((\d+)\)', t)
                ev.append('[' + ('v' + a.group(1) if a else '') + ('l' + b.group(1) if b else '') + ']')
            elif re.match(r'(global|buffer|flat|scratch)_', t): ev.append('{' + t.split()[0] + '}')
        out = ''.join(ev)
        for ch in 'MDptT':
            out = re.sub('(' + ch + '+)', lambda m: ch + str(len(m.group(1))) + ' ', out)
        print(src[i])
        print(out)
        print()

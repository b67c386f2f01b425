#!/usr/bin/env python3
"""Count the s_waitcnt forms, DMA pieces, MFMAs and barriers of one kernel in a hipcc -S dump: isa_waits.py file.s name_substring"""
import re
import sys
from collections import Counter

src = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2]
i = 0
while i < len(src):
    m = re.match(r'^(\S*' + re.escape(pat) + r'\S*):', src[i])
    if m and not src[i].startswith('\t'):
        j = i + 1
        while j < len(src) and not src[j].startswith('.Lfunc_end'):
            j += 1
        body = src[i:j]
        c = Counter(re.sub(r'\s+', ' ', l.split(';')[0].strip()) for l in body if 's_waitcnt' in l)
        text = '\n'.join(body)
        print(m.group(1), len(body), 'lines')
        for k, v in sorted(c.items(), key=lambda x: -x[1]):
            print('   ', v, k)
        print('    glds x4', text.count('global_load_lds_dwordx4'), '| glds dword', len(re.findall(r'global_load_lds_dword\s', text)),
              '| mfma', text.count('v_mfma'), '| s_barrier', text.count('s_barrier'), '| ds_read_b64_tr', text.count('ds_read_b64_tr_b16'),
              '| scratch', text.count('scratch_'))
        i = j
    i += 1

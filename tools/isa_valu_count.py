#!/usr/bin/env python3
"""VALU work of the attention kernels' key loops from a hipcc -S dump: for every kernel whose name contains the pattern, the innermost
loop that holds v_exp_f32 — its plain VALU instructions, transcendentals, MFMAs, LDS reads and cross-lane ops per iteration.
Usage: hipcc -S --cuda-device-only ... attention.hip -o a.s; isa_valu_count.py a.s 'attn_fwd_kernelILi32ELi32ELb0ELi1'"""
import re
import sys

src = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2]
for i, l in enumerate(src):
    if re.match(r'^\S*' + re.escape(pat) + r'\S*:', l):
        j = i
        while not src[j].startswith('.Lfunc_end'):
            j += 1
        body = [x.strip() for x in src[i:j]]
        labels = {m.group(1): k for k, t in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', t)] if m}
        loops = []
        for k, t in enumerate(body):
            m = re.match(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', t) or re.match(r's_branch\s+(\.LBB\d+_\d+)', t)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                loops.append((labels[m.group(1)], k))
        best = None
        for (a, b) in loops:
            seg = body[a:b + 1]
            if any(t.startswith('v_exp_f32') for t in seg) and any(t.startswith('v_mfma') for t in seg):
                if best is None or (b - a) < (best[1] - best[0]):
                    best = (a, b)
        print(l.split(':')[0])
        if best is None:
            print('  no loop with v_exp_f32 + v_mfma found')
            continue
        seg = [t for t in body[best[0]:best[1] + 1] if t and not t.startswith(('.', ';', '//')) and not t.endswith(':')]
        mfma = sum(t.startswith('v_mfma') for t in seg)
        trans = sum(t.startswith(('v_exp_f32', 'v_log_f32', 'v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32')) for t in seg)
        xlane = sum(t.startswith(('v_permlane', 'ds_bpermute', 'ds_swizzle', 'v_readlane', 'v_readfirstlane')) or 'dpp' in t for t in seg)
        valu = sum(t.startswith('v_') for t in seg) - mfma - trans
        pk = sum(t.startswith('v_pk_') for t in seg)
        lds = sum(t.startswith('ds_read') for t in seg)
        salu = sum(t.startswith('s_') for t in seg)
        print(f'  innermost exp loop: {len(seg)} instructions: {valu} plain VALU (of them {pk} packed, {xlane} cross-lane), {trans} transcendental, {mfma} MFMA, {lds} LDS reads, {salu} scalar')

#!/usr/bin/env python3
"""Instruction mix of the largest loop of every kernel whose mangled name contains one of the given substrings.
usage: isa_loop_stats.py <file.s> <substr> [<substr> ...]   (the .s comes from hipcc -save-temps=obj)"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
funcs = re.split(r'\n(?=_Z\S+:)', s)
for f in funcs:
    name = f.split(':')[0]
    if not any(k in name for k in sys.argv[2:]):
        continue
    labels = [(m.start(), m.group(1)) for m in re.finditer(r'\n(\.LBB\d+_\d+):', f)]
    best = None
    for pos, lab in labels:
        for m in re.finditer(r's_cbranch_\w+ ' + re.escape(lab) + r'\b', f):
            if m.start() > pos:
                seg = f[pos:m.start()]
                n = len(seg.split('\n'))
                if best is None or n > best[0]:
                    best = (n, lab, seg)
    if not best:
        continue
    ins = [l.strip().split()[0] for l in best[2].split('\n')
           if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().split()[0].endswith(':')]
    c = Counter()
    for i in ins:
        if i.startswith('v_mfma'):
            c['mfma'] += 1
        elif i.startswith('v_'):
            c['valu'] += 1
        elif i.startswith('s_waitcnt'):
            c['waitcnt'] += 1
        elif i.startswith('s_'):
            c['salu'] += 1
        elif i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
            c['vmem'] += 1
        elif i.startswith('ds_'):
            c['lds'] += 1
        else:
            c['other'] += 1
    print(name[:70], 'loop', best[1], dict(c))
    print('   top valu:', Counter(i for i in ins if i.startswith('v_') and not i.startswith('v_mfma')).most_common(16))
    print('   lds:', Counter(i for i in ins if i.startswith('ds_')).most_common(6), 'vmem:', Counter(i for i in ins if i.startswith(('global_', 'buffer_'))).most_common(6))

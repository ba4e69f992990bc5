#!/usr/bin/env python3
"""Static instruction mix of one kernel from hipcc's -S output.  usage: isa_mix.py file.s kernel_substring [top]
(loops are counted once: use SQ counters for dynamic counts)"""
import re, sys
from collections import Counter
txt = open(sys.argv[1]).read()
sub = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
for f in re.split(r'\n(?=_Z\w+:)', txt):
    name = f.split(':')[0].strip()
    if sub not in name:
        continue
    lines = [l.strip() for l in f.split('\n')]
    ins = [l for l in lines if l and not l.startswith(('.', ';', '//')) and not l.endswith(':')]
    c = Counter(l.split()[0] for l in ins)
    v = sum(n for k, n in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
    print(name, 'total', sum(c.values()), 'valu', v, 'mfma', sum(n for k, n in c.items() if k.startswith('v_mfma')),
          'salu', sum(n for k, n in c.items() if k.startswith('s_')), 'vmem', sum(n for k, n in c.items() if k.startswith(('global_', 'buffer_', 'scratch_', 'flat_'))),
          'lds', sum(n for k, n in c.items() if k.startswith('ds_')))
    for k, n in c.most_common(top):
        print('    %-28s %d' % (k, n))

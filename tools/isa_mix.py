#!/usr/bin/env python3
"""Static instruction mix of kernels in a hipcc -S listing: isa_mix.py file.s <substring> [...]"""
import re, sys, collections
lines = open(sys.argv[1]).read().splitlines()
pats = sys.argv[2:]
i = 0
while i < len(lines):
    m = re.match(r'(_Z\w+):', lines[i])
    if m and any(p in m.group(1) for p in pats):
        name = m.group(1); c = collections.Counter(); i += 1
        while i < len(lines) and 's_endpgm' not in lines[i]:
            t = lines[i].strip()
            mm = re.match(r'([a-z][a-z_0-9]+)(\s|$)', t)
            if mm and not t.endswith(':'): c[mm.group(1)] += 1
            i += 1
        tot = sum(c.values())
        cls = collections.Counter()
        for k, v in c.items():
            cls['v_pk' if k.startswith('v_pk') else 'valu' if k.startswith('v_') else 'ds' if k.startswith('ds_') else 'vmem' if k.startswith(('global_', 'buffer_', 'scratch_', 'flat_')) else 'salu'] += v
        print(name, tot, dict(cls)); print('   ', c.most_common(30))
    i += 1

#!/usr/bin/env python3
"""Instruction statistics of one kernel in a hipcc -save-temps .s file: isa_stats.py file.s kernel-name-substring [--loops]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
start = next(i for i, l in enumerate(s) if re.match(r'^_Z\S*%s\S*:' % re.escape(name), l))
end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
meta_end = next(i for i in range(end, len(s)) if '.end_amdhsa_kernel' in s[i])
code = [l.strip() for l in s[start + 1:end + 1]]
ins = [l for l in code if l and not l.startswith(';') and not l.startswith('.')]
c = Counter(l.split()[0] for l in ins)
print('instructions', len(ins), ' valu', sum(v for k, v in c.items() if k.startswith('v_')), ' salu', sum(v for k, v in c.items() if k.startswith('s_')))
print({k: v for k, v in c.items() if k.split('_')[0] in ('flat', 'scratch', 'ds', 'global', 'buffer')})
for l in s[end:meta_end]:
    if any(k in l for k in ('next_free_vgpr', 'next_free_sgpr', 'group_segment_fixed', 'private_segment_fixed', 'accum_offset')):
        print(l.strip())
if len(sys.argv) > 3:
    # innermost loops: backward branches
    labels = {m.group(1): i for i, l in enumerate(code) for m in [re.match(r'^(\.LBB\S+):', l)] if m}
    for i, l in enumerate(code):
        m = re.match(r's_cbranch_\w+ (\S+)|s_branch (\S+)', l)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt] < i:
                seg = [x for x in code[labels[tgt]:i + 1] if x and not x.startswith(';') and not x.startswith('.')]
                cc = Counter(x.split()[0].split('_')[0] for x in seg)
                print(f'loop {tgt}: {len(seg)} instructions', dict(cc))

#!/usr/bin/env python3
"""Static instruction counts of every innermost loop of a kernel that contains a given marker instruction (default v_perm_b32: the z-streaming
plane loops), from `hipcc -S --cuda-device-only` output - to see whether an edit elsewhere in the kernel changed the code of its hot loops.
    python tools/isa_loops.py file.s <mangled kernel name prefix> [marker]"""
import collections, re, sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 else 'v_perm_b32'
i = s.index(name + ''); i = s.index(':', s.index('\n' + name, 0) if ('\n' + name) in s else i); j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
labs = {}
for n, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labs[m.group(1)] = n
loops = []
for n, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labs and labs[m.group(1)] < n:
        loops.append((labs[m.group(1)], n))
marks = [n for n, l in enumerate(body) if marker in l]
inner = []
for h, t in loops:
    if not any(h < m < t for m in marks): continue
    if any(h <= h2 and t2 <= t and (h2, t2) != (h, t) and any(h2 < m < t2 for m in marks) for h2, t2 in loops): continue
    inner.append((h, t))
def stats(lines):
    ops = collections.Counter()
    for l in lines:
        t = l.strip().split()
        if not t or t[0].startswith(('.', ';')) or t[0].endswith(':'): continue
        ops[t[0]] += 1
    tot = collections.Counter()
    for o, c in ops.items():
        tot['VALU' if o.startswith('v_') else 'SALU' if o.startswith('s_') else 'LDS' if o.startswith('ds_') else 'VMEM'] += c
    return ops, tot
for m in re.finditer(r'; (NumVgprs|NumSgprs|ScratchSize|Occupancy|SGPRSpill|VGPRSpill)[^\n]*', s[j:j + 4000]):
    print(m.group(0))
for h, t in inner:
    ops, tot = stats(body[h:t + 1])
    sp = sum(c for o, c in ops.items() if 'scratch' in o) 
    print(f"# loop lines {h}-{t} ({t - h + 1}): " + ", ".join(f"{k} {v}" for k, v in sorted(tot.items())) + f"; scratch ops {sp}; v_readlane/writelane {ops.get('v_readlane_b32', 0) + ops.get('v_writelane_b32', 0)}; s_waitcnt {ops.get('s_waitcnt', 0)}")

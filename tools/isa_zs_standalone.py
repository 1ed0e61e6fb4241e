#!/usr/bin/env python3
"""Static instruction counts of the z-streaming loop of the STAND-ALONE kernel (tools/zbench.hip -S --cuda-device-only): two plane steps of R rows,
with a cycle estimate from the measured per-instruction costs (profiles/r05a_mfma_coissue_and_op_costs.txt, 4 waves/SIMD)."""
import collections, re, sys
s = open(sys.argv[1]).read()
name = sys.argv[2] if len(sys.argv) > 2 else '_ZN3trx21affine_zstream_kernelILi0E'
i = s.index(name); i = s.index(':', i); j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
perm = [n for n, l in enumerate(body) if 'v_perm_b32' in l]
a, b = max(0, perm[0] - 600), min(len(body), perm[-1] + 300)
labs = {}
for n in range(a, b):
    m = re.match(r'^(\.LBB\d+_\d+):', body[n])
    if m: labs[m.group(1)] = n
best = None
for n in range(a, b):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', body[n])
    if m and m.group(1) in labs and labs[m.group(1)] < n:
        h = labs[m.group(1)]
        if h < perm[0] and n > perm[-1] and (best is None or n - h < best[1] - best[0]): best = (h, n)
loop = body[best[0]:best[1] + 1]
ops = collections.Counter(); cyc = 0.0; sg = 0
fast = ('v_add_f32', 'v_sub_f32', 'v_mul_f32', 'v_fmac_f32', 'v_fma_f32', 'v_mov_b32', 'v_add_u32', 'v_and_b32', 'v_max_f32', 'v_min_f32', 'v_or_b32')
for l in loop:
    t = l.strip().split()
    if not t or t[0].startswith(('.', ';')) or t[0].endswith(':'): continue
    ops[t[0]] += 1
    if t[0].startswith('v_'):
        base = t[0].replace('_e32', '').replace('_e64', '')
        has_s = bool(re.search(r'[ ,]s\d+|s\[\d+:\d+\]|vcc|0x[0-9a-f]+', ' '.join(t[1:])))
        if base in fast and not has_s: cyc += 2.5
        else: cyc += 4.4 if base not in ('v_readlane_b32',) else 4.5
        if base in fast and has_s: sg += 1
tot = collections.Counter()
for o, c in ops.items():
    tot['VALU' if o.startswith('v_') else 'SALU' if o.startswith('s_') else 'LDS' if o.startswith('ds_') else 'VMEM'] += c
nv = 8
print(f"# loop: {len(loop)} lines; per voxel-wave (/{nv}): " + ", ".join(f"{k} {v / nv:.1f}" for k, v in sorted(tot.items())) + f"; VALU cycle estimate {cyc / nv:.1f} per voxel-wave; fast-class ops with a scalar operand {sg / nv:.2f}")
for o, c in sorted(ops.items(), key=lambda kv: -kv[1]):
    if c >= 2: print(f"{o:28s} {c:5d} {c / nv:8.2f}")

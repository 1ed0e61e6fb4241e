#!/usr/bin/env python3
"""Itemise the z-streaming F1 loop: static instruction counts of ONE pair of plane steps of affine_tile_dual_kernel<0,0>'s
z-streaming body, by opcode and by what the opcode does in this loop.  Input: the device assembly of affine.hip
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -S --cuda-device-only torchregister_amd/csrc/affine.hip -o affine.s
    python3 tools/isa_zstream.py affine.s > profiles/r04_zstream_loop_isa.txt
The loop body holds TWO plane steps (tvA / tvB register sets), each 4 rows per thread: per voxel-wave = counts / 8."""
import collections, re, sys
s = open(sys.argv[1]).read()
name = '_ZN3trx23affine_tile_dual_kernelILi0ELi0EEEv'
i = s.index(name); i = s.index(':', i); j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
perm = [n for n, l in enumerate(body) if 'v_perm_b32' in l]
a, b = perm[0] - 600, perm[-1] + 300
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
ops = collections.Counter()
for l in loop:
    t = l.strip().split()
    if not t or t[0].startswith(('.', ';')) or t[0].endswith(':'): continue
    ops[t[0]] += 1
role = {
    'v_pk_fma_f32': 'trilinear lerp (z, y on pairs) + accumulation of sum(q g), sum(q g yn), moments', 'v_pk_add_f32': 'lerp differences + accumulation', 'v_pk_mul_f32': 'accumulation',
    'v_fmac_f32_e32': 'lerp in x / gradient, moments', 'v_fma_f32': 'lerp in x / gradient', 'v_sub_f32_e32': 'lerp differences', 'v_mul_f32_e32': 'yn g for the yn-weighted sums',
    'v_add_f32_e32': 'coordinates (per row: base + row term) and U += running sums (9 per plane)', 'v_cvt_flr_i32_f32': 'floor of the three coordinates',
    'v_fract_f32_e32': 'interpolation fractions', 'v_perm_b32': 'ring slot of floor(z), floor(z) + 1 from the byte tables', 'v_mad_u32_u24': 'LDS address: slot * plane pitch',
    'v_mad_i32_i24': 'LDS address: row * row pitch', 'v_lshl_add_u32': 'LDS address: x * 4 + base', 'v_add_u32_e32': 'byte selector of v_perm (floor(z) + selc)',
    'v_readlane_b32': 'per-plane constants (zn, unnorm(zn)) from the lane tables', 'ds_read2_b32': 'the four x-pairs of a voxel', 's_barrier': 'one per plane step',
    'global_load_dword': 'targets of the next plane (4 rows)', 'global_load_lds_dwordx4': 'ring DMA: two pieces per wave and plane',
}
tot = collections.Counter()
for o, c in ops.items():
    tot['VALU' if o.startswith('v_') else 'SALU' if o.startswith('s_') else 'LDS' if o.startswith('ds_') else 'VMEM'] += c
print(f"# z-streaming loop of affine_tile_dual_kernel<0,0>: {len(loop)} lines of assembly, two plane steps of 4 rows per thread (static counts; per voxel-wave = / 8)")
print("# class totals:", ", ".join(f"{k} {v} ({v / 8:.1f} per voxel-wave)" for k, v in sorted(tot.items())))
print(f"{'opcode':28s} {'count':>5s} {'per voxel-wave':>15s}  role in this loop")
for o, c in sorted(ops.items(), key=lambda kv: -kv[1]):
    if c >= 2 or o in role: print(f"{o:28s} {c:5d} {c / 8:15.2f}  {role.get(o, '')}")

#!/usr/bin/env python3
"""Extension of eft_lds_banks.py: LDS cycles of the exact-footprint gather for wave shapes with a y extent (x * y * z lanes; VERDICT r4 item 4 ii).  CPU model."""
import itertools, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from eft_plan_check import plan, rot
from eft_lds_banks import cycles32, packed, box

def sim3(A, layout, shape, ntiles=3, seed=0):
    nx, ny, nz = shape
    rng = np.random.default_rng(seed)
    tot = n = 0
    lanes = [(lx, ly, lz) for lz in range(nz) for ly in range(ny) for lx in range(nx)]
    for _ in range(ntiles):
        bf = rng.random(3)
        for ox, oy, oz in itertools.product(range(0, 16, nx), range(0, 16, ny), range(0, 16, nz)):
            if rng.random() > 0.25: continue
            q = np.array([[ox + lx, oy + ly, oz + lz] for lx, ly, lz in lanes], float)
            fi = np.floor(q @ A.T + bf).astype(int)
            for dx, dy, dz in itertools.product((0, 1), repeat=3):
                ad = [layout(x + dx, y + dy, z + dz) for x, y, z in fi]
                tot += cycles32(ad[:32]) + cycles32(ad[32:])
            n += 1
    return tot / n

poses = {"value_rot pose": rot(.5, .4, .3) @ np.diag([1.05, .95, 1.02]), "rigid rand-init pose": rot(0.4963, 0.7682, 0.0885), "R(.3,.3,.3)": rot(.3, .3, .3), "Rz(.6)": rot(0, 0, .6)}
for name, A in poses.items():
    L = packed(A)
    out = []
    for shape in [(16,1,4),(8,1,8),(4,4,4),(8,2,4),(8,4,2),(4,8,2),(16,2,2),(16,4,1),(8,8,1),(2,8,4),(4,2,8)]:
        out.append(f"{'x'.join(map(str,shape))}: {sim3(A, L, shape):.1f}")
    print(f"{name:22s} " + "  ".join(out), flush=True)

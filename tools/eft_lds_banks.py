#!/usr/bin/env python3
"""LDS bank-conflict model of the exact-footprint tile's gather (runs on the CPU).  For the bench's rotated poses it
counts the LDS cycles of the 8 data dwords a voxel-wave reads (4 x ds_read2_b32: two passes of 32 lanes each, a pass costs the largest number
of DISTINCT dwords on one bank), for
  * the kernel's layout (rows packed one after the other in 16-byte granules, iy fastest) under three wave shapes,
  * the bounding-box layout of the tile kernels (28-dword pitch),
  * the packed layout with each z-plane of rows started on a chosen bank offset ("steering"; 0 ... 28 dwords of padding per plane),
  * and the best a bank-linear layout (bank = x + cy y + cz z mod 32, any cy, cz; distinct cells keep distinct addresses) could do for that pose - a lower bound for row pitches,
    paddings and XOR swizzles of the linear family.
The ideal is 16 cycles (8 dwords x 2 passes).  The measured counter (profiles/r04f_eft_kernel_pmc.txt) is 54 LDS cycles per voxel-wave of
which 27 are conflicts, table reads included; this model covers the data reads only.
      python3 tools/eft_lds_banks.py"""
import itertools, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from eft_plan_check import plan, rot


def cycles32(addrs):
    banks = {}
    for a in addrs: banks.setdefault(a % 32, set()).add(a)
    return max(len(v) for v in banks.values())


def sim(A, layout, lanes_of_waves, ntiles=4, seed=0):
    rng = np.random.default_rng(seed)
    tot = n = 0
    for _ in range(ntiles):
        bf = rng.random(3)
        for j in range(0, 16, 3):
            for w in range(len(lanes_of_waves) // 64):
                q = np.array([[lx, j, lz] for lx, lz in lanes_of_waves[w * 64:(w + 1) * 64]], float)
                fi = np.floor(q @ A.T + bf).astype(int)
                for dx, dy, dz in itertools.product((0, 1), repeat=3):
                    ad = [layout(x + dx, y + dy, z + dz) for x, y, z in fi]
                    tot += cycles32(ad[:32]) + cycles32(ad[32:])
                n += 1
    return tot / n


def lanemap(nx, nz):
    return [(wx * nx + lx, wz * nz + lz) for wz in range(16 // nz) for wx in range(16 // nx) for lz in range(nz) for lx in range(nx)]


def packed(A, plane_step=None):
    rows, _, _ = plan(A)
    E, off, prev = {}, 0, None
    for z in sorted(set(k[1] for k in rows)):
        ys = sorted(k[0] for k in rows if k[1] == z)
        if plane_step is not None and prev is not None:
            off += (((prev + plane_step) - (off - rows[(ys[0], z)][0])) % 32 + 3) // 4 * 4 % 32
        prev = (off - rows[(ys[0], z)][0]) % 32
        for y in ys:
            wlo, whi = rows[(y, z)]
            E[(y, z)] = off - wlo
            off += -(-(whi - wlo + 1) // 4) * 4
    return lambda x, y, z: E.get((y, z), 0) + x


def box(A, bw=28, bh=27):
    lo = np.floor(np.minimum(A * 15, 0).sum(1) - 0.05).astype(int); lo[0] &= ~3
    return lambda x, y, z: (x - lo[0]) + bw * ((y - lo[1]) + bh * (z - lo[2]))


if __name__ == "__main__":
    poses = {"value_rot pose": rot(.5, .4, .3) @ np.diag([1.05, .95, 1.02]), "rigid rand-init pose": rot(0.4963, 0.7682, 0.0885), "R(.7,.8,.6)": rot(.7, .8, .6),
             "R(.3,.3,.3)": rot(.3, .3, .3), "Rz(.6)": rot(0, 0, .6)}
    kernel = lanemap(16, 4)
    for name, A in poses.items():
        shapes = "  ".join(f"{s}: {sim(A, packed(A), lanemap(*map(int, s.split('x')))):.1f}" for s in ("16x4", "8x8", "4x16"))
        steer = min((sim(A, packed(A, cz), kernel), cz) for cz in range(0, 32, 4))
        linear = min((sim(A, (lambda x, y, z, cy=cy, cz=cz: x + (4096 + cy) * y + (4096 * 64 + cz) * z), kernel, ntiles=2), cy, cz) for cy in range(0, 32, 2) for cz in range(0, 32, 2))
        print(f"{name:22s} ideal 16.0 | packed rows, wave shape (x by z lanes) {shapes} | bounding box 28-pitch {sim(A, box(A), kernel):.1f}"
              f" | best plane steering {steer[0]:.1f} (step {steer[1]}) | best bank-linear layout {linear[0]:.1f} (cy {linear[1]}, cz {linear[2]})", flush=True)

#!/usr/bin/env python3
"""L2 model of the exact-footprint kernel's fetches (runs on the CPU).  One 256^3 pair at the bench's rotated pose: every
XCD is an LRU cache of 4 MiB in 128-byte lines; its 64 resident blocks walk their columns of 16^3 tiles, each tile requesting the lines of its
plan's granules (tools/eft_plan_check.py) and of its 16 x 16 target rows of 64 bytes.  Blocks run with a random phase of up to `drift`
tile-times against each other, a tile's lines arriving in four chunks.  Printed: misses per voxel-wave (x 128 B = bytes fetched from the
fabric), for the product's column-to-XCD map (slabs of 16 x 2 columns) and for compact 8 x 4 patches.
Measured on the GPU (profiles/r04h_pose_pmc.txt): 12.9 requests and 7.4 misses per voxel-wave = 2.04 GB per 8-pair launch.
      python3 tools/eft_l2_model.py"""
import os, sys
from collections import OrderedDict
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from eft_plan_check import plan, rot

S, T = 256, 16
A = rot(.5, .4, .3) @ np.diag([1.05, .95, 1.02])
CEN = np.array([S / 2] * 3)
ROWS, _, _ = plan(A)
RK = np.array([(dy, dz, wlo, whi) for (dy, dz), (wlo, whi) in ROWS.items()])


def tile_lines(X, Y, Z):
    R = np.floor(A @ (np.array([X * T, Y * T, Z * T], float) - CEN) + CEN).astype(int)
    y, z = R[1] + RK[:, 0], R[2] + RK[:, 1]
    x0 = R[0] + RK[:, 2]
    x1 = x0 + ((RK[:, 3] - RK[:, 2] + 4) // 4) * 4 - 1
    ok = (y >= 0) & (y < S) & (z >= 0) & (z < S) & (x1 >= 0) & (x0 < S)
    y, z, x0, x1 = y[ok], z[ok], np.clip(x0[ok], 0, S - 1), np.clip(x1[ok], 0, S - 1)
    base = (z * S + y) * S
    l0, l1 = (base + x0) // 32, (base + x1) // 32
    mov = np.unique(np.concatenate([l0, l1[l1 > l0], l0[l1 > l0 + 1] + 1]))
    yy, zz = np.meshgrid(np.arange(Y * T, Y * T + T), np.arange(Z * T, Z * T + T))
    return mov, -1 - np.unique(((zz * S + yy) * S + X * T).ravel() // 32)


def slabs(xcd):      # the product: block g of the flat grid -> column (g & 7) * 32 + (g >> 3), x fastest; two y segments of 8 tiles
    out = []
    for j in range(64):
        g = xcd + 8 * j
        yseg, cb = g // 256, g % 256
        col = (cb & 7) * 32 + (cb >> 3)
        out.append([(col % 16, yseg * 8 + t, col // 16) for t in range(8)])
    return out


def patches(xcd):    # TRX_EF_PATCH = 1
    PX, PZ = xcd % 2, xcd // 2
    return [[(PX * 8 + ix, yseg * 8 + t, PZ * 4 + iz) for t in range(8)] for yseg in range(2) for iz in range(4) for ix in range(8)]


def simulate(order, drift, cache_lines=4 * 1024 * 1024 // 128, chunks=4, seed=0):
    rng = np.random.default_rng(seed)
    miss = [0, 0]
    req = 0
    for xcd in range(8):
        blocks = order(xcd)
        phase = {bi: rng.random() * drift for bi in range(len(blocks))}
        ev = sorted((t + phase[bi] + k / chunks, bi, t, k) for bi, b in enumerate(blocks) for t in range(len(b)) for k in range(chunks))
        cache, memo = OrderedDict(), {}
        for _, bi, t, k in ev:
            tile = blocks[bi][t]
            if tile not in memo: memo[tile] = tile_lines(*tile)
            for which, arr in enumerate(memo[tile]):
                part = arr[k::chunks]
                req += len(part)
                for l in part.tolist():
                    if l in cache: cache.move_to_end(l)
                    else:
                        cache[l] = 1
                        if len(cache) > cache_lines: cache.popitem(last=False)
                        miss[which] += 1
    nvw = S ** 3 / 64
    return miss[0] / nvw, miss[1] / nvw, req / nvw


if __name__ == "__main__":
    print("unique lines per voxel-wave: moving 1.70 (part of the pre-image lies outside the volume), target 2.00")
    for drift in (0.0, 1.0, 2.0, 3.0, 4.0):
        for name, order in (("slabs 16 x 2 (product)", slabs), ("patches 8 x 4", patches)):
            m, t, r = simulate(order, drift)
            print(f"drift {drift:3.1f} tile-times  {name:24s} line requests {r:5.2f}  misses: moving {m:4.2f} + target {t:4.2f} = {m + t:4.2f} per voxel-wave = {(m + t) * 128 * S ** 3 / 64 * 8 / 1e9:4.2f} GB per 8-pair launch", flush=True)

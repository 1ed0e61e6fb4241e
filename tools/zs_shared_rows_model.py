#!/usr/bin/env python3
"""CPU model for VERDICT r3 item 2's first candidate: could a z-streaming thread derive floor / fract / ring address of its rows 1..3 from row 0
(rows differ by exactly one ring row next to the identity), behind a wave-uniform ballot test?  The shortcut is valid for a wave and plane when,
for every lane and each coordinate c, floor(i_c(row j)) == floor(i_c(row 0)) + j [c = y] / + 0 [c = x, z] for j = 1..3.  This counts the share of
(wave, plane) instances of one 256^3 volume where it holds, for a pose theta = I + eps * (theta* - I) along the bench's trajectory.
      python3 tools/zs_shared_rows_model.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import THETA_STAR

S = 256
ths = np.array(THETA_STAR, dtype=np.float64)
I = np.eye(3, 4)
xn = (2 * np.arange(S) + 1) / S - 1      # align_corners=False normalised centres
def unnorm(c): return ((c + 1) * S - 1) / 2
rng = np.random.default_rng(0)
for eps in (0.0, 0.01, 0.05, 0.1, 0.25, 0.5, 1.0):
    th = I + eps * (ths - I)
    ok = tot = 0
    for z in rng.choice(S, 24, replace=False):
        for y0 in range(0, S, 4):               # a thread's 4 rows: y0 .. y0 + 3 (the kernel's row groups are interleaved differently; the test is per 4 consecutive rows of one x)
            for x0 in range(0, S, 64):
                x = xn[x0:x0 + 64]
                good = np.ones(64, bool)
                base = None
                for j in range(4):
                    yy = xn[y0 + j]; zz = xn[z]
                    c = [unnorm(th[r, 0] * x + th[r, 1] * yy + th[r, 2] * zz + th[r, 3]) for r in range(3)]   # source x, y, z
                    fl = [np.floor(v) for v in c]
                    if j == 0: base = fl
                    else: good &= (fl[0] == base[0]) & (fl[1] == base[1] + j) & (fl[2] == base[2])
                ok += good.all(); tot += 1
    print(f"theta = I + {eps:4.2f} (theta* - I): the shared-row shortcut is valid for {100 * ok / tot:5.1f} % of the (wave, plane, row group) instances")

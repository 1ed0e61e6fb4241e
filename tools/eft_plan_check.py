#!/usr/bin/env python3
"""numpy prototype of the exact-footprint plan (csrc/affine_eft.h: ef_row_window, ef_dims) and its checks - runs on the CPU, no GPU needed:
  * the plan's per-row x-windows against a brute-force union of the cells a 16^3 tile touches over random fractional tile origins
    (no violation allowed), for the bench poses and random rotations x zooms x shears;
  * granule counts: the plan (G, what the kernel stages per tile), the exact per-tile mean, the bounding box;
  * the closed-form zonotope estimate of G that was considered for a lane-parallel offer test (plan / estimate = 0.79 ... 0.90).
      python3 tools/eft_plan_check.py [n_random]"""
import itertools, math, sys
import numpy as np
EPS = 0.05


def rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def plan(A, T=16):
    """{(dy, dz): (wlo, whi)} relative to R = floor(image of the tile's corner), granule count"""
    ex = T - 1
    E = A * ex
    ext_lo = np.minimum(E, 0).sum(1); ext_hi = np.maximum(E, 0).sum(1)
    N = np.linalg.inv(A)
    dy0 = int(np.floor(ext_lo[1] - EPS)); dz0 = int(np.floor(ext_lo[2] - EPS))
    ny = int(np.floor(ext_hi[1] + EPS)) + 3 - dy0; nz = int(np.floor(ext_hi[2] + EPS)) + 3 - dz0
    rows = {}
    for iz in range(nz):
        for iy in range(ny):
            dy, dz = dy0 + iy, dz0 + iz
            yc, zc = dy - 0.5, dz - 0.5
            lo, hi, ok = ext_lo[0] - EPS, ext_hi[0] + EPS, True
            for a in range(3):
                gc = N[a, 1] * yc + N[a, 2] * zc
                gh = (abs(N[a, 1]) + abs(N[a, 2])) * (1.5 + EPS)
                l, h, n = -gc - gh, ex - gc + gh, N[a, 0]
                if abs(n) < 1e-6:
                    ok = ok and l <= 1e-3 and h >= -1e-3
                else:
                    a0, a1 = l / n, h / n
                    lo, hi = max(lo, min(a0, a1)), min(hi, max(a0, a1))
            if ok and lo <= hi:
                rows[(dy, dz)] = (int(np.floor(lo - EPS)), int(np.floor(hi + EPS)) + 2)
    return rows, sum(-(-(whi - wlo + 1) // 4) for wlo, whi in rows.values()), (ny, nz)


def check(A, rows, T=16, n=12, seed=0):
    """cells touched by the 2x2x2 neighbourhoods of every voxel of the tile, over fractional origins, that the plan misses"""
    q = np.stack(np.meshgrid(np.arange(T), np.arange(T), np.arange(T), indexing='ij'), -1).reshape(-1, 3)
    rng = np.random.default_rng(seed)
    v, bad, sizes = q @ A.T, 0, []
    for i in range(n):
        bf = rng.random(3) if i > 1 else (np.zeros(3) if i == 0 else np.full(3, 0.99999))
        s = np.floor(v + bf).astype(int)
        cur = {}
        for d in itertools.product((0, 1), repeat=3):
            for x, y, z in s + np.array(d):
                r = rows.get((y, z))
                if r is None or x < r[0] or x > r[1]: bad += 1
                c = cur.setdefault((y, z), [x, x]); c[0] = min(c[0], x); c[1] = max(c[1], x)
        sizes.append(sum(-(-(c[1] - c[0] + 1) // 4) for c in cur.values()))
    return bad, float(np.mean(sizes))


def zonotope_estimate(A, T=16, hx=5.2, hyz=3.1):
    E = [A[:, a] * (T - 1) for a in range(3)]
    gens = E + [np.array([hx, 0, 0]), np.array([0, hyz, 0]), np.array([0, 0, hyz])]
    V = sum(abs(np.linalg.det(np.stack([gens[a], gens[b], gens[c]]))) for a, b, c in itertools.combinations(range(6), 3))
    g2 = [np.array([e[1], e[2]]) for e in E] + [np.array([hyz, 0]), np.array([0, hyz])]
    R = sum(abs(g2[a][0] * g2[b][1] - g2[a][1] * g2[b][0]) for a, b in itertools.combinations(range(5), 2))
    return (V + 3 * R) / 4


if __name__ == "__main__":
    nrand = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(5)
    tests = {"value_rot pose": rot(.5, .4, .3) @ np.diag([1.05, .95, 1.02]), "rigid rand-init pose": rot(0.4963, 0.7682, 0.0885), "R(.7,.8,.6)": rot(.7, .8, .6),
             "identity": np.eye(3), "Rz(.6)": rot(0, 0, .6), "Rx(.6)": rot(.6, 0, 0), "R(.5,.4,.3) x 1.1": rot(.5, .4, .3) * 1.1}
    for i in range(nrand):
        a = rng.random(3) * 1.2 * rng.choice([-1, 1], 3)
        tests[f"random {i}"] = rot(*a) @ np.diag(0.85 + 0.3 * rng.random(3)) + 0.08 * (rng.random((3, 3)) - 0.5)
    worst = 0.0
    for name, A in tests.items():
        rows, G, dims = plan(A)
        bad, exact = check(A, rows)
        bb = np.ptp(np.array([(r[0], k[0], k[1]) for k, r in rows.items()] + [(r[1], k[0], k[1]) for k, r in rows.items()]), axis=0) + 1
        est = zonotope_estimate(A)
        worst = max(worst, G / est)
        print(f"{name:22s} rows {len(rows):4d} {dims}  plan G = {G:5d} granules ({4 * G / 4096:.2f} floats / voxel)  exact per tile {exact:6.0f}  bounding box {np.prod(bb) / 4096:.2f} floats / voxel"
              f"  missed cells {bad}  G / zonotope estimate {G / est:.3f}")
        assert bad == 0
    print("largest plan / estimate:", round(worst, 3))

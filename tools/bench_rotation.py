#!/usr/bin/env python3
"""Affine-mode F1 steps against the rotation of theta ("no cliff"): 8 x 256^3, affine + NCC, lr = 0 so that theta stays
where it is put; rotations about z, about a general axis, and with the anisotropic zoom of bench.py's theta*.  us per pair-iteration."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
from bench import blobs_gpu, THETA_STAR

def rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx

dev = torch.device("cuda")
S, B = 256, 8
tgt = torch.cat([blobs_gpu((S,) * 3, 1000 + b, dev) for b in range(B)])
mov = torch.cat([blobs_gpu((S,) * 3, 2000 + b, dev) for b in range(B)])
cases = [("identity", torch.eye(3))] + [(f"Rz({a})", rot(0, 0, a)) for a in (0.02, 0.05, 0.1, 0.15, 0.2, 0.3, 0.6, 1.0)] + \
        [(f"Rx({a})", rot(a, 0, 0)) for a in (0.3, 0.6)] + [(f"Ry({a})", rot(0, a, 0)) for a in (0.3, 0.6)] + \
        [(f"R({a},{a},{a})", rot(a, a, a)) for a in (0.05, 0.1, 0.2, 0.3, 0.4)] + [("R(0.1,0.1,0.5)", rot(0.1, 0.1, 0.5)), ("R(0.7,0.8,0.6)", rot(0.7, 0.8, 0.6))] + \
        [("theta* of bench.py", torch.tensor(THETA_STAR)[:, :3])]
FLAGS = int(sys.argv[1]) if len(sys.argv) > 1 else 0   # trx_volumes.flags, e.g. 16 = TRX_FLAG_NO_ROT_DEEP_TILE
base = None
for name, R in cases:
    th = torch.cat([R.float(), torch.tensor([[0.01], [-0.02], [0.015]])], dim=1)[None].expand(B, 3, 4).contiguous()
    s = tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), lr=0.0, init=th, capacity=400, flags=FLAGS)
    s.run(60); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); s.run(100); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 100 / B
    base = base or us
    print(f"{name:24s} {us:7.1f} us per pair-iteration   x{us / base:4.2f} of identity   {8 * S**3 / us / 1e6:5.2f} TB/s algorithmic")

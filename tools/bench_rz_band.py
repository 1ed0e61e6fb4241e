#!/usr/bin/env python3
"""Development: 8 x 256^3 affine + NCC steps at rotations about z (and z plus a little x / y) between 0.2 and 1.0 rad - where the step kernels
hand a pair to GeomD, GeomRD or the exact-footprint kernel (dual_choice / eft_wants).  us per pair-iteration and rows_used[0]
(negative: the exact-footprint kernel took the pair).   python3 tools/bench_rz_band.py [flags]"""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
from bench import blobs_gpu, rot

dev = torch.device("cuda")
S, B = 256, 8
tgt = torch.cat([blobs_gpu((S,) * 3, 1000 + b, dev) for b in range(B)])
mov = torch.cat([blobs_gpu((S,) * 3, 2000 + b, dev) for b in range(B)])
cases = [(f"Rz({a})", rot(0, 0, a)) for a in (0.2, 0.22, 0.25, 0.28, 0.3, 0.35, 0.4, 0.45, 0.5, 0.6, 0.8, 1.0)] + \
        [(f"R(0.05,0.05,{a})", rot(0.05, 0.05, a)) for a in (0.3, 0.45, 0.6)] + [(f"R(0.1,0,{a})", rot(0.1, 0.0, a)) for a in (0.3, 0.45, 0.6)] + \
        [(f"Rx({a})", rot(a, 0, 0)) for a in (0.08, 0.1, 0.12, 0.15)] + [(f"Ry({a})", rot(0, a, 0)) for a in (0.08, 0.1, 0.12, 0.15)] + \
        [("R(0.08,0.08,0)", rot(0.08, 0.08, 0.0)), ("R(0.05,0.05,0.25)", rot(0.05, 0.05, 0.25)), ("R(0.03,0.03,0.3)", rot(0.03, 0.03, 0.3))] + \
        [(f"zoom {z}", z * torch.eye(3)) for z in (1.04, 1.06, 1.1)] + [("diag(1.05,.95,1.02)", torch.diag(torch.tensor([1.05, 0.95, 1.02]))),
         ("Rz(0.3) x 1.05", 1.05 * rot(0, 0, 0.3)), ("Rz(0.5) x 1.05", 1.05 * rot(0, 0, 0.5)), ("Rz(0.3) x 0.95", 0.95 * rot(0, 0, 0.3))]
FLAGS = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for name, R in cases:
    th = torch.cat([R.float(), torch.tensor([[0.01], [-0.02], [0.015]])], dim=1)[None].expand(B, 3, 4).contiguous()
    s = tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), lr=0.0, init=th, capacity=400, flags=FLAGS)
    s.run(60); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); s.run(100); e1.record(); torch.cuda.synchronize()
    print(f"{name:20s} {e0.elapsed_time(e1) * 1e3 / 100 / B:7.1f} us per pair-iteration   rows_used[0] = {s.rows_used().tolist()[0]}")

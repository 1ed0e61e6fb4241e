#!/usr/bin/env python3
"""Which launches should be offered the exact-footprint kernel?  Rotated pose R(0.5, 0.4, 0.3) diag(1.05, 0.95, 1.02), B pairs of S^3,
affine + NCC, lr = 0: us per step with the default offer rule, with the kernel forced (TRX_FLAG_EFT) and without it (TRX_FLAG_NO_EFT)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
from bench import blobs_gpu, pose_rot
dev = torch.device("cuda")
for B, S in ((1, 128), (2, 128), (4, 128), (8, 128), (16, 128), (1, 192), (2, 192), (4, 192), (1, 256), (2, 256), (3, 256), (4, 256), (1, 320), (16, 192)):
    shp = (S,) * 3
    tgt = torch.cat([blobs_gpu(shp, 1000 + b, dev) for b in range(B)])
    mov = torch.cat([blobs_gpu(shp, 2000 + b, dev) for b in range(B)])
    out = []
    for fl in (0, 1024 | 8, 512):
        s = tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), lr=0.0, init=pose_rot(dev, B), capacity=400, flags=fl)
        s.run(40); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); s.run(100); e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 10)
    print(f"{B:3d} x {S}^3: default {out[0]:7.1f} us per step | forced {out[1]:7.1f} | without {out[2]:7.1f}   (rows used {s.rows_used().tolist()[0]})")

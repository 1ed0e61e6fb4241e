#!/usr/bin/env python3
"""The step kernels' F1 launch alone at a fixed theta = I + eps * pattern, 8 x 256^3 (for PMC passes): python tools/f1_at_eps.py eps [n]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchregister_amd as tr
import bench
eps = float(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mov, tgt = bench.make_batch(0, torch.device("cuda"), 256, 8)
k = torch.arange(12, dtype=torch.float64).reshape(3, 4)
th = (torch.eye(3, 4, dtype=torch.float64) + eps * torch.sin(1.2345 * (k + 1.0))).float()[None].repeat(8, 1, 1)
s = tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=0.0, init=th, capacity=4)
for _ in range(n): s.accumulate_only()
torch.cuda.synchronize()

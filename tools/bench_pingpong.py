#!/usr/bin/env python3
"""Headline workload (8 x 256^3 affine + NCC, Adam lr 1e-4) with and without the alternating walk direction (TRX_FLAG_NO_PINGPONG), solvers
timed alternately on one box: ms per step of 200-step runs."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
from torchregister_amd import _lib
import bench
dev = torch.device("cuda")
mov, tgt = bench.make_batch(0, dev, 256, 8)
def solver(flags): return tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=1e-4, capacity=1000, flags=flags)
sv = {"pingpong": solver(0), "one-way": solver(_lib.FLAG_NO_PINGPONG)}
for s in sv.values(): s.run(120)
torch.cuda.synchronize()
for rnd in range(4):
    for name, s in sv.items():
        torch.cuda.synchronize(); t0 = time.perf_counter(); s.run(200); torch.cuda.synchronize()
        print(f"{name:9s} {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per step", flush=True)

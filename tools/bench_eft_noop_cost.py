#!/usr/bin/env python3
"""What the headline pays for the exact-footprint kernel's empty launch: bench.py's batch (8 x 256^3 affine + NCC, Adam, from the identity), 200 steps
after 120, with the kernel offered (default) and never offered (TRX_FLAG_NO_EFT), alternating."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
from torchregister_amd import _lib
from bench import make_batch
dev = torch.device("cuda")
mov, tgt = make_batch(0, dev)
def run(flags):
    s = tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=1e-4, capacity=400, flags=flags)
    s.run(120); torch.cuda.synchronize()
    t0 = time.perf_counter(); s.run(200); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 200 * 1e6
run(0)
for r in range(4):
    a, b = run(0), run(_lib.FLAG_NO_EFT)
    print(f"round {r}: offered {a:7.2f} us per step, never offered {b:7.2f} us per step  ({100 * (a / b - 1):+.2f} %)")

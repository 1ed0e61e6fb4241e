#!/usr/bin/env python3
"""Round 6 probe: does a PAIR-GROUP-MAJOR schedule keep a group's volumes in the 256 MiB Infinity Cache?
The headline batch (8 x 256^3 affine + NCC, Adam lr 1e-4) is run as 8 / g independent solvers of g pairs each; a sweep runs every solver for K
iterations before moving to the next one (g = 8: the product's pair-minor order).  Prints us per pair-iteration."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
import bench
dev = torch.device("cuda")
mov, tgt = bench.make_batch(0, dev, 256, 8)
def solvers(g):
    return [tr.AffineSolver(mov[i:i + g], tgt[i:i + g], mode="affine", loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=1e-4, capacity=4000) for i in range(0, 8, g)]
warm = solvers(8)[0]
warm.run(150)
torch.cuda.synchronize()
TOTAL = 96   # iterations per pair and measurement
for rnd in range(2):
    for g in (8, 4, 2, 1):
        for K in (1, 4, 16, 48):
            sv = solvers(g)
            for s in sv: s.run(8)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(TOTAL // K):
                for s in sv: s.run(K)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            print(f"round {rnd} group {g} pairs x K {K:2d}: {el / (8 * TOTAL) * 1e6:7.2f} us per pair-iteration  bodies {sv[0].bodies()[:2]}", flush=True)
            del sv

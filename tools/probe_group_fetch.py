#!/usr/bin/env python3
"""Round 6: HBM-side traffic and time of the headline batch run pair-group-major (groups of g pairs x K iterations) - run under rocprofv3 --pmc FETCH_SIZE.
    python tools/probe_group_fetch.py g K [iterations per pair]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
import bench
g, K = int(sys.argv[1]), int(sys.argv[2])
TOTAL = int(sys.argv[3]) if len(sys.argv) > 3 else 96
dev = torch.device("cuda")
mov, tgt = bench.make_batch(0, dev, 256, 8)
sv = [tr.AffineSolver(mov[i:i + g], tgt[i:i + g], mode="affine", loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=1e-4, capacity=TOTAL + 16) for i in range(0, 8, g)]
for s in sv: s.run(8)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(TOTAL // K):
    for s in sv: s.run(K)
torch.cuda.synchronize()
print(f"group {g} x K {K}: {(time.perf_counter() - t0) / (8 * TOTAL) * 1e6:.2f} us per pair-iteration", flush=True)

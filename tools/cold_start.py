#!/usr/bin/env python3
"""where does a registration that starts on an idle GPU recover the settled rate, and does a pre-roll of the F1 kernel at
solver construction help?  8 x 256^3 affine + NCC (Adam), 2 s of idle GPU, then 300 iterations timed in groups of 10 (events);
variants: no pre-roll / N untimed F1 launches (accumulate_only) in front of the first iteration."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
import bench
dev = torch.device("cuda")
mov, tgt = bench.make_batch(0, dev)
def run(preroll, iters=300, grp=10):
    s = tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=1e-4, capacity=iters + 8)
    torch.cuda.synchronize(); time.sleep(2.0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters // grp + 1)]
    t0 = time.perf_counter()
    for _ in range(preroll): s.accumulate_only()
    ev[0].record()
    for g in range(iters // grp):
        s.run(grp); ev[g + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = [ev[g].elapsed_time(ev[g + 1]) / grp for g in range(iters // grp)]
    return ms, wall
for pre in (0, 8, 32, 100):
    ms, wall = run(pre)
    first20 = 8 * 20 / (sum(ms[:2]) * 10e-3); first100 = 8 * 100 / (sum(ms[:10]) * 10e-3); first200 = 8 * 200 / (sum(ms[:20]) * 10e-3)
    settled = sum(ms[-5:]) / 5
    rec = next((g * 10 for g in range(len(ms)) if all(m < settled * 1.03 for m in ms[g:g + 3])), None)
    print(f"pre-roll {pre:3d} launches: ms per step in groups of 10: " + " ".join(f"{m:.3f}" for m in ms[:14]) + f" ... settled {settled:.3f}")
    print(f"      pair-it/s over the first 20 / 100 / 200 iterations: {first20:.0f} / {first100:.0f} / {first200:.0f}; within 3 % of the settled rate from iteration {rec}; wall incl. pre-roll {wall * 1e3:.1f} ms")

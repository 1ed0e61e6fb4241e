#!/usr/bin/env python3
"""Round 6: the headline workload (8 x 256^3 affine + NCC, Adam lr 1e-4) and the converging run (lr 2e-3, 300 iterations) with the one-kernel form of a
step (TRX_FLAG_ONE_KERNEL, AffineSolver's "auto" policy) against the three-kernel form, solvers timed alternately on one box."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
import bench
dev = torch.device("cuda")
mov, tgt = bench.make_batch(0, dev, 256, 8)
def solver(ok, lr=1e-4, cap=1200): return tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=lr, capacity=cap, one_kernel=ok)
sv = {"one-kernel": solver("auto"), "three-kernel": solver(False)}
for s in sv.values(): s.run(120)
torch.cuda.synchronize()
for rnd in range(4):
    for name, s in sv.items():
        torch.cuda.synchronize(); t0 = time.perf_counter(); s.run(200); torch.cuda.synchronize()
        print(f"headline {name:12s} {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per step  (flag {s.one_kernel}, bodies {sorted(set(s.bodies()))})", flush=True)
for rnd in range(3):
    for name, ok in (("one-kernel", "auto"), ("three-kernel", False)):
        s = solver(ok, lr=2e-3, cap=300)
        torch.cuda.synchronize(); t0 = time.perf_counter(); s.run(300); torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"converging run {name:12s} {8 * 300 / el:8.0f} pair-it/s  loss ratio worst {(s.losses[:, 299] / s.losses[:, 0]).max().item():.4f}  bodies at the end {sorted(set(s.bodies()))}", flush=True)

# smaller launches that still fill the chip (the z-streaming kernel in front): where two empty launches would weigh most
for shape, nb in (((64, 128, 128), 16), ((256, 256, 256), 2), ((128, 128, 128), 8)):
    t = torch.cat([bench.blobs_gpu(shape, 1000 + i, dev) for i in range(nb)])
    m = tr.get_affine_warp(torch.tensor(bench.THETA_STAR, device=dev)[None].expand(nb, 3, 4).contiguous(), t)
    sv = {k: tr.AffineSolver(m, t, mode="affine", loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=1e-5, capacity=5000, one_kernel=ok) for k, ok in (("one-kernel", "auto"), ("three-kernel", False))}
    for s in sv.values(): s.run(200)
    torch.cuda.synchronize()
    for rnd in range(3):
        for name, s in sv.items():
            torch.cuda.synchronize(); t0 = time.perf_counter(); s.run(1000); torch.cuda.synchronize()
            print(f"{nb} x {shape} {name:12s} {(time.perf_counter() - t0) / 1000 * 1e6:.2f} us per step  (flag {s.one_kernel}, bodies {sorted(set(s.bodies()))})", flush=True)

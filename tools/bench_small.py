#!/usr/bin/env python3
"""Launch-bound sizes: one pair of S^3 (affine + NCC, SGD), us per iteration of run(400) - with the finalise folded into the next iteration's kernel (default, round 6)
and as two launches per iteration (TRX_FLAG_NO_CARRY) - and of the F1 launch alone."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
from bench import blobs_gpu, THETA_STAR
dev = torch.device("cuda")
for S in (64, 96, 128, 160, 192):
    tgt = blobs_gpu((S,) * 3, 1000, dev); mov = tr.get_affine_warp(torch.tensor(THETA_STAR, device=dev)[None], tgt)
    for flags, name in ((0, "default"), (65536, "no_carry")):   # (65536 = TRX_FLAG_NO_CARRY: a step kernel + a finalise kernel per iteration, the form of rounds 1-5)
        s = tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), lr=1e-6, capacity=1000, flags=flags)
        s.run(100); torch.cuda.synchronize()
        t0 = time.perf_counter(); s.run(400); torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 400
        for _ in range(50): s.accumulate_only()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): s.accumulate_only()
        e1.record(); torch.cuda.synchronize()
        print(f"1 x {S}^3 {name:8s}: {t * 1e6:6.1f} us per iteration, F1 launch alone {e0.elapsed_time(e1) * 5:6.1f} us, rows {s.rows_used().tolist()[0]}")

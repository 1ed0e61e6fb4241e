#!/usr/bin/env python3
"""Launch-bound sizes: the tile kernels against the un-tiled gather kernel (TRX_FLAG_GATHER_PATH), one pair of S^3, us per iteration of run(400)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
from bench import blobs_gpu, THETA_STAR
dev = torch.device("cuda")
for S in (32, 48, 64, 96, 128):
    tgt = blobs_gpu((S,) * 3, 1000, dev); mov = tr.get_affine_warp(torch.tensor(THETA_STAR, device=dev)[None], tgt)
    for flags, name in ((0, "tiles+carry"), (65536, "tiles"), (1, "gather")):
        s = tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), lr=1e-6, capacity=1000, flags=flags)
        s.run(100); torch.cuda.synchronize()
        t0 = time.perf_counter(); s.run(400); torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 400
        print(f"1 x {S}^3 {name:12s}: {t * 1e6:6.1f} us per iteration", flush=True)

#!/usr/bin/env python3
"""Host enqueue cost of one affine step (no sync inside the timed region) against its GPU time, 1 x 64^3."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
from bench import blobs_gpu, THETA_STAR
dev = torch.device("cuda")
S = 64
tgt = blobs_gpu((S,) * 3, 1000, dev); mov = tr.get_affine_warp(torch.tensor(THETA_STAR, device=dev)[None], tgt)
s = tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), lr=1e-6, capacity=5000)
s.run(200); torch.cuda.synchronize()
t0 = time.perf_counter(); s.run(2000); t_enq = time.perf_counter() - t0; torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"1 x {S}^3: enqueue {t_enq / 2000 * 1e6:.1f} us per iteration (host), complete {t_all / 2000 * 1e6:.1f} us per iteration")

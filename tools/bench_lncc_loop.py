#!/usr/bin/env python3
"""256^3 direct flow + local-window NCC + smoothness, Adam: the device-side loop (trx_flow_lncc_run) against the round-2 composition
(HIP warp -> LocalNCCLoss autograd.Function -> torch.optim.Adam).  us per iteration."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
from torchregister_amd.warpings import smooth_regulariser
from bench import blobs_gpu
dev = torch.device("cuda")
shape = (256,) * 3
tgt = blobs_gpu(shape, 1000, dev); mov = blobs_gpu(shape, 1001, dev)
for w in (5, 9):
    s = tr.FlowSolver(mov, tgt, optimizer="adam", lr=0.01, capacity=200, smooth_weight=1.0, lncc=dict(window=w))
    s.run(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); s.run(50); torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 50
    # round-2 composition
    fl = torch.zeros(1, 3, *shape, device=dev, requires_grad=True)
    opt = torch.optim.Adam([fl], 0.01)
    st = tr.SpatialTransformer(shape); crit = tr.LocalNCCLoss(window=w)
    def it():
        opt.zero_grad()
        e = crit(tgt, st(mov, fl)) + smooth_regulariser(fl, 1.0)
        e.backward(); opt.step()
    for _ in range(5): it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): it()
    torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) / 20
    print(f"256^3 flow + LNCC(w={w}) + smooth, Adam: device loop {t * 1e6:.0f} us / iteration, autograd composition {t2 * 1e6:.0f} us / iteration")

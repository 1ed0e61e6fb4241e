#!/usr/bin/env python3
"""BASELINE.json configs that fit one GPU, timed end to end through the public API (development numbers for DESIGN.md).
cfg1: 2-D 256x256 rigid+MSE 500 it; cfg2: 3-D 128^3 affine+NCC 200 it; cfg3: 3-D 256^3 direct flow + NCC + smoothness 100 it."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch.nn as nn
import TorchRegister as tr
import phantoms as ph
from bench import blobs_gpu, THETA_STAR

dev = "cuda"
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best

# cfg1
tgt2 = ph.blobs((256, 256), 1000).to(dev)
mov2 = tr.get_affine_warp(torch.tensor(ph.THETA_STAR2, device=dev)[None], tgt2)
init2 = torch.tensor([0.1, 0.02, -0.03])
def cfg1():
    r = tr.Register("rigid", device=dev, criterion=[nn.MSELoss()], weight=[1.0], init=init2); r.optim(mov2, tgt2, lr=1e-2, max_epochs=500); return r
t = timed(cfg1); print(f"cfg1 2D 256^2 rigid+MSE 500 it: {t*1e3:.1f} ms -> {500/t:.0f} it/s")
# cfg2
tgt3 = blobs_gpu((128,)*3, 1000, dev); mov3 = tr.get_affine_warp(torch.tensor(THETA_STAR, device=dev)[None], tgt3)
def cfg2():
    r = tr.Register("affine", device=dev, criterion=[tr.NCCLoss()], weight=[1.0], honor_criterion=True); r.optim(mov3, tgt3, lr=1e-6, max_epochs=200); return r
t = timed(cfg2); print(f"cfg2 3D 128^3 affine+NCC 200 it: {t*1e3:.1f} ms -> {200/t:.0f} it/s")
# cfg3
tgt4 = blobs_gpu((256,)*3, 1000, dev); mov4 = tr.get_affine_warp(torch.tensor(THETA_STAR, device=dev)[None], tgt4)
for sw, opt in ((0.0, "sgd"), (1.0, "adam")):
    def cfg3():
        r = tr.Register("flow", device=dev, criterion=[tr.NCCLoss()], weight=[1.0], flow_model="direct", optimizer=opt, smooth_weight=sw)
        r.optim(mov4, tgt4, lr=1.0 if opt == "sgd" else 0.01, max_epochs=100); return r
    t = timed(cfg3); print(f"cfg3 3D 256^3 direct flow+NCC {opt} smooth={sw} 100 it: {t*1e3:.1f} ms -> {100/t:.0f} it/s")

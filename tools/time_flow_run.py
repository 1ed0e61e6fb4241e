#!/usr/bin/env python3
"""256^3 dense flow: microseconds per iteration inside one run(100) call (fused steps).  python tools/time_flow_run.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
from bench import blobs_gpu
dev = torch.device("cuda")
shape = (256,) * 3
tgt = blobs_gpu(shape, 1000, dev); mov = blobs_gpu(shape, 1001, dev)
for opt, lr, sm in (("sgd", 1.0, 0.0), ("adam", 0.01, 0.0), ("adam", 0.01, 1.0)):
    s = tr.FlowSolver(mov, tgt, loss=tr.LossSpec(w_ncc=1.0), optimizer=opt, lr=lr, capacity=300, smooth_weight=sm)
    s.run(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); s.run(100); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"256^3 flow NCC+{opt} smooth={sm}: {t * 1e4:.1f} us per iteration")

#!/usr/bin/env python3
"""Rigid-mode steps (finalise kernel with the Theta chain rule) for rocprofv3 --kernel-trace --stats: python tools/prof_rigid.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
from bench import blobs_gpu
dev = torch.device("cuda")
for shape, B, it in (((128, 128, 128), 1, 200), ((256, 256), 1, 300)):
    tgt = torch.cat([blobs_gpu(shape, 1000 + b, dev) if len(shape) == 3 else blobs_gpu(shape + (1,), 1000 + b, dev).reshape(1, 1, *shape) for b in range(B)])
    mov = torch.roll(tgt, 2, dims=-1).contiguous()
    npose = 6 if len(shape) == 3 else 3
    s = tr.AffineSolver(mov, tgt, mode="rigid", loss=tr.LossSpec(w_mse=1.0), lr=1e-6, init=0.02 * torch.rand(B, npose), capacity=it)
    s.run(it); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s2 = tr.AffineSolver(mov, tgt, mode="rigid", loss=tr.LossSpec(w_mse=1.0), lr=1e-6, init=0.02 * torch.rand(B, npose), capacity=it)
    e0.record(); s2.run(it); e1.record(); torch.cuda.synchronize()
    print(shape, "rigid:", e0.elapsed_time(e1) / it * 1e3, "us per iteration")

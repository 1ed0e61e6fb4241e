#!/usr/bin/env python3
"""Local-window NCC kernels alone (loss + gradient wrt the warped volume): us per call at 256^3 for 1 and 8 pairs, windows 5 and 9; and the
share of the 8 TB/s roofline at the kernels' algorithmic bytes (44 B per voxel: DESIGN.md 4.5)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchregister_amd import _engine as eng
from bench import blobs_gpu

dev = torch.device("cuda")
shape = (256,) * 3
for B in (1, 8):
    tgt = torch.cat([blobs_gpu(shape, 1000 + i, dev) for i in range(B)])
    wrp = torch.cat([blobs_gpu(shape, 2000 + i, dev) for i in range(B)])
    for w in (5, 9):
        for grad in (True, False):
            for _ in range(3):
                eng.local_ncc_loss_grad(tgt, wrp, w, need_grad=grad)
            torch.cuda.synchronize()
            n = 20
            t0 = time.perf_counter()
            for _ in range(n):
                eng.local_ncc_loss_grad(tgt, wrp, w, need_grad=grad)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / n * 1e6
            byts = (44 if grad else 20) * B * 256 ** 3
            print(f"{B} x 256^3 window {w} {'loss + gradient' if grad else 'loss only      '}: {us:8.1f} us per call, {us / B:7.1f} per pair, {byts / us / 1e6:5.2f} TB/s algorithmic = {byts / us / 8e6:.3f} of 8 TB/s")

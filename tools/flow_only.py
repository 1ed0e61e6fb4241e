#!/usr/bin/env python3
"""The flow loop alone, for profilers: one 256^3 pair (or a 64-plane 512^2 slab), direct flow + NCC, N iterations in one trx_flow_run call.
    python tools/flow_only.py [adam|sgd] [smooth_weight] [iters] [D H W]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
from bench import blobs_gpu
opt = sys.argv[1] if len(sys.argv) > 1 else "adam"
sm = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 60
shape = tuple(int(v) for v in sys.argv[4:7]) if len(sys.argv) > 6 else (256, 256, 256)
dev = torch.device("cuda")
tgt = blobs_gpu(shape, 1000, dev); mov = blobs_gpu(shape, 1001, dev)
s = tr.FlowSolver(mov, tgt, loss=tr.LossSpec(w_ncc=1.0), optimizer=opt, lr=1.0 if opt == "sgd" else 0.01, capacity=iters + 40, smooth_weight=sm)
s.run(20); torch.cuda.synchronize()
t0 = time.perf_counter(); s.run(iters); torch.cuda.synchronize(); t = time.perf_counter() - t0
nv = shape[0] * shape[1] * shape[2]
alg = (80 if opt == "adam" else 32) + (20 if sm else 0)   # fused step: flow r/w 24, target 4, moving 4 (+ adam m, v r/w 48); with smoothness the moments pass stays fused, the double buffer does not add bytes
print(f"{shape} flow NCC+{opt} smooth={sm}: {t / iters * 1e6:.1f} us per iteration")

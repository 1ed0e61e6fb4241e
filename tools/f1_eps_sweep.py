#!/usr/bin/env python3
"""F1 launch time of the step kernels against the distance of theta from the identity (8 x 256^3): theta = I + eps * pattern, and theta = I + t * (theta* - I)
(the path a headline run takes); us per launch, partial rows per pair (64: z-streaming body, 128 / 256: tile geometries)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
import bench
dev = torch.device("cuda")
mov, tgt = bench.make_batch(0, dev, 256, 8)
k = torch.arange(12, dtype=torch.float64).reshape(3, 4)
pat = torch.sin(1.2345 * (k + 1.0))
star = torch.tensor(bench.THETA_STAR, dtype=torch.float64) - torch.eye(3, 4, dtype=torch.float64)
def t_of(th):
    s = tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=0.0, init=th.float()[None].repeat(8, 1, 1), capacity=4)
    for _ in range(30): s.accumulate_only()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): s.accumulate_only()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 20, s.rows_used().tolist()[0]
for _ in range(2): t_of(torch.eye(3, 4, dtype=torch.float64))
for eps in (0.0, 0.002, 0.005, 0.008, 0.012, 0.016, 0.02, 0.03, 0.05):
    us, rows = t_of(torch.eye(3, 4, dtype=torch.float64) + eps * pat)
    print(f"I + {eps:5.3f} * pattern : {us:7.1f} us per launch, rows {rows}")
for t in (0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.7, 1.0):
    us, rows = t_of(torch.eye(3, 4, dtype=torch.float64) + t * star)
    print(f"I + {t:4.2f} * (theta* - I): {us:7.1f} us per launch, rows {rows}")

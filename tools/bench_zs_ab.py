"""A/B of the step kernels' bodies on the headline batch: the F1 launch alone (events), flags 0 / NO_ZSTREAM / ZSTREAM, at several thetas."""
import sys, os, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchregister_amd as tr
from torchregister_amd import _lib
import bench

def time_f1(s, reps=200):
    for _ in range(150): s.accumulate_only()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): s.accumulate_only()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
mov, tgt = bench.make_batch(0, torch.device("cuda"), size, B)
for eps in (0.0, 0.004, 0.01, 0.02, 0.03):
    k = torch.arange(12, dtype=torch.float64).reshape(3, 4)
    th = (torch.eye(3, 4, dtype=torch.float64) + eps * torch.sin(1.2345 * (k + 1.0))).float()[None].repeat(B, 1, 1)
    row = []
    for name, fl in (("default", 0), ("no_zs", _lib.FLAG_NO_ZSTREAM), ("zs", _lib.FLAG_ZSTREAM)):
        s = tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=0.0, init=th, capacity=4, flags=fl)
        s.run(1); torch.cuda.synchronize()
        ru = s.rows_used().tolist()
        us = time_f1(s)
        row.append(f"{name}: {us:7.1f} us rows {ru[0]}")
    print(f"B={B} {size}^3 eps={eps}: " + " | ".join(row), flush=True)

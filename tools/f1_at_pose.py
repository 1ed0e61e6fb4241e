#!/usr/bin/env python3
"""The step kernels' F1 launch alone at a fixed ROTATED pose, 8 x 256^3 (for rocprofv3 passes):
    python tools/f1_at_pose.py rot|rigid [n] [flags]
rot   = affine mode at theta = R(0.5, 0.4, 0.3) diag(1.05, 0.95, 1.02) (bench.py's value_rot pose)
rigid = rigid mode at the reference's torch.manual_seed(0); torch.rand(6) pose (bench.py's value_rigid_randinit pose)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchregister_amd as tr
import bench
which = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda")
mov, tgt = bench.make_batch(0, dev, 256, 8)
if which == "rot":
    s = tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=0.0, init=bench.pose_rot(dev), capacity=4, flags=flags)
elif which == "rigid":
    s = tr.AffineSolver(mov, tgt, mode="rigid", loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=0.0, init=bench.pose_rigid_randinit(dev), capacity=4, flags=flags)
else:
    raise SystemExit("rot | rigid")
for _ in range(n): s.accumulate_only()
torch.cuda.synchronize()

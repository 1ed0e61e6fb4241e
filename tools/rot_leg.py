#!/usr/bin/env python3
"""development: 300 affine + NCC steps of 8 x 256^3 at the bench's rotated pose, for rocprofv3 --kernel-trace --stats (which kernels a rotated step is made of).
   cd /tmp && rocprofv3 --kernel-trace --stats -d <dir> -o p -- python3 $GRAFT_REPO_ROOT/tools/rot_leg.py"""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
from bench import blobs_gpu


def rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


dev = torch.device("cuda")
S, B = 256, 8
tgt = torch.cat([blobs_gpu((S,) * 3, 1000 + b, dev) for b in range(B)])
mov = torch.cat([blobs_gpu((S,) * 3, 2000 + b, dev) for b in range(B)])
R = rot(0.5, 0.4, 0.3) @ torch.diag(torch.tensor([1.05, 0.95, 1.02]))
th = torch.cat([R.float(), torch.tensor([[0.01], [-0.02], [0.015]])], dim=1)[None].expand(B, 3, 4).contiguous()
s = tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), lr=0.0, init=th, capacity=400)
s.run(300)
torch.cuda.synchronize()
print("bodies", s.bodies())

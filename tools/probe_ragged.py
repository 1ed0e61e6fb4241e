#!/usr/bin/env python3
"""Round 6 probe: what do real-world (ragged) shapes get next to the shapes the z-streaming kernel tiles (W % 64 == 0, H % 32 == 0)?  8 pairs, affine + NCC, Adam lr 1e-4."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
import bench
dev = torch.device("cuda")
for shape in ((182, 218, 182), (192, 224, 192), (160, 192, 224), (160, 192, 256), (256, 256, 256)):
    tgt = torch.cat([bench.blobs_gpu(shape, 1000 + i, dev) for i in range(8)])
    mov = tr.get_affine_warp(torch.tensor(bench.THETA_STAR, device=dev)[None].expand(8, 3, 4).contiguous(), tgt)
    s = tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=1e-4, capacity=700)
    s.run(150); torch.cuda.synchronize()
    t0 = time.perf_counter(); s.run(300); torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 300
    nv = 8 * shape[0] * shape[1] * shape[2]
    print(f"8 x {shape}: {t * 1e6:7.1f} us per step  {t * 1e12 / nv:6.2f} ps per voxel  {8 * nv / t / 1e12:5.2f} TB/s algorithmic = {8 * nv / t / 8e12:.3f}  bodies {sorted(set(s.bodies()))}", flush=True)

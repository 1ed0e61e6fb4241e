#!/usr/bin/env python3
"""Secondary timings (development): flow step (SGD/Adam/smooth), forward warps, loss-only; 256^3."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchregister_amd as tr
from torchregister_amd import _engine as eng
from bench import blobs_gpu, THETA_STAR

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
shape = (S,) * 3
N = S ** 3
tgt = blobs_gpu(shape, 1000, dev)
mov = tr.get_affine_warp(torch.tensor(THETA_STAR, device=dev)[None], tgt)
def rep(name, us, nbytes):
    print(f"{name:38s} {us:10.1f} us   {nbytes / us / 1e6:6.2f} TB/s algorithmic")
for opt, bpv in (("sgd", 52), ("adam", 100)):
    fs = tr.FlowSolver(mov, tgt, loss=tr.LossSpec(w_ncc=1.0), optimizer=opt, lr=1.0 if opt == "sgd" else 0.01, capacity=64)
    rep(f"flow step NCC+{opt} (1 pair)", timeit(lambda: fs.run(1)), bpv * N)
fs = tr.FlowSolver(mov, tgt, loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=0.01, capacity=64, smooth_weight=1.0)
rep("flow step NCC+adam+smooth (1 pair)", timeit(lambda: fs.run(2)) / 2, 124 * N)
fl = fs.flow
rep("flow_warp 1ch", timeit(lambda: eng.flow_warp(mov, fl)), 20 * N)
th = torch.tensor(THETA_STAR, device=dev)[None]
rep("affine_warp 1ch (1 pair)", timeit(lambda: eng.affine_warp(th, mov)), 8 * N)
mov8 = mov.expand(8, 1, *shape).contiguous()
rep("affine_warp 1ch (8 pairs)", timeit(lambda: eng.affine_warp(th.expand(8, 3, 4).contiguous(), mov8)), 8 * N * 8)
s = tr.AffineSolver(mov, tgt, loss=tr.LossSpec(w_ncc=1.0), lr=1e-6, capacity=64)
rep("affine step (1 pair)", timeit(lambda: s.run(1)), 8 * N)
rep("affine loss-only (1 pair)", timeit(lambda: s.eval_loss()), 8 * N)
# local-window NCC (extension): loss + gradient wrt the warped volume, 52 B/voxel algorithmic
wrp = eng.affine_warp(torch.eye(3, 4, device=dev)[None], mov)
for win in (9, 5):
    rep(f"local NCC w={win} loss+grad (1 pair)", timeit(lambda: eng.local_ncc_loss_grad(tgt, wrp, win)), 52 * N)
rep("local NCC w=9 loss only (1 pair)", timeit(lambda: eng.local_ncc_loss_grad(tgt, wrp, 9, need_grad=False)), 24 * N)
tgt8, wrp8 = tgt.expand(8, 1, *shape).contiguous(), wrp.expand(8, 1, *shape).contiguous()
rep("local NCC w=9 loss+grad (8 pairs)", timeit(lambda: eng.local_ncc_loss_grad(tgt8, wrp8, 9), 5), 52 * N * 8)
# generic-path pieces: warp backward wrt theta from an arbitrary grad_out (row-walking gather kernel)
go = torch.rand_like(mov)
rep("affine_warp_backward (1 pair)", timeit(lambda: eng.affine_warp_backward(th, mov, go)), 12 * N)
go8 = torch.rand_like(mov8)
rep("affine_warp_backward (8 pairs)", timeit(lambda: eng.affine_warp_backward(th.expand(8, 3, 4).contiguous(), mov8, go8), 5), 12 * N * 8)
fl1 = torch.zeros(1, 3, *shape, device=dev) + 0.3
rep("flow_warp_backward (1 pair)", timeit(lambda: eng.flow_warp_backward(mov, fl1, go)), 36 * N)
# rigid steps from the reference's kind of initial pose (every angle uniform in [0,1) rad, ref:utils.py:316-330): 8 pairs, 8 different poses
g = torch.Generator().manual_seed(7)
tgt8r = torch.cat([blobs_gpu(shape, 1000 + b, dev) for b in range(8)])
poses = torch.rand(8, 6, generator=g)
for name, init in (("random poses U[0,1)^6", poses), ("small poses (0.02 rad)", 0.02 * poses)):
    sr = tr.AffineSolver(mov8, tgt8r, mode="rigid", loss=tr.LossSpec(w_mse=1.0), lr=1e-6, init=init, capacity=64)
    rep(f"rigid step, {name} (8 pairs)", timeit(lambda: sr.run(1), 20), 8 * N * 8)
# PCIe-inclusive rate of the headline workload: 8 pairs handed over as pinned host buffers, 200 affine+NCC iterations, theta read back
import time
hm, ht = mov8.cpu().pin_memory(), tgt8r.cpu().pin_memory()
torch.cuda.synchronize()
for it in (200, 1000):
    t0 = time.perf_counter()
    dm, dt = hm.to(dev, non_blocking=True), ht.to(dev, non_blocking=True)
    sp = tr.AffineSolver(dm, dt, loss=tr.LossSpec(w_ncc=1.0), lr=1e-4, optimizer="adam", capacity=it)
    sp.run(it)
    th_host = sp.best_theta.cpu()
    dt_s = time.perf_counter() - t0
    print(f"PCIe-inclusive, 8 pairs x {it} iterations: {dt_s * 1e3:8.1f} ms  -> {8 * it / dt_s:9.0f} pair-iterations/s (1 GiB host->device included)")

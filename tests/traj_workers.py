"""Arbiter runs of tests/test_gpu_baseline_trajectories.py as picklable jobs: oracle/compose.py loops (ATen CPU ops at the reference's call
sites) in one precision each, executed side by side in worker processes - ATen's CPU grid_sample threads poorly over one volume, the GPU
box has the cores to run all of them at once."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]


def theta0_np(eps=2.5e-3, seed=0):
    import numpy as np
    k = np.arange(12, dtype=np.float64).reshape(3, 4)
    return (np.eye(3, 4) + eps * np.sin(1.2345 * (k + 1.0) + 0.77 * seed)).astype(np.float32)


def rigid_pose0(i):
    import torch
    torch.manual_seed(0)
    return (torch.rand(6) + 1e-3 * torch.sin(0.7 * torch.arange(6) + i)).float()


def smooth_flow0(shape):
    import torch
    ax = [torch.arange(n, dtype=torch.float64) for n in shape]
    comp = lambda a, b, c: (torch.sin(a * ax[0])[:, None, None] + torch.cos(b * ax[1])[None, :, None] + torch.sin(c * ax[2] + 0.4)[None, None, :])  # noqa: E731
    return torch.stack([0.7 * comp(0.021, 0.017, 0.013), 0.5 * comp(0.011, 0.023, 0.019), 0.6 * comp(0.015, 0.012, 0.027)]).float()[None] + 0.37


def pair(shape, seed):
    import torch
    import phantoms as ph
    from oracle import compose
    tgt = ph.blobs(shape, seed)
    return compose.affine_warp(torch.tensor(ph.THETA_STAR3)[None], tgt), tgt


def run(job):
    """job = (kind, shape, seed, dtype name, optimizer, lr, iters, extra) -> dict of numpy arrays"""
    import torch
    from oracle import compose
    torch.set_num_threads(8)   # (16 jobs side by side under the rest of the GPU suite: leave the host cores that suite's own oracle calls need)
    kind, shape, seed, dtn, optimizer, lr, iters, extra = job
    dt = getattr(torch, dtn)
    mov, tgt = pair(shape, seed)
    if kind == "rigid":   # extra = the pair's index: the reference's initial pose (torch.manual_seed(0); torch.rand(6)) + a pair-dependent nudge
        r = compose.affine_loop(mov.to(dt), tgt.to(dt), lr, iters, optimizer=optimizer, pose0=rigid_pose0(extra), w_ncc=1.0)
        return dict(losses=r["losses"].numpy(), thetas=r["thetas"].double().numpy(), best_idx=r["best_idx"])
    if kind == "affine":
        r = compose.affine_loop(mov.to(dt), tgt.to(dt), lr, iters, optimizer=optimizer, theta0=torch.from_numpy(theta0_np(seed=extra)), w_ncc=1.0)
        return dict(losses=r["losses"].numpy(), thetas=r["thetas"].double().numpy(), best_idx=r["best_idx"])
    r = compose.flow_loop(mov.to(dt), tgt.to(dt), lr, iters, optimizer=optimizer, flow0=smooth_flow0(shape), smooth_weight=extra, w_ncc=1.0)
    return dict(losses=r["losses"].numpy(), flow=r["flow"].double().numpy())

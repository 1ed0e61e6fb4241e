#!/usr/bin/env python3
"""Randomised parity sweep of the fused F1 step (and the forward warp) against the C oracle in fp64.
   python tests/fuzz_affine.py [cases] [seed]
Random shapes (tiny to ~100^3, ragged, W % 4 != 0), random theta (near identity ... large rotations / zoom / flips /
mostly-out-of-bounds), random loss weights and batch sizes.  Tolerances as in tests/test_gpu_tile_paths.py."""
import math, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # test infrastructure: the oracle may only be used from tests/
import oracle
import phantoms as ph
import torchregister_amd._engine as eng


def rand_theta(rng, kind):
    a = {"tiny": 0.02, "small": 0.12, "medium": 0.4, "large": 1.2}[kind]
    ax, ay, az = rng.uniform(-a, a, 3)
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    sc = np.diag(rng.uniform(0.8, 1.25, 3) if kind != "tiny" else rng.uniform(0.97, 1.03, 3))
    if kind == "large" and rng.random() < 0.3:
        sc = sc @ np.diag(rng.choice([-1.0, 1.0], 3))
    m = rz @ ry @ rx @ sc
    t = rng.uniform(-0.5, 0.5, 3) if kind in ("medium", "large") else rng.uniform(-0.06, 0.06, 3)
    th = np.concatenate([m, t[:, None]], axis=1)
    return th + 1e-3 * rng.standard_normal(th.shape)      # generic: keeps samples off exact voxel positions


def smooth(shape, f):
    ax = [torch.arange(n, dtype=torch.float64) for n in shape]
    v = torch.sin(f * ax[0])[:, None, None] * torch.cos(0.7 * f * ax[1])[None, :, None] + torch.sin(1.3 * f * ax[2] + 0.5)[None, None, :]
    return v.float().view(1, 1, *shape)


def run(n, seed, grad_bar=2e-4, verbose=True):
    rng = np.random.default_rng(seed)
    worst = {"loss": 0.0, "grad": 0.0, "warp": 0.0}
    fails = 0
    for it in range(n):
        big = rng.random() < 0.3
        shape = tuple(int(v) for v in (rng.integers(3, 100, 3) if big else rng.integers(3, 48, 3)))
        B = int(rng.integers(1, 4))
        kind = rng.choice(["tiny", "small", "medium", "large"], p=[0.3, 0.35, 0.2, 0.15])
        kw = dict(w_ncc=float(rng.uniform(0, 1)), w_mse=float(rng.uniform(0, 1)))
        # smooth phantoms only: a component that oscillates from plane to plane (e.g. cos of the flat index) turns the one
        # sample in ~10^5 that lands within an ulp of an integer coordinate into a large one-sided-derivative difference
        # between fp32 and fp64 (value continuous, derivative not) - an fp32 property of trilinear sampling, not a defect
        tgt = torch.cat([ph.blobs(shape, 500 + 7 * it + b) + 0.05 * smooth(shape, 0.31 + 0.01 * b) for b in range(B)])
        mov = torch.cat([ph.blobs(shape, 900 + 5 * it + b) + 0.1 * smooth(shape, 0.23) for b in range(B)])
        ths = np.stack([rand_theta(rng, kind) for _ in range(B)])
        th = torch.tensor(ths, dtype=torch.float32)
        s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1)
        s.run(1)
        wrp_t = eng.affine_warp(th.cuda(), mov.cuda())
        # generic warp backward (tile kernel MODE 2) with grad_out = dMSE/dwarped must reproduce the MSE gradient
        go = 2.0 * (wrp_t - tgt.cuda()) / float(np.prod(shape))
        dth_b = eng.affine_warp_backward(th.cuda(), mov.cuda(), go).cpu().numpy()
        wrp = wrp_t.cpu().numpy()
        torch.cuda.synchronize()
        tabs64, tabs32 = oracle.base_tables(shape, np.float64), oracle.base_tables(shape, np.float32)
        for b in range(B):
            tu = th[b].double().numpy()
            total, _, dth, _ = oracle.c_affine_loss_grad(mov[b, 0].double().numpy(), tgt[b, 0].double().numpy(), tu, oracle.wts(**kw), tabs64)
            loss = s.losses[b, 0].item(); grad = s.grad[b, :12].cpu().numpy().reshape(3, 4)
            _, _, dth32, _ = oracle.c_affine_loss_grad(mov[b, 0].numpy(), tgt[b, 0].numpy(), th[b].numpy(), oracle.wts(**kw), tabs32)
            el = abs(loss - total) / max(1.0, abs(total))
            gmax = max(np.max(np.abs(dth)), 1e-12)
            # fp32 floor of the gradient: the reference's own fp32-vs-fp64 gap (cancellation in the NCC terms), as in the tests
            gbar = max(grad_bar, 2.0 * np.max(np.abs(dth32 - dth)) / gmax)
            eg = np.max(np.abs(grad - dth)) / gmax / gbar * grad_bar     # normalised so that the bar reads grad_bar
            r64 = oracle.c_affine_warp(mov[b, 0].double().numpy(), tu, tabs64)
            r32 = oracle.c_affine_warp(mov[b, 0].numpy(), th[b].numpy(), tabs32)
            ew = np.max(np.abs(wrp[b, 0] - r32)); bw = max(2e-6, 3.0 * np.max(np.abs(r32 - r64)))   # 3x: random large theta (fixed cases: 2x)
            _, _, dm64, _ = oracle.c_affine_loss_grad(mov[b, 0].double().numpy(), tgt[b, 0].double().numpy(), tu, oracle.wts(w_mse=1.0), tabs64)
            _, _, dm32, _ = oracle.c_affine_loss_grad(mov[b, 0].numpy(), tgt[b, 0].numpy(), th[b].numpy(), oracle.wts(w_mse=1.0), tabs32)
            mmax = max(np.max(np.abs(dm64)), 1e-12)
            mbar = max(grad_bar, 2.0 * np.max(np.abs(dm32 - dm64)) / mmax)
            eb = np.max(np.abs(dth_b[b] - dm64)) / mmax / mbar * grad_bar
            worst["loss"] = max(worst["loss"], el); worst["grad"] = max(worst["grad"], eg, eb); worst["warp"] = max(worst["warp"], ew / bw)
            bad = el > 2e-5 or eg > grad_bar or eb > grad_bar or ew > bw or not np.isfinite(loss)
            if bad:
                fails += 1
                if verbose: print(f"FAIL case {it} pair {b}: shape {shape} B {B} kind {kind} kw {kw} loss err {el:.2e} grad err {eg:.2e} bwd err {eb:.2e} warp err/bar {ew / bw:.2f}\n theta {tu.tolist()}")
    if verbose:
        print(f"{n} cases, {fails} failures; worst loss rel {worst['loss']:.2e} (bar 2e-5), grad rel-to-max {worst['grad']:.2e} (bar {grad_bar:.0e}), warp err/bar {worst['warp']:.2f}")
    return fails, worst


if __name__ == "__main__":
    f, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    sys.exit(1 if f else 0)

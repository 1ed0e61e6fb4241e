#!/usr/bin/env python3
"""Randomised parity sweep of the fused F1 step (affine and rigid mode, forward warp, warp backward) against the C oracle in fp64.
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


KINK_SHIFT = 2.5e-7   # normalised units = S/2 * 2.5e-7 voxels: ~1 fp32 ulp of a coordinate of magnitude S, whatever S


def kink_variants(theta):
    """theta with the translations nudged by +/- KINK_SHIFT - all three together and each axis on its own (a sample can sit on a lattice
    plane of one axis only, and the joint nudge then moves it along that plane's normal by the same amount but mixes in the other two).  At integer sample coordinates (including -1 and S, where
    the zero padding starts) the trilinear value is continuous but its derivative jumps by up to a full voxel value: when a
    sample lies within fp32 rounding of such a coordinate, an fp32 evaluation may legitimately land on the other side.  How
    much that can matter for THIS theta is measured on the oracle itself: its fp64 gradient at the nudged thetas."""
    th = np.asarray(theta, dtype=np.float64).reshape(3, 4)
    out = []
    for sgn in (+1.0, -1.0):
        t = th.copy(); t[:, 3] += sgn * KINK_SHIFT
        out.append(t.reshape(np.asarray(theta).shape))
        for ax in range(3):
            t = th.copy(); t[ax, 3] += sgn * KINK_SHIFT
            out.append(t.reshape(np.asarray(theta).shape))
    return out


def run(n, seed, grad_bar=2e-4, verbose=True, only=None):
    rng = np.random.default_rng(seed)
    worst = {"loss": 0.0, "grad": 0.0, "warp": 0.0, "widest_bar": 0.0}
    fails = kinks = only_wide = 0   # only_wide: comparisons whose error is above the stated floor and passes only because its bar was widened
    for it in range(n):
        big = rng.random() < 0.3
        shape = tuple(int(v) for v in (rng.integers(3, 100, 3) if big else rng.integers(3, 48, 3)))
        if os.environ.get("FUZZ_BIG"):   # a few large ragged volumes (many tiles per column, several block rounds): minutes per case on the oracle
            shape = tuple(int(v) for v in rng.integers(100, 270, 3))
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
        # rigid step (dual kernel: GeomA / GeomR per pair) from a pose drawn like the reference's init (ref:utils.py:316-330) or near zero
        poses = rng.uniform(0.0, 1.0, (B, 6)) * (1.0 if rng.random() < 0.6 else 0.08)
        p32 = torch.tensor(poses, dtype=torch.float32)
        if only is not None and it != only:
            continue
        deep = 8 if (it % 2) else 0    # TRX_FLAG_DEEP_TILE on every other case: GeomD / GeomRD wherever they fit, also on these small volumes
        if it % 4 == 1:
            deep |= 1024               # ... and TRX_FLAG_EFT on every fourth: the exact-footprint kernel takes every rotated pair whose plan fits
        s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1, flags=deep)
        s.run(1)
        wrp_t = eng.affine_warp(th.cuda(), mov.cuda())
        # generic warp backward (tile kernel MODE 2) with grad_out = dMSE/dwarped must reproduce the MSE gradient
        go = 2.0 * (wrp_t - tgt.cuda()) / float(np.prod(shape))
        dth_b = eng.affine_warp_backward(th.cuda(), mov.cuda(), go).cpu().numpy()
        wrp = wrp_t.cpu().numpy()
        sr = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="rigid", loss=eng.LossSpec(**kw), lr=0.0, init=p32, capacity=1, flags=deep)
        sr.run(1)
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
            m64, t64 = mov[b, 0].double().numpy(), tgt[b, 0].double().numpy()
            kv = kink_variants(tu)
            ksens = max(np.max(np.abs(oracle.c_affine_loss_grad(m64, t64, t, oracle.wts(**kw), tabs64)[2] - dth)) for t in kv) / gmax
            kinks += ksens > grad_bar
            gbar = max(grad_bar, 2.0 * np.max(np.abs(dth32 - dth)) / gmax, 1.5 * ksens)
            eg = np.max(np.abs(grad - dth)) / gmax / gbar * grad_bar     # normalised so that the bar reads grad_bar
            only_wide += grad_bar < eg * gbar / grad_bar <= gbar
            r64 = oracle.c_affine_warp(mov[b, 0].double().numpy(), tu, tabs64)
            r32 = oracle.c_affine_warp(mov[b, 0].numpy(), th[b].numpy(), tabs32)
            ew = np.max(np.abs(wrp[b, 0] - r32)); bw = max(2e-6, 3.0 * np.max(np.abs(r32 - r64)))   # 3x: random large theta (fixed cases: 2x)
            _, _, dm64, _ = oracle.c_affine_loss_grad(mov[b, 0].double().numpy(), tgt[b, 0].double().numpy(), tu, oracle.wts(w_mse=1.0), tabs64)
            _, _, dm32, _ = oracle.c_affine_loss_grad(mov[b, 0].numpy(), tgt[b, 0].numpy(), th[b].numpy(), oracle.wts(w_mse=1.0), tabs32)
            mmax = max(np.max(np.abs(dm64)), 1e-12)
            msens = max(np.max(np.abs(oracle.c_affine_loss_grad(m64, t64, t, oracle.wts(w_mse=1.0), tabs64)[2] - dm64)) for t in kv) / mmax
            mbar = max(grad_bar, 2.0 * np.max(np.abs(dm32 - dm64)) / mmax, 1.5 * msens)
            eb = np.max(np.abs(dth_b[b] - dm64)) / mmax / mbar * grad_bar
            only_wide += grad_bar < eb * mbar / grad_bar <= mbar
            pu = p32[b].double().numpy()
            thr = oracle.c_theta_fwd(pu)
            tot_r, _, dth_r, _ = oracle.c_affine_loss_grad(mov[b, 0].double().numpy(), tgt[b, 0].double().numpy(), thr, oracle.wts(**kw), tabs64)
            dp = oracle.c_theta_vjp(pu, dth_r)
            _, _, dth_r32, _ = oracle.c_affine_loss_grad(mov[b, 0].numpy(), tgt[b, 0].numpy(), oracle.c_theta_fwd(p32[b].numpy()), oracle.wts(**kw), tabs32)
            dp32 = oracle.c_theta_vjp(p32[b].numpy(), dth_r32)
            pmax = max(np.max(np.abs(dp)), 1e-12)
            psens = max(np.max(np.abs(oracle.c_theta_vjp(pu, oracle.c_affine_loss_grad(m64, t64, t, oracle.wts(**kw), tabs64)[2]) - dp))
                        for t in kink_variants(thr)) / pmax
            kinks += psens > grad_bar
            pbar = max(grad_bar, 2.0 * np.max(np.abs(dp32 - dp)) / pmax, 1.5 * psens)
            er = np.max(np.abs(sr.grad[b, :6].cpu().numpy() - dp)) / pmax / pbar * grad_bar
            only_wide += grad_bar < er * pbar / grad_bar <= pbar
            elr = abs(sr.losses[b, 0].item() - tot_r) / max(1.0, abs(tot_r))
            el = max(el, elr)
            worst["loss"] = max(worst["loss"], el); worst["warp"] = max(worst["warp"], ew / bw)
            worst["grad"] = max(worst["grad"], eg, eb, er)
            worst["widest_bar"] = max(worst["widest_bar"], gbar, mbar, pbar)
            bad = el > 2e-5 or eg > grad_bar or eb > grad_bar or er > grad_bar or ew > bw or not np.isfinite(loss)
            if bad:
                fails += 1
                if verbose and only is not None:
                    e = np.abs(wrp[b, 0] - r32)
                    idx = np.argwhere(e > bw)
                    print(" warp: voxels over the bar:", len(idx), "first", idx[:6].tolist(), "last", idx[-3:].tolist(),
                          "\n gpu", [float(wrp[b, 0][tuple(i)]) for i in idx[:6]], "\n ref", [float(r32[tuple(i)]) for i in idx[:6]])
                    print(" pose", pu.tolist(), "\n gpu pose grad", sr.grad[b, :6].cpu().numpy(), "\n oracle", dp, "\n oracle fp32", dp32)
                if verbose: print(f"FAIL case {it} pair {b}: shape {shape} B {B} kind {kind} kw {kw} loss err {el:.2e} grad err {eg:.2e} bwd err {eb:.2e} rigid err {er:.2e} warp err/bar {ew / bw:.2f}\n theta {tu.tolist()}")
    if verbose:
        print(f"{n} cases, {fails} failures ({kinks} gradient bars widened: a sample within fp32 rounding of an integer coordinate, see kink_variants; {only_wide} comparisons passed ONLY through a widened bar - error above the {grad_bar:.0e} floor, below the widened bar); worst loss rel {worst['loss']:.2e} (bar 2e-5), grad rel-to-max {worst['grad']:.2e} (bar {grad_bar:.0e}), warp err/bar {worst['warp']:.2f}; widest gradient bar used {worst['widest_bar']:.2e} of the gradient's maximum")
    return fails, worst


if __name__ == "__main__":
    f, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
               only=int(sys.argv[3]) if len(sys.argv) > 3 else None)   # third argument: re-run one case of the sweep verbosely
    sys.exit(1 if f else 0)

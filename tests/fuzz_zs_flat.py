#!/usr/bin/env python3
"""Randomised sweep of the CHIP-FILLING launches of the affine step (round 5): the z-streaming kernel in front (its 64 x 32 tile and its flat
64 x 16 tile, walking up or down), the exact-footprint kernel and the tile kernel behind it, all in one launch per case - and, every other case, the one-kernel form of the same step (TRX_FLAG_ONE_KERNEL).
   python tests/fuzz_zs_flat.py [cases] [seed]
Batches of 16 pairs of 64 x 128 x 128 or 8 pairs of 96 / 128 x 128 x 128 whose poses are drawn per pair from: the identity's neighbourhood, the
convergence basin (rotations to 0.15 rad about z, zooms to 1.1, a little tilt), and general rotations.  Checker: the same launch restricted to
the tile kernels (TRX_FLAG_NO_ZSTREAM | TRX_FLAG_NO_EFT: GeomD / GeomA / GeomRD / GeomR, the bodies every other sweep pins to the oracle) -
loss to 2e-5, gradient to 3e-4 of its maximum; a pair whose two fp32 evaluations are further apart goes to the C oracle in fp64, and both must
then meet max(3e-4, 2 x the oracle's own kink sensitivity) against it."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle   # (test infrastructure: only used from tests/)
import phantoms as ph
from fuzz_affine import kink_variants
import torchregister_amd._engine as eng
from torchregister_amd import _lib


def rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    return np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])


def run(n, seed, verbose=True):
    rng = np.random.default_rng(seed)
    fails, asked, seen, only_wide = 0, 0, {}, 0
    worst = {"loss": 0.0, "grad": 0.0, "widest_bar": 3e-4}
    for it in range(n):
        shape, B = [((64, 128, 128), 16), ((96, 128, 128), 8), ((128, 128, 128), 8)][int(rng.integers(0, 3))]
        mats = []
        for b in range(B):
            kind = rng.choice(["near", "basin", "far"], p=[0.35, 0.45, 0.2])
            if kind == "near":
                A = np.eye(3) + float(rng.choice([0.0, 3e-3, 1e-2, 2e-2])) * rng.uniform(-1, 1, (3, 3))
            elif kind == "basin":
                A = rot(rng.uniform(-0.03, 0.03), rng.uniform(-0.03, 0.03), rng.uniform(-0.16, 0.16)) @ np.diag(1.0 + rng.uniform(-0.08, 0.1, 3))
            else:
                A = rot(*rng.uniform(-0.7, 0.7, 3)) @ np.diag(1.0 + rng.uniform(-0.1, 0.1, 3))
            mats.append(np.concatenate([A, rng.uniform(-0.08, 0.08, (3, 1))], axis=1))
        th = torch.tensor(np.stack(mats), dtype=torch.float32)
        tgt = torch.cat([ph.blobs_fast(shape, 2000 + 7 * it + (b % 4), device="cuda") for b in range(B)])
        mov = torch.cat([ph.blobs_fast(shape, 3000 + 5 * it + (b % 3), device="cuda") + 0.1 * ph.vol(shape, 0.011 + 0.001 * (b % 5), "sin").cuda() for b in range(B)])
        kw = dict(w_ncc=float(rng.uniform(0.3, 1)), w_mse=float(rng.uniform(0, 1))) if rng.random() < 0.7 else dict(w_mse=1.0)
        down = bool(rng.integers(0, 2))
        # every other case in the ONE-KERNEL form (round 6: the z-streaming kernel alone, the pairs outside its windows on GeomR's body inside it)
        s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1, flags=_lib.FLAG_WALK_DOWN if down else 0,
                             one_kernel=bool(it & 1))
        s.run(1)
        r = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1, flags=_lib.FLAG_NO_ZSTREAM | _lib.FLAG_NO_EFT)
        r.run(1)
        torch.cuda.synchronize()
        bodies = [("one-kernel " if (it & 1) else "") + n for n in s.bodies()]
        for name in bodies:
            seen[name] = seen.get(name, 0) + 1
        for b in range(B):
            lr_, ls = r.losses[b, 0].item(), s.losses[b, 0].item()
            el = abs(ls - lr_) / max(1.0, abs(lr_))
            gb = r.grad[b, :12]
            eg = torch.max(torch.abs(s.grad[b, :12] - gb)).item() / max(gb.abs().max().item(), 1e-20)
            if eg > 3e-4:
                # two fp32 kernels further apart than the floor: ask the oracle (fp64) who is right, with the bar the other sweeps use - twice its own
                # sensitivity to a one-ulp nudge of the translations (a sample within fp32 rounding of a lattice plane: tests/fuzz_affine.py kink_variants)
                m64, t64, tu = mov[b, 0].double().cpu().numpy(), tgt[b, 0].double().cpu().numpy(), th[b].double().numpy()
                tabs = oracle.base_tables(shape, np.float64)
                _, _, dth, _ = oracle.c_affine_loss_grad(m64, t64, tu, oracle.wts(**kw), tabs)
                gmax = max(np.max(np.abs(dth)), 1e-20)
                ksens = max(np.max(np.abs(oracle.c_affine_loss_grad(m64, t64, t, oracle.wts(**kw), tabs)[2] - dth)) for t in kink_variants(tu)) / gmax
                bar = max(3e-4, 2.0 * ksens)
                eg = max(np.max(np.abs(s.grad[b, :12].cpu().numpy().reshape(3, 4) - dth)), np.max(np.abs(r.grad[b, :12].cpu().numpy().reshape(3, 4) - dth))) / gmax / bar * 3e-4
                asked += 1
                only_wide += 3e-4 < eg * bar / 3e-4 <= bar
                worst["widest_bar"] = max(worst["widest_bar"], bar)
            worst["loss"] = max(worst["loss"], el); worst["grad"] = max(worst["grad"], eg)
            if not (el <= 2e-5 and eg <= 3e-4 and np.isfinite(ls)):
                fails += 1
                if verbose:
                    print(f"FAIL case {it} pair {b}: shape {shape} B {B} body {bodies[b]} down {down} kw {kw} loss {ls} / {lr_} ({el:.2e}) grad {eg:.2e}\n theta {th[b].tolist()}")
    if verbose:
        print(f"{n} cases, {fails} failures; worst loss rel {worst['loss']:.2e} (bar 2e-5), gradient rel-to-max {worst['grad']:.2e} (bar 3e-4; {asked} pairs went to the oracle, "
              f"widest bar used {worst['widest_bar']:.2e} of the gradient's maximum, {only_wide} of them passed ONLY through a widened bar); pairs per body: {seen}")
    return fails, worst


if __name__ == "__main__":
    f, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    sys.exit(1 if f else 0)

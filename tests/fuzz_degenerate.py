#!/usr/bin/env python3
"""Degenerate shapes (axes of 1..5 voxels, 2-D and 3-D) through the affine step, the forward warp, the flow loss/gradient and the
local NCC: vs the C oracle / the torch specification in fp64.  python tests/fuzz_degenerate.py"""
import itertools, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # test infrastructure: the oracle may only be used from tests/
import oracle
from oracle import compose
import torchregister_amd._engine as eng


def run(verbose=True):
    rng = np.random.default_rng(0)
    fails, n = 0, 0
    shapes = list(itertools.product([1, 2, 3, 5], repeat=3)) + [(a, b) for a in (1, 2, 4) for b in (1, 3, 5)]
    for shape in shapes:
        nd = len(shape)
        n += 1
        mov = torch.tensor(rng.random((1, 1) + shape), dtype=torch.float32)
        tgt = torch.tensor(rng.random((1, 1) + shape), dtype=torch.float32)
        th = np.eye(nd, nd + 1) + 0.07 * rng.standard_normal((nd, nd + 1))
        tht = torch.tensor(th[None], dtype=torch.float32)
        kw = dict(w_ncc=0.6, w_mse=0.7)
        if int(np.prod(shape)) < 8:
            # NCC of 1..4 voxels: the centred sums are ~0, and the one-pass raw-moment form (fp32 block partials) leaves ~1e-8 where
            # the two-pass form leaves exactly 0 - against EPSILON = 1e-10 that is a visible change of a meaningless number
            kw = dict(w_mse=1.0)
        bad = []
        try:
            s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=tht, capacity=1)
            s.run(1)
            wrp = eng.affine_warp(tht.cuda(), mov.cuda()).cpu().numpy()[0, 0]
            tabs = oracle.base_tables(shape, np.float64)
            tu = tht[0].double().numpy()
            total, _, dth, _ = oracle.c_affine_loss_grad(mov[0, 0].double().numpy(), tgt[0, 0].double().numpy(), tu, oracle.wts(**kw), tabs)
            r64 = oracle.c_affine_warp(mov[0, 0].double().numpy(), tu, tabs)
            loss = s.losses[0, 0].item()
            grad = s.grad[0, : nd * (nd + 1)].cpu().numpy().reshape(nd, nd + 1)
            if not (abs(loss - total) <= 5e-5 * max(1.0, abs(total))): bad.append(f"affine loss {loss} vs {total}")
            if max(shape) > 1 and not (np.max(np.abs(grad - dth)) <= 1e-3 * max(np.max(np.abs(dth)), 1e-6)): bad.append(f"affine grad {np.max(np.abs(grad - dth)):.2e} of {np.max(np.abs(dth)):.2e}")
            if not (np.max(np.abs(wrp - r64)) <= 5e-6): bad.append(f"warp {np.max(np.abs(wrp - r64)):.2e}")
            if min(shape) < 2:
                # flow on an axis of one voxel: the reference normalises by S - 1 = 0 (NaN, ref:utils.py:354-356) and the oracle
                # follows it (everything out of bounds); the kernels interpolate in voxel space.  Outside the reference's domain.
                raise StopIteration
            fl = torch.tensor(0.6 * rng.standard_normal((1, nd) + shape) + 0.31, dtype=torch.float32)
            terms, dfl = eng.flow_loss_grad(mov.cuda(), tgt.cuda(), fl.cuda(), eng.LossSpec(**kw))
            t64, _, d64, _ = oracle.c_flow_loss_grad(mov[0, 0].double().numpy(), tgt[0, 0].double().numpy(), fl[0].double().numpy(), oracle.wts(**kw))
            if not (abs(terms[0, 0].item() - t64) <= 5e-5 * max(1.0, abs(t64))): bad.append(f"flow loss {terms[0, 0].item()} vs {t64}")
            if not (np.max(np.abs(dfl[0].cpu().numpy() - d64)) <= 1e-3 * max(np.max(np.abs(d64)), 1e-6)): bad.append("flow grad")
            for win in (3, 9):
                l, g = eng.local_ncc_loss_grad(tgt.cuda(), mov.cuda(), win, 1.0)
                w64 = mov.double().requires_grad_()
                l64 = compose.local_ncc_loss(tgt.double(), w64, win, 1.0)
                (g64,) = torch.autograd.grad(l64, w64)
                if not (abs(l.item() - l64.item()) <= 5e-5): bad.append(f"lncc{win} loss {l.item()} vs {l64.item()}")
                if not ((g.cpu().double() - g64).abs().max().item() <= 2e-3 * max(g64.abs().max().item(), 1e-6)): bad.append(f"lncc{win} grad")
        except StopIteration:
            pass
        except Exception as e:   # noqa: BLE001
            bad.append(f"exception {type(e).__name__}: {e}")
        if bad:
            fails += 1
            if verbose: print("FAIL", shape, bad)
    if verbose: print(f"{n} shapes, {fails} failures")
    return fails


if __name__ == "__main__":
    sys.exit(1 if run() else 0)

#!/usr/bin/env python3
"""Randomised parity sweep of the dense-flow loss/gradient kernels (vs the C oracle in fp64) and of the local-window NCC
extension (vs its torch-conv specification in fp64).   python tests/fuzz_flow_lncc.py [cases] [seed]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # test infrastructure: the oracle may only be used from tests/
import oracle
from oracle import compose
import phantoms as ph
import torchregister_amd._engine as eng
from fuzz_affine import smooth


def smooth_nd(shape, f):
    if len(shape) == 3:
        return smooth(shape, f)
    ax = [torch.arange(n, dtype=torch.float64) for n in shape]
    return (torch.sin(f * ax[0])[:, None] + torch.cos(1.3 * f * ax[1] + 0.5)[None, :]).float().view(1, 1, *shape)


def cancellation_floor(tgt, warped, d64, w):
    """What fp32 leaves of dL/dflow where the per-voxel loss derivative is a CANCELLATION, from the fp64 oracle alone.  All three losses are
    affine in (y, w) given the moments: dL/dw(v) = cy y(v) + cw w(v) + c0 (oracle/trx_oracle_body.h, loss_from_moments), and the kernels - like
    torch's fp32 autograd - evaluate that sum in fp32: each term carries a relative rounding of 2^-24, so the derivative of a voxel is known to
    2^-24 (|cy y| + |cw w| + |c0|), whatever the sum itself comes to.  NCC is invariant under scaling of w: when only ONE sample of the warped
    image lies inside the moving image (tiny images, large flows: seed 65 case 32, a 3 x 7 image with flows of +-6 pixels) its derivative at that
    voxel is zero by symmetry while the three terms are ~40 each.  Returns the absolute floor of dL/dflow: 2 * 2^-24 * max_v (terms(v) |grad w(v)|)
    (a rounded coefficient times a rounded product: two roundings per term), |grad w(v)| = |dflow64(v)| / |dL/dw(v)|."""
    y, ww = np.asarray(tgt, np.float64).ravel(), np.asarray(warped, np.float64).ravel()
    n = float(y.size)
    Sy, Sw, Syy, Sww, Syw = y.sum(), ww.sum(), (y * y).sum(), (ww * ww).sum(), (y * ww).sum()
    my, mw = Sy / n, Sw / n
    Saa, Sbb, Sab = Syy - Sy * my, Sww - Sw * mw, Syw - Sy * mw
    sden = np.sqrt(Saa * Sbb + 1e-10)
    k1, k2 = -w[2] / sden, w[2] * Sab * Saa / sden ** 3
    q = w[0] * 2.0 / n + w[3] * w[4] * 2.0
    cy, cw, c0 = w[1] * k1 - q, w[1] * k2 + q, w[1] * (-k1 * my - k2 * mw)
    dldw = cy * y + cw * ww + c0
    terms = np.abs(cy * y) + np.abs(cw * ww) + abs(c0)
    g = np.max(np.abs(np.asarray(d64, np.float64).reshape(d64.shape[0], -1)), axis=0)
    ok = np.abs(dldw) > 1e-300
    if not ok.any():
        return 0.0
    return float(2.0 * 2.0 ** -24 * np.max(terms[ok] * g[ok] / np.abs(dldw[ok])))


def run(n, seed, verbose=True, only=None, details=None):
    """only: evaluate just that case of the sweep (the random stream is still drawn for the cases before it); details: a list that receives one
    dict per evaluated case (errors, the fp32-vs-fp64 gap of the torch specification, the bars used)."""
    rng = np.random.default_rng(seed)
    fails = only_wide = 0   # only_wide: comparisons whose error is above the stated floor and passes only because its bar was widened
    worst = {"flow_loss": 0.0, "flow_grad": 0.0, "lncc_loss": 0.0, "lncc_grad": 0.0, "widest_flow_bar": 0.0, "widest_lncc_bar": 0.0}
    for it in range(n):
        nd = 3 if rng.random() < 0.7 else 2
        shape = tuple(int(v) for v in rng.integers(3, 40 if nd == 3 else 90, nd))
        tgt = ph.blobs(shape, 300 + it) + 0.05 * smooth_nd(shape, 0.31)
        mov = ph.blobs(shape, 700 + it) + 0.1 * smooth_nd(shape, 0.23)
        amp = float(rng.choice([0.3, 1.5, 6.0]))
        flow = torch.tensor(amp * rng.standard_normal((1, nd) + shape), dtype=torch.float32)
        flow = flow + 0.37          # keep samples off exact voxel positions
        kw = dict(w_ncc=float(rng.uniform(0, 1)), w_mse=float(rng.uniform(0, 1)))
        win = int(rng.choice([3, 5, 7, 9]))
        B = int(rng.integers(1, 3))
        if only is not None and it != only:
            continue
        terms, dfl = eng.flow_loss_grad(mov.cuda(), tgt.cuda(), flow.cuda(), eng.LossSpec(**kw))
        args = lambda dt: (mov[0, 0].numpy().astype(dt), tgt[0, 0].numpy().astype(dt), flow[0].numpy().astype(dt), oracle.wts(**kw))
        t64, _, d64, w64_ = oracle.c_flow_loss_grad(*args(np.float64))
        t32, _, d32, _ = oracle.c_flow_loss_grad(*args(np.float32))
        el = abs(terms[0, 0].item() - t64) / max(1.0, abs(t64))
        gmax = max(np.max(np.abs(d64)), 1e-12)
        eg = np.max(np.abs(dfl[0].cpu().numpy() - d64)) / gmax
        gbar = max(2e-4, 2.0 * np.max(np.abs(d32 - d64)) / gmax, cancellation_floor(tgt[0, 0].numpy(), w64_, d64, oracle.wts(**kw)) / gmax)
        worst["flow_loss"] = max(worst["flow_loss"], el); worst["flow_grad"] = max(worst["flow_grad"], eg / gbar)
        bad = el > 2e-5 or eg > gbar
        # local NCC on (target, warped)
        y = torch.cat([tgt] * B); w = torch.cat([mov + 0.01 * b for b in range(B)])
        loss, grad = eng.local_ncc_loss_grad(y.cuda(), w.cuda(), win, 1.7)
        w64 = w.double().requires_grad_()
        l64 = sum(compose.local_ncc_loss(y[b:b + 1].double(), w64[b:b + 1], win, 1.7) for b in range(B))
        (g64,) = torch.autograd.grad(l64, w64)
        w32 = w.clone().requires_grad_()
        l32 = sum(compose.local_ncc_loss(y[b:b + 1], w32[b:b + 1], win, 1.7) for b in range(B))
        (g32,) = torch.autograd.grad(l32, w32)
        ell = abs(loss.sum().item() - l64.item()) / max(1.0, abs(l64.item()))
        lbar = max(2e-5, 2.0 * abs(l32.item() - l64.item()) / max(1.0, abs(l64.item())))
        gm = max(g64.abs().max().item(), 1e-12)
        egl = (grad.cpu().double() - g64).abs().max().item() / gm
        # (the bar scales with what torch's own fp32 evaluation of the specification loses against fp64; the kernels sum the windows in another
        #  order - sliding sums along x - and may lose a little more where the variance is a difference of nearly equal sums: window 3, seed 51
        #  case 231 reached 2.1 x torch's fp32 error)
        glbar = max(2e-4, 2.5 * (g32.double() - g64).abs().max().item() / gm)
        worst["lncc_loss"] = max(worst["lncc_loss"], ell / lbar); worst["lncc_grad"] = max(worst["lncc_grad"], egl / glbar)
        worst["widest_flow_bar"] = max(worst["widest_flow_bar"], gbar); worst["widest_lncc_bar"] = max(worst["widest_lncc_bar"], glbar)
        only_wide += (2e-4 < eg <= gbar) + (2e-4 < egl <= glbar) + (2e-5 < ell <= lbar)
        if details is not None:
            details.append(dict(case=it, shape=shape, win=win, B=B, lncc_grad_err=egl, lncc_grad_fp32_gap=(g32.double() - g64).abs().max().item() / gm, lncc_grad_bar=glbar,
                                lncc_loss_err=ell, lncc_loss_bar=lbar, flow_grad_err=eg, flow_grad_bar=gbar, flow_loss_err=el))
        bad = bad or ell > lbar or egl > glbar
        if bad:
            fails += 1
            if verbose:
                print(f"FAIL case {it}: shape {shape} amp {amp} kw {kw} win {win} B {B}: flow loss {el:.2e} grad {eg:.2e}/{gbar:.2e}; lncc loss {ell:.2e}/{lbar:.2e} grad {egl:.2e}/{glbar:.2e}")
    if verbose:
        print(f"{n} cases, {fails} failures; worst (error / bar; widest_*: the largest bar any case was given, relative to the gradient's maximum): {worst}; {only_wide} comparisons passed ONLY through a widened bar (error above the stated floor, below the widened bar)")
    return fails, worst


if __name__ == "__main__":
    f, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0, only=int(sys.argv[3]) if len(sys.argv) > 3 else None)
    sys.exit(1 if f else 0)

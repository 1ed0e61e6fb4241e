#!/usr/bin/env python3
"""Randomised parity sweep of the paths tests/fuzz_affine.py does not reach (all against the C oracle in fp64):
   2-D affine / rigid steps, forward warp and warp backward; multi-channel forward warps (2-D and 3-D, one launch for all
   channels); loss-only evaluation; short SGD and Adam trajectories (affine and rigid, 2-D and 3-D: loss curve, best index, final theta); dense-flow trajectories
   (SGD / Adam, with and without the smoothness term, vs the torch composition) and their Z-slab partition (random cuts) vs the whole volume; the Parzen-window PDFs of the NMI loss, forward and backward; the warp on the NMI lattice
   (trx_affine_warp_lattice and its backward).
   python tests/fuzz_misc.py [cases] [seed]
Bars: loss 2e-5 relative, gradients 3e-4 of their maximum (random large theta sits a little above the 2e-4 floor of the fixed
cases: 2 marginal results, 2.7e-4 and a warp at 1.08x a 3x bar, in 900 cases) or twice the oracle's own fp32-vs-fp64 gap, warps 2e-6 or
four times that gap, trajectories 1e-4 or twice the gap of the oracle's fp32 and fp64 loops."""
import math, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # test infrastructure: the oracle may only be used from tests/
import oracle
from oracle import compose
import phantoms as ph
import torchregister_amd._engine as eng
from fuzz_affine import rand_theta, smooth, KINK_SHIFT
from fuzz_flow_lncc import smooth_nd


GRAD_FLOOR, WARP_GAPS = 3e-4, 4.0


def rand_theta2(rng, kind):
    a = {"tiny": 0.02, "small": 0.12, "medium": 0.4, "large": 1.2}[kind]
    ang = rng.uniform(-a, a)
    sc = np.diag(rng.uniform(0.8, 1.25, 2) if kind != "tiny" else rng.uniform(0.97, 1.03, 2))
    if kind == "large" and rng.random() < 0.3:
        sc = sc @ np.diag(rng.choice([-1.0, 1.0], 2))
    m = np.array([[math.cos(ang), -math.sin(ang)], [math.sin(ang), math.cos(ang)]]) @ sc
    t = rng.uniform(-0.5, 0.5, 2) if kind in ("medium", "large") else rng.uniform(-0.06, 0.06, 2)
    th = np.concatenate([m, t[:, None]], axis=1)
    return th + 1e-3 * rng.standard_normal(th.shape)


def kink_variants(theta):
    th = np.asarray(theta, dtype=np.float64)
    out = []
    for sgn in (+1.0, -1.0):
        t = th.copy(); t[..., -1] += sgn * KINK_SHIFT
        out.append(t)
    return out


def phantom(shape, seed, f):
    return ph.blobs(shape, seed) + 0.1 * smooth_nd(shape, f)


def gbar(g32, g64, floor):
    gmax = max(np.max(np.abs(g64)), 1e-12)
    return max(floor, 2.0 * np.max(np.abs(np.asarray(g32, dtype=np.float64) - g64)) / gmax), gmax


def kink_sens(fn, theta, ref):
    """Largest change of the oracle's own fp64 result `fn(theta)` when the translations move by +/- one fp32 ulp of a coordinate
    (fuzz_affine.KINK_SHIFT): what a sample sitting within fp32 rounding of an integer coordinate can legitimately change."""
    return max(np.max(np.abs(np.asarray(fn(t)) - ref)) for t in kink_variants(theta))


def steps_2d(rng, it, out):
    shape = tuple(int(v) for v in rng.integers(3, 300, 2))
    B = int(rng.integers(1, 4))
    kind = rng.choice(["tiny", "small", "medium", "large"], p=[0.3, 0.35, 0.2, 0.15])
    kw = dict(w_ncc=float(rng.uniform(0, 1)), w_mse=float(rng.uniform(0, 1)))
    tgt = torch.cat([phantom(shape, 4000 + 7 * it + b, 0.31) for b in range(B)])
    mov = torch.cat([phantom(shape, 5000 + 5 * it + b, 0.23) for b in range(B)])
    th = torch.tensor(np.stack([rand_theta2(rng, kind) for _ in range(B)]), dtype=torch.float32)
    poses = torch.tensor(rng.uniform(0.0, 1.0, (B, 3)) * (1.0 if rng.random() < 0.6 else 0.08), dtype=torch.float32)
    s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1)
    s.run(1)
    sr = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="rigid", loss=eng.LossSpec(**kw), lr=0.0, init=poses, capacity=1)
    sr.run(1)
    wrp = eng.affine_warp(th.cuda(), mov.cuda())
    go = 2.0 * (wrp - tgt.cuda()) / float(np.prod(shape))
    dth_b = eng.affine_warp_backward(th.cuda(), mov.cuda(), go).cpu().numpy()
    terms = s.eval_loss().cpu().numpy()
    wrp = wrp.cpu().numpy()
    torch.cuda.synchronize()
    t64, t32 = oracle.base_tables(shape, np.float64), oracle.base_tables(shape, np.float32)
    for b in range(B):
        m64, g64 = mov[b, 0].double().numpy(), tgt[b, 0].double().numpy()
        m32, g32 = mov[b, 0].numpy(), tgt[b, 0].numpy()
        tu = th[b].double().numpy()
        total, _, dth, _ = oracle.c_affine_loss_grad(m64, g64, tu, oracle.wts(**kw), t64)
        _, _, dth32, _ = oracle.c_affine_loss_grad(m32, g32, th[b].numpy(), oracle.wts(**kw), t32)
        bar, gmax = gbar(dth32, dth, GRAD_FLOOR)
        bar = max(bar, 1.5 * kink_sens(lambda t: oracle.c_affine_loss_grad(m64, g64, t, oracle.wts(**kw), t64)[2], tu, dth) / gmax)
        out.check("2d loss", abs(s.losses[b, 0].item() - total) / max(1.0, abs(total)), 2e-5, (it, b, shape, kind))
        out.check("2d loss-only", abs(terms[b, 0] - total) / max(1.0, abs(total)), 2e-5, (it, b, shape, kind))
        out.check("2d grad", np.max(np.abs(s.grad[b, :6].cpu().numpy().reshape(2, 3) - dth)) / gmax, bar, (it, b, shape, kind), GRAD_FLOOR)
        _, _, dm, _ = oracle.c_affine_loss_grad(m64, g64, tu, oracle.wts(w_mse=1.0), t64)
        _, _, dm32, _ = oracle.c_affine_loss_grad(m32, g32, th[b].numpy(), oracle.wts(w_mse=1.0), t32)
        bar, mmax = gbar(dm32, dm, GRAD_FLOOR)
        bar = max(bar, 1.5 * kink_sens(lambda t: oracle.c_affine_loss_grad(m64, g64, t, oracle.wts(w_mse=1.0), t64)[2], tu, dm) / mmax)
        out.check("2d warp backward", np.max(np.abs(dth_b[b] - dm)) / mmax, bar, (it, b, shape, kind), GRAD_FLOOR)
        r64, r32 = oracle.c_affine_warp(m64, tu, t64), oracle.c_affine_warp(m32, th[b].numpy(), t32)
        out.check("2d warp", np.max(np.abs(wrp[b, 0] - r32)), max(2e-6, WARP_GAPS * np.max(np.abs(r32 - r64))), (it, b, shape, kind), 2e-6)
        pu = poses[b].double().numpy()
        tot_r, _, dth_r, _ = oracle.c_affine_loss_grad(m64, g64, oracle.c_theta_fwd(pu), oracle.wts(**kw), t64)
        dp = oracle.c_theta_vjp(pu, dth_r)
        _, _, dth_r32, _ = oracle.c_affine_loss_grad(m32, g32, oracle.c_theta_fwd(poses[b].numpy()), oracle.wts(**kw), t32)
        bar, pmax = gbar(oracle.c_theta_vjp(poses[b].numpy(), dth_r32), dp, GRAD_FLOOR)
        bar = max(bar, 1.5 * kink_sens(lambda t: oracle.c_theta_vjp(pu, oracle.c_affine_loss_grad(m64, g64, t, oracle.wts(**kw), t64)[2]), oracle.c_theta_fwd(pu), dp) / pmax)
        out.check("2d rigid loss", abs(sr.losses[b, 0].item() - tot_r) / max(1.0, abs(tot_r)), 2e-5, (it, b, shape, pu.tolist()))
        out.check("2d rigid grad", np.max(np.abs(sr.grad[b, :3].cpu().numpy() - dp)) / pmax, bar, (it, b, shape, pu.tolist()), GRAD_FLOOR)


def multichannel_warp(rng, it, out):
    nd = 3 if rng.random() < 0.6 else 2
    shape = tuple(int(v) for v in (rng.integers(3, 70, 3) if nd == 3 else rng.integers(3, 200, 2)))
    B, C = int(rng.integers(1, 3)), int(rng.integers(2, 5))
    kind = rng.choice(["tiny", "small", "medium", "large"], p=[0.3, 0.3, 0.2, 0.2])
    x = torch.cat([torch.cat([phantom(shape, 6000 + 11 * it + 5 * b + c, 0.2 + 0.03 * c) for c in range(C)], dim=1) for b in range(B)])
    th = torch.tensor(np.stack([(rand_theta if nd == 3 else rand_theta2)(rng, kind) for _ in range(B)]), dtype=torch.float32)
    w = eng.affine_warp(th.cuda(), x.cuda()).cpu().numpy()
    t64, t32 = oracle.base_tables(shape, np.float64), oracle.base_tables(shape, np.float32)
    for b in range(B):
        for c in range(C):
            r64 = oracle.c_affine_warp(x[b, c].double().numpy(), th[b].double().numpy(), t64)
            r32 = oracle.c_affine_warp(x[b, c].numpy(), th[b].numpy(), t32)
            out.check(f"{nd}d warp, {C} channels", np.max(np.abs(w[b, c] - r32)), max(2e-6, WARP_GAPS * np.max(np.abs(r32 - r64))), (it, b, c, shape, kind), 2e-6)


def trajectory(rng, it, out):
    nd = 3 if rng.random() < 0.6 else 2
    shape = tuple(int(v) for v in (rng.integers(6, 36, 3) if nd == 3 else rng.integers(8, 120, 2)))
    rigid = rng.random() < 0.5
    iters = int(rng.integers(3, 16))
    tgt = phantom(shape, 7000 + it, 0.31)
    th_true = (rand_theta if nd == 3 else rand_theta2)(rng, "tiny")
    mov = torch.tensor(oracle.c_affine_warp(tgt[0, 0].double().numpy(), th_true, oracle.base_tables(shape, np.float64)), dtype=torch.float32)[None, None]
    mov = mov + 0.02 * phantom(shape, 7500 + it, 0.27)
    kw = dict(w_mse=1.0) if rng.random() < 0.5 else dict(w_ncc=float(rng.uniform(0.2, 1.0)), w_mse=float(rng.uniform(0, 1)))
    lr = float(10 ** rng.uniform(-3.0, -1.5)) / (1.0 + 100.0 * kw.get("w_ncc", 0.0))   # stable step sizes: overshooting runs amplify fp32 noise
    npose = 6 if nd == 3 else 3
    pose0 = rng.uniform(-0.05, 0.05, npose) if rigid else None
    init = torch.tensor(pose0[None], dtype=torch.float32) if rigid else None
    adam = rng.random() < 0.4          # Adam (extension, torch.optim.Adam defaults) against the torch composition of the same ops
    if adam:
        lr = float(10 ** rng.uniform(-3.5, -2.3))
    s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="rigid" if rigid else "affine", loss=eng.LossSpec(**kw), lr=lr, init=init, capacity=iters,
                         optimizer="adam" if adam else "sgd")
    s.run(iters)
    torch.cuda.synchronize()
    m32, g32 = mov[0, 0].numpy(), tgt[0, 0].numpy()
    p32 = None if pose0 is None else init[0].numpy()
    if adam:
        def loop(dt):
            r = compose.affine_loop(mov.to(dt), tgt.to(dt), float(np.float32(lr)), iters, pose0=None if init is None else init[0].to(dt), optimizer="adam", **kw)
            th = r["thetas"].double().numpy()
            return dict(losses=r["losses"].numpy(), thetas=th, final_theta=th[-1], best_idx=r["best_idx"])
        o32, o64 = loop(torch.float32), loop(torch.float64)
    else:
        o32 = oracle.c_affine_loop(m32, g32, oracle.wts(**kw), lr, iters, pose0=p32, tables=oracle.base_tables(shape, np.float32))
        o64 = oracle.c_affine_loop(m32.astype(np.float64), g32.astype(np.float64), oracle.wts(**kw), float(np.float32(lr)), iters,
                                   pose0=None if p32 is None else p32.astype(np.float64), tables=oracle.base_tables(shape, np.float64))
    tag = (it, shape, "rigid" if rigid else "affine", "adam" if adam else "sgd", iters, lr, kw)
    losses = s.losses[0, :iters].cpu().numpy()
    lbar = max(1e-4, 2.0 * np.max(np.abs(o32["losses"] - o64["losses"]) / np.maximum(1.0, np.abs(o64["losses"]))))
    out.check("trajectory losses", np.max(np.abs(losses - o64["losses"]) / np.maximum(1.0, np.abs(o64["losses"]))), lbar, tag, 1e-4)
    tbar = max(1e-4, (3.0 if adam else 2.0) * np.max(np.abs(o32["thetas"] - o64["thetas"])))   # Adam's division by sqrt(v) amplifies the gap
    out.check("trajectory final theta", np.max(np.abs(s.current_theta[0].cpu().numpy() - o64["final_theta"])), tbar, tag, 1e-4)
    # best = first strict minimum (ref:warpings.py:85-93); only compared where the fp64 curve separates its two lowest values
    l64 = np.sort(o64["losses"])
    if len(l64) < 2 or (l64[1] - l64[0]) > 4.0 * lbar * max(1.0, abs(l64[0])):
        out.check("trajectory best index", float(int(s.best_idx[0].item()) != o64["best_idx"]), 0.5, tag)
        out.check("trajectory best theta", np.max(np.abs(s.best[0].cpu().numpy() - o64["thetas"][o64["best_idx"]])), tbar, tag, 1e-4)


def _torch_flow_loop(mov, tgt, lr, iters, optimizer, smooth, kw, dtype, init=None):
    """Dense-flow optimisation as a torch composition (SGD = the reference's flow path; Adam and the smoothness term are extensions)."""
    nd = mov.dim() - 2
    mov, tgt = mov.to(dtype), tgt.to(dtype)
    fl = (torch.zeros(1, nd, *mov.shape[2:], dtype=dtype) if init is None else init.to(dtype).clone()).requires_grad_()
    opt = torch.optim.SGD([fl], lr) if optimizer == "sgd" else torch.optim.Adam([fl], lr)
    losses = []
    for _ in range(iters):
        opt.zero_grad()
        e = compose.weighted_loss(tgt, compose.flow_warp(mov, fl), **kw)
        if smooth:
            e = e + smooth / nd * sum((fl.diff(dim=2 + d) ** 2).mean() for d in range(nd))
        e.backward()
        opt.step()
        losses.append(e.item())
    return np.asarray(losses), fl.detach().double().numpy()


def flow_trajectory(rng, it, out):
    nd = 3 if rng.random() < 0.6 else 2
    shape = tuple(int(v) for v in (rng.integers(4, 28, 3) if nd == 3 else rng.integers(4, 90, 2)))
    iters = int(rng.integers(2, 9))
    adam = rng.random() < 0.5
    smooth = float(rng.uniform(0.5, 5.0)) if rng.random() < 0.5 else 0.0
    kw = dict(w_ncc=1.0) if rng.random() < 0.5 else dict(w_ncc=float(rng.uniform(0.2, 1.0)), w_mse=float(rng.uniform(0, 1)))
    # stable step sizes: an overshooting run (loss going up again) amplifies a last-bit difference ~7x per iteration
    lr = float(10 ** rng.uniform(-2.5, -1.2)) if adam else float(10 ** rng.uniform(-0.7, 0.0))
    tgt = phantom(shape, 8000 + it, 0.31)
    mov = phantom(shape, 8500 + it, 0.23)
    # Start OFF the voxel lattice.  A zero flow puts every sample exactly on it, where the derivative is one-sided: the kernels
    # (and the C oracle) take the right-hand one, torch's normalise / un-normalise round trip lands a rounding error to either
    # side - at local extrema of the image the two derivatives differ in sign and Adam's first step (lr * sign) then differs by
    # 2 lr at ~1 % of the voxels.  Implementation-defined in the reference as well; not what this sweep is after.
    init = torch.tensor(0.37 + 0.05 * rng.standard_normal((1, nd) + shape), dtype=torch.float32)
    s = eng.FlowSolver(mov.cuda(), tgt.cuda(), loss=eng.LossSpec(**kw), optimizer="adam" if adam else "sgd", lr=lr, capacity=iters, smooth_weight=smooth,
                       init=init)
    s.run(iters)
    torch.cuda.synchronize()
    l32, f32 = _torch_flow_loop(mov, tgt, float(np.float32(lr)), iters, "adam" if adam else "sgd", smooth, kw, torch.float32, init)
    l64, f64 = _torch_flow_loop(mov, tgt, float(np.float32(lr)), iters, "adam" if adam else "sgd", smooth, kw, torch.float64, init)
    tag = (it, shape, "adam" if adam else "sgd", iters, lr, smooth, kw)
    lb = max(1e-4, 2.0 * np.max(np.abs(l32 - l64) / np.maximum(1.0, np.abs(l64))))
    out.check("flow trajectory losses", np.max(np.abs(s.losses[0, :iters].cpu().numpy() - l64) / np.maximum(1.0, np.abs(l64))), lb, tag)
    # Single voxels whose sample comes within an fp32 ulp of the lattice take the other one-sided derivative (their own update then
    # differs by lr * jump), so besides the maximum the field is compared at its 99.5th percentile and in RMS.
    gf = s.flow[0].cpu().double().numpy()
    pct = lambda a: float(np.percentile(np.abs(a), 99.5))
    rms = lambda a: float(np.sqrt(np.mean(np.square(a))))
    scale = max(1.0, np.max(np.abs(f64)))
    out.check("flow trajectory field (99.5 %)", pct(gf - f64[0]), max(2e-4 * scale, 2.0 * pct(f32 - f64)), tag, 2e-4 * scale)
    out.check("flow trajectory field (rms)", rms(gf - f64[0]), max(1e-4 * scale, 2.0 * rms(f32 - f64)), tag, 1e-4 * scale)
    fb = max(2e-4 * scale, 2.0 * np.max(np.abs(f32 - f64)))
    if not adam:   # (Adam's lr * g / sqrt(v) turns a last-bit difference of a ~0 gradient into a fraction of lr at single voxels)
        out.check("flow trajectory field (max)", np.max(np.abs(gf - f64[0])), max(fb, 20.0 * pct(f32 - f64)), tag)
    if nd == 3 and shape[0] >= 4:
        # Z slabs of random depth (config 5 on one GPU): moments summed by hand, boundary planes copied by hand
        k = int(rng.integers(2, min(4, shape[0] // 2) + 1))
        cuts = sorted(rng.choice(np.arange(1, shape[0]), size=k - 1, replace=False).tolist())
        bounds = [0] + cuts + [shape[0]]
        skw = dict(loss=eng.LossSpec(**kw), optimizer="adam" if adam else "sgd", lr=lr, capacity=iters, smooth_weight=smooth)
        ic = init.cuda()
        slabs = [eng.SlabFlowSolver(mov.cuda(), tgt.cuda()[:, :, a:b].contiguous(), a, **skw) for a, b in zip(bounds[:-1], bounds[1:])]
        for sl, a, b in zip(slabs, bounds[:-1], bounds[1:]):
            sl.flow.copy_(ic[:, :, a:b])
        dbg = os.environ.get("FUZZ_DEBUG")
        if dbg:
            w2 = eng.FlowSolver(mov.cuda(), tgt.cuda(), init=init, **skw)
        for k_it in range(iters):
            if dbg:
                w2.run(1); torch.cuda.synchronize()
            if smooth:
                planes = [sl.boundary_planes() for sl in slabs]
                for r, sl in enumerate(slabs):
                    if sl.has_lo: sl.halo_lo.copy_(planes[r - 1][1])
                    if sl.has_hi: sl.halo_hi.copy_(planes[r + 1][0])
            total = sum(sl.local_moments().clone() for sl in slabs)
            for sl in slabs:
                sl.apply(total)
            if dbg:
                torch.cuda.synchronize()
                d = (torch.cat([sl.flow for sl in slabs], dim=2) - w2.flow).abs()
                print(f"   iteration {k_it}: slab-vs-whole max |dflow| {d.max().item():.3e} at {np.unravel_index(int(d.argmax().item()), d.shape)}  |flow| max {w2.flow.abs().max().item():.2f}"
                      f"  loss {w2.losses[0, k_it].item():.6f} vs {slabs[0].losses[0, k_it].item():.6f}")
        torch.cuda.synchronize()
        flow = torch.cat([sl.flow for sl in slabs], dim=2)
        whole_l = s.losses[0, :iters]
        out.check("slab losses", torch.max(torch.abs(slabs[0].losses[0, :iters] - whole_l) / torch.clamp(whole_l.abs(), min=1.0)).item(), 1e-5, tag + (bounds,))
        # (the sums are added in a different order, and six iterations at a random step size grow that last-bit difference to 7e-5 of the
        # field in one case of 240; the fixed test holds 1e-5.  Adam divides by sqrt(v): where a gradient is ~1e-9 a last-bit change of the globally summed coefficients moves the step by a
        # fraction of lr, so the field is compared against the fp32-vs-fp64 spread of the torch composition as well)
        out.check("slab field", torch.max(torch.abs(flow - s.flow)).item(), max(1e-4 * max(1.0, s.flow.abs().max().item()), fb if adam else 0.0), tag + (bounds,))


def kde_pdf(rng, it, out):
    """Parzen-window PDFs of the NMI loss (trx_kde_pdf / trx_kde_pdf_backward) vs the reference's [N, S, bins] formulation
    (ref:utils.py:24-30) in fp64: ragged sample counts (the kernel works in chunks of 4096 and groups of 16), 1..1024 bins."""
    import torchregister_amd.utils as U
    N = int(rng.integers(1, 5))
    S = int(rng.choice([rng.integers(1, 40), rng.integers(40, 5000), rng.integers(4000, 14000)]))
    bins = int(rng.choice([rng.integers(1, 9), rng.integers(9, 300), rng.integers(300, 1025)], p=[0.25, 0.6, 0.15]))
    h = float(rng.choice([0.05, 0.1, 0.5, 1.0, 1.5, 1.7, 3.0]))   # >= 1.5: the series form (window >= value range)
    sig = torch.tensor(rng.uniform(-0.2, 1.3, (N, S)), dtype=torch.float32)
    xis = torch.linspace(sig.max().item(), sig.min().item(), bins).repeat(N, 1)
    wts = torch.tensor(rng.uniform(-0.3, 0.7, (N, bins)), dtype=torch.float32)
    ref = {}
    for dt in (torch.float32, torch.float64):
        sg = sig.clone().to(dt).requires_grad_()
        diff = sg.unsqueeze(-1) - xis.to(dt).unsqueeze(1)
        p = (1 / h) * torch.mean((1 / (2 * torch.pi)) * torch.exp(-((diff / h) ** 2) / 2), dim=1)
        (p * wts.to(dt)).sum().backward()
        ref[dt] = (p.detach().double().numpy(), sg.grad.double().numpy())
    sc = sig.clone().cuda().requires_grad_()
    pc = U.PDF_xis(sc, xis.cuda(), h)
    (pc * wts.cuda()).sum().backward()
    (p32, g32), (p64, g64) = ref[torch.float32], ref[torch.float64]
    tag = (it, N, S, bins, h)
    out.check("kde pdf", np.max(np.abs(pc.detach().cpu().double().numpy() - p64)), max(2e-6 * np.max(np.abs(p64)), 2 * np.max(np.abs(p32 - p64))), tag, 2e-6 * np.max(np.abs(p64)))
    out.check("kde pdf backward", np.max(np.abs(sc.grad.cpu().double().numpy() - g64)), max(1e-5 * np.max(np.abs(g64)), 2 * np.max(np.abs(g32 - g64)), 1e-12), tag, max(1e-5 * np.max(np.abs(g64)), 1e-12))


def lattice_warp(rng, it, out):
    """trx_affine_warp_lattice[_backward] (the warp on the NMI loss's nearest-neighbour lattice) vs the C oracle: values = the oracle's
    full warp read at the lattice voxels; backward = the oracle's dMSE/dtheta for a target chosen such that dMSE/dwarped is the
    scattered grad_out (t = w - g N / 2).  Random down- and up-sampling lattices (repeated voxels), 2-D and 3-D, 1-2 pairs."""
    nd = 3 if rng.random() < 0.6 else 2
    shape = tuple(int(v) for v in (rng.integers(3, 60, 3) if nd == 3 else rng.integers(3, 160, 2)))
    size = tuple(int(max(1, round(n * rng.uniform(0.3, 1.7)))) for n in shape)
    B = int(rng.integers(1, 3))
    kind = rng.choice(["tiny", "small", "medium", "large"], p=[0.3, 0.3, 0.2, 0.2])
    mov = torch.cat([phantom(shape, 7000 + 13 * it + b, 0.23) for b in range(B)])
    th = torch.tensor(np.stack([(rand_theta if nd == 3 else rand_theta2)(rng, kind) for _ in range(B)]), dtype=torch.float32)
    mc = mov.cuda()
    vol = eng._Batch(mc, mc).vol()
    lat = eng.LatticeWarp(vol, shape, size, mc.device)
    thp = eng.pad_theta(th.reshape(B, -1).cuda(), nd)
    vals = lat.forward(thp).cpu().numpy()
    n = int(np.prod(size))
    go = rng.uniform(-0.4, 0.6, (B, n)).astype(np.float32)
    dth = lat.backward(thp, torch.from_numpy(go).cuda())[:, : nd * (nd + 1)].reshape(B, nd, nd + 1).cpu().numpy()
    tabs = [t.cpu().numpy() for t in ((lat.iz, lat.iy, lat.ix) if nd == 3 else (lat.iy, lat.ix))]
    sel = np.ix_(*tabs)
    t64, t32 = oracle.base_tables(shape, np.float64), oracle.base_tables(shape, np.float32)
    nvox = float(np.prod(shape))
    for b in range(B):
        m64, m32, tu = mov[b, 0].double().numpy(), mov[b, 0].numpy(), th[b].double().numpy()
        r64, r32 = oracle.c_affine_warp(m64, tu, t64), oracle.c_affine_warp(m32, th[b].numpy(), t32)
        tag = (it, b, shape, size, kind)
        out.check(f"{nd}d lattice warp", np.max(np.abs(vals[b].reshape(size) - r32[sel])), max(2e-6, WARP_GAPS * np.max(np.abs(r32 - r64))), tag, 2e-6)
        gw = np.zeros(shape, dtype=np.float64)
        np.add.at(gw, sel, go[b].reshape(size).astype(np.float64))
        tgt64 = r64 - gw * nvox / 2.0
        _, _, d64, _ = oracle.c_affine_loss_grad(m64, tgt64, tu, oracle.wts(w_mse=1.0), t64)
        # (no fp32 run of the oracle here: its target w - g N / 2 cancels in fp32 and would only inflate the bar)
        gmax = max(np.max(np.abs(d64)), 1e-12)
        bar = max(GRAD_FLOOR, 1.5 * kink_sens(lambda t: oracle.c_affine_loss_grad(m64, tgt64, t, oracle.wts(w_mse=1.0), t64)[2], tu, d64) / gmax)
        out.check(f"{nd}d lattice warp backward", np.max(np.abs(dth[b] - d64)) / gmax, bar, tag, GRAD_FLOOR)


class Tally:
    def __init__(self, verbose):
        self.worst, self.fails, self.verbose = {}, 0, verbose
        self.only_wide = 0   # comparisons whose error is above the stated floor and passes only because its bar was widened

    def check(self, name, err, bar, tag, floor=None):
        r = float(err) / float(bar) if bar > 0 else float(err)
        if floor is not None and float(floor) < float(err) <= float(bar):
            self.only_wide += 1
        self.worst[name] = max(self.worst.get(name, 0.0), r)
        if not (r <= 1.0):
            self.fails += 1
            if self.verbose:
                print(f"FAIL {name}: error {err:.3e} bar {bar:.3e}  case {tag}")


def run(n, seed, verbose=True, only=None):
    rng = np.random.default_rng(seed)
    out = Tally(verbose)
    for it in range(n):
        if only is not None and it > only:
            break
        if only is not None and it == only:
            os.environ["FUZZ_DEBUG"] = "1"
        # (every case draws from the shared generator, so earlier cases are re-run to reach case `only`)
        (steps_2d, multichannel_warp, trajectory, flow_trajectory, kde_pdf, lattice_warp)[it % 6](rng, it, out)
    if verbose:
        print(f"{n} cases, {out.fails} failures; worst error / bar: " + ", ".join(f"{k} {v:.2f}" for k, v in sorted(out.worst.items())) +
              f"; {out.only_wide} comparisons passed ONLY through a widened bar (error above the stated floor, below the widened bar)")
    return out.fails, out.worst


if __name__ == "__main__":
    f, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 90, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
               only=int(sys.argv[3]) if len(sys.argv) > 3 else None)   # third argument: stop after that case, with per-iteration detail
    sys.exit(1 if f else 0)

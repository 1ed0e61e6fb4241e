#!/usr/bin/env python3
"""Generate the golden fixtures by IMPORTING the reference (build container only).

    python tests/golden/make_golden.py          # writes tests/golden/*.npz

The reference (/root/reference, AgamChopra/TorchRegister v0.2.3) never travels to the GPU
box and none of its text is stored: the fixtures hold only its OUTPUTS (fp32, plus an fp64
re-run of the same reference functions as arbiter).  Inputs are closed-form
(tests/phantoms.py) and are re-created at test time.

Entry points exercised (ref: = /root/reference/src/TorchRegister/):
  ref:warpings.py:18-26   get_affine_warp          (single-step cases, composed loops)
  ref:utils.py:186-221    NCCLoss / SSDLoss        (single-step cases)
  ref:utils.py:280-330    Theta / Regressor        (KAT C, rigid trajectories)
  ref:utils.py:333-365    SpatialTransformer       (flow cases)
  ref:warpings.py:30-174  affine_register / rigid_register (trajectories, [final,best])
  ref:torchregister.py    Register.optim/__call__  (trajectories through the public API)
The loss curve is captured with the plt.plot hook described in SURVEY.md Q12.
"""
import contextlib
import io
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF = "/root/reference/src"
sys.path[:0] = [REF, os.path.join(REF, "TorchRegister")]
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import warnings  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

warnings.filterwarnings("ignore")
import TorchRegister as tr  # noqa: E402
import utils as rutils  # noqa: E402
import warpings  # noqa: E402

import phantoms as ph  # noqa: E402

torch.set_num_threads(8)


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        yield


class CurveHook:
    """Replace warpings.plt.plot; the last call receives the complete loss list (Q12)."""

    def __init__(self):
        self.curve = None

    def __call__(self, *args, **kw):
        self.curve = list(args[0])
        return []


def npf(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------- single step
def affine_single(out, name, shape, theta):
    for dt, tag in ((torch.float32, "32"), (torch.float64, "64")):
        mov = ph.vol(shape, 0.37, "sin", torch.float64).to(dt)
        tgt = ph.vol(shape, 0.23, "cos", torch.float64).to(dt)
        if dt == torch.float32:  # inputs are the fp32-rounded values in both precisions
            mov32, tgt32 = mov, tgt
        else:
            mov, tgt = mov32.double(), tgt32.double()
        th = torch.tensor(theta, dtype=dt)[None].requires_grad_()
        w = tr.get_affine_warp(th, mov)
        out[f"{name}/warped{tag}"] = npf(w)
        for lname, crit in (("ncc", tr.NCCLoss()), ("mse", nn.MSELoss()), ("ssd", tr.SSDLoss())):
            th.grad = None
            w = tr.get_affine_warp(th, mov)
            e = crit(tgt, w)
            e.backward()
            out[f"{name}/{lname}{tag}"] = npf(e)
            out[f"{name}/d{lname}{tag}"] = npf(th.grad[0])
    out[f"{name}/theta"] = np.asarray(theta, dtype=np.float64)
    out[f"{name}/shape"] = np.asarray(shape)


def flow_single(out, name, shape, amp):
    for dt, tag in ((torch.float32, "32"), (torch.float64, "64")):
        mov = ph.vol(shape, 0.37, "sin").to(dt)
        tgt = ph.vol(shape, 0.23, "cos").to(dt)
        st = rutils.SpatialTransformer(shape)
        if dt == torch.float64:
            st = st.double()
        fl = ph.flow_field(shape, amp).to(dt).requires_grad_()
        w = st(mov, fl)
        out[f"{name}/warped{tag}"] = npf(w)
        for lname, crit in (("ncc", tr.NCCLoss()), ("mse", nn.MSELoss())):
            fl.grad = None
            e = crit(tgt, st(mov, fl))
            e.backward()
            out[f"{name}/{lname}{tag}"] = npf(e)
            out[f"{name}/d{lname}{tag}"] = npf(fl.grad)
    out[f"{name}/amp"] = np.float64(amp)
    out[f"{name}/shape"] = np.asarray(shape)


def theta_single(out):
    th = rutils.Theta()
    for name, x in (("theta3", [0.3, -0.2, 0.5, 0.1, -0.4, 0.25]), ("theta2", [0.3, -0.2, 0.5])):
        for dt, tag in ((torch.float32, "32"), (torch.float64, "64")):
            xv = torch.tensor(x, dtype=dt, requires_grad=True)
            o = th(xv)
            out[f"{name}/out{tag}"] = npf(o)
            g = torch.arange(1, o.numel() + 1, dtype=dt)
            (o * g).sum().backward()
            out[f"{name}/jtv{tag}"] = npf(xv.grad)
        out[f"{name}/x"] = np.asarray(x)


# ----------------------------------------------------------------------------- trajectories
def run_driver(fn, mov, tgt, lr, epochs, per, seed):
    """Call ref affine_register / rigid_register directly: returns [final,best] warped + theta."""
    hook = CurveHook()
    old = warpings.plt.plot
    warpings.plt.plot = hook
    try:
        torch.manual_seed(seed)
        with quiet():
            warped, theta = fn(mov, tgt, lr=lr, epochs=epochs, per=per, device="cpu", debug=True,
                               criterions=[nn.MSELoss()], weights=[1.0], grad_edges=False)
    finally:
        warpings.plt.plot = old
    return warped, theta, hook.curve


def run_register(mode, mov, tgt, lr, epochs, per, seed, criterion, weight):
    hook = CurveHook()
    old = warpings.plt.plot
    warpings.plt.plot = hook
    try:
        torch.manual_seed(seed)
        reg = tr.Register(mode=mode, device="cpu", criterion=criterion, weight=weight, debug=True)
        with quiet():
            reg.optim(mov, tgt, lr=lr, max_epochs=epochs, per=per)
            w = reg(torch.cat([mov, 0.5 * mov + 0.25], dim=1))
    finally:
        warpings.plt.plot = old
    return reg, w, hook.curve


def seeded_pose(seed, n):
    torch.manual_seed(seed)
    return torch.rand(n)



def compose_loop(mov32, tgt32, loss, lr, epochs, dt, pose0=None):
    """SGD loop built only from the reference's public pieces (get_affine_warp, NCCLoss, Theta)."""
    nd = mov32.dim() - 2
    mov, tgt = mov32.to(dt), tgt32.to(dt)
    crit = tr.NCCLoss() if loss == "ncc" else nn.MSELoss()
    if pose0 is not None:
        p = pose0.clone().to(dt).requires_grad_()
        th_mod = rutils.Theta()
        make = lambda: th_mod(p).view(1, nd, nd + 1)  # noqa: E731
    else:
        p = torch.eye(nd, nd + 1, dtype=dt)[None].clone().requires_grad_()
        make = lambda: p  # noqa: E731
    opt = torch.optim.SGD([p], lr)
    losses, thetas = [], []
    for _ in range(epochs):
        opt.zero_grad()
        th = make()
        thetas.append(npf(th)[0].copy())
        e = crit(tgt, tr.get_affine_warp(th, mov))
        e.backward()
        opt.step()
        losses.append(e.item())
    thetas.append(npf(make())[0].copy())
    return np.asarray(losses, dtype=np.float64), np.asarray(thetas), npf(tr.get_affine_warp(make(), mov))

def traj_driver(out, name, mode, shape, lr, epochs, seed, per=0.125):
    nd = len(shape)
    tgt = ph.blobs(shape, 1000 + seed)
    star = ph.THETA_STAR3 if nd == 3 else ph.THETA_STAR2
    mov = tr.get_affine_warp(torch.tensor(star)[None], tgt).detach()
    fn = warpings.rigid_register if mode == "rigid" else warpings.affine_register
    warped, theta, curve = run_driver(fn, mov, tgt, lr, epochs, per, seed)
    reg, wcall, curve2 = run_register(mode, mov, tgt, lr, epochs, per, seed, [nn.MSELoss()], [1.0])
    assert np.array_equal(np.float64(curve), np.float64(curve2)), name
    assert torch.equal(reg.theta, theta[1]), name
    out[f"{name}/losses"] = np.asarray(curve, dtype=np.float64)
    out[f"{name}/final_theta"] = npf(theta[0])
    out[f"{name}/best_theta"] = npf(theta[1])
    out[f"{name}/best_warped"] = npf(warped[1])
    out[f"{name}/final_warped"] = npf(warped[0])
    out[f"{name}/call2c"] = npf(wcall)
    out[f"{name}/moving"] = npf(mov)  # moving itself is a reference output (warp of target)
    pose0 = seeded_pose(seed, 6 if nd == 3 else 3) if mode == "rigid" else None
    if mode == "rigid":
        out[f"{name}/init"] = npf(pose0)
    # fp64 arbiter = same loop composed from the reference's public pieces; its fp32 run
    # must reproduce the driver exactly (proves the composition IS the driver, Q3).
    l32, t32, _ = compose_loop(mov, tgt, "mse", lr, epochs, torch.float32, pose0)
    assert np.array_equal(l32, np.asarray(curve, dtype=np.float64)), name
    assert np.array_equal(t32[-1], npf(theta[0])[0]), name
    l64, t64, w64 = compose_loop(mov, tgt, "mse", lr, epochs, torch.float64, pose0)
    out[f"{name}/losses64"], out[f"{name}/thetas64"], out[f"{name}/final_warped64"] = l64, t64, w64
    out[f"{name}/meta"] = np.asarray([lr, epochs, seed, per], dtype=np.float64)
    out[f"{name}/shape"] = np.asarray(shape)
    print(f"  {name}: loss {curve[0]:.6f} -> {curve[-1]:.6f} best {min(curve):.6f}")


def traj_default_crit(out, name, shape, weight, lr, epochs, seed):
    """affine, criterion=None (=> MSE, NCC, NMI with user weights; Q2); 2D only (Q5)."""
    tgt = ph.blobs(shape, 1000 + seed)
    mov = tr.get_affine_warp(torch.tensor(ph.THETA_STAR2)[None], tgt).detach()
    reg, wcall, curve = run_register("affine", mov, tgt, lr, epochs, 0.125, seed, None, weight)
    out[f"{name}/losses"] = np.asarray(curve, dtype=np.float64)
    out[f"{name}/best_theta"] = npf(reg.theta)
    out[f"{name}/call2c"] = npf(wcall)
    out[f"{name}/moving"] = npf(mov)
    out[f"{name}/meta"] = np.asarray([lr, epochs, seed, 0.125] + list(weight), dtype=np.float64)
    out[f"{name}/shape"] = np.asarray(shape)
    print(f"  {name}: loss {curve[0]:.6f} -> {curve[-1]:.6f}")


def composed_affine(out, name, shape, loss, lr, epochs, seed, rigid=False):
    """Loops the Register API cannot reach (Q2): reference public pieces + torch SGD."""
    nd = len(shape)
    tgt32 = ph.blobs(shape, 1000 + seed)
    star = ph.THETA_STAR3 if nd == 3 else ph.THETA_STAR2
    mov32 = tr.get_affine_warp(torch.tensor(star)[None], tgt32).detach()
    out[f"{name}/moving"] = npf(mov32)
    pose0 = seeded_pose(seed, 6 if nd == 3 else 3) if rigid else None
    if rigid:
        out[f"{name}/init"] = npf(pose0)
    for dt, tag in ((torch.float32, "32"), (torch.float64, "64")):
        ls, ths, w = compose_loop(mov32, tgt32, loss, lr, epochs, dt, pose0)
        out[f"{name}/losses{tag}"], out[f"{name}/thetas{tag}"], out[f"{name}/final_warped{tag}"] = ls, ths, w
    out[f"{name}/meta"] = np.asarray([lr, epochs, seed], dtype=np.float64)
    out[f"{name}/shape"] = np.asarray(shape)
    l32, l64 = out[f"{name}/losses32"], out[f"{name}/losses64"]
    print(f"  {name}: loss {l32[0]:.6f} -> {l32[-1]:.6f}; max|l32-l64| {np.abs(l32 - l64).max():.2e}; step0 gap {abs(l32[0]-l64[0]):.2e}")


def composed_flow(out, name, shape, loss, lr, epochs, seed):
    nd = len(shape)
    tgt32 = ph.blobs(shape, 1000 + seed)
    star = ph.THETA_STAR3 if nd == 3 else ph.THETA_STAR2
    mov32 = tr.get_affine_warp(torch.tensor(star)[None], tgt32).detach()
    out[f"{name}/moving"] = npf(mov32)
    for dt, tag in ((torch.float32, "32"), (torch.float64, "64")):
        mov, tgt = mov32.to(dt), tgt32.to(dt)
        st = rutils.SpatialTransformer(shape)
        if dt == torch.float64:
            st = st.double()
        crit = tr.NCCLoss() if loss == "ncc" else nn.MSELoss()
        fl = torch.zeros(1, nd, *shape, dtype=dt, requires_grad=True)
        opt = torch.optim.SGD([fl], lr)
        losses = []
        for _ in range(epochs):
            opt.zero_grad()
            e = crit(tgt, st(mov, fl))
            e.backward()
            opt.step()
            losses.append(e.item())
        out[f"{name}/losses{tag}"] = np.asarray(losses, dtype=np.float64)
        out[f"{name}/flow{tag}"] = npf(fl)
        out[f"{name}/final_warped{tag}"] = npf(st(mov, fl))
    out[f"{name}/meta"] = np.asarray([lr, epochs, seed], dtype=np.float64)
    out[f"{name}/shape"] = np.asarray(shape)
    l32, l64 = out[f"{name}/losses32"], out[f"{name}/losses64"]
    print(f"  {name}: loss {l32[0]:.6f} -> {l32[-1]:.6f}; max|l32-l64| {np.abs(l32 - l64).max():.2e}")


def flow_deform_semantics(out):
    """ref:warpings.py:238-242 deform + ref:torchregister.py:123-126 per-channel __call__."""
    shape = (6, 7, 8)
    fr = warpings.flow_register.__new__(warpings.flow_register)
    nn.Module.__init__(fr)
    fr.warp = rutils.SpatialTransformer(shape)
    fr.flow = ph.flow_field(shape, 1.1, 0.07)
    reg = tr.Register(mode="flow")
    reg.theta, reg.warp = fr.flow, fr.deform
    x = torch.cat([ph.vol(shape, 0.37, "sin"), ph.vol(shape, 0.23, "cos")], dim=1)
    out["deform/call2c"] = npf(reg(x))
    out["deform/shape"] = np.asarray(shape)


def flow_unet(out, name, shape, crit_names, weights, lr, epochs, seed):
    """The reference's own flow mode (attention U-Net generates the flow; ref:warpings.py:178-242,
    ref:utils.py:409-559) through Register, n=32.  The flow and warp are stored on a stride-4 lattice."""
    nd = len(shape)
    tgt = ph.blobs(shape, 1000 + seed)
    star = ph.THETA_STAR3 if nd == 3 else ph.THETA_STAR2
    mov = tr.get_affine_warp(torch.tensor(star)[None], tgt).detach()
    crits = [{"ncc": tr.NCCLoss(), "mse": nn.MSELoss()}[c] for c in crit_names]
    hook = CurveHook()
    old = warpings.plt.plot
    warpings.plt.plot = hook
    try:
        torch.manual_seed(seed)
        reg = tr.Register(mode="flow", device="cpu", criterion=crits, weight=weights, debug=True)
        with quiet():
            reg.optim(mov, tgt, lr=lr, max_epochs=epochs, n=32)
            w = reg(torch.cat([mov, 0.5 * mov + 0.25], dim=1))
    finally:
        warpings.plt.plot = old
    sl = (slice(None), slice(None)) + (slice(None, None, 4),) * nd
    out[f"{name}/losses"] = np.asarray(hook.curve, dtype=np.float64)
    out[f"{name}/flow_s4"] = npf(reg.theta)[sl]
    out[f"{name}/call2c_s4"] = npf(w)[sl]
    out[f"{name}/flow_absmax"] = np.float64(npf(reg.theta).__abs__().max())
    out[f"{name}/moving"] = npf(mov)
    out[f"{name}/meta"] = np.asarray([lr, epochs, seed] + list(weights), dtype=np.float64)
    out[f"{name}/shape"] = np.asarray(shape)
    print(f"  {name}: loss {hook.curve[0]:.6f} -> {hook.curve[-1]:.6f}; |flow|max {out[f'{name}/flow_absmax']:.3f}")


def check_kats(ss, tj):
    """SURVEY §8c KAT values (produced by the same import in the survey session)."""
    def close(a, b, tol=2e-6):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        assert np.allclose(a, b, rtol=tol, atol=tol), (a, b)
    close(ss["A3/warped32"].sum(dtype=np.float64), 3.05003738, 1e-5)
    close(ss["A3/warped32"][0, 0, 2, 3, 4], 0.35492158)
    close(ss["A3/ncc32"], 105.54127502)
    close(ss["A3/mse32"], 0.77542049)
    close(ss["A3/ssd32"], 488.514893)
    close(ss["A3/dncc32"][1], [-4.41250515, 71.93289948, -2.20623469, -70.22976685], 1e-5)
    close(ss["A3/dmse32"][2], [-0.76303124, -0.02561814, 0.70878500, 0.63431865], 1e-5)
    close(ss["B2/warped32"].sum(dtype=np.float64), 5.07647371, 1e-5)
    close(ss["B2/mse32"], 0.83573323)
    close(ss["B2/ncc32"], 105.32128906)
    close(ss["B2/dncc32"], [[-7.66944885, 13.65703583, -9.27043343],
                           [32.23183060, 95.43234253, -121.40481567]], 1e-5)
    close(ss["theta3/out32"][:4], [0.93629342, 0.31320453, 0.15892664, 0.02491700])
    close(ss["theta3/jtv32"], [-11.21278095, 7.76327038, -6.22798157, 0.99006629, 1.71127760, 2.82004452], 1e-5)
    close(ss["theta2/out32"], [0.95533651, -0.29552022, -0.2, 0.29552022, 0.95533651, 0.5])
    close(ss["D3/warped32"].sum(dtype=np.float64), 5.62683439, 1e-5)
    close(ss["D3/ncc32"], 94.50695038)
    close(ss["D3/dncc32"].sum(dtype=np.float64), -14.63764477, 1e-4)
    close(ss["D3/dncc32"][0, :, 2, 3, 4], [-1.96595061, -0.50358868, 0.00286643], 1e-5)
    close(ss["D2/warped32"].sum(dtype=np.float64), -2.16666865, 1e-5)
    close(tj["katF_rigid/best_theta"][0], [[0.87562358, -0.48299414, 0.76699811],
                                           [0.48299414, 0.87562358, 0.07750706]], 1e-5)
    close(tj["katF_affine/best_theta"][0][0], [1.00475943, 0.00083895, 0.00022245, 0.00952367], 1e-5)
    print("KATs OK")


def kat_f(out):
    mov2, tgt2 = ph.vol((6, 7), 0.37, "sin"), ph.vol((6, 7), 0.23, "cos")
    reg, _, curve = run_register("rigid", mov2, tgt2, 1e-2, 5, 0.1, 0, [nn.MSELoss()], [1.0])
    out["katF_rigid/best_theta"] = npf(reg.theta)
    out["katF_rigid/losses"] = np.asarray(curve, dtype=np.float64)
    out["katF_rigid/init"] = npf(seeded_pose(0, 3))
    mov3, tgt3 = ph.vol((8, 8, 8), 0.37, "sin"), ph.vol((8, 8, 8), 0.23, "cos")
    reg, _, curve = run_register("affine", mov3, tgt3, 1e-2, 5, 0.125, 0, [nn.MSELoss()], [1.0])
    out["katF_affine/best_theta"] = npf(reg.theta)
    out["katF_affine/losses"] = np.asarray(curve, dtype=np.float64)


def main():
    ss = {}
    print("single-step cases")
    affine_single(ss, "A3", (5, 6, 7), ph.THETA_A)
    affine_single(ss, "B2", (6, 7), ph.THETA_B)
    affine_single(ss, "OOB3", (16, 16, 16), ph.THETA_OOB3)
    affine_single(ss, "ROT3", (24, 20, 28), ph.THETA_ROT3)
    affine_single(ss, "OOB2", (32, 32), ph.THETA_OOB2)
    affine_single(ss, "ID3", (9, 8, 10), [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]])
    affine_single(ss, "ID2", (11, 13), [[1, 0, 0], [0, 1, 0]])
    flow_single(ss, "D3", (5, 6, 7), 0.8)
    flow_single(ss, "D2", (6, 7), 0.8)
    flow_single(ss, "F3", (12, 10, 14), 3.0)
    flow_single(ss, "F2", (20, 24), 2.5)
    theta_single(ss)
    flow_deform_semantics(ss)
    ss["ncc_self"] = npf(tr.NCCLoss()(ph.vol((5, 6, 7), 0.37), ph.vol((5, 6, 7), 0.37)))
    ss["norm124"] = npf(tr.norm(torch.tensor([1.0, 2.0, 4.0])))
    ss["nmi2d"] = npf(tr.NMILoss()(ph.blobs((32, 32), 1), ph.blobs((32, 32), 2)))

    tj = {}
    print("trajectories")
    kat_f(tj)
    traj_driver(tj, "rigid2d_mse", "rigid", (64, 64), 1.0, 100, 0)
    traj_driver(tj, "rigid3d_mse", "rigid", (24, 24, 24), 5.0, 60, 1)
    traj_driver(tj, "affine3d_mse", "affine", (16, 16, 16), 1e-1, 60, 2)
    traj_driver(tj, "affine2d_mse", "affine", (32, 32), 1e-1, 60, 3)
    traj_default_crit(tj, "affine2d_w010", (32, 32), [0.0, 1.0, 0.0], 1e-4, 40, 4)
    traj_default_crit(tj, "affine2d_w550", (32, 32), [0.5, 0.5, 0.0], 1e-4, 40, 5)
    traj_default_crit(tj, "affine2d_default", (32, 32), [0.33, 0.33, 0.33], 1e-5, 12, 14)   # NMI active (Parzen KDE)
    composed_affine(tj, "c_affine3d_ncc", (24, 24, 24), "ncc", 3e-5, 40, 6)
    composed_affine(tj, "c_affine2d_ncc", (48, 40), "ncc", 1e-4, 40, 7)
    composed_affine(tj, "c_rigid3d_ncc", (20, 24, 28), "ncc", 2e-4, 40, 8, rigid=True)
    composed_flow(tj, "c_flow3d_ncc", (12, 14, 16), "ncc", 2.0, 30, 9)
    composed_flow(tj, "c_flow3d_mse", (12, 14, 16), "mse", 2000.0, 30, 10)
    composed_flow(tj, "c_flow2d_ncc", (24, 28), "ncc", 1.0, 30, 11)
    flow_unet(tj, "unet2d_ncc", (160, 160), ["ncc"], [1.0], 1e-3, 8, 12)
    flow_unet(tj, "unet2d_mix", (156, 172), ["mse", "ncc"], [0.5, 0.5], 1e-3, 6, 13)

    check_kats(ss, tj)
    np.savez_compressed(os.path.join(HERE, "single_step.npz"), **ss)
    np.savez_compressed(os.path.join(HERE, "trajectories.npz"), **tj)
    for f in ("single_step.npz", "trajectories.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()

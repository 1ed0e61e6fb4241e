"""GPU parity of the local-window NCC extension (trx_lncc_loss_grad) against its specification, the torch
composition oracle/compose.py::local_ncc_loss (conv box filters + autograd).  The reference has no local NCC
("parity unpinned"), so the torch composition is the arbiter: fp64 on the CPU, with the fp32 run of the same
composition giving the bar = max(floor, 2 x |ref32 - ref64|).  Floors: loss 2e-5 rel, gradient 2e-4 of max."""
import numpy as np
import pytest
import torch

import phantoms as ph
from conftest import bar
from oracle import compose

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torchregister_amd._engine as e
    assert torch.cuda.is_available()
    return e


def _ref(tgt, wrp, window, alpha, dtype):
    y = tgt.to(dtype)
    w = wrp.to(dtype).clone().requires_grad_()
    losses, grads = [], []
    for b in range(y.shape[0]):      # per pair: the kernel returns one loss per pair
        l = compose.local_ncc_loss(y[b:b + 1], w[b:b + 1], window, alpha)
        (g,) = torch.autograd.grad(l, w)
        losses.append(l.item())
        grads.append(g[b:b + 1].numpy())
    return np.asarray(losses), np.concatenate(grads)


def _pair(shape, B=1):
    nd = len(shape)
    tgt = torch.cat([ph.blobs(shape, 300 + b) + 0.05 * ph.vol(shape, 0.37 + 0.01 * b, "sin") for b in range(B)])
    if nd == 3:
        wrp = torch.cat([compose.affine_warp(torch.tensor(ph.THETA_STAR3)[None], tgt[b:b + 1]) for b in range(B)])
    else:
        wrp = torch.cat([compose.affine_warp(torch.tensor(ph.THETA_STAR2)[None], tgt[b:b + 1]) for b in range(B)])
    return tgt, wrp + 0.02 * ph.vol(tuple(wrp.shape[2:]), 0.23, "cos")


CASES = [((70, 20, 36), 9, 1), ((97, 12, 33), 5, 2), ((20, 18, 44), 9, 1), ((9, 33, 36), 9, 2), ((12, 40, 70), 5, 1), ((7, 7, 7), 9, 1), ((16, 16, 32), 3, 1), ((24, 40, 33), 7, 1),
         ((40, 52), 9, 2), ((33, 70), 5, 1), ((8, 8), 9, 1)]


@pytest.mark.parametrize("shape,window,B", CASES)
def test_lncc_loss_and_grad_vs_torch(eng, shape, window, B):
    tgt, wrp = _pair(shape, B)
    alpha = 2.5
    l32, g32 = _ref(tgt, wrp, window, alpha, torch.float32)
    l64, g64 = _ref(tgt, wrp, window, alpha, torch.float64)
    loss, grad = eng.local_ncc_loss_grad(tgt.cuda(), wrp.cuda(), window, alpha)
    loss, grad = loss.cpu().numpy(), grad.cpu().numpy()
    assert np.max(np.abs(loss - l32)) <= bar(l32, l64, 2e-5 * np.max(np.abs(l64)))
    assert np.max(np.abs(grad - g32)) <= bar(g32, g64, 2e-4 * np.max(np.abs(g64)))


def test_lncc_identical_images_and_loss_only(eng):
    """cc = 1 wherever the window has variance: loss ~ alpha * (fraction of flat windows); no gradient requested."""
    x = ph.blobs((16, 20, 36), 5).cuda() + 0.1 * ph.vol((16, 20, 36), 0.37, "sin").cuda()
    loss, grad = eng.local_ncc_loss_grad(x, x, 9, 1.0, need_grad=False)
    assert grad is None
    ref = compose.local_ncc_loss(x.cpu().double(), x.cpu().double()).item()
    assert abs(loss.item() - ref) <= 1e-4


def test_local_ncc_criterion_autograd(eng):
    """tr.LocalNCCLoss as a criterion: batch mean, gradient flows to the warped image only."""
    import torchregister_amd as tr
    tgt, wrp = _pair((12, 16, 40), 2)
    crit = tr.LocalNCCLoss(window=5, alpha=3.0)
    w = wrp.cuda().requires_grad_()
    loss = crit(tgt.cuda(), w)
    loss.backward()
    w64 = wrp.double().requires_grad_()
    ref = compose.local_ncc_loss(tgt.double(), w64, 5, 3.0)
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 2e-5 * abs(ref.item()) + 1e-6
    assert torch.max(torch.abs(w.grad.cpu().double() - w64.grad)).item() <= 2e-4 * w64.grad.abs().max().item()
    with pytest.raises(Exception):
        crit(tgt, wrp)          # CPU tensors: no fallback


def test_register_affine_with_local_ncc(eng):
    """Register(mode='affine') driven by the local NCC through the generic path (HIP warp + HIP criterion + SGD on theta)
    against the same loop composed from torch CPU ops in fp64: loss curve to 1e-4 rel, final theta to 1e-4."""
    import torchregister_amd as tr
    shape = (24, 28, 32)
    tgt = ph.blobs(shape, 41)
    mov = compose.affine_warp(torch.tensor(ph.THETA_STAR3)[None], tgt)
    lr, iters = 5e-2, 30
    reg = tr.Register(mode="affine", device="cuda", criterion=[tr.LocalNCCLoss(window=9)], weight=[1.0], honor_criterion=True)
    reg.optim(mov.cuda(), tgt.cuda(), lr=lr, max_epochs=iters)
    losses = np.asarray(reg.losses, dtype=np.float64)

    def cpu_loop(dtype):
        th = torch.eye(3, 4, dtype=dtype)[None].clone().requires_grad_()
        opt = torch.optim.SGD([th], lr)
        out = []
        for _ in range(iters):
            opt.zero_grad()
            e = compose.local_ncc_loss(tgt.to(dtype), compose.affine_warp(th, mov.to(dtype)))
            e.backward()
            opt.step()
            out.append(e.item())
        return np.asarray(out), th.detach().numpy()[0]

    l64, th64 = cpu_loop(torch.float64)
    l32, th32 = cpu_loop(torch.float32)
    assert losses[-1] < losses[0]
    assert np.max(np.abs(losses - l32)) <= bar(l32, l64, 1e-4 * np.max(np.abs(l64)))
    assert np.max(np.abs(reg.final_theta.cpu().numpy().reshape(3, 4) - th32)) <= bar(th32, th64, 1e-4)


def test_lncc_full_size_properties(eng):
    """256^3: deterministic, loss in [0, alpha], gradient finite; scaling the warped image leaves cc (hence the loss) unchanged."""
    shape = (256, 256, 256)
    tgt = ph.blobs(shape, 1000).cuda()
    wrp = ph.blobs(shape, 1001).cuda()
    l1, g1 = eng.local_ncc_loss_grad(tgt, wrp, 9, 1.0)
    l2, g2 = eng.local_ncc_loss_grad(tgt, wrp, 9, 1.0)
    assert torch.equal(l1, l2) and torch.equal(g1, g2)
    assert 0.0 <= l1.item() <= 1.0 and torch.isfinite(g1).all()


# ----------------------------------------------------------------------------- round 3: the local-NCC loop on the device
def _torch_lncc_flow_loop(mov, tgt, lr, iters, optimizer, smooth, window, flow0, dtype):
    """arbiter: oracle/compose.py (local_ncc_loss, flow_warp, smooth_regulariser) under torch autograd + torch.optim, on the CPU"""
    from oracle import compose
    mov, tgt = mov.to(dtype), tgt.to(dtype)
    fl = flow0.to(dtype).clone().requires_grad_()
    opt = torch.optim.SGD([fl], lr) if optimizer == "sgd" else torch.optim.Adam([fl], lr)
    losses = []
    for _ in range(iters):
        opt.zero_grad()
        e = compose.local_ncc_loss(tgt, compose.flow_warp(mov, fl), window=window)
        if smooth:
            e = e + compose.smooth_regulariser(fl, smooth)
        e.backward()
        opt.step()
        losses.append(e.item())
    return np.asarray(losses), fl.detach().numpy()


@pytest.mark.parametrize("optimizer,lr,smooth,window", [("sgd", 30.0, 0.0, 5), ("adam", 0.05, 2.0, 9), ("adam", 0.05, 0.0, 7), ("sgd", 20.0, 3.0, 9)])
def test_lncc_flow_loop_vs_torch_autograd(optimizer, lr, smooth, window):
    """trx_flow_lncc_run (warp -> window sums -> gradient -> smoothness -> SGD / Adam, all on the device) against the same objective
    under torch autograd in fp64; bar = max(floor, 2 x the arbiter's own fp32-vs-fp64 gap).  Starts off the voxel lattice."""
    import torchregister_amd._engine as eng
    from conftest import bar
    shape, iters = (20, 24, 28), 12
    tgt = ph.blobs(shape, 61) + 0.05 * ph.vol(shape, 0.031, "sin")
    mov = ph.blobs(shape, 62) + 0.05 * ph.vol(shape, 0.027, "cos")
    ax = [torch.arange(n, dtype=torch.float64) for n in shape]
    comp = lambda a, b, c: (torch.sin(a * ax[0])[:, None, None] + torch.cos(b * ax[1])[None, :, None] + torch.sin(c * ax[2] + 0.4)[None, None, :])  # noqa: E731
    f0 = torch.stack([0.4 * comp(0.21, 0.17, 0.13), 0.3 * comp(0.11, 0.23, 0.19), 0.35 * comp(0.15, 0.12, 0.27)]).float()[None] + 0.17
    l32, f32 = _torch_lncc_flow_loop(mov, tgt, lr, iters, optimizer, smooth, window, f0, torch.float32)
    l64, f64 = _torch_lncc_flow_loop(mov, tgt, lr, iters, optimizer, smooth, window, f0, torch.float64)
    s = eng.FlowSolver(mov.cuda(), tgt.cuda(), optimizer=optimizer, lr=lr, init=f0, capacity=iters, smooth_weight=smooth, lncc=dict(window=window))
    s.run(iters)
    torch.cuda.synchronize()
    e, b = np.max(np.abs(s.losses[0].cpu().numpy() - l64)), bar(l32, l64, 2e-5 * np.max(np.abs(l64)))
    assert e <= b, ("loss curve", e, b)
    e, b = np.max(np.abs(s.flow.cpu().numpy() - f64)), bar(f32, f64, 2e-4)
    assert e <= b, ("flow", e, b)
    assert l64[-1] < l64[0]


def test_lncc_flow_loop_through_flow_register_and_batch_independence():
    """flow_register(flow_model='direct', criterions=[LocalNCCLoss]) takes the device-side loop (no torch optimiser): same result as the
    solver, and the pairs of a batch are independent registrations (pair 1 of a batch == the same pair alone, bit for bit)."""
    import torchregister_amd as tr
    shape, iters = (16, 24, 32), 6
    tgt = torch.cat([ph.blobs(shape, 71), ph.blobs(shape, 72)]).cuda()
    mov = torch.cat([ph.blobs(shape, 73), ph.blobs(shape, 74)]).cuda()
    reg = tr.flow_register(shape, criterions=[tr.LocalNCCLoss(window=5, alpha=2.0)], weights=[0.5], lr=0.05, max_epochs=iters, stop_crit=-1.0,
                           flow_model="direct", optimizer="adam", smooth_weight=1.0)
    reg.optimize(mov, tgt, debug=False)
    assert reg.solver.lncc == (5, 1.0, 1e-5) and reg.losses.shape == (2, iters)
    solo = tr.FlowSolver(mov[1:], tgt[1:], optimizer="adam", lr=0.05, capacity=iters, smooth_weight=1.0, lncc=dict(window=5, alpha=1.0))
    solo.run(iters)
    torch.cuda.synchronize()
    assert torch.equal(reg.final_flow[1], solo.flow[0]) and torch.equal(reg.losses[1], solo.losses[0, :iters])
    assert torch.isfinite(reg.losses).all() and (reg.losses[:, -1] < reg.losses[:, 0]).all()

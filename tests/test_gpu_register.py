"""GPU parity of the public API: tr.Register(mode).optim(...) / reg(x) vs the reference's own runs
(golden fixtures produced by tests/golden/make_golden.py through the reference's Register)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import phantoms as ph
from conftest import bar

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tr():
    import TorchRegister as tr   # the drop-in import name (ref:README.md:26)
    assert torch.cuda.is_available()
    return tr


def _mov_tgt(g, name):
    shape = tuple(g[f"{name}/shape"])
    seed = int(g[f"{name}/meta"][2])
    return torch.from_numpy(g[f"{name}/moving"]).cuda(), ph.blobs(shape, 1000 + seed).cuda()


def test_kat_f_rigid_and_affine(tr, trajectories):
    """SURVEY §8c KAT F: 5 iterations through Register, rigid 2-D and affine 3-D."""
    g = trajectories
    mov2, tgt2 = ph.vol((6, 7), 0.37, "sin").cuda(), ph.vol((6, 7), 0.23, "cos").cuda()
    reg = tr.Register("rigid", device="cuda", criterion=[nn.MSELoss()], weight=[1.0], init=torch.from_numpy(g["katF_rigid/init"]))
    reg.optim(mov2, tgt2, lr=1e-2, max_epochs=5)
    assert np.allclose(reg.theta.cpu().numpy(), g["katF_rigid/best_theta"], atol=2e-6)
    assert np.allclose(reg.losses[0].cpu().numpy(), g["katF_rigid/losses"], rtol=1e-5)
    mov3, tgt3 = ph.vol((8, 8, 8), 0.37, "sin").cuda(), ph.vol((8, 8, 8), 0.23, "cos").cuda()
    reg = tr.Register("affine", device="cuda", criterion=[nn.MSELoss()], weight=[1.0])
    reg.optim(mov3, tgt3, lr=1e-2, max_epochs=5, per=0.125)
    assert np.allclose(reg.losses[0].cpu().numpy(), g["katF_affine/losses"], rtol=2e-4)
    assert np.allclose(reg.theta.cpu().numpy(), g["katF_affine/best_theta"], atol=2e-4)


@pytest.mark.parametrize("name,mode", [("rigid2d_mse", "rigid"), ("rigid3d_mse", "rigid"), ("affine3d_mse", "affine"), ("affine2d_mse", "affine")])
def test_register_trajectory_and_call(tr, trajectories, name, mode):
    g = trajectories
    lr, iters = float(g[f"{name}/meta"][0]), int(g[f"{name}/meta"][1])
    mov, tgt = _mov_tgt(g, name)
    init = torch.from_numpy(g[f"{name}/init"]) if mode == "rigid" else None
    # the user criterion is discarded by the reference (Q2): NCCLoss given, MSE optimised
    reg = tr.Register(mode, device="cuda", criterion=[tr.NCCLoss()], weight=[7.0], init=init)
    reg.optim(mov, tgt, lr=lr, max_epochs=iters, per=0.1)     # per=0.1 crashes the reference (Q4); ignored here
    l32, l64 = g[f"{name}/losses"], g[f"{name}/losses64"]
    assert np.max(np.abs(reg.losses[0].cpu().numpy() - l32)) <= bar(l32, l64, 1e-4 * np.max(np.abs(l64)))
    tbar = bar(g[f"{name}/final_theta"][0], g[f"{name}/thetas64"][-1], 1e-4)
    assert reg.theta.shape == (1,) + g[f"{name}/best_theta"].shape[1:]
    assert np.max(np.abs(reg.theta.cpu().numpy() - g[f"{name}/best_theta"])) <= tbar
    assert np.max(np.abs(reg.final_theta.cpu().numpy() - g[f"{name}/final_theta"])) <= tbar
    # __call__ on a 2-channel volume (ref:torchregister.py:123-128)
    x = torch.cat([mov, 0.5 * mov + 0.25], dim=1)
    w = reg(x).cpu().numpy()
    assert w.shape == g[f"{name}/call2c"].shape
    assert np.max(np.abs(w - g[f"{name}/call2c"])) <= max(1e-4, 4 * tbar)


@pytest.mark.parametrize("name", ["affine2d_w010", "affine2d_w550"])
def test_register_default_criterion_with_weights(tr, trajectories, name):
    """criterion=None -> [MSE, NCC, NMI] with the user's weights (NMI weight 0 here): fused MSE+NCC."""
    g = trajectories
    lr, iters = float(g[f"{name}/meta"][0]), int(g[f"{name}/meta"][1])
    weight = [float(v) for v in g[f"{name}/meta"][4:7]]
    mov, tgt = _mov_tgt(g, name)
    reg = tr.Register("affine", device="cuda", criterion=None, weight=weight)
    reg.optim(mov, tgt, lr=lr, max_epochs=iters, per=0.125)
    gl = g[f"{name}/losses"]
    # no fp64 arbiter for this fixture: stated tolerance = 5e-4 relative on the curve, 5e-4 abs on theta
    assert np.max(np.abs(reg.losses[0].cpu().numpy() - gl)) <= 5e-4 * np.max(np.abs(gl))
    assert np.max(np.abs(reg.theta.cpu().numpy() - g[f"{name}/best_theta"])) <= 5e-4
    x = torch.cat([mov, 0.5 * mov + 0.25], dim=1)
    assert np.max(np.abs(reg(x).cpu().numpy() - g[f"{name}/call2c"])) <= 2e-3


def test_register_generic_criterion_path(tr):
    """A criterion with no fused form (L1) drives the HIP warp through autograd; compare with the
    same loop on torch's CPU ops (oracle composition)."""
    from oracle import compose
    shape = (16, 16, 16)
    tgt = ph.blobs(shape, 1003)
    mov = compose.affine_warp(torch.tensor(ph.THETA_STAR3)[None], tgt)
    th = torch.eye(3, 4)[None].clone().requires_grad_()
    opt = torch.optim.SGD([th], 0.05)
    ref = []
    for _ in range(10):
        opt.zero_grad()
        e = nn.functional.l1_loss(compose.affine_warp(th, mov), tgt)
        e.backward(); opt.step(); ref.append(e.item())
    reg = tr.Register("affine", device="cuda", criterion=[nn.L1Loss()], weight=[1.0], honor_criterion=True)
    reg.optim(mov.cuda(), tgt.cuda(), lr=0.05, max_epochs=10)
    assert np.allclose(reg.losses.cpu().numpy().ravel(), ref, rtol=2e-3)
    assert np.allclose(reg.final_theta.cpu().numpy(), th.detach().numpy(), atol=2e-3)


@pytest.mark.parametrize("mode", ["affine", "rigid"])
def test_register_mixed_criterion_list(tr, mode):
    """A list that mixes fused terms (NCC, MSE) with one that has no fused form (L1): the fused terms come from one F1 launch
    (warpings._FusedLossFn), L1 from the generic autograd path, and the sum must follow the same loop on torch's CPU ops in fp64."""
    from oracle import compose
    shape = (18, 16, 20)
    tgt = ph.blobs(shape, 1005)
    mov = compose.affine_warp(torch.tensor(ph.THETA_STAR3)[None], tgt) + 0.02 * ph.blobs(shape, 1006)
    w = [0.4, 1.0, 0.7]
    pose0 = torch.tensor([0.02, -0.03, 0.01, 0.05, 0.0, -0.04])
    ref = {}
    for dt in (torch.float32, torch.float64):
        p = (pose0.to(dt).clone() if mode == "rigid" else torch.eye(3, 4, dtype=dt)[None].clone()).requires_grad_()
        opt = torch.optim.SGD([p], 2e-3)
        ls = []
        for _ in range(8):
            opt.zero_grad()
            th = compose.pose_to_theta(p) if mode == "rigid" else p
            y = compose.affine_warp(th, mov.to(dt))
            e = w[0] * compose.ncc_loss(tgt.to(dt), y) + w[1] * nn.functional.l1_loss(y, tgt.to(dt)) + w[2] * compose.mse_loss(tgt.to(dt), y)
            e.backward(); opt.step(); ls.append(e.item())
        ref[dt] = (np.asarray(ls), (compose.pose_to_theta(p) if mode == "rigid" else p).detach().double().numpy().reshape(3, 4))
    reg = tr.Register(mode, device="cuda", criterion=[tr.NCCLoss(), nn.L1Loss(), nn.MSELoss()], weight=w, honor_criterion=True,
                      init=pose0 if mode == "rigid" else None)
    reg.optim(mov.cuda(), tgt.cuda(), lr=2e-3, max_epochs=8)
    (l32, t32), (l64, t64) = ref[torch.float32], ref[torch.float64]
    got = reg.losses.cpu().numpy().ravel()
    assert np.max(np.abs(got - l64) / np.maximum(1.0, np.abs(l64))) <= max(1e-4, 2.0 * np.max(np.abs(l32 - l64) / np.maximum(1.0, np.abs(l64))))
    assert np.max(np.abs(reg.final_theta.cpu().numpy().reshape(3, 4) - t64)) <= max(1e-4, 2.0 * np.max(np.abs(t32 - t64)))


def test_get_affine_warp_autograd(tr, single_step):
    g = single_step
    mov, tgt = ph.vol((5, 6, 7), 0.37, "sin").cuda(), ph.vol((5, 6, 7), 0.23, "cos").cuda()
    th = torch.tensor(g["A3/theta"], dtype=torch.float32, device="cuda")[None].requires_grad_()
    e = tr.NCCLoss()(tgt, tr.get_affine_warp(th, mov))
    e.backward()
    assert abs(e.item() - float(g["A3/ncc32"])) <= 2e-4
    assert np.max(np.abs(th.grad[0].cpu().numpy() - g["A3/dncc32"])) <= bar(g["A3/dncc32"], g["A3/dncc64"], 1e-4 * np.abs(g["A3/dncc64"]).max())
    # flat theta form (ref:warpings.py:19-23)
    w2 = tr.get_affine_warp(th.detach().reshape(1, 12), mov)
    assert torch.equal(w2, tr.get_affine_warp(th.detach(), mov))


def test_register_flow_mode(tr, trajectories, single_step):
    """mode='flow' (direct flow field): loss curve / flow / deform vs the reference composition."""
    g = trajectories
    name = "c_flow3d_ncc"
    lr, iters = float(g[f"{name}/meta"][0]), int(g[f"{name}/meta"][1])
    mov, tgt = _mov_tgt(g, name)
    reg = tr.Register("flow", device="cuda", criterion=[tr.NCCLoss()], weight=[1.0], flow_model="direct")
    reg.optim(mov, tgt, lr=lr, max_epochs=iters)
    l32, l64 = g[f"{name}/losses32"], g[f"{name}/losses64"]
    assert np.max(np.abs(reg.losses[0].cpu().numpy() - l32)) <= bar(l32, l64, 1e-4 * np.max(np.abs(l64)))
    f32, f64 = g[f"{name}/flow32"], g[f"{name}/flow64"]
    assert np.max(np.abs(reg.final_theta.cpu().numpy() - f32)) <= bar(f32, f64, 1e-4)
    # Register.theta is the flow of the LAST FORWARD (before the last update), like flowreg.flow
    assert not torch.equal(reg.theta, reg.final_theta)
    w = reg(torch.cat([mov, mov * 2], dim=1))
    assert w.shape == (1, 2) + tuple(mov.shape[2:])
    assert torch.allclose(w[:, 1], 2 * w[:, 0], atol=1e-6)
    # SpatialTransformer module + autograd wrt flow
    st = tr.SpatialTransformer(tuple(single_step["D3/shape"])).cuda()
    fl = ph.flow_field(tuple(single_step["D3/shape"]), 0.8).cuda().requires_grad_()
    m, t = ph.vol((5, 6, 7), 0.37, "sin").cuda(), ph.vol((5, 6, 7), 0.23, "cos").cuda()
    e = tr.NCCLoss()(t, st(m, fl))
    e.backward()
    assert abs(e.item() - float(single_step["D3/ncc32"])) <= 2e-4
    assert np.max(np.abs(fl.grad.cpu().numpy() - single_step["D3/dncc32"])) <= bar(single_step["D3/dncc32"], single_step["D3/dncc64"], 1e-4 * np.abs(single_step["D3/dncc64"]).max())


def test_register_batch_extension(tr):
    shape = (24, 24, 24)
    tgts = torch.cat([ph.blobs(shape, 30 + i) for i in range(4)]).cuda()
    movs = tr.get_affine_warp(torch.tensor(ph.THETA_STAR3)[None].cuda(), tgts)
    reg = tr.Register("affine", device="cuda", criterion=[nn.MSELoss()], weight=[1.0])
    reg.optim(movs, tgts, lr=0.1, max_epochs=50)
    assert reg.theta.shape == (4, 3, 4) and reg.losses.shape == (4, 50)
    assert (reg.losses[:, -1] < reg.losses[:, 0]).all()
    assert reg(movs).shape == movs.shape


@pytest.mark.parametrize("name,crit", [("unet2d_ncc", ["ncc"]), ("unet2d_mix", ["mse", "ncc"])])
def test_register_flow_mode_unet_vs_reference(tr, trajectories, name, crit):
    """mode='flow' as the reference runs it: the attention U-Net (same seed -> same weights) generates the flow,
    warp + loss + backward in the fused HIP kernels.

    (a) vs the reference's own run (golden): the first two losses agree to 2e-4 — i.e. the initial forward and
        the first full backward + SGD step.  Later iterations cannot be compared across conv back-ends: the
        reference's U-Net instance-normalises 2x2-element feature maps at its bottleneck (160^2 input) and a
        random-init flow of ~10 voxels hops over interpolation kinks, so MIOpen-vs-MKL-DNN rounding is amplified
        to percent level by iteration 3 (measured: 5.6301 vs 5.6949).
    (b) vs the SAME model driven by plain torch GPU ops (F.grid_sample warp + torch losses, identical
        convolutions): loss and all parameter gradients of one backward agree, which isolates and validates
        the HIP warp/loss autograd boundary."""
    from oracle import compose
    g = trajectories
    lr, iters, seed = float(g[f"{name}/meta"][0]), int(g[f"{name}/meta"][1]), int(g[f"{name}/meta"][2])
    weights = [float(v) for v in g[f"{name}/meta"][3:]]
    shape = tuple(g[f"{name}/shape"])
    mov, tgt = torch.from_numpy(g[f"{name}/moving"]).cuda(), ph.blobs(shape, 1000 + seed).cuda()
    crits = [{"ncc": tr.NCCLoss(), "mse": nn.MSELoss()}[c] for c in crit]
    torch.manual_seed(seed)
    reg = tr.Register("flow", device="cuda", criterion=crits, weight=weights)
    reg.optim(mov, tgt, lr=lr, max_epochs=iters, n=32)
    gl = g[f"{name}/losses"]
    mine = reg.losses[0].cpu().numpy()
    assert len(mine) == len(gl)
    assert np.max(np.abs(mine[:2] - gl[:2])) <= 2e-4 * np.max(np.abs(gl))
    # (b) same seed, same module: parameter gradients of ONE backward through the fused HIP loss vs through
    #     plain torch GPU ops (F.grid_sample warp + torch losses).  (Whole trajectories are not comparable even
    #     between these two: torch-CPU, torch-GPU and this path agree on iterations 0-1 and then separate.)
    from torchregister_amd.warpings import _FlowLossFn, loss_spec_from
    torch.manual_seed(seed)
    net = tr.Attention_UNet(shape, "bilinear", in_c=1, n=32).cuda()
    fl = net.features(mov)
    e_ref = sum(w * c(tgt, compose.flow_warp(mov, fl)) for c, w in zip(crits, weights))
    g_ref = torch.autograd.grad(e_ref, list(net.parameters()), retain_graph=True)
    e_hip = _FlowLossFn.apply(fl, mov, tgt, loss_spec_from(crits, weights))
    g_hip = torch.autograd.grad(e_hip, list(net.parameters()))
    assert abs(e_hip.item() - e_ref.item()) <= 2e-5 * abs(e_ref.item())
    num = max((a - b).abs().max().item() for a, b in zip(g_hip, g_ref))
    den = max(b.abs().max().item() for b in g_ref)
    assert num <= 2e-3 * den, (num, den)
    w = reg(torch.cat([mov, 0.5 * mov + 0.25], dim=1))
    assert w.shape == (1, 2) + shape
    assert torch.allclose(w[:, :1], compose.flow_warp(mov, reg.theta), atol=1e-5)


def test_unet_seeded_weights_and_flow_match_reference_on_cpu_shapes(tr):
    """Attention_UNet: parameter names/shapes as in the reference (n=32: 31 278 parameters in 2-D, 89 189 in 3-D)."""
    m2 = tr.Attention_UNet((160, 160), "bilinear", in_c=1, n=32)
    assert sum(p.numel() for p in m2.parameters()) == 31278
    m3 = tr.Attention_UNet((156, 156, 156), "bilinear", in_c=1, n=32)
    assert sum(p.numel() for p in m3.parameters()) == 89189
    names = [n for n, _ in m2.named_parameters()]
    assert names[0] == "layer1.0.weight" and "skip1.input_filter.weight" in names and names[-1] == "out.bias"


def test_register_fully_default_criterion_with_nmi(tr, trajectories):
    """Register('affine', criterion=None, weight=[.33,.33,.33]): MSE + NCC + the Parzen-window NMI (torch ops on the
    GPU) drive the HIP warp through autograd (generic path).  Tolerance 2e-3 rel on the curve: the KDE sums 10^4 x 256
    exponentials per patch in a different order on the GPU."""
    g = trajectories
    name = "affine2d_default"
    lr, iters = float(g[f"{name}/meta"][0]), int(g[f"{name}/meta"][1])
    weight = [float(v) for v in g[f"{name}/meta"][4:7]]
    mov, tgt = _mov_tgt(g, name)
    reg = tr.Register("affine", device="cuda", criterion=None, weight=weight)
    reg.optim(mov, tgt, lr=lr, max_epochs=iters, per=0.125)
    gl = g[f"{name}/losses"]
    mine = reg.losses.cpu().numpy().ravel()
    assert np.max(np.abs(mine - gl)) <= 2e-3 * np.max(np.abs(gl)), (mine, gl)
    assert np.max(np.abs(reg.theta.cpu().numpy() - g[f"{name}/best_theta"])) <= 2e-3


def test_readme_pipeline_rigid_affine_flow():
    """The usage of ref:README.md:58-83 through the drop-in import name: rigid -> affine -> flow, each stage registering the
    previous stage's warp (`reg(moving)`), default criterion (MSE + NCC + NMI weights, ref:torchregister.py:52-60) for the
    parametric stages.  Checks the plumbing end to end: every stage lowers the mismatch to the target."""
    import TorchRegister as tr
    shape = (40, 48, 44)
    target = ph.blobs(shape, 77).cuda()
    th = torch.tensor([[0.98, -0.06, 0.03, 0.04], [0.07, 1.03, -0.02, -0.03], [-0.02, 0.03, 0.97, 0.02]])
    moving = tr.get_affine_warp(th[None].cuda(), target) + 0.0

    def mismatch(x):
        return torch.mean((x - target) ** 2).item()

    m0 = mismatch(moving)
    torch.manual_seed(3)
    warping = tr.Register(mode='rigid', device='cuda', debug=False, init=torch.zeros(6))
    warping.optim(moving, target, max_epochs=60, lr=2e-3)
    warped1 = warping(moving)
    assert warped1.shape == moving.shape and mismatch(warped1) < m0
    warping = tr.Register(mode='affine', device='cuda', debug=False)
    warping.optim(warped1.detach(), target, max_epochs=60, lr=2e-3)
    warped2 = warping(warped1.detach())
    assert mismatch(warped2) < mismatch(warped1)
    assert tuple(warping.theta.shape) == (1, 3, 4)
    warping = tr.Register(mode='flow', device='cuda', debug=False, criterion=[tr.NCCLoss()], weight=[1.0], flow_model='direct')
    warping.optim(warped2.detach(), target, lr=2.0, max_epochs=60)
    warped3 = warping(warped2.detach())
    assert mismatch(warped3) < mismatch(warped2)
    assert tuple(warping.theta.shape) == (1, 3, *shape)          # the flow field, voxel units (ref:README.md:80)
    assert torch.isfinite(tr.norm(torch.abs(warping.theta))).all()


@pytest.mark.parametrize("mode,shape", [("affine", (40, 44, 48)), ("rigid", (40, 44, 48)), ("affine", (72, 80)), ("rigid", (72, 80))])
def test_default_criterion_fast_loop_equals_generic_autograd_loop(tr, monkeypatch, mode, shape):
    """criterion=None (MSE + NCC + NMI, ref:warpings.py:33-35) runs a dedicated loop without autograd and without a per-iteration host
    sync (warpings._nmi_affine_loop); it must follow the generic loop (HIP warp as an autograd op + torch NMI algebra + torch SGD,
    sample lines from .item() extrema like the reference) - loss curve 2e-5 of its maximum, theta 2e-6."""
    import torchregister_amd.warpings as w
    nd = len(shape)
    tgt = ph.blobs(shape, 77).cuda()
    th = torch.tensor(ph.THETA_STAR3 if nd == 3 else ph.THETA_STAR2)[None].cuda()
    mov = tr.get_affine_warp(th, tgt) + 0.0
    init = torch.tensor([0.05, -0.03, 0.04, 0.1, -0.1, 0.05][: 6 if nd == 3 else 3]) if mode == "rigid" else None
    runs = {}
    for which in ("fast", "generic"):
        if which == "generic":
            monkeypatch.setattr(w, "_nmi_fast_path", lambda *a, **k: None)
        reg = tr.Register(mode, device="cuda", criterion=None, weight=[0.3, 0.5, 0.2], init=init)
        reg.optim(mov, tgt, lr=2e-5, max_epochs=8)
        runs[which] = (reg.losses.detach().flatten().cpu().double().numpy(), reg.final_theta.cpu().double().numpy(), reg.theta.cpu().double().numpy())
    lf, tf, bf = runs["fast"]
    lg, tg, bg = runs["generic"]
    assert len(lf) == len(lg) == 8 and lg[-1] < lg[0]
    assert np.max(np.abs(lf - lg)) <= 2e-5 * np.max(np.abs(lg)), (lf, lg)
    assert np.max(np.abs(tf - tg)) <= 2e-6 and np.max(np.abs(bf - bg)) <= 2e-6

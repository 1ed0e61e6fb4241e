"""GPU parity: the HIP dense-flow path (SpatialTransformer semantics) vs oracle + golden vectors.

Tolerances as in test_gpu_affine.py (fp32; bar = max(stated floor, 2x the reference's own
fp32-vs-fp64 gap)).  The smoothness regulariser and Adam are extensions that the reference does
not have ("parity unpinned"): they are checked against plain torch autograd / torch.optim.Adam.
"""
import numpy as np
import pytest
import torch

import oracle
import phantoms as ph
from conftest import bar

pytestmark = pytest.mark.gpu

FLOW_CASES = ["D3", "D2", "F3", "F2"]
LOSSES = {"ncc": dict(w_ncc=1.0), "mse": dict(w_mse=1.0)}


@pytest.fixture(scope="module")
def eng():
    import torchregister_amd._engine as e
    assert torch.cuda.is_available()
    return e


def _mt(shape):
    return ph.vol(shape, 0.37, "sin"), ph.vol(shape, 0.23, "cos")


@pytest.mark.parametrize("case", FLOW_CASES)
def test_flow_warp_and_grad_vs_golden(eng, single_step, case):
    g = single_step
    shape = tuple(g[f"{case}/shape"])
    mov, tgt = _mt(shape)
    fl = ph.flow_field(shape, float(g[f"{case}/amp"]))
    w = eng.flow_warp(mov.cuda(), fl.cuda()).cpu().numpy()
    g32, g64 = g[f"{case}/warped32"], g[f"{case}/warped64"]
    assert np.max(np.abs(w - g32)) <= bar(g32, g64, 2e-6)
    for lname, kw in LOSSES.items():
        terms, dfl = eng.flow_loss_grad(mov.cuda(), tgt.cuda(), fl.cuda(), eng.LossSpec(**kw))
        l32, l64 = float(g[f"{case}/{lname}32"]), float(g[f"{case}/{lname}64"])
        d32, d64 = g[f"{case}/d{lname}32"], g[f"{case}/d{lname}64"]
        assert abs(terms[0, 0].item() - l32) <= bar(l32, l64, 2e-5 * max(1.0, abs(l64)))
        assert np.max(np.abs(dfl.cpu().numpy() - d32)) <= bar(d32, d64, 1e-4 * np.max(np.abs(d64)))


@pytest.mark.parametrize("case", ["D3", "F2"])
def test_flow_warp_backward_generic(eng, single_step, case):
    """Generic warp backward with grad_out = dMSE/dwarped reproduces the golden dMSE/dflow."""
    g = single_step
    shape = tuple(g[f"{case}/shape"])
    mov, tgt = _mt(shape)
    fl = ph.flow_field(shape, float(g[f"{case}/amp"])).cuda()
    w = eng.flow_warp(mov.cuda(), fl)
    go = 2.0 * (w - tgt.cuda()) / w.numel()
    dfl = eng.flow_warp_backward(mov.cuda(), fl, go).cpu().numpy()
    d32, d64 = g[f"{case}/dmse32"], g[f"{case}/dmse64"]
    assert np.max(np.abs(dfl - d32)) <= bar(d32, d64, 1e-4 * np.max(np.abs(d64)))


def test_flow_multichannel_deform(eng, single_step):
    """Register.__call__ in flow mode: every channel warped by the same flow (ref:torchregister.py:123-126)."""
    g = single_step
    shape = tuple(g["deform/shape"])
    x = torch.cat([ph.vol(shape, 0.37, "sin"), ph.vol(shape, 0.23, "cos")], dim=1)
    fl = ph.flow_field(shape, 1.1, 0.07)
    w = eng.flow_warp(x.cuda(), fl.cuda()).cpu().numpy()
    assert np.max(np.abs(w - g["deform/call2c"])) <= 2e-6


@pytest.mark.parametrize("name,loss", [("c_flow3d_ncc", "ncc"), ("c_flow3d_mse", "mse"), ("c_flow2d_ncc", "ncc")])
def test_flow_trajectories_vs_golden(eng, trajectories, name, loss):
    g = trajectories
    lr, iters = float(g[f"{name}/meta"][0]), int(g[f"{name}/meta"][1])
    shape = tuple(g[f"{name}/shape"])
    seed = int(g[f"{name}/meta"][2])
    mov, tgt = torch.from_numpy(g[f"{name}/moving"]), ph.blobs(shape, 1000 + seed)
    s = eng.FlowSolver(mov.cuda(), tgt.cuda(), loss=eng.LossSpec(**LOSSES[loss]), lr=lr, capacity=iters)
    s.run(iters)
    torch.cuda.synchronize()
    losses = s.losses[0].cpu().numpy().astype(np.float64)
    l32, l64 = g[f"{name}/losses32"], g[f"{name}/losses64"]
    assert np.max(np.abs(losses - l32)) <= bar(l32, l64, 1e-4 * np.max(np.abs(l64)))
    f32, f64 = g[f"{name}/flow32"], g[f"{name}/flow64"]
    assert np.max(np.abs(s.flow.cpu().numpy() - f32)) <= bar(f32, f64, 1e-4)
    w = eng.flow_warp(mov.cuda(), s.flow).cpu().numpy()
    assert np.max(np.abs(w - g[f"{name}/final_warped32"])) <= bar(g[f"{name}/final_warped32"], g[f"{name}/final_warped64"], 1e-4)


def _torch_flow_ref(mov, tgt, lr, iters, optimizer, smooth, dtype):
    """Plain-torch reference of the extensions (Adam, smoothness) on CPU."""
    from oracle import compose
    nd = mov.dim() - 2
    mov, tgt = mov.to(dtype), tgt.to(dtype)
    fl = torch.zeros(1, nd, *mov.shape[2:], dtype=dtype, requires_grad=True)
    opt = torch.optim.SGD([fl], lr) if optimizer == "sgd" else torch.optim.Adam([fl], lr)
    losses = []
    for _ in range(iters):
        opt.zero_grad()
        e = compose.ncc_loss(tgt, compose.flow_warp(mov, fl))
        if smooth:
            reg = 0.0
            for d in range(nd):
                df = fl.diff(dim=2 + d)
                reg = reg + (df * df).mean()
            e = e + smooth * reg / nd
        e.backward()
        opt.step()
        losses.append(e.item())
    return np.asarray(losses), fl.detach().numpy()


@pytest.mark.parametrize("optimizer,smooth", [("sgd", 5.0), ("adam", 0.0), ("adam", 2.0)])
def test_flow_extensions_vs_torch(eng, optimizer, smooth):
    shape = (12, 14, 16)
    from oracle import compose
    tgt = ph.blobs(shape, 1009)
    mov = compose.affine_warp(torch.tensor(ph.THETA_STAR3)[None], tgt)
    lr, iters = (1.0, 15) if optimizer == "sgd" else (0.05, 15)
    l32, f32 = _torch_flow_ref(mov, tgt, lr, iters, optimizer, smooth, torch.float32)
    l64, f64 = _torch_flow_ref(mov, tgt, lr, iters, optimizer, smooth, torch.float64)
    s = eng.FlowSolver(mov.cuda(), tgt.cuda(), loss=eng.LossSpec(w_ncc=1.0), optimizer=optimizer, lr=lr, capacity=iters,
                       smooth_weight=smooth)
    s.run(iters)
    torch.cuda.synchronize()
    assert np.max(np.abs(s.losses[0].cpu().numpy() - l32)) <= bar(l32, l64, 1e-4 * np.max(np.abs(l64)))
    assert np.max(np.abs(s.flow.cpu().numpy() - f32)) <= bar(f32, f64, 2e-4)


def test_flow_headline_size_vs_oracle(eng):
    """BASELINE config 3 at full size - one 256^3 pair, dense flow + NCC: loss and dL/dflow of one evaluation against the C oracle in
    fp64 (bar: 1e-4 of the gradient's maximum or twice the oracle's own fp32-vs-fp64 gap, as in the small cases)."""
    shape = (256, 256, 256)
    tgt, mov = ph.blobs(shape, 1000), ph.blobs(shape, 1001)
    ax = [torch.arange(n, dtype=torch.float64) for n in shape]
    comp = lambda a, b, c: (torch.sin(a * ax[0])[:, None, None] + torch.cos(b * ax[1])[None, :, None] + torch.sin(c * ax[2] + 0.4)[None, None, :])
    fl = torch.stack([1.3 * comp(0.021, 0.017, 0.013), 0.9 * comp(0.011, 0.023, 0.019), 1.1 * comp(0.015, 0.012, 0.027)]).float()[None] + 0.37
    kw = dict(w_ncc=1.0)
    terms, dfl = eng.flow_loss_grad(mov.cuda(), tgt.cuda(), fl.cuda(), eng.LossSpec(**kw))
    torch.cuda.synchronize()
    args = lambda dt: (mov[0, 0].numpy().astype(dt), tgt[0, 0].numpy().astype(dt), fl[0].numpy().astype(dt), oracle.wts(**kw))
    t64, _, d64, _ = oracle.c_flow_loss_grad(*args(np.float64))
    t32, _, d32, _ = oracle.c_flow_loss_grad(*args(np.float32))
    assert abs(terms[0, 0].item() - t64) <= 2e-5 * max(1.0, abs(t64))
    assert np.max(np.abs(dfl[0].cpu().numpy() - d64)) <= max(1e-4 * np.max(np.abs(d64)), 2.0 * np.max(np.abs(d32.astype(np.float64) - d64)))


@pytest.mark.parametrize("smooth", [0.0, 2.5])
@pytest.mark.parametrize("shape", [(40, 36, 44), (70, 90)])
@pytest.mark.parametrize("optimizer,lr", [("sgd", 1.0), ("adam", 0.02)])
def test_fused_next_moments_are_bitwise_the_two_pass_steps(eng, optimizer, lr, shape, smooth):
    """Inside one trx_flow_run call the update kernel of iteration i also produces the moments of iteration i + 1 (no
    smoothness term; 2-D and 3-D).  Same voxel order, same arithmetic as the stand-alone moments pass: run(12) and 12 x run(1) (which cannot
    fuse) must agree bit for bit - loss curve, flow and, for Adam, the optimiser state."""
    tgt, mov = ph.blobs(shape, 1).cuda(), ph.blobs(shape, 2).cuda()
    kw = dict(loss=eng.LossSpec(w_ncc=1.0, w_mse=0.3), optimizer=optimizer, lr=lr, capacity=13, smooth_weight=smooth)
    a = eng.FlowSolver(mov, tgt, **kw)
    a.run(7)        # (with the smoothness term the regulariser part of each recorded loss arrives one coefficient kernel later and
    a.run(2)        #  the last one of a call by a flush: several calls of different lengths, odd and even for the double buffer)
    a.run(1)
    a.run(3)
    b = eng.FlowSolver(mov, tgt, **kw)
    for _ in range(13):
        b.run(1)
    torch.cuda.synchronize()
    assert torch.equal(a.losses, b.losses) and torch.equal(a.flow, b.flow)
    if optimizer == "adam":
        assert torch.equal(a.adam_m, b.adam_m) and torch.equal(a.adam_v, b.adam_v)


def test_flow_full_size_properties(eng):
    """256^3: zero flow reproduces moving exactly; integer shift flow == slicing; run is deterministic."""
    shape = (256, 256, 256)
    mov = ph.blobs(shape, 1000).cuda()
    zero = torch.zeros(1, 3, *shape, device="cuda")
    assert torch.equal(eng.flow_warp(mov, zero), mov)
    fl = zero.clone()
    fl[:, 0] = 2.0
    fl[:, 2] = -3.0
    w = eng.flow_warp(mov, fl)
    assert torch.equal(w[0, 0, :-2, :, 3:], mov[0, 0, 2:, :, :-3])
    assert torch.count_nonzero(w[0, 0, -2:]).item() == 0 and torch.count_nonzero(w[0, 0, :, :, :3]).item() == 0
    tgt = ph.blobs(shape, 1001).cuda()
    outs = []
    for _ in range(2):
        s = eng.FlowSolver(mov, tgt, loss=eng.LossSpec(w_ncc=1.0), lr=10.0, capacity=3)
        s.run(3)
        torch.cuda.synchronize()
        outs.append((s.losses.clone(), s.flow.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][0][0, 2].item() < outs[0][0][0, 0].item()


@pytest.mark.parametrize("optimizer,lr,smooth", [("sgd", 2.0, 0.0), ("adam", 0.05, 0.0), ("sgd", 1.0, 4.0), ("adam", 0.05, 2.0)])
def test_slab_partition_equals_whole_volume(eng, optimizer, lr, smooth):
    """Z-slab mode (config 5) emulated on one GPU: three slabs of unequal depth; the moments are summed by hand (what the
    all-reduce does) and, with the smoothness regulariser, the neighbours' boundary flow planes are copied by hand
    (what the xGMI halo exchange does).  Must reproduce the un-partitioned FlowSolver: loss curve to 1e-5 rel
    (fp64 sums in a different order), flow to 1e-5."""
    shape = (36, 28, 40)
    tgt = ph.blobs(shape, 1021).cuda()
    mov = ph.blobs(shape, 1022).cuda()
    iters = 6
    kw = dict(loss=eng.LossSpec(w_ncc=1.0, w_mse=0.3), optimizer=optimizer, lr=lr, capacity=iters, smooth_weight=smooth)
    whole = eng.FlowSolver(mov, tgt, **kw)
    whole.run(iters)
    bounds = [0, 10, 25, 36]
    slabs = [eng.SlabFlowSolver(mov, tgt[:, :, a:b].contiguous(), a, **kw) for a, b in zip(bounds[:-1], bounds[1:])]
    for _ in range(iters):
        if smooth:   # halo exchange between Z neighbours
            planes = [s.boundary_planes() for s in slabs]
            for r, s in enumerate(slabs):
                if s.has_lo:
                    s.halo_lo.copy_(planes[r - 1][1])
                if s.has_hi:
                    s.halo_hi.copy_(planes[r + 1][0])
        total = sum(s.local_moments().clone() for s in slabs)
        for s in slabs:
            s.apply(total)
    torch.cuda.synchronize()
    for s in slabs:
        assert torch.allclose(s.losses, whole.losses, rtol=1e-5, atol=1e-6)       # every rank records the whole-volume loss
    flow = torch.cat([s.flow for s in slabs], dim=2)
    # Adam divides by sqrt(v): where the gradient is ~1e-9 (smoothness term of a near-zero flow) a last-bit change of
    # the coefficients (the global sums are added in a different order) moves the step by a fraction of lr
    tol = 2e-3 if (optimizer == "adam" and smooth) else 1e-5
    assert torch.max(torch.abs(flow - whole.flow)).item() <= tol * max(1.0, whole.flow.abs().max().item())
    # single-rank run() path (no process group): one full-depth slab == the plain solver
    one = eng.SlabFlowSolver(mov, tgt, 0, **kw)
    one.run(iters)
    torch.cuda.synchronize()
    assert torch.allclose(one.losses, whole.losses, rtol=1e-6, atol=1e-7)
    assert torch.allclose(one.flow, whole.flow, atol=1e-6)


def test_slab_boundary_smooth_term_split(eng):
    """Pass A with the upper neighbour's plane == pass A without it + trx_flow_slab_boundary_smooth (the split that lets the halo
    exchange run on a side stream beside pass A, SlabFlowSolver.run)."""
    shape = (20, 28, 36)
    tgt, mov = ph.blobs(shape, 31).cuda(), ph.blobs(shape, 32).cuda()
    kw = dict(loss=eng.LossSpec(w_ncc=1.0), optimizer="sgd", lr=1.0, capacity=2, smooth_weight=2.0)
    lo = eng.SlabFlowSolver(mov, tgt[:, :, :12].contiguous(), 0, **kw)
    fl = (0.7 * ph.flow_field(shape, 1.0, 0.05)).cuda()
    lo.flow.copy_(fl[:, :, :12])
    lo.halo_hi.copy_(fl[0, :, 12])                      # what the rank above would send: its lowest plane
    with_halo = lo.local_moments().clone()
    without = lo.local_moments_without_halo().clone()
    lo.add_boundary_smooth(without)
    torch.cuda.synchronize()
    assert not torch.equal(with_halo[0, 5], lo.local_moments_without_halo()[0, 5])      # the term is not zero
    assert torch.allclose(with_halo, without, rtol=2e-6, atol=0)        # (pass A carries the term in fp32 block partials, the small kernel in fp64)
    edge = ((fl[0, :, 12] - fl[0, :, 11]).double() ** 2).sum().item()
    assert abs((with_halo[0, 5] - lo.local_moments_without_halo()[0, 5]).item() - edge) <= 1e-6 * edge

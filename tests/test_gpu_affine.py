"""GPU parity: the HIP affine/rigid path (through the C ABI) vs the oracle and the golden vectors.

Tolerances (fp32 path): warped <= max(2e-6, 2x ref fp32-fp64 gap) abs; loss <=
max(2e-5*|L|, 2x gap); gradient <= max(1e-4*max|g|, 2x gap); trajectories: loss <= max(1e-4*max|L|,
2x gap), theta <= max(1e-4, 2x gap).  "gap" is the reference's own fp32-vs-fp64 difference stored in
the golden fixtures.
"""
import numpy as np
import pytest
import torch

import oracle
import phantoms as ph
from conftest import bar

pytestmark = pytest.mark.gpu

AFFINE_CASES = ["A3", "B2", "OOB3", "ROT3", "OOB2", "ID3", "ID2"]
LOSSES = {"ncc": dict(w_ncc=1.0), "mse": dict(w_mse=1.0), "ssd": dict(w_ssd=1.0)}


@pytest.fixture(scope="module")
def eng():
    import torchregister_amd._engine as e
    assert torch.cuda.is_available()
    return e


def _mt(shape):
    return ph.vol(shape, 0.37, "sin"), ph.vol(shape, 0.23, "cos")


@pytest.mark.parametrize("case", AFFINE_CASES)
def test_affine_warp_vs_golden(eng, single_step, case):
    g = single_step
    shape = tuple(g[f"{case}/shape"])
    mov, _ = _mt(shape)
    th = torch.tensor(g[f"{case}/theta"], dtype=torch.float32)[None].cuda()
    w = eng.affine_warp(th, mov.cuda()).cpu().numpy()
    g32, g64 = g[f"{case}/warped32"], g[f"{case}/warped64"]
    assert np.max(np.abs(w - g32)) <= bar(g32, g64, 2e-6)


@pytest.mark.parametrize("case", AFFINE_CASES)
@pytest.mark.parametrize("lname", ["ncc", "mse", "ssd"])
def test_affine_step_loss_grad_vs_golden(eng, single_step, case, lname):
    """One fused step: recorded loss and dL/dtheta against the reference's autograd values."""
    g = single_step
    shape = tuple(g[f"{case}/shape"])
    nd = len(shape)
    mov, tgt = _mt(shape)
    th = torch.tensor(g[f"{case}/theta"], dtype=torch.float32)[None]
    s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(**LOSSES[lname]), lr=0.0, init=th, capacity=2)
    s.run(1)
    torch.cuda.synchronize()
    loss = s.losses[0, 0].item()
    grad = s.grad[0, : nd * (nd + 1)].cpu().numpy().reshape(nd, nd + 1)
    l32, l64 = float(g[f"{case}/{lname}32"]), float(g[f"{case}/{lname}64"])
    d32, d64 = g[f"{case}/d{lname}32"], g[f"{case}/d{lname}64"]
    assert abs(loss - l32) <= bar(l32, l64, 2e-5 * max(1.0, abs(l64)))
    assert np.max(np.abs(grad - d32)) <= bar(d32, d64, 1e-4 * np.max(np.abs(d64)))
    # lr = 0: parameters unchanged, best == this theta
    assert torch.equal(s.current_theta.cpu(), th)
    assert s.best_idx[0].item() == 0 and torch.equal(s.best.cpu(), th)
    # loss-only entry point agrees with the step
    terms = s.eval_loss(th.cuda()).cpu().numpy()[0]
    assert abs(terms[0] - loss) <= 1e-6 * max(1.0, abs(loss))


@pytest.mark.parametrize("case", ["A3", "B2", "OOB3", "ROT3", "OOB2"])
def test_affine_step_vs_c_oracle_weighted_mix(eng, single_step, case):
    """Weighted MSE+NCC+SSD mix against the C oracle (fp64 arbiter)."""
    g = single_step
    shape = tuple(g[f"{case}/shape"])
    nd = len(shape)
    mov, tgt = _mt(shape)
    kw = dict(w_mse=0.4, w_ncc=0.7, ncc_alpha=50.0, w_ssd=0.01, ssd_alpha=2.0)
    th = torch.tensor(g[f"{case}/theta"], dtype=torch.float32)[None]
    s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=2)
    s.run(1)
    torch.cuda.synchronize()
    total, _, dth, _ = oracle.c_affine_loss_grad(mov[0, 0].double().numpy(), tgt[0, 0].double().numpy(), g[f"{case}/theta"],
                                                 oracle.wts(**kw), oracle.base_tables(shape, np.float64))
    assert abs(s.losses[0, 0].item() - total) <= 2e-5 * max(1.0, abs(total))
    grad = s.grad[0, : nd * (nd + 1)].cpu().numpy().reshape(nd, nd + 1)
    assert np.max(np.abs(grad - dth)) <= 2e-4 * np.max(np.abs(dth))


@pytest.mark.parametrize("case", ["A3", "B2", "ROT3", "OOB2"])
def test_affine_warp_backward_vs_oracle(eng, single_step, case):
    g = single_step
    shape = tuple(g[f"{case}/shape"])
    mov, tgt = _mt(shape)
    th = torch.tensor(g[f"{case}/theta"], dtype=torch.float32)[None]
    # grad_out = dMSE/dwarped -> dtheta must equal the golden dMSE/dtheta
    w = eng.affine_warp(th.cuda(), mov.cuda())
    go = 2.0 * (w - tgt.cuda()) / w.numel()
    dth = eng.affine_warp_backward(th.cuda(), mov.cuda(), go).cpu().numpy()[0]
    d32, d64 = g[f"{case}/dmse32"], g[f"{case}/dmse64"]
    assert np.max(np.abs(dth - d32)) <= bar(d32, d64, 1e-4 * np.max(np.abs(d64)))


def _mov_tgt(g, name):
    shape = tuple(g[f"{name}/shape"])
    seed = int(g[f"{name}/meta"][2])
    return torch.from_numpy(g[f"{name}/moving"]), ph.blobs(shape, 1000 + seed)


@pytest.mark.parametrize("name", ["rigid2d_mse", "rigid3d_mse", "affine3d_mse", "affine2d_mse"])
def test_driver_trajectories_vs_golden(eng, trajectories, name):
    """Whole loops of ref rigid_register / affine_register (MSE, SGD, best tracking)."""
    g = trajectories
    lr, iters = float(g[f"{name}/meta"][0]), int(g[f"{name}/meta"][1])
    mov, tgt = _mov_tgt(g, name)
    rigid = f"{name}/init" in g
    init = torch.from_numpy(g[f"{name}/init"])[None] if rigid else None
    s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="rigid" if rigid else "affine", loss=eng.LossSpec(w_mse=1.0), lr=lr,
                         init=init, capacity=iters)
    s.run(iters)
    torch.cuda.synchronize()
    losses = s.losses[0].cpu().numpy().astype(np.float64)
    l32, l64 = g[f"{name}/losses"], g[f"{name}/losses64"]
    assert np.max(np.abs(losses - l32)) <= bar(l32, l64, 1e-4 * np.max(np.abs(l64)))
    tbar = bar(g[f"{name}/final_theta"][0], g[f"{name}/thetas64"][-1], 1e-4)
    assert np.max(np.abs(s.current_theta[0].cpu().numpy() - g[f"{name}/final_theta"][0])) <= tbar
    # best = first strict minimum of the recorded curve; theta of that forward
    bi = s.best_idx[0].item()
    assert bi == int(np.argmin(losses))
    assert abs(s.best_loss[0].item() - losses[bi]) == 0.0
    gold_bi = int(np.argmin(l32))
    if bi == gold_bi:
        assert np.max(np.abs(s.best[0].cpu().numpy() - g[f"{name}/best_theta"][0])) <= tbar


@pytest.mark.parametrize("name,rigid", [("c_affine3d_ncc", False), ("c_affine2d_ncc", False), ("c_rigid3d_ncc", True)])
def test_ncc_trajectories_vs_golden(eng, trajectories, name, rigid):
    g = trajectories
    lr, iters = float(g[f"{name}/meta"][0]), int(g[f"{name}/meta"][1])
    mov, tgt = _mov_tgt(g, name)
    init = torch.from_numpy(g[f"{name}/init"])[None] if rigid else None
    s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="rigid" if rigid else "affine", loss=eng.LossSpec(w_ncc=1.0), lr=lr,
                         init=init, capacity=iters)
    s.run(iters)
    torch.cuda.synchronize()
    losses = s.losses[0].cpu().numpy().astype(np.float64)
    l32, l64 = g[f"{name}/losses32"], g[f"{name}/losses64"]
    assert np.max(np.abs(losses - l32)) <= bar(l32, l64, 1e-4 * np.max(np.abs(l64)))
    t32, t64 = g[f"{name}/thetas32"], g[f"{name}/thetas64"]
    assert np.max(np.abs(s.current_theta[0].cpu().numpy() - t32[-1])) <= bar(t32, t64, 1e-4)


def test_batch_equals_singles(eng):
    """B pairs in one launch == the same pairs solved one by one (bitwise: fixed-order reductions)."""
    shape = (20, 24, 28)
    movs = torch.cat([ph.blobs(shape, 10 + i) for i in range(3)]).cuda()
    tgts = torch.cat([ph.blobs(shape, 20 + i) for i in range(3)]).cuda()
    sb = eng.AffineSolver(movs, tgts, mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=1e-4, capacity=5)
    sb.run(5)
    for i in range(3):
        s1 = eng.AffineSolver(movs[i:i + 1], tgts[i:i + 1], mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=1e-4, capacity=5)
        s1.run(5)
        torch.cuda.synchronize()
        assert torch.allclose(sb.losses[i], s1.losses[0], rtol=1e-5, atol=1e-6)
        assert torch.allclose(sb.current_theta[i], s1.current_theta[0], rtol=0, atol=1e-6)


def test_determinism(eng):
    shape = (32, 32, 32)
    mov, tgt = ph.blobs(shape, 1).cuda(), ph.blobs(shape, 2).cuda()
    runs = []
    for _ in range(2):
        s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=1e-4, capacity=10)
        s.run(10)
        torch.cuda.synchronize()
        runs.append((s.losses.clone(), s.theta.clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])


def test_adam_vs_torch_adam(eng):
    """Adam extension: trajectory vs torch.optim.Adam on the oracle composition (CPU, fp64 arbiter bar)."""
    from oracle import compose
    shape = (16, 20, 24)
    tgt = ph.blobs(shape, 1003)
    mov = compose.affine_warp(torch.tensor(ph.THETA_STAR3)[None], tgt)
    r32 = compose.affine_loop(mov, tgt, 1e-3, 25, optimizer="adam", w_ncc=1.0)
    r64 = compose.affine_loop(mov.double(), tgt.double(), 1e-3, 25, optimizer="adam", w_ncc=1.0)
    s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(w_ncc=1.0), optimizer="adam", lr=1e-3, capacity=25)
    s.run(25)
    torch.cuda.synchronize()
    l32, l64 = r32["losses"].numpy(), r64["losses"].numpy()
    assert np.max(np.abs(s.losses[0].cpu().numpy() - l32)) <= bar(l32, l64, 1e-4 * np.max(np.abs(l64)))
    assert np.max(np.abs(s.current_theta[0].cpu().numpy() - r32["thetas"][-1].numpy())) <= bar(r32["thetas"].numpy(), r64["thetas"].numpy(), 1e-4)


def test_full_size_properties(eng):
    """256^3 (BASELINE size): identity theta reproduces moving; NCC(x,x)=0; loss-only == step loss."""
    shape = (256, 256, 256)
    tgt = ph.blobs(shape, 1000).cuda()
    ident = torch.eye(3, 4)[None].cuda()
    w = eng.affine_warp(ident, tgt)
    assert torch.max(torch.abs(w - tgt)).item() <= 1e-5
    s = eng.AffineSolver(tgt, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0, w_mse=1.0), lr=0.0, capacity=1)
    s.run(1)
    torch.cuda.synchronize()
    assert abs(s.losses[0, 0].item()) <= 1e-3
    # linearity of the warp in the image: warp(a*x + b*y) == a*warp(x) + b*warp(y)
    th = torch.tensor(ph.THETA_STAR3)[None].cuda()
    other = ph.blobs(shape, 1001).cuda()
    lhs = eng.affine_warp(th, 0.3 * tgt + 0.7 * other)
    rhs = 0.3 * eng.affine_warp(th, tgt) + 0.7 * eng.affine_warp(th, other)
    assert torch.max(torch.abs(lhs - rhs)).item() <= 1e-5
    # fused loss at full size vs torch GPU ops on the HIP-warped volume (fp64 reduction)
    s2 = eng.AffineSolver(eng.affine_warp(th, tgt), tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=0.0, capacity=1)
    s2.run(1)
    torch.cuda.synchronize()
    wv = eng.affine_warp(ident, s2.batch.moving).double()
    y = tgt.double()
    a, b = y - y.mean(), wv - wv.mean()
    ncc = 100.0 * (1 - (a * b).sum() / ((a * a).sum() * (b * b).sum() + 1e-10).sqrt())
    assert abs(s2.losses[0, 0].item() - ncc.item()) <= 2e-4 * max(1.0, abs(ncc.item()))


def test_headline_size_step_vs_oracle(eng):
    """The bench workload itself - 256^3, affine + NCC, theta near the bench's theta* - against the C oracle in fp64 (and its fp32
    run for the bar): loss 2e-5 relative, dL/dtheta 2e-4 of its maximum or twice the oracle's own fp32-vs-fp64 gap.  Two pairs with
    different theta in one launch (the oracle needs ~10 s per evaluation at this size, so not the full batch of 8)."""
    shape = (256, 256, 256)
    tgt = torch.cat([ph.blobs(shape, 1000), ph.blobs(shape, 1001)])
    ths = np.stack([np.asarray(ph.THETA_STAR3, dtype=np.float64).reshape(3, 4),
                    np.eye(3, 4) + 0.02 * np.sin(1.3 * np.arange(12)).reshape(3, 4)])
    th = torch.tensor(ths, dtype=torch.float32)
    mov = eng.affine_warp(torch.tensor(np.stack([np.eye(3, 4) + 0.03 * np.cos(0.9 * np.arange(12)).reshape(3, 4)] * 2), dtype=torch.float32).cuda(),
                          tgt.cuda()).cpu() + 0.05 * torch.cat([ph.blobs(shape, 1002), ph.blobs(shape, 1003)])
    kw = dict(w_ncc=1.0)
    s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1)
    s.run(1)
    torch.cuda.synchronize()
    t64, t32 = oracle.base_tables(shape, np.float64), oracle.base_tables(shape, np.float32)
    for b in range(2):
        total, _, dth, _ = oracle.c_affine_loss_grad(mov[b, 0].double().numpy(), tgt[b, 0].double().numpy(), th[b].double().numpy(), oracle.wts(**kw), t64)
        _, _, dth32, _ = oracle.c_affine_loss_grad(mov[b, 0].numpy(), tgt[b, 0].numpy(), th[b].numpy(), oracle.wts(**kw), t32)
        assert abs(s.losses[b, 0].item() - total) <= 2e-5 * max(1.0, abs(total))
        gmax = np.max(np.abs(dth))
        assert np.max(np.abs(s.grad[b, :12].cpu().numpy().reshape(3, 4) - dth)) <= max(2e-4 * gmax, 2.0 * np.max(np.abs(dth32 - dth)))


@pytest.mark.parametrize("shape", [(1, 1, 4), (2, 3, 4), (1, 16, 32), (3, 1, 8), (4, 4), (1, 7), (2, 2, 2)])
def test_tiny_and_degenerate_shapes(eng, shape):
    """Smallest volumes (single voxel rows/planes, sizes below every tile dimension) against the C oracle."""
    nd = len(shape)
    mov, tgt = _mt(shape)
    th64 = np.eye(nd, nd + 1) + 0.01 * np.sin(1.7 * np.arange(nd * (nd + 1))).reshape(nd, nd + 1)
    th = torch.tensor(th64, dtype=torch.float32)[None]
    w = eng.affine_warp(th.cuda(), mov.cuda()).cpu().numpy()[0, 0]
    ref = oracle.c_affine_warp(mov[0, 0].double().numpy(), th[0].double().numpy(), oracle.base_tables(shape, np.float64))
    assert np.max(np.abs(w - ref)) <= 2e-6
    s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(w_mse=1.0, w_ssd=0.1), lr=0.0, init=th, capacity=1)
    s.run(1)
    torch.cuda.synchronize()
    total, _, dth, _ = oracle.c_affine_loss_grad(mov[0, 0].double().numpy(), tgt[0, 0].double().numpy(), th[0].double().numpy(),
                                                 oracle.wts(w_mse=1.0, w_ssd=0.1), oracle.base_tables(shape, np.float64))
    assert abs(s.losses[0, 0].item() - total) <= 2e-5 * max(1.0, abs(total))
    g = s.grad[0, : nd * (nd + 1)].cpu().numpy().reshape(nd, nd + 1)
    assert np.max(np.abs(g - dth)) <= 2e-4 * max(1e-6, np.max(np.abs(dth)))


def test_graph_capture_and_side_stream(eng):
    """The C ABI neither allocates nor synchronises: a run can be captured into a HIP graph and replayed, and it
    follows torch's current stream."""
    shape = (32, 32, 32)
    mov, tgt = ph.blobs(shape, 41).cuda(), ph.blobs(shape, 42).cuda()
    ref = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=1e-4, capacity=12)
    ref.run(12)
    torch.cuda.synchronize()
    s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=1e-4, capacity=12)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        s.run(4)                       # eager, on a side stream
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            s.run(4)                   # captured: 8 kernel nodes, nothing executes yet
        g.replay()
        g.replay()
    side.synchronize()
    assert torch.equal(s.step.cpu(), torch.tensor([12], dtype=torch.int32))
    assert torch.equal(s.losses, ref.losses) and torch.equal(s.theta, ref.theta)


LATTICE_CASES = [  # spatial, lattice size, theta (row-major nd x (nd+1))
    ((40, 36, 52), (20, 20, 20), [[1.02, 0.03, -0.02, 0.05], [-0.03, 0.98, 0.02, -0.04], [0.01, -0.02, 1.03, 0.02]]),
    ((23, 31, 37), (10, 12, 14), [[0.9, 0.3, 0.1, 0.4], [-0.3, 0.9, 0.05, -0.3], [0.1, -0.1, 1.1, 0.2]]),         # samples leave the volume (zero padding)
    ((16, 18, 20), (24, 20, 40), [[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0]]),             # up-sampling: lattice points repeat
    ((48, 40), (20, 20), [[1.01, 0.04, 0.03], [-0.05, 0.97, -0.02]]),
    ((33, 57), (50, 16), [[0.8, 0.5, 0.3], [-0.5, 0.8, -0.2]]),
]


@pytest.mark.parametrize("spatial,size,theta", LATTICE_CASES)
def test_warp_lattice_equals_warp_then_nearest(eng, spatial, size, theta):
    """trx_affine_warp_lattice[_backward] (the NMI loss's view of the warp, ref:utils.py:236-252) against the chain it replaces:
    full-volume HIP warp (golden / oracle-checked above) -> F.interpolate(nearest) and, backward, the adjoint of that gather ->
    trx_affine_warp_backward.  Backward: 1e-4 of max|dtheta| (fp32 partial sums in a different order)."""
    import ctypes
    import torch.nn.functional as F
    nd = len(spatial)
    B = 2
    mov = torch.stack([ph.vol(spatial, 0.37 + 0.11 * b, "sin")[0] for b in range(B)]).cuda()
    th = torch.tensor(theta, dtype=torch.float32)[None].repeat(B, 1, 1)
    th[1, 0, -1] += 0.07
    th = th.cuda()
    batch = eng._Batch(mov, mov)
    vol = batch.vol()
    lat = eng.LatticeWarp(vol, spatial, size, mov.device)
    thp = eng.pad_theta(th.reshape(B, -1), nd)
    got = lat.forward(thp)
    full = eng.affine_warp(th, mov)
    ref = F.interpolate(full, size=size, mode="nearest")
    assert got.shape == (B, int(np.prod(size)))
    # the phantom changes by O(1) between neighbouring voxels and the LDS-tiled warp kernel forms its coordinates as corner + slope
    # (1e-5 voxels from the plain fma chain): 2e-5 abs against it, and EXACT equality with the un-tiled kernel (same arithmetic)
    assert torch.max(torch.abs(got - ref.detach().reshape(B, -1))).item() <= 2e-5
    import ctypes as ct
    from torchregister_amd import _lib
    gb = eng._Batch(mov, mov, flags=_lib.FLAG_GATHER_PATH)
    gvol, plain = gb.vol(), torch.empty_like(mov)
    _lib.check(_lib.load().trx_affine_warp(ct.byref(gvol), _lib.ptr(thp), 1, _lib.ptr(plain), _lib.current_stream(mov.device)), "trx_affine_warp")
    assert torch.equal(got, F.interpolate(plain, size=size, mode="nearest").reshape(B, -1))
    g = torch.Generator().manual_seed(sum(spatial))
    go = (torch.rand(B, int(np.prod(size)), generator=g) - 0.4).cuda()
    dth = lat.backward(thp, go)[:, : nd * (nd + 1)].reshape(B, nd, nd + 1).cpu().numpy()
    # adjoint of the nearest gather, built from the lattice tables.  (NOT torch.autograd through F.interpolate on the GPU: ATen's
    # device kernel for the nearest BACKWARD inverts the index map with its own float arithmetic - ceil(d * (out / in)) - and for
    # non-integer ratios sends some gradients to a neighbouring voxel, e.g. 52 -> 20: output 5 reads input 13, its gradient lands on
    # input 12.  ATen's CPU kernel, which the reference runs on, uses the forward index function in both directions.)
    gw = torch.zeros_like(full)
    idx = torch.meshgrid(*[t.long() for t in ((lat.iz, lat.iy, lat.ix) if nd == 3 else (lat.iy, lat.ix))], indexing="ij")
    for b in range(B):
        gw[b, 0].index_put_(idx, go[b].view(size), accumulate=True)
    # through the un-tiled kernel (same coordinate arithmetic as the lattice kernel): the derivative of trilinear interpolation jumps at
    # cell faces, and a sample within 1e-5 voxels of one may fall on different sides in the tiled kernel (one such sample moves an entry
    # by ~1 %; seen in the first case) - the tiled path gets the looser bar
    want = torch.zeros(B, eng.PSTRIDE, device=mov.device)
    ws = torch.empty(int(_lib.load().trx_affine_workspace_bytes(ct.byref(gvol))), dtype=torch.uint8, device=mov.device)
    gvol.target = gw.data_ptr()
    _lib.check(_lib.load().trx_affine_warp_backward(ct.byref(gvol), _lib.ptr(thp), 1, _lib.ptr(gw), _lib.ptr(want), _lib.ptr(ws), ws.numel(),
                                                    _lib.current_stream(mov.device)), "trx_affine_warp_backward")
    want = want[:, : nd * (nd + 1)].reshape(B, nd, nd + 1).cpu().numpy()
    assert np.max(np.abs(dth - want)) <= 1e-4 * np.max(np.abs(want))
    tiled = eng.affine_warp_backward(th, mov, gw.contiguous()).cpu().numpy()
    assert np.max(np.abs(dth - tiled)) <= 2e-2 * np.max(np.abs(tiled))
    assert ctypes.sizeof(vol) > 0   # (vol must stay alive while lat is used)


def test_warp_lattice_argument_checks(eng):
    import ctypes
    from torchregister_amd import _lib
    mov = ph.vol((8, 9, 10), 0.3, "sin").cuda()
    vol = eng._Batch(mov, mov).vol()
    lat = eng.LatticeWarp(vol, (8, 9, 10), (4, 4, 4), mov.device)
    th = eng.pad_theta(torch.eye(3, 4).reshape(1, -1).cuda(), 3)
    out = torch.empty(1, 64, device="cuda")
    lib = _lib.load()
    rc = lib.trx_affine_warp_lattice(ctypes.byref(vol), _lib.ptr(th), None, 4, _lib.ptr(lat.iy), 4, _lib.ptr(lat.ix), 4, _lib.ptr(out), None)
    assert rc != 0   # 3-D without a z table
    rc = lib.trx_affine_warp_lattice_backward(ctypes.byref(vol), _lib.ptr(th), _lib.ptr(lat.iz), 4, _lib.ptr(lat.iy), 4, _lib.ptr(lat.ix), 4,
                                              _lib.ptr(out), _lib.ptr(lat.dtheta), _lib.ptr(lat.ws), 16, None)
    assert rc != 0   # workspace too small


def test_compose_theta_one_warp_equals_the_two_stage_pipeline(eng):
    """SURVEY 8f.2 (optional extra): rigid -> affine as ONE resampling.  On a smooth phantom the one-warp result and the chain of two warps
    differ only by the second interpolation (and by what the intermediate image lost at its border): 1 % of the range in the interior."""
    import torchregister_amd as tr
    shape = (48, 56, 40)
    ax = [torch.arange(n, dtype=torch.float32) for n in shape]   # smooth (wavelength >= 40 voxels): the second interpolation costs ~0.3 %
    x = (torch.sin(0.15 * ax[0])[:, None, None] * torch.cos(0.11 * ax[1])[None, :, None] + torch.sin(0.13 * ax[2] + 0.5)[None, None, :]).view(1, 1, *shape).cuda()
    first = torch.tensor(oracle.c_theta_fwd(np.array([0.05, -0.04, 0.06, 0.03, -0.02, 0.04])), dtype=torch.float32).reshape(1, 3, 4).cuda()
    second = torch.tensor([[1.03, 0.02, -0.01, 0.02], [-0.02, 0.97, 0.03, -0.03], [0.01, -0.02, 1.02, 0.01]])[None].cuda()
    chain = tr.get_affine_warp(second, tr.get_affine_warp(first, x))
    once = tr.get_affine_warp(tr.compose_theta(first, second), x)
    inner = (slice(None), slice(None), slice(6, -6), slice(6, -6), slice(6, -6))
    rng_ = (x.max() - x.min()).item()
    assert torch.max(torch.abs(chain[inner] - once[inner])).item() <= 1e-2 * rng_
    # and it is not the other order
    wrong = tr.get_affine_warp(tr.compose_theta(second, first), x)
    assert torch.max(torch.abs(chain[inner] - wrong[inner])).item() > torch.max(torch.abs(chain[inner] - once[inner])).item()

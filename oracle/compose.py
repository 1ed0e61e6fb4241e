"""ORACLE — TEST INFRASTRUCTURE ONLY (second checker + the timed CPU baseline).

The reference's hot loop re-composed from torch's own CPU ops at exactly the call sites the
reference uses, so on any box with torch the *same third-party kernels* the reference would
run (ATen affine_grid_generator / grid_sampler / SGD) act as arbiter:
  affine_warp     <- ref:warpings.py:18-26  (F.affine_grid + F.grid_sample, align_corners=False)
  flow_warp       <- ref:utils.py:350-365   (identity grid + flow, normalise, align_corners=True)
  ncc_loss        <- ref:utils.py:197-205   (global NCC, EPSILON=1e-10, alpha=100)
  nmi_loss        <- ref:utils.py:18-79, :224-259 (Parzen PDFs, NMI, NMILoss; pinned by trajectories_r2b.npz and the 2-D default-criterion run)
  pose_to_theta   <- ref:utils.py:287-310
  sgd loops       <- ref:warpings.py:67-93, :138-159, :208-233 (best = first strict minimum)
Validated against tests/golden/ in tests/test_oracle_golden.py.  This is what
bench.py reports as cpu_baseline (kind "port": the reference itself cannot travel to the GPU box).
"""
import torch
import torch.nn.functional as F

EPSILON = 1e-10


def affine_warp(theta, moving):
    nd = moving.dim() - 2
    theta = theta.reshape(-1, nd, nd + 1)
    grid = F.affine_grid(theta, list(moving.shape), align_corners=False)
    return F.grid_sample(moving, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


def identity_grid(shape, dtype=torch.float32, device="cpu"):
    axes = [torch.arange(0, s, dtype=dtype, device=device) for s in shape]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"))[None]


def flow_warp(src, flow, grid=None, mode="bilinear"):
    """ref:src/TorchRegister/utils.py:350-365 (SpatialTransformer.forward): identity grid + flow, normalised to [-1, 1] with S - 1,
    channels flipped to (x, y[, z]), grid_sample(align_corners=True, mode=mode) - `mode` 'bilinear' or 'nearest' as the reference passes it."""
    shape = flow.shape[2:]
    nd = len(shape)
    if grid is None:
        grid = identity_grid(shape, flow.dtype, flow.device)
    loc = grid + flow
    loc = torch.stack([2 * (loc[:, i] / (shape[i] - 1) - 0.5) for i in range(nd)], dim=1)
    loc = loc.movedim(1, -1).flip(-1)  # channel-last, (x, y[, z]) order
    return F.grid_sample(src, loc, mode=mode, padding_mode="zeros", align_corners=True)


def flow_warp_nearest_voxel_space(src, flow):
    """The same nearest warp restated in VOXEL space (what the HIP kernel computes): out = src[rint(voxel + flow)], zeros outside, rint =
    round half to even.  Differs from flow_warp(mode='nearest') only where voxel + flow is within ~1e-4 of a half-integer (the reference's
    normalise / un-normalise round trip decides the side there)."""
    shape = flow.shape[2:]
    nd = len(shape)
    pos = identity_grid(shape, flow.dtype, flow.device) + flow
    idx = torch.round(pos)   # torch.round rounds half to even
    ok = torch.ones_like(idx[:, 0], dtype=torch.bool)
    lin = torch.zeros_like(idx[:, 0], dtype=torch.long)
    for i in range(nd):
        ok &= (idx[:, i] >= 0) & (idx[:, i] < shape[i])
        lin = lin * shape[i] + idx[:, i].clamp(0, shape[i] - 1).long()
    flat = src.reshape(src.shape[0], src.shape[1], -1)
    out = torch.gather(flat, 2, lin.reshape(lin.shape[0], 1, -1).expand(-1, src.shape[1], -1)).reshape(src.shape)
    return out * ok[:, None].to(src.dtype)


def ncc_loss(y, yp, alpha=100.0):
    a = y - y.mean()
    b = yp - yp.mean()
    ncc = (a * b).sum() / ((a * a).sum() * (b * b).sum() + EPSILON) ** 0.5
    return (1 - ncc) * alpha


def local_ncc_loss(y, yp, window=9, alpha=1.0, eps=1e-5):
    """Box-window NCC (VoxelMorph-style) - the definition the HIP extension trx_lncc_loss_grad implements; the
    reference has no local NCC (its NCCLoss, ref:src/TorchRegister/utils.py:182-205, is global), so this torch
    composition IS the specification ("parity unpinned").  y = target, yp = warped, [B,1,*spatial]; returns the
    mean over the batch of alpha * (1 - mean_q cc(q))."""
    nd = y.dim() - 2
    conv = F.conv3d if nd == 3 else F.conv2d
    filt = torch.ones(1, 1, *([window] * nd), dtype=y.dtype, device=y.device)
    n = float(window ** nd)

    def box(x):
        return conv(x, filt, padding=window // 2)

    s_i, s_j, s_ii, s_jj, s_ij = box(y), box(yp), box(y * y), box(yp * yp), box(y * yp)
    cross = s_ij - s_i * s_j / n
    var_i = s_ii - s_i * s_i / n
    var_j = s_jj - s_j * s_j / n
    cc = cross * cross / (var_i * var_j + eps)
    return alpha * (1 - cc.mean())


def mse_loss(y, yp):
    return F.mse_loss(yp, y)


def ssd_loss(y, yp, alpha=3.0):
    return ((y - yp) ** 2).sum() * alpha


def parzen_pdf(data, steps=256, bandwidth=3.0):
    """ref:utils.py:18-51 (K_gauss, PDF_xis, get_pdf): per row of `data` (flattened) the Parzen estimate at `steps` points running from the
    largest to the smallest value of ALL rows - the reference calls them (min_val, max_val) = (max, min) - on a float32 sample line whatever
    the data's dtype; kernel exp(-u^2 / 2) / (2 pi) (the reference's constant), mean over the samples, / h."""
    signals = torch.flatten(data, start_dim=1)
    line = torch.linspace(signals.max().item(), signals.min().item(), steps, dtype=torch.float32).expand(signals.shape[0], steps)
    u = (signals.unsqueeze(-1) - line.unsqueeze(1)) / bandwidth                       # [N, S, steps]
    return (1.0 / bandwidth) * torch.mean(torch.exp(-(u ** 2) / 2) / (2 * torch.pi), dim=1)


def nmi_loss(y, yp, alpha=1000.0, bins=256, patch_size=100, bandwidth=3.0):
    """ref:utils.py:53-79 (NMI) behind ref:utils.py:224-259 (NMILoss.forward): nearest re-sampling of both images to (2 patch)^nd, viewed
    as 2^nd "patches" of patch^nd samples (a plain reshape of the re-sampled volume), Parzen PDFs of target, warped and of the two stacked,
    Shannon terms with the reference's sign convention, NMI = 2 MI / (E1 + E2), loss = mean |NMI - 1| alpha."""
    nd = y.dim() - 2
    r = 2 * patch_size
    q = [F.interpolate(t, size=(r,) * nd, mode="nearest").view((2 ** nd) * t.shape[0] * t.shape[1], *([patch_size] * nd)) for t in (y, yp)]
    hists = [parzen_pdf(q[0], bins, bandwidth), parzen_pdf(q[1], bins, bandwidth), parzen_pdf(torch.stack((q[0], q[1]), dim=1), bins, bandwidth)]
    ent = []
    for hst in hists:
        p = hst / hst.sum(dim=1, keepdim=True)
        ent.append(torch.sum(p * torch.log2(p + EPSILON), dim=1))
    mi = ent[0] + ent[1] - ent[2]
    nmi = 2 * mi / (ent[0] + ent[1])
    return torch.mean(torch.abs(nmi - 1.0) * alpha)


def weighted_loss(y, yp, w_mse=0.0, w_ncc=0.0, ncc_alpha=100.0, w_ssd=0.0, ssd_alpha=3.0):
    e = 0.0
    if w_mse:
        e = e + w_mse * mse_loss(y, yp)
    if w_ncc:
        e = e + w_ncc * ncc_loss(y, yp, ncc_alpha)
    if w_ssd:
        e = e + w_ssd * ssd_loss(y, yp, ssd_alpha)
    return e


def pose_to_theta(x, max_translate=0.25):
    if x.numel() == 6:
        psi, th, phi = x[0], x[1], x[2]
        c, s = torch.cos, torch.sin
        t = max_translate * torch.tanh(x[3:6])
        rows = [c(psi) * c(th), s(phi) * s(psi) * c(th) - c(phi) * s(th), c(phi) * s(psi) * c(th) + s(phi) * s(th), t[0],
                c(psi) * s(th), s(phi) * s(psi) * s(th) + c(phi) * c(th), c(phi) * s(psi) * s(th) - s(phi) * c(th), t[1],
                -s(psi), s(phi) * c(psi), c(phi) * c(psi), t[2]]
        return torch.stack(rows).view(1, 3, 4)
    a = x[0]
    return torch.stack([torch.cos(a), -torch.sin(a), x[1], torch.sin(a), torch.cos(a), x[2]]).view(1, 2, 3)


def affine_loop(moving, target, lr, iters, pose0=None, optimizer="sgd", theta0=None, **loss_kw):
    """SGD (or Adam, extension) on theta / pose; returns losses, thetas[iters+1], best idx.  theta0: start of the affine mode (default
    identity, the reference's start: ref:src/TorchRegister/warpings.py:42-55)."""
    nd = moving.dim() - 2
    dt = moving.dtype
    if pose0 is not None:
        p = pose0.clone().to(dt).requires_grad_()
        make = lambda: pose_to_theta(p)  # noqa: E731
    else:
        p = (torch.eye(nd, nd + 1, dtype=dt) if theta0 is None else theta0.to(dt).reshape(nd, nd + 1))[None].clone().requires_grad_()
        make = lambda: p  # noqa: E731
    opt = torch.optim.SGD([p], lr) if optimizer == "sgd" else torch.optim.Adam([p], lr)
    losses, thetas, best, best_idx = [], [], None, -1
    for t in range(iters):
        opt.zero_grad()
        th = make()
        thetas.append(th.detach()[0].clone())
        e = weighted_loss(target, affine_warp(th, moving), **loss_kw)
        e.backward()
        opt.step()
        v = e.item()
        losses.append(v)
        if best is None or v < best:
            best, best_idx = v, t
    thetas.append(make().detach()[0].clone())
    return dict(losses=torch.tensor(losses, dtype=torch.float64), thetas=torch.stack(thetas), best_idx=best_idx)


def smooth_regulariser(flow, weight):
    """Extension (not in the reference): weight / ndim * sum_d mean_{c,p} (forward difference of the flow along d)^2 - the definition
    of the fused flow kernels' smoothness term (torchregister_amd/csrc/flow.hip, flow_coef_kernel), restated with torch ops."""
    nd = flow.dim() - 2
    reg = 0.0
    for d in range(nd):
        df = flow.diff(dim=2 + d)
        reg = reg + (df * df).mean()
    return weight * reg / nd


def flow_loop(moving, target, lr, iters, optimizer="sgd", flow0=None, smooth_weight=0.0, **loss_kw):
    """ref:src/TorchRegister/warpings.py:208-233 with the flow field itself as the parameter (flow_model='direct'); flow0: its start
    (default zero); smooth_weight: the smoothness extension above."""
    nd = moving.dim() - 2
    shape = moving.shape[2:]
    fl = (torch.zeros(1, nd, *shape, dtype=moving.dtype) if flow0 is None else flow0.to(moving.dtype).clone()).requires_grad_()
    grid = identity_grid(shape, moving.dtype)
    opt = torch.optim.SGD([fl], lr) if optimizer == "sgd" else torch.optim.Adam([fl], lr)
    losses = []
    for _ in range(iters):
        opt.zero_grad()
        e = weighted_loss(target, flow_warp(moving, fl, grid), **loss_kw)
        if smooth_weight:
            e = e + smooth_regulariser(fl, smooth_weight)
        e.backward()
        opt.step()
        losses.append(e.item())
    return dict(losses=torch.tensor(losses, dtype=torch.float64), flow=fl.detach(),
                final_warped=flow_warp(moving, fl.detach(), grid))

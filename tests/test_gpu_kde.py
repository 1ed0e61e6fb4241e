"""GPU parity of the Parzen-window PDF kernels (trx_kde_pdf / trx_kde_pdf_backward) that stand in for the reference's
PDF_xis (ref:utils.py:24-30) inside NMI: against the reference's own formulation (the [N, S, bins] difference tensor)
evaluated with torch in fp64 on the CPU; bar = max(floor, 2 x the fp32 run's own gap).  Floors: pdf 2e-6 rel (fp32 sums of thousands of exp2 terms), gradient
1e-5 of max."""
import numpy as np
import pytest
import torch

import phantoms as ph

pytestmark = pytest.mark.gpu


def _ref_pdf(signals, xis, h):
    diff = signals.unsqueeze(-1) - xis.unsqueeze(1)
    return (1 / h) * torch.mean((1 / (2 * torch.pi)) * torch.exp(-((diff / h) ** 2) / 2), dim=1)


@pytest.mark.parametrize("series", ["1", "0"])
@pytest.mark.parametrize("N,S,bins,h", [(1, 1000, 256, 3.0), (3, 4096 + 17, 256, 0.1), (2, 30000, 64, 0.5), (8, 5000, 256, 3.0), (1, 7, 5, 1.0),
                                        (2, 9000, 256, 1.5), (1, 4097, 1024, 1.6), (3, 33, 3, 2.0)])
def test_kde_pdf_forward_backward(N, S, bins, h, series, monkeypatch):
    """Both forms of the kernels: one exponential per (sample, bin) pair, and - where the window is at least as wide as the value range
    (h >= 1.5 here) - the series form (25 power sums + a polynomial per bin); TRX_KDE_SERIES=0 forces the first."""
    import torchregister_amd.utils as U
    monkeypatch.setenv("TRX_KDE_SERIES", series)
    g = torch.Generator().manual_seed(S + bins)
    sig = torch.rand(N, S, generator=g) * 1.5 - 0.2
    hi, lo = sig.max().item(), sig.min().item()
    xis = torch.linspace(hi, lo, bins).repeat(N, 1)
    wts = torch.rand(N, bins, generator=g) - 0.3          # arbitrary downstream gradient
    outs = {}
    for dt in (torch.float32, torch.float64):
        s = sig.detach().clone().to(dt).requires_grad_()
        p = _ref_pdf(s, xis.to(dt), h)
        (p * wts.to(dt)).sum().backward()
        outs[dt] = (p.detach().numpy(), s.grad.numpy())
    sc = sig.detach().clone().cuda().requires_grad_()
    pc = U.PDF_xis(sc, xis.cuda(), h)
    (pc * wts.cuda()).sum().backward()
    p32, g32 = outs[torch.float32]
    p64, g64 = outs[torch.float64]
    ep, bp = np.max(np.abs(pc.detach().cpu().numpy() - p64)), max(2e-6 * np.max(np.abs(p64)), 2 * np.max(np.abs(p32 - p64)))
    eg, bg = np.max(np.abs(sc.grad.cpu().numpy() - g64)), max(1e-5 * np.max(np.abs(g64)), 2 * np.max(np.abs(g32 - g64)))
    assert ep <= bp, (ep, bp)
    assert eg <= bg, (eg, bg)


def test_nmi_loss_matches_torch_formulation():
    """NMILoss (ref:utils.py:224-259) on 2-D images: the HIP-backed PDF inside NMI vs the all-torch formulation in fp64."""
    import torchregister_amd.utils as U
    shape = (64, 80)
    y = ph.blobs(shape, 5)
    yp = (ph.blobs(shape, 6) + 0.1 * ph.vol(shape, 0.37, "sin")).requires_grad_()
    crit = U.NMILoss()
    # reference formulation on the CPU (PDF_xis takes the torch branch for CPU tensors) in fp64 and in fp32: |NMI - 1| * 1000 of
    # the nearly flat PDFs that bandwidth 3 produces is ill-conditioned, the fp32 run's own gap sets the bar
    refs = {}
    for dt in (torch.float32, torch.float64):
        ypr = yp.detach().to(dt).clone().requires_grad_()
        r = crit(y.to(dt), ypr)
        r.backward()
        refs[dt] = (r.item(), ypr.grad.double())
    ypc = yp.detach().cuda().requires_grad_()
    got = crit(y.cuda(), ypc)
    got.backward()
    l32, g32 = refs[torch.float32]
    l64, g64 = refs[torch.float64]
    assert abs(got.item() - l64) <= max(1e-5 * abs(l64), 2 * abs(l32 - l64)), (got.item(), l32, l64)
    gmax = g64.abs().max().item()
    assert (ypc.grad.cpu().double() - g64).abs().max().item() <= max(1e-4 * gmax, 2 * (g32 - g64).abs().max().item())


@pytest.mark.parametrize("N,bins", [(4, 256), (8, 256), (1, 64), (3, 1000)])
def test_nmi_algebra_kernel_vs_torch_autograd(N, bins):
    """trx_nmi_from_pdfs (normalisation, entropies, NMI, alpha * mean|NMI - 1| and d loss / d PDFs in one kernel) against the same
    algebra written with torch in fp64 (ref:utils.py:62-79, :257-258)."""
    import torchregister_amd._engine as eng
    g = torch.Generator().manual_seed(N * 1000 + bins)
    hs = [(0.05 + torch.rand(N, bins, generator=g)) * (0.5 + torch.rand(N, 1, generator=g)) for _ in range(3)]
    hs[2] = 0.5 * (hs[0] + hs[1]) * (0.7 + 0.6 * torch.rand(N, bins, generator=g))
    alpha = 1000.0
    hd = [h.double().requires_grad_() for h in hs]
    ps = [h / h.sum(dim=1, keepdim=True) for h in hd]
    e = [torch.sum(p * torch.log2(p + 1e-10), dim=1) for p in ps]
    mi = e[0] + e[1] - e[2]
    nmi = 2 * mi / (e[0] + e[1])
    loss = torch.mean(torch.abs(nmi - 1.0) * alpha)
    loss.backward()
    gn, gm, terms, grads = eng.nmi_from_pdfs(*[h.cuda() for h in hs], alpha)
    assert np.max(np.abs(gn.cpu().numpy() - nmi.detach().numpy())) <= 2e-6 * np.max(np.abs(nmi.detach().numpy()))
    assert np.max(np.abs(gm.cpu().numpy() - mi.detach().numpy())) <= 2e-6 * np.max(np.abs(mi.detach().numpy())) + 1e-7
    assert abs(terms.sum().item() - loss.item()) <= 2e-6 * abs(loss.item())
    for got, h in zip(grads, hd):
        ref = h.grad.numpy()
        assert np.max(np.abs(got.cpu().numpy() - ref)) <= 2e-6 * np.max(np.abs(ref))


def test_nmi_loss_caches_the_target_side_only_while_it_is_unchanged():
    import torchregister_amd.utils as U
    shape = (48, 48, 48)
    y, yp = ph.blobs(shape, 5).cuda(), ph.blobs(shape, 6).cuda()
    crit = U.NMILoss()
    a = crit(y, yp).item()
    assert crit._cache.get("yref") is y and "h1" in crit._cache
    b = crit(y, yp).item()                       # second call: the target's patches / extrema / PDF come from the cache
    assert a == b
    fresh = U.NMILoss()(y, yp).item()
    assert a == fresh
    y.mul_(1.5)                                  # in-place change of the target: the cache must not be used
    c = crit(y, yp).item()
    assert c == U.NMILoss()(y, yp).item() and c != a


def test_nmi_patches_gradient_is_the_gather_adjoint():
    """NMILoss's nearest down-sampling (ref:utils.py:236-252) on the GPU: values = F.interpolate(nearest), gradient = the exact adjoint
    of that gather (what ATen's CPU kernel - the reference's platform - computes; the device kernel's own backward misplaces some
    gradients for non-integer ratios, which this test also demonstrates so that the work-around can be dropped when that changes)."""
    import torch.nn.functional as F
    import torchregister_amd.utils as U
    nmi = U.NMILoss(patch_size=10)
    x = torch.rand(1, 1, 40, 36, 52, device="cuda").requires_grad_()
    p = nmi._patches(x)
    assert torch.equal(p.reshape(-1), F.interpolate(x.detach(), size=(20, 20, 20), mode="nearest").reshape(-1))
    w = torch.rand_like(p)
    (p * w).sum().backward()
    xc = x.detach().cpu().requires_grad_()
    (F.interpolate(xc, size=(20, 20, 20), mode="nearest").reshape(p.shape) * w.cpu()).sum().backward()
    assert torch.equal(x.grad.cpu(), xc.grad)
    xd = x.detach().clone().requires_grad_()
    (F.interpolate(xd, size=(20, 20, 20), mode="nearest").reshape(p.shape) * w).sum().backward()
    if torch.equal(xd.grad.cpu(), xc.grad):
        pytest.skip("this ATen build's device nearest-backward agrees with the CPU kernel: the gather work-around is no longer needed")


def test_kde_series_cached_sums_and_strided_lines():
    """trx_kde_pdf_series_cached: the PDF of the same samples on another (possibly strided) sample line from the power sums an earlier
    call left behind = a fresh trx_kde_pdf_series call, bit for bit (same kernel, same sums)."""
    import torchregister_amd._engine as eng
    g = torch.Generator().manual_seed(5)
    sig = (torch.rand(4, 20000, generator=g) * 0.9 + 0.05).cuda()
    h, center = 3.0, 0.5
    sums = eng.KdeSums(sig, h, center)
    wide = torch.rand(4, 2 * 64, generator=g).cuda()
    for xis in (wide[:, :64], wide[:, 64:], wide):
        want = eng.kde_pdf(sig, xis.contiguous(), h, center)
        assert torch.equal(sums.pdf(xis), want)


def test_nmi_from_pdfs_pooled_equals_manual_pooling():
    import torchregister_amd._engine as eng
    g = torch.Generator().manual_seed(6)
    N, bins = 8, 256
    h1 = (torch.rand(N, bins, generator=g) + 0.1).cuda()
    pdf_w = (torch.rand(N, 2 * bins, generator=g) + 0.1).cuda()
    pdf_t = (torch.rand(N, bins, generator=g) + 0.1).cuda()
    terms, gw = eng.nmi_from_pdfs_pooled(h1, pdf_w, pdf_t, 0.7)
    hj = 0.5 * (pdf_w[:, bins:] + pdf_t)
    _, _, terms2, (_, g2, gj) = eng.nmi_from_pdfs(h1, pdf_w[:, :bins].contiguous(), hj, 0.7)
    assert torch.equal(terms, terms2)
    assert torch.equal(gw, torch.cat([g2, 0.5 * gj], dim=1))


@pytest.mark.parametrize("spatial,size", [((40, 36, 52), (20, 20, 20)), ((33, 57), (50, 16))])
def test_nmi_lattice_lines_vs_torch_composition(spatial, size):
    """trx_nmi_lattice_lines: the lattice values equal trx_affine_warp_lattice's and the two sample lines equal aminmax + lerp +
    maximum / minimum + cat of the torch composition (to an ulp of the line's span: the device may contract a + w * d)."""
    import torchregister_amd._engine as eng
    nd = len(spatial)
    mov = ph.vol(spatial, 0.37, "sin").cuda() * 0.5 + 0.5
    th = torch.tensor([[1.02, 0.03, -0.02, 0.05], [-0.03, 0.98, 0.02, -0.04], [0.01, -0.02, 1.03, 0.02]] if nd == 3 else [[1.01, 0.04, 0.03], [-0.05, 0.97, -0.02]])
    thp = eng.pad_theta(th.reshape(1, -1).cuda(), nd)
    vol = eng._Batch(mov, mov).vol()
    lat = eng.LatticeWarp(vol, spatial, size, mov.device)
    P, bins = 2 ** nd, 256
    for tlo, thi in ((0.2, 0.7), (-0.5, 1.5)):      # target extrema inside / outside the warped range
        mm_t = torch.tensor([[tlo, thi]], device="cuda")
        vals, xis = lat.forward_lines(thp, mm_t, P, bins)
        assert torch.equal(vals, lat.forward(thp))
        plo, phi = torch.aminmax(vals)
        ramp = (torch.arange(bins, device="cuda", dtype=torch.float32) / (bins - 1)).expand(P, bins).contiguous()
        want = torch.cat([torch.lerp(phi, plo, ramp), torch.lerp(torch.maximum(phi, mm_t[0, 1]), torch.minimum(plo, mm_t[0, 0]), ramp)], dim=1)
        assert xis.shape == want.shape
        assert torch.max(torch.abs(xis - want)).item() <= 2.5e-7 * max(1.0, thi - tlo)
        assert xis[0, 0].item() == phi.item() and xis[0, bins - 1].item() == plo.item()   # the end points are the extrema themselves


@pytest.mark.parametrize("nd,rigid", [(3, False), (3, True), (2, False), (2, True)])
def test_nmi_loop_update_vs_torch_ops(nd, rigid):
    """trx_nmi_loop_update = hist_theta[t] <- theta; loss <- sum(terms) + fused loss; theta <- theta - lr (g_a + g_b)  (rigid: through
    trx_theta_chain, as the loop did with separate launches before)."""
    import ctypes
    import torchregister_amd._engine as eng
    from torchregister_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(10 * nd + rigid)
    PS = eng.PSTRIDE
    nt, npose = nd * (nd + 1), (6 if nd == 3 else 3)
    pose = torch.zeros(1, PS, device="cuda"); pose[0, :npose] = (torch.rand(npose, generator=g) * 0.5).cuda()
    theta = torch.zeros(1, PS, device="cuda")
    if rigid:
        _lib.check(lib.trx_theta_chain(_lib.ptr(pose), None, nd, 1, _lib.ptr(theta), None, None), "chain")
    else:
        theta[0, :nt] = (torch.eye(nd, nd + 1).reshape(-1) + 0.05 * torch.rand(nt, generator=g)).cuda()
    ga, gb = torch.zeros(1, PS, device="cuda"), torch.zeros(1, PS, device="cuda")
    ga[0, :nt] = torch.randn(nt, generator=g).cuda(); gb[0, :nt] = torch.randn(nt, generator=g).cuda()
    terms = torch.rand(2 ** nd, generator=g).cuda()
    loss_b = torch.rand(1, generator=g).cuda()
    lr = 0.013
    # expected, with the separate launches the loop used before
    want_loss = terms.sum() + loss_b[0]
    gsum = ga + gb
    if rigid:
        dpose, p2, th2 = torch.zeros_like(pose), pose.clone(), torch.zeros_like(theta)
        _lib.check(lib.trx_theta_chain(_lib.ptr(pose), _lib.ptr(gsum), nd, 1, None, _lib.ptr(dpose), None), "chain")
        p2.sub_(dpose, alpha=lr)
        _lib.check(lib.trx_theta_chain(_lib.ptr(p2), None, nd, 1, _lib.ptr(th2), None, None), "chain")
    else:
        th2 = theta - lr * gsum
    th_before = theta.clone()
    hist_loss, hist_theta, param = torch.zeros(1, device="cuda"), torch.zeros(PS, device="cuda"), torch.zeros(1, PS, device="cuda")
    _lib.check(lib.trx_nmi_loop_update(nd, _lib.ptr(theta), _lib.ptr(pose) if rigid else None, _lib.ptr(ga), _lib.ptr(gb), lr, _lib.ptr(terms), 2 ** nd,
                                       _lib.ptr(loss_b), _lib.ptr(hist_loss), _lib.ptr(hist_theta), _lib.ptr(param), None), "update")
    torch.cuda.synchronize()
    assert torch.equal(hist_theta, th_before[0])
    assert abs(hist_loss.item() - want_loss.item()) <= 1e-6 * abs(want_loss.item())
    assert torch.max(torch.abs(theta[0, :nt] - th2[0, :nt])).item() <= 2e-7
    assert torch.equal(param, theta)
    if rigid:
        assert torch.max(torch.abs(pose - p2)).item() <= 1e-7

"""Building blocks mirrored from the reference's utils module (same names and argument meaning).

ref: = /root/reference/src/TorchRegister/utils.py.  Losses are ordinary torch modules (they run
on the GPU through PyTorch-ROCm) so they can be used stand-alone exactly like the reference's;
inside Register.optim the MSE / NCC / SSD family is recognised and replaced by the fused HIP
kernels (see warpings.loss_spec_from).  SpatialTransformer is backed by the HIP flow kernels.
"""
import os

import torch
import torch.nn as nn
from torch.nn import functional as F

from . import _engine

EPSILON = 1e-10  # ref:utils.py:15


def norm(x):
    """Min-max normalise to [0,1] (ref:utils.py:262-267; its own epsilon is 1e-9)."""
    try:
        return (x - torch.min(x)) / ((torch.max(x) - torch.min(x)) + 1e-9)
    except Exception:
        print("WARNING: Input could not be normalized!")


def padNd(input_, target, device="cpu", mode="constant", value=0):
    """Centre-pad input_ to target's spatial size; an odd difference puts the extra voxel at the END of
    the axis (the reference flips its [ceil, floor] pairs before handing them to F.pad, ref:utils.py:271-277)."""
    dims = input_.dim() - 2
    pads = []
    for i in reversed(range(dims)):
        delta = target.shape[2 + i] - input_.shape[2 + i]
        back = -(-delta // 2)
        pads += [delta - back, back]
    return F.pad(input_, tuple(pads), mode=mode, value=value).to(dtype=torch.float, device=device)


class NCCLoss(nn.Module):
    """alpha * (1 - global NCC) (ref:utils.py:186-205). `grad_edges` / `device` are accepted and ignored
    like in the reference.  Call order is (target, warped)."""

    def __init__(self, alpha=100, grad_edges=True, device="cpu"):
        super().__init__()
        self.NCC = None
        self.alpha = alpha

    def forward(self, y, yp):
        a = y - torch.mean(y)
        b = yp - torch.mean(yp)
        self.NCC = torch.sum(a * b) / ((torch.sum(a ** 2) * torch.sum(b ** 2) + EPSILON) ** 0.5)
        return (1 - self.NCC) * self.alpha


class _LocalNCCFn(torch.autograd.Function):
    """loss [B] = local-window NCC of (target, warped) through the HIP kernels; gradient wrt warped only."""

    @staticmethod
    def forward(ctx, y, yp, window, alpha, eps):
        from . import _engine
        loss, grad = _engine.local_ncc_loss_grad(y, yp, window, alpha, eps, need_grad=yp.requires_grad)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, gl):
        (grad,) = ctx.saved_tensors
        if grad is None:
            return None, None, None, None, None
        return None, grad * gl.view(-1, *([1] * (grad.dim() - 1))), None, None, None


class LocalNCCLoss(nn.Module):
    """alpha * (1 - mean of the squared local NCC over `window`^nd boxes) - the box-window NCC of VoxelMorph-style
    registration.  EXTENSION: the reference's NCCLoss is global; this criterion exists only here (north_star,
    SURVEY 8f.3) and is defined by oracle/compose.py::local_ncc_loss.  Runs on the GPU only (HIP kernels
    trx_lncc_loss_grad); call order (target, warped) like every criterion of the package; batch mean."""

    def __init__(self, window=9, alpha=1.0, eps=1e-5):
        super().__init__()
        if window not in (3, 5, 7, 9):
            raise ValueError("window must be 3, 5, 7 or 9")
        self.window, self.alpha, self.eps = int(window), float(alpha), float(eps)

    def forward(self, y, yp):
        return _LocalNCCFn.apply(y, yp, self.window, self.alpha, self.eps).mean()


class SSDLoss(nn.Module):
    """alpha * sum((y - yp)^2) (ref:utils.py:208-221)."""

    def __init__(self, alpha=3):
        super().__init__()
        self.SSD = None
        self.alpha = alpha

    def forward(self, y, yp):
        self.SSD = torch.sum((y - yp) ** 2)
        return self.SSD * self.alpha


# ---- Parzen-window NMI (ref:utils.py:18-79, 224-259): plain torch ops, NOT on the fused path --------
def K_gauss(input_):
    return (1 / (2 * torch.pi)) * torch.exp(-(input_ ** 2) / 2)


class _KdePdfFn(torch.autograd.Function):
    """PDF_xis on the GPU without the [N, S, bins] tensor (HIP kernels trx_kde_pdf / trx_kde_pdf_backward)."""

    @staticmethod
    def forward(ctx, signals, xis, h, value_range=None):
        ctx.save_for_backward(signals, xis)
        ctx.h = float(h)
        # window wide against the data (the NMI loss's bandwidth 3 on normalised intensities): series form, O(S + bins) instead of O(S * bins)
        ctx.center = None
        if os.environ.get("TRX_KDE_SERIES", "1") != "0":
            if value_range is not None:      # (lo, hi) of signals and xis already known on the host (get_pdf): no extra sync
                lo, hi = value_range
                ctx.center = 0.5 * (lo + hi) if (hi - lo) <= float(h) else None
            else:
                ctx.center = _engine.kde_series_center(signals, xis, h)
        return _engine.kde_pdf(signals, xis, h, ctx.center)

    @staticmethod
    def backward(ctx, g):
        signals, xis = ctx.saved_tensors
        gs = _engine.kde_pdf_backward(signals, xis, g, ctx.h, ctx.center) if ctx.needs_input_grad[0] else None
        return gs, None, None, None


def PDF_xis(signals, xis, h=3, value_range=None):
    if signals.is_cuda and signals.dim() == 2 and xis.dim() == 2 and xis.shape[1] <= 1024 and signals.dtype == torch.float32:
        return _KdePdfFn.apply(signals, xis, h, value_range)  # same numbers, 4 B per sample instead of 4 * bins
    diff = signals.unsqueeze(-1) - xis.unsqueeze(1)          # [N, S, bins] (the reference's formulation)
    return (1 / h) * torch.mean(K_gauss(diff / h), dim=1)


def PDF(signals, Xs, h=3, value_range=None):
    return PDF_xis(signals, Xs, h, value_range)


def get_pdf(data, steps=256, bandwidth=2, value_range=None):
    signals = torch.flatten(data, start_dim=1)
    # the reference names these (min, max) but takes (max, min): the sample line runs max -> min
    if value_range is None:
        hi, lo = torch.max(signals).item(), torch.min(signals).item()
    else:                     # (hi, lo) already on the host (NMI fetches the extrema of both images in one sync)
        hi, lo = value_range
    line = torch.linspace(hi, lo, steps, dtype=torch.float, device=signals.device) * torch.ones(
        (len(data), steps), dtype=torch.float, device=signals.device)
    return PDF(signals, line, h=bandwidth, value_range=(lo, hi))   # the sample line spans exactly the signals' range


def _nmi_pdfs(img1, img2, bins, bandwidth, cache=None):
    """The three Parzen "histograms" of NMI (ref:utils.py:55-60).  The six .item() calls of the reference's three get_pdf
    (ref:utils.py:40-48) are one host sync (the joint sample's extrema are those of the two images).  `cache`: a dict owned by the
    caller in which the quantities that depend on img1 only (its extrema and PDF) survive from call to call while img1 is the same
    unmodified tensor - in a registration loop img1 is the fixed target."""
    key = None
    if cache is not None and not img1.requires_grad:
        key = (img1._version, int(bins), float(bandwidth))
    # identity, not address: the cache holds a reference to img1, so `is` cannot be fooled by an allocator that hands a freed
    # tensor's address (and version 0) to the next one of the same shape
    if key is not None and cache.get("img1") is img1 and cache.get("key") == key:
        hi1, lo1, h1 = cache["hi1"], cache["lo1"], cache["h1"]
        hi2, lo2 = torch.stack([img2.detach().amax(), img2.detach().amin()]).tolist()
    else:
        hi1, lo1, hi2, lo2 = torch.stack([img1.detach().amax(), img1.detach().amin(), img2.detach().amax(), img2.detach().amin()]).tolist()
        h1 = get_pdf(img1, steps=bins, bandwidth=bandwidth, value_range=(hi1, lo1))
        if key is not None:
            cache.update(key=key, img1=img1, hi1=hi1, lo1=lo1, h1=h1.detach())
    h2 = get_pdf(img2, steps=bins, bandwidth=bandwidth, value_range=(hi2, lo2))
    hj = get_pdf(torch.stack((img1, img2), dim=1), steps=bins, bandwidth=bandwidth, value_range=(max(hi1, hi2), min(lo1, lo2)))
    return h1, h2, hj


class _NmiAlgebraFn(torch.autograd.Function):
    """alpha * mean|NMI - 1| from the three PDFs and its gradient wrt them, ONE kernel (trx_nmi_from_pdfs) in place of ~25 torch
    launches forward and ~40 backward on [N, 256] tensors.  Returns (loss, nmi, mi); only `loss` carries a gradient."""

    @staticmethod
    def forward(ctx, h1, h2, hj, alpha):
        nmi, mi, terms, grads = _engine.nmi_from_pdfs(h1, h2, hj, alpha, need_grad=True)
        ctx.save_for_backward(*grads)
        ctx.mark_non_differentiable(nmi, mi)
        return terms.sum(), nmi, mi

    @staticmethod
    def backward(ctx, gl, _gn, _gm):
        g1, g2, gj = ctx.saved_tensors
        need = ctx.needs_input_grad
        return (g1 * gl if need[0] else None), (g2 * gl if need[1] else None), (gj * gl if need[2] else None), None


def NMI(img1, img2, bins=256, bandwidth=0.1):
    """ref:utils.py:53-79: (normalised mutual information, mutual information) per sample of the batch, differentiable."""
    h1, h2, hj = _nmi_pdfs(img1, img2, bins, bandwidth)
    p1 = h1 / h1.sum(dim=1, keepdim=True)
    p2 = h2 / h2.sum(dim=1, keepdim=True)
    pj = hj / hj.sum(dim=1, keepdim=True)
    # sign convention of the reference: E = -sum(p * -log2(p + eps)) = +sum(p log2 p)
    e1 = torch.sum(p1 * torch.log2(p1 + EPSILON), dim=1)
    e2 = torch.sum(p2 * torch.log2(p2 + EPSILON), dim=1)
    ej = torch.sum(pj * torch.log2(pj + EPSILON), dim=1)
    mi = e1 + e2 - ej
    return 2 * mi / (e1 + e2), mi


class NMILoss(nn.Module):
    """alpha * mean|NMI - 1| on 2^d patches of a nearest-resampled 200^d copy (ref:utils.py:224-259).  On the GPU the PDFs are HIP
    kernels (csrc/kde.hip) and everything behind them - normalisation, entropies, NMI, |NMI - 1|, and the backward of all of it - is one
    more (trx_nmi_from_pdfs); the target's patches and PDF are kept between calls while the target tensor is unchanged."""

    def __init__(self, alpha=1000, bins=256, patch_size=100, bandwidth=3):
        super().__init__()
        self.bins, self.alpha, self.patch, self.bandwidth = bins, alpha, patch_size, bandwidth
        self._cache = {}
        self._lattice = {}

    def _patches(self, t):
        r = self.patch * 2
        nd = t.dim() - 2
        if t.is_cuda and t.requires_grad:
            # Same values as F.interpolate(nearest) (the index tables come from that very call), but differentiated as a gather.
            # ATen's DEVICE kernel for the nearest backward inverts the index map with its own float arithmetic and, for non-integer
            # size ratios, hands some gradients to a neighbouring voxel (52 -> 20: output 5 reads input 13, its gradient lands on 12);
            # the CPU kernel the reference runs on uses the forward index function both ways, as index_select's backward does.
            from ._engine import nearest_lattice
            key = (tuple(t.shape[2:]), r, t.device)
            tabs = self._lattice.get(key)
            if tabs is None:
                tabs = self._lattice[key] = [i.long() for i in nearest_lattice(t.shape[2:], (r,) * nd, t.device)]
            for d, tab in enumerate(tabs):
                t = t.index_select(2 + d, tab)
            t = t.contiguous()
        else:
            t = F.interpolate(t, size=(r,) * nd, mode="nearest")
        return t.view((2 ** nd) * t.shape[0] * t.shape[1], *([self.patch] * nd))

    def forward(self, y, yp):
        fused = y.is_cuda and yp.is_cuda and y.dtype == torch.float32 and self.bins <= 1024
        if fused and not y.requires_grad:
            # the target's patches / extrema / PDF survive while `y` is the SAME tensor object, unmodified: the cache keeps `y` itself
            # (a strong reference: its storage cannot be recycled under the cache) and compares identity + version counter
            if self._cache.get("yref") is not y or self._cache.get("yver") != y._version:
                self._cache = {"yref": y, "yver": y._version, "y": self._patches(y)}
            yq = self._cache["y"]
        else:
            yq = self._patches(y)
        ypq = self._patches(yp)
        if fused:
            h1, h2, hj = _nmi_pdfs(yq, ypq, self.bins, self.bandwidth, cache=self._cache)
            return _NmiAlgebraFn.apply(h1, h2, hj, float(self.alpha))[0]
        nmi, _ = NMI(yq, ypq, self.bins, self.bandwidth)
        return torch.mean(torch.abs(nmi - 1.0) * self.alpha)


class Theta(nn.Module):
    """Pose vector -> affine matrix entries (ref:utils.py:280-310).

    3-D: x = (psi, theta, phi, tx, ty, tz): R = Rz(theta) Ry(psi) Rx(phi), t = max_translate*tanh(.)
    2-D: x = (angle, tx, ty): [[cos, -sin, tx], [sin, cos, ty]] (translation not bounded)."""

    def __init__(self):
        super().__init__()
        self.sin, self.cos, self.tanh = torch.sin, torch.cos, torch.tanh

    def forward(self, x, max_translate=0.25):
        if len(x) > 3:
            out = _engine.pose_to_theta(x)
            if max_translate != 0.25:
                out = out.clone()
                out[3::4] = max_translate * torch.tanh(x[3:6])
            return out.flatten()
        return _engine.pose_to_theta(x).flatten()


class Regressor(nn.Module):
    """Learnable pose, torch.rand initialised on `device` (ref:utils.py:313-330)."""

    def __init__(self, moving, device):
        super().__init__()
        n = 6 if moving.dim() == 5 else 3
        self.reg = nn.Parameter(torch.rand((n), device=device), requires_grad=True)
        self.thetas = Theta()

    def forward(self):
        theta = self.thetas(self.reg)
        return theta.view(1, 3, 4) if theta.shape[-1] == 12 else theta.view(1, 2, 3)


class _FlowWarpNearestFn(torch.autograd.Function):
    """mode='nearest': piecewise constant in the flow - autograd gets the zeros ATen's grid_sampler backward produces for the grid."""

    @staticmethod
    def forward(ctx, src, flow):
        ctx.flow_meta = (flow.shape, flow.dtype, flow.device)
        return _engine.flow_warp(src, flow, nearest=True)

    @staticmethod
    def backward(ctx, grad_out):
        shape, dtype, device = ctx.flow_meta
        return None, (torch.zeros(shape, dtype=dtype, device=device) if ctx.needs_input_grad[1] else None)


class _FlowWarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, flow):
        ctx.save_for_backward(src, flow)
        return _engine.flow_warp(src, flow)

    @staticmethod
    def backward(ctx, grad_out):
        src, flow = ctx.saved_tensors
        dflow = _engine.flow_warp_backward(src, flow, grad_out) if ctx.needs_input_grad[1] else None
        return None, dflow   # no gradient flows to the moving image on this path


class SpatialTransformer(nn.Module):
    """N-D spatial transformer: sample src at voxel + flow, zeros outside (ref:utils.py:333-365).

    Same constructor/forward signature as the reference.  The reference's identity-grid buffer (nd x volume
    floats, 200 MB at 256^3, built on the CPU and copied over) is NOT created: the HIP kernel adds the voxel
    index itself.  Gradient wrt `flow` is provided (HIP backward); gradient wrt `src` is not (the reference
    never needs it)."""

    def __init__(self, size, mode="bilinear"):
        super().__init__()
        if mode not in ("bilinear", "nearest"):   # the two modes grid_sample has for 5-D inputs (bicubic is 4-D only and unused by the reference)
            raise NotImplementedError(f"mode={mode!r}: the HIP warp implements 'bilinear' and 'nearest'")
        self.mode = mode
        self.size = tuple(int(s) for s in size)

    def forward(self, src, flow):
        if tuple(flow.shape[2:]) != tuple(src.shape[2:]):
            raise ValueError("flow and src spatial sizes differ")
        if src.shape[0] != flow.shape[0]:
            flow = flow.expand(src.shape[0], *flow.shape[1:])
        return (_FlowWarpNearestFn if self.mode == "nearest" else _FlowWarpFn).apply(src, flow)


# ----------------------------------------------------------------------------------------------
# Attention U-Net that GENERATES the flow in the reference's flow mode (ref:utils.py:368-559).
# Host-side PyTorch-ROCm code (MIOpen convolutions) — SURVEY section 8f row 1; the warp at its end and
# its backward are the HIP flow kernels (SpatialTransformer above).  Modules are created in the same
# order, with the same shapes, as the reference so that a given torch seed yields the same weights.
# ----------------------------------------------------------------------------------------------
def _conv(dims):
    return (nn.Conv3d, nn.ConvTranspose3d, nn.InstanceNorm3d, nn.MaxPool3d) if dims == 3 else (nn.Conv2d, nn.ConvTranspose2d, nn.InstanceNorm2d, nn.MaxPool2d)


class attention_grid(nn.Module):
    """Attention gate: 1x1 strided filter of the skip tensor + 1x1 filter of the gating tensor -> sigmoid map,
    resampled (nearest) to the skip tensor and applied to it, then instance-normalised (ref:utils.py:368-406)."""

    def __init__(self, x_c, g_c, i_c, stride=3, mode='nearest', dims=3):
        super().__init__()
        Conv, _, Norm, _ = _conv(dims)
        self.input_filter = Conv(in_channels=x_c, out_channels=i_c, kernel_size=1, stride=stride, bias=False)
        self.gate_filter = Conv(in_channels=g_c, out_channels=i_c, kernel_size=1, stride=1, bias=True)
        self.psi = Conv(in_channels=i_c, out_channels=1, kernel_size=1, stride=1, bias=True)
        self.bnorm = Norm(i_c)
        self.mode = mode

    def forward(self, x, g, device):
        feats = [self.input_filter(x), self.gate_filter(g)]   # strided view of the skip tensor, gating signal: sizes differ by a voxel or two
        big = max(feats, key=lambda t: t.shape[-1])
        gate = sum(t if t.shape[-1] == big.shape[-1] else padNd(t, big, device) for t in feats)
        att = F.interpolate(torch.sigmoid(self.psi(F.relu(gate))), size=x.shape[2:], mode=self.mode)
        return self.bnorm(x * att), att


class Attention_UNet(nn.Module):
    """9-stage valid-convolution U-Net with attention-gated skips; channels (64,128,256,512,1024)/n; input =
    the moving image, output = (warped, flow[nd]) (ref:utils.py:409-559)."""

    def __init__(self, img_size, mode='nearest', in_c=1, n=1):
        super().__init__()
        dims = len(img_size)
        Conv, ConvT, Norm, Pool = _conv(dims)
        c = [int(v / n) for v in (64, 128, 256, 512, 1024)]

        def double(i, o):
            return [Conv(in_channels=i, out_channels=o, kernel_size=3), nn.ReLU(), Norm(o),
                    Conv(in_channels=o, out_channels=o, kernel_size=3), nn.ReLU(), Norm(o)]

        def up(i, o):
            return [ConvT(in_channels=i, out_channels=o, kernel_size=2, stride=2), nn.ReLU(), Norm(o)]

        # creation order matters for seeded-initialisation parity with the reference
        self.layer1 = nn.Sequential(*double(in_c, c[0]))
        self.skip1 = attention_grid(c[0], c[0], c[0], dims=dims)
        self.layer2 = nn.Sequential(*double(c[0], c[1]))
        self.skip2 = attention_grid(c[1], c[1], c[1], dims=dims)
        self.layer3 = nn.Sequential(*double(c[1], c[2]))
        self.skip3 = attention_grid(c[2], c[2], c[2], dims=dims)
        self.layer4 = nn.Sequential(*double(c[2], c[3]))
        self.skip4 = attention_grid(c[3], c[3], c[3], dims=dims)
        self.layer5 = nn.Sequential(*(double(c[3], c[4]) + up(c[4], c[3])))
        self.layer6 = nn.Sequential(*(double(c[4], c[3]) + up(c[3], c[2])))
        self.layer7 = nn.Sequential(*(double(c[3], c[2]) + up(c[2], c[1])))
        self.layer8 = nn.Sequential(*(double(c[2], c[1]) + up(c[1], c[0])))
        self.layer9 = nn.Sequential(*double(c[1], c[0]))
        self.out = Conv(in_channels=c[0], out_channels=dims, kernel_size=1)
        self.maxpool = Pool(kernel_size=2, stride=2)
        self.warp = SpatialTransformer(img_size, mode)

    def features(self, x, device=None):
        """Everything up to the flow head: returns flow [B, nd, *spatial] (no warp)."""
        device = x.device if device is None else device
        y1 = self.layer1(x)
        y2 = self.layer2(self.maxpool(y1))
        y3 = self.layer3(self.maxpool(y2))
        y4 = self.layer4(self.maxpool(y3))
        y = self.layer5(self.maxpool(y4))
        for skip, ys, layer in ((self.skip4, y4, self.layer6), (self.skip3, y3, self.layer7), (self.skip2, y2, self.layer8),
                                (self.skip1, y1, self.layer9)):
            ys, _ = skip(ys, y, device=device)
            y = layer(torch.cat((ys, padNd(y, ys, device=device)), dim=1))
        return self.out(padNd(y, x, device=device))

    def forward(self, x, device=None, out_att=False):
        flow = self.features(x, device)
        return self.warp(x, flow), flow

"""Closed-form synthetic inputs shared by the golden generator, the tests and bench.py.

Nothing here touches /root/reference: the golden fixtures store only OUTPUTS of the
reference; the inputs are re-created from these formulas on any machine.
"""
import numpy as np
import torch


def vol(shape, f, fn="sin", dtype=torch.float32):
    """fn(arange(n) * f) in fp64, rounded once to `dtype`, viewed as [1,1,*shape]."""
    n = int(np.prod(shape))
    i = torch.arange(n, dtype=torch.float64) * f
    v = torch.sin(i) if fn == "sin" else torch.cos(i)
    return v.to(dtype).view(1, 1, *shape)


def blobs(shape, seed, nblob=6, dtype=torch.float32):
    """Sum of `nblob` Gaussian blobs on a [-1,1]^d lattice (SURVEY Appendix A recipe).

    Per blob, drawn in this order from torch.Generator().manual_seed(seed):
    centre = rand(d) - 0.5, sigma = 0.05 + 0.2 * rand(1), amplitude = rand(1).
    """
    g = torch.Generator().manual_seed(int(seed))
    d = len(shape)
    axes = [torch.linspace(-1, 1, s, dtype=torch.float64) for s in shape]
    grids = torch.meshgrid(*axes, indexing="ij")
    img = torch.zeros(shape, dtype=torch.float64)
    for _ in range(nblob):
        c = torch.rand(d, generator=g) - 0.5
        sig = 0.05 + 0.2 * torch.rand(1, generator=g)
        a = torch.rand(1, generator=g)
        r2 = sum((grids[k] - float(c[k])) ** 2 for k in range(d))
        img += float(a) * torch.exp(-r2 / (2 * float(sig) ** 2))
    return img.to(dtype).view(1, 1, *shape)


def blobs_fast(shape, seed, nblob=6, device="cpu"):
    """The same phantom family as `blobs` (same draws from the same generator) evaluated separably - exp(-r^2 / 2 s^2) as the outer product of
    three 1-D factors, fp64, rounded once to fp32 - on `device`: the values differ from `blobs` in the last bits, so golden fixtures keep
    `blobs`; tests that feed the SAME tensor to the kernels and to the oracle use this one for big volumes (16 volumes of 192^3 take 50 s
    with `blobs` on the host, well under a second with this on the GPU)."""
    g = torch.Generator().manual_seed(int(seed))
    d = len(shape)
    axes = [torch.linspace(-1, 1, s, dtype=torch.float64, device=device) for s in shape]
    img = torch.zeros(shape, dtype=torch.float64, device=device)
    for _ in range(nblob):
        c = torch.rand(d, generator=g) - 0.5
        sig = 0.05 + 0.2 * torch.rand(1, generator=g)
        a = torch.rand(1, generator=g)
        f = [torch.exp(-(axes[k] - float(c[k])) ** 2 / (2 * float(sig) ** 2)) for k in range(d)]
        term = f[0].view(-1, *([1] * (d - 1)))
        for k in range(1, d):
            term = term * f[k].view(*([1] * k), -1, *([1] * (d - 1 - k)))
        img += float(a) * term
    return img.to(torch.float32).view(1, 1, *shape)


def flow_field(shape, amp=0.8, f=0.11, dtype=torch.float32):
    """amp * sin(f * i) viewed as [1, ndim, *shape] (voxel units, channel i moves along dim i)."""
    nd = len(shape)
    n = nd * int(np.prod(shape))
    i = torch.arange(n, dtype=torch.float64) * f
    return (amp * torch.sin(i)).to(dtype).view(1, nd, *shape)


# theta constants used by fixtures / KATs
THETA_A = [[0.9, 0.1, -0.05, 0.02], [-0.08, 1.05, 0.03, -0.04], [0.06, -0.02, 0.95, 0.01]]
THETA_B = [[0.9, 0.1, 0.02], [-0.08, 1.05, -0.04]]
# strongly out-of-bounds: zoom-out + shift so many corners fall outside the volume.  Generic
# (non-dyadic) entries: sample coordinates must not land exactly on integers, where the
# trilinear derivative is one-sided and the side taken depends on last-bit rounding.
THETA_OOB3 = [[1.437, 0.213, -0.118, 0.3531], [0.1519, 1.3127, 0.1093, -0.2977], [-0.2071, 0.1037, 1.4911, 0.2543]]
THETA_OOB2 = [[1.4113, 0.3071, 0.2939], [-0.2531, 1.3517, -0.3467]]
# rotation-like
THETA_ROT3 = [[0.8013, -0.5527, 0.1031, 0.0517], [0.5009, 0.8219, -0.1523, -0.0211], [-0.0507, 0.2017, 0.9713, 0.0307]]
# ground-truth perturbations used to synthesise "moving" from "target" (Appendix A)
THETA_STAR3 = [[0.95, -0.1, 0.02, 0.05], [0.1, 0.97, 0.0, -0.03], [0.0, 0.03, 1.02, 0.02]]
THETA_STAR2 = [[0.98, -0.17, 0.05], [0.17, 0.98, -0.03]]

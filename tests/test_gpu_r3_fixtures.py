"""Round-3 fixtures on the GPU (tests/golden/fixtures_r3.npz, generated from the reference by make_golden_r3.py):
  * SpatialTransformer(mode='nearest') - the HIP nearest warp against the reference's own outputs (exact off the half-integer ties),
    its zero gradient, and the autograd plumbing;
  * Attention_UNet with DEFAULT arguments (mode='nearest'): constructs (the reference does: ref:utils.py:409-410,520), parameter names and
    shapes are the reference's, one forward follows the reference's flow and - through the fixture's own flow - its warped image exactly;
  * NMILoss: a target freed and replaced by another tensor at the same address is NOT mistaken for the cached one."""
import os
import zlib

import numpy as np
import pytest
import torch

import phantoms as ph
from test_oracle_golden import _smooth_flow

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fixtures_r3.npz"))


@pytest.fixture(scope="module")
def tr():
    import torchregister_amd as t
    assert torch.cuda.is_available()
    return t


@pytest.mark.parametrize("name,shape", [("st_nearest_2d", (48, 64)), ("st_nearest_3d", (20, 36, 28))])
def test_spatial_transformer_nearest_vs_reference(tr, g, name, shape):
    amp, seed = g[f"{name}/meta"]
    src = (ph.blobs(shape, 901) + 0.2 * ph.vol(shape, 0.021, "sin")).cuda()
    flow = _smooth_flow(shape, int(seed), float(amp)).cuda().requires_grad_()
    st = tr.SpatialTransformer(shape, mode="nearest")
    out = st(src, flow)
    ref = torch.from_numpy(g[f"{name}/warped"]).cuda()
    tie = torch.from_numpy(g[f"{name}/tie"]).cuda()[:, None]
    assert torch.equal(out[~tie], ref[~tie])            # bit for bit: a nearest warp copies voxels
    # two channels share the coordinates; the gradient wrt the flow is ATen's: zeros
    out2 = st(torch.cat([src, 2.0 * src], dim=1), flow)
    assert torch.equal(out2[:, 1:][~tie], 2.0 * ref[~tie])
    out.sum().backward()
    assert flow.grad is not None and torch.count_nonzero(flow.grad) == 0


def test_attention_unet_default_constructor_matches_reference(tr, g):
    torch.manual_seed(3)
    m = tr.Attention_UNet((64, 64))                      # default mode='nearest', n=1 - raised NotImplementedError in round 2
    names = [n for n, _ in m.named_parameters()]
    assert [zlib.crc32(n.encode()) for n in names] == g["unet_default_ctor/param_name_crc32"].tolist()
    shapes = np.array([list(p.shape) + [0] * (5 - p.dim()) for _, p in m.named_parameters()], dtype=np.int64)
    assert np.array_equal(shapes, g["unet_default_ctor/param_shapes"])
    assert [int(m.warp.mode == "nearest"), int(m.skip1.mode == "nearest")] == g["unet_default_ctor/modes_are_nearest"].tolist()


def test_attention_unet_default_mode_forward_vs_reference(tr, g):
    shape = (156, 156)
    torch.manual_seed(7)
    m = tr.Attention_UNet(shape, n=16)                   # same seed, same creation order -> the reference's weights
    assert np.array_equal(m.layer1[0].weight.detach().numpy()[:4], g["unet2d_nearest/first_weight"])
    x = (ph.blobs(shape, 911) + 0.1 * ph.vol(shape, 0.017, "cos")).cuda()
    m = m.cuda()
    with torch.no_grad():
        y, flow = m(x)
    rflow = torch.from_numpy(g["unet2d_nearest/flow"]).cuda()
    rng = float(rflow.abs().max())
    # MIOpen vs ATen-CPU convolutions through 18 layers + instance norms: 1e-3 of the flow's range
    assert float((flow - rflow).abs().max()) <= 2e-3 * max(rng, 1.0), (float((flow - rflow).abs().max()), rng)
    # the warp itself, fed the reference's flow, reproduces the reference's warped image except at half-integer ties of THAT flow
    w = m.warp(x, rflow)
    rw = torch.from_numpy(g["unet2d_nearest/warped"]).cuda()
    pos = torch.stack(torch.meshgrid(*[torch.arange(s, dtype=torch.float32, device="cuda") for s in shape], indexing="ij"))[None] + rflow
    tie = ((pos - torch.floor(pos) - 0.5).abs() < 2e-4).any(dim=1, keepdim=True)
    assert torch.equal(w[~tie], rw[~tie])
    assert float(tie.float().mean()) < 0.01
    # and the network's own output differs from the reference's at few pixels only (a nearest warp flips where its flow differs by 1e-3)
    assert float((y != rw).float().mean()) < 0.05


def test_nmi_loss_cache_is_keyed_on_identity_not_address(tr):
    """A persistent NMILoss must not reuse the previous target's patches / PDF when a NEW target tensor lands on the freed one's address
    with the same version counter (the caching allocator does exactly that)."""
    shape = (1, 1, 40, 40, 40)
    loss = tr.NMILoss(patch_size=10)
    yp = ph.blobs(shape[2:], 21).cuda()
    for seed in (31, 32, 33, 34):
        y = ph.blobs(shape[2:], seed).cuda()
        ptr = y.data_ptr()
        a = loss(y, yp).item()
        b = tr.NMILoss(patch_size=10)(y, yp).item()
        assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), (seed, a, b)
        assert loss._cache["yref"] is y                  # the cache pins the tensor itself ...
        del y
        y2 = ph.blobs(shape[2:], seed + 100).cuda()
        assert y2.data_ptr() != ptr                      # ... so its storage cannot be handed to the next target while it is cached
        a2 = loss(y2, yp).item()
        b2 = tr.NMILoss(patch_size=10)(y2, yp).item()
        assert abs(a2 - b2) <= 1e-6 * max(1.0, abs(b2)), (seed, a2, b2)
        del y2
    # same storage, same version counter, different tensor object (what a recycled address looks like to an address-keyed cache):
    # identity says "not the cached tensor" and the entry is rebuilt
    y = ph.blobs(shape[2:], 77).cuda()
    loss(y, yp)
    alias = y.view(shape)
    assert alias.data_ptr() == y.data_ptr() and alias._version == y._version and alias is not y
    before = loss._cache["y"]
    loss(alias, yp)
    assert loss._cache["yref"] is alias and loss._cache["y"] is not before

"""BASELINE.json's multi-GPU configurations at their REAL per-GPU sizes, against the oracle ("configs_untested").

config 4  "3D 256^3 batch=64 affine+NCC sharded 8 per GPU": one GPU's share is B = 8 pairs of 256^3 in one launch.
config 5  "3D 512^3 single-volume flow-field+NCC, Z-slab partitioned across 8 GPUs": the 512^3 volume as one full-depth slab and as
          8 slabs of 64 planes driven exactly as 8 ranks would drive them (moments summed by hand = the all-reduce, boundary flow
          planes copied by hand = the xGMI halo exchange), first evaluation against the C oracle in fp64, later iterations slab-vs-whole.
The oracle needs ~10 s per 256^3 and ~80 s per 512^3 evaluation on the GPU box's host cores.
"""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torchregister_amd._engine as e
    assert torch.cuda.is_available()
    return e


def blobs_gpu(shape, seed, nblob=6):
    """Sum of separable Gaussian blobs built on the GPU in fp32 (a 512^3 fp64 lattice on the host would take minutes): the tests feed
    exactly these fp32 values to the oracle, so how they were made does not matter."""
    g = torch.Generator().manual_seed(seed)
    axes = [torch.linspace(-1, 1, s, device="cuda") for s in shape]
    img = torch.zeros(shape, device="cuda")
    for _ in range(nblob):
        c = (torch.rand(3, generator=g) - 0.5).tolist()
        sig = float(0.05 + 0.2 * torch.rand(1, generator=g))
        a = float(torch.rand(1, generator=g))
        ez, ey, ex = (torch.exp(-(axes[k] - c[k]) ** 2 / (2 * sig * sig)) for k in range(3))
        img += a * ez[:, None, None] * ey[None, :, None] * ex[None, None, :]
    return img[None, None]


def smooth_flow_gpu(shape, amp=1.0):
    """A smooth flow of ~1 voxel that keeps every sample OFF the voxel lattice (trilinear kinks)."""
    ax = [torch.arange(n, device="cuda", dtype=torch.float32) for n in shape]
    comp = lambda a, b, c: (torch.sin(a * ax[0])[:, None, None] + torch.cos(b * ax[1])[None, :, None] + torch.sin(c * ax[2] + 0.4)[None, None, :])  # noqa: E731
    return amp * torch.stack([1.3 * comp(0.021, 0.017, 0.013), 0.9 * comp(0.011, 0.023, 0.019), 1.1 * comp(0.015, 0.012, 0.027)])[None] + 0.37


def test_config4_share_b8_256_vs_oracle_and_singles(eng):
    """B = 8 pairs of 256^3, affine + NCC, one launch (bench.py's workload).  (a) two of the eight against the C oracle in fp64 (loss
    2e-5 rel, dL/dtheta 2e-4 of its maximum or twice the oracle's own fp32-vs-fp64 gap); (b) a pair's result does not depend on its
    slot in the batch (bit for bit, pairs permuted); (c) the batch equals eight single-pair launches to the fp32 floors (a single
    pair splits its columns differently and may run another kernel body); (d) three Adam iterations of the batch
    stay bit-for-bit reproducible."""
    shape = (256, 256, 256)
    B = 8
    base = [blobs_gpu(shape, 2000 + i) for i in range(4)]
    tgt = torch.cat(base + [b.flip(2) for b in base[:2]] + [b.flip(3) for b in base[2:]])          # 8 different volumes
    ths = np.stack([np.eye(3, 4) + 0.025 * np.sin(1.3 * np.arange(12) + 0.7 * b).reshape(3, 4) for b in range(B)])
    th = torch.tensor(ths, dtype=torch.float32)
    gen = np.stack([np.eye(3, 4) + 0.03 * np.cos(0.9 * np.arange(12) + 0.5 * b).reshape(3, 4) for b in range(B)])
    mov = eng.affine_warp(torch.tensor(gen, dtype=torch.float32).cuda(), tgt) + 0.05 * tgt.roll(1, 0)
    kw = dict(w_ncc=1.0)
    s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1)
    s.run(1)
    torch.cuda.synchronize()
    # (a)
    t64, t32 = oracle.base_tables(shape, np.float64), oracle.base_tables(shape, np.float32)
    for b in (1, 6):
        m, t = mov[b, 0].cpu().numpy(), tgt[b, 0].cpu().numpy()
        total, _, dth, _ = oracle.c_affine_loss_grad(m.astype(np.float64), t.astype(np.float64), th[b].double().numpy(), oracle.wts(**kw), t64)
        _, _, dth32, _ = oracle.c_affine_loss_grad(m, t, th[b].numpy(), oracle.wts(**kw), t32)
        assert abs(s.losses[b, 0].item() - total) <= 2e-5 * max(1.0, abs(total)), b
        gmax = np.max(np.abs(dth))
        assert np.max(np.abs(s.grad[b, :12].cpu().numpy().reshape(3, 4) - dth)) <= max(2e-4 * gmax, 2.0 * np.max(np.abs(dth32 - dth))), b
    # (b)
    perm = torch.tensor([3, 0, 7, 5, 1, 6, 2, 4])
    sp = eng.AffineSolver(mov[perm], tgt[perm], mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th[perm], capacity=1)
    sp.run(1)
    torch.cuda.synchronize()
    assert torch.equal(sp.losses[:, 0], s.losses[perm.cuda(), 0]) and torch.equal(sp.grad, s.grad[perm.cuda()])
    # (c)
    for b in range(B):
        s1 = eng.AffineSolver(mov[b:b + 1], tgt[b:b + 1], mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th[b:b + 1], capacity=1)
        s1.run(1)
        torch.cuda.synchronize()
        # a single pair may run another kernel body than the batch (the z-streaming body in 8 z segments, or a tile geometry, where the batch
        # streams whole columns): the bars are the fp32 floors between bodies (tests/fuzz_zstream.py: body vs tiles), not a summation-order bar
        assert abs(s1.losses[0, 0].item() - s.losses[b, 0].item()) <= 2e-5 * max(1.0, abs(s.losses[b, 0].item()))   # (the NCC loss is 100 (1 - ncc))
        gb = s.grad[b, :12]
        assert torch.max(torch.abs(s1.grad[0, :12] - gb)).item() <= 2e-4 * gb.abs().max().item()
    # (d)
    runs = []
    for _ in range(2):
        r = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(**kw), optimizer="adam", lr=1e-4, init=th, capacity=3)
        r.run(3)
        torch.cuda.synchronize()
        runs.append((r.losses.clone(), r.theta.clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert (runs[0][0][:, 2] < runs[0][0][:, 0]).all()


@pytest.fixture(scope="module")
def vol512(eng):
    """The 512^3 pair, an off-lattice starting flow, and the C oracle's first evaluation (fp64 arbiter + fp32 for the bar): ~3 min of
    host time, shared by the parametrisations below (the oracle covers the data term; the regulariser is checked against torch)."""
    shape = (512, 512, 512)
    tgt, mov = blobs_gpu(shape, 3000), blobs_gpu(shape, 3001)
    fl0 = smooth_flow_gpu(shape)
    m, t, f = mov[0, 0].cpu().numpy(), tgt[0, 0].cpu().numpy(), fl0[0].cpu().numpy()
    t64, _, d64, _ = oracle.c_flow_loss_grad(m.astype(np.float64), t.astype(np.float64), f.astype(np.float64), oracle.wts(w_ncc=1.0))
    _, _, d32, _ = oracle.c_flow_loss_grad(m, t, f, oracle.wts(w_ncc=1.0))
    gap = float(np.max(np.abs(d32.astype(np.float64) - d64)))
    del d32, m, t, f
    # the whole-volume evaluation entry point against it
    terms, dfl = eng.flow_loss_grad(mov, tgt, fl0, eng.LossSpec(w_ncc=1.0))
    torch.cuda.synchronize()
    gmax = float(np.max(np.abs(d64)))
    assert abs(terms[0, 0].item() - t64) <= 2e-5 * max(1.0, abs(t64))
    assert np.max(np.abs(dfl[0].cpu().numpy() - d64)) <= max(1e-4 * gmax, 2.0 * gap)
    return dict(mov=mov, tgt=tgt, fl0=fl0, t64=t64, d64=d64, gap=gap, gmax=gmax)


@pytest.mark.parametrize("smooth", [0.0, 2.0e5])
def test_config5_512_cubed_slabs_vs_oracle_and_whole(eng, vol512, smooth):
    """512^3, direct flow + NCC (+ smoothness regulariser), SGD: whole-volume solver, ONE full-depth slab and EIGHT slabs of 64
    planes (the 8-GPU partition) for two iterations.  First evaluation against oracle.c_flow_loss_grad in fp64 (loss, and dL/dflow
    recovered from the first SGD update); second iteration slab-vs-whole."""
    from torchregister_amd.warpings import smooth_regulariser
    mov, tgt, fl0, t64, d64, gap, gmax = (vol512[k] for k in ("mov", "tgt", "fl0", "t64", "d64", "gap", "gmax"))
    reg0 = smooth_regulariser(fl0.double(), smooth).item() if smooth else 0.0
    greg, gtot = None, gmax
    if smooth:
        f0 = fl0.double().requires_grad_()
        (greg,) = torch.autograd.grad(smooth_regulariser(f0, smooth), f0)
        assert greg.abs().max().item() > 0.02 * gmax        # the regulariser matters in this test
        gtot = gmax + greg.abs().max().item()
        del f0
    lr = 0.2 / gtot                              # dL/dflow is ~1e-6 per voxel at this size: a first step of at most 0.2 voxel
    kw = dict(loss=eng.LossSpec(w_ncc=1.0), optimizer="sgd", lr=lr, capacity=2, smooth_weight=smooth)
    whole = eng.FlowSolver(mov, tgt, init=fl0, **kw)
    whole.run(2)
    torch.cuda.synchronize()
    assert abs(whole.losses[0, 0].item() - (t64 + reg0)) <= 2e-5 * max(1.0, abs(t64 + reg0))

    def check_first_update(flow_after, what):
        """(flow0 - flow1) / lr = dL/dflow of the first evaluation (with the regulariser its torch gradient is removed); recovering
        it from an fp32 flow costs eps * |flow| / lr ~ 2.5e-6 of the gradient's maximum on top of the kernel's own error"""
        g = (fl0.double() - flow_after.double()) / lr
        if greg is not None:
            g = g - greg
        err = float(np.max(np.abs(g[0].cpu().numpy() - d64)))
        assert err <= max(1e-4 * gmax, 2.0 * gap) + 1e-5 * gmax, (what, err, gmax, gap)

    # ---- one full-depth slab == the whole-volume solver
    one = eng.SlabFlowSolver(mov, tgt, 0, **kw)
    one.flow.copy_(fl0)
    one.run(1)
    torch.cuda.synchronize()
    check_first_update(one.flow, "one slab")
    one.run(1)
    torch.cuda.synchronize()
    assert torch.allclose(one.losses, whole.losses, rtol=1e-6, atol=1e-7)
    assert torch.max(torch.abs(one.flow - whole.flow)).item() <= 1e-5
    del one

    # ---- eight slabs of 64 planes, driven as eight ranks would drive them
    bounds = list(range(0, 513, 64))
    slabs = [eng.SlabFlowSolver(mov, tgt[:, :, a:b].contiguous(), a, **kw) for a, b in zip(bounds[:-1], bounds[1:])]
    for s, a, b in zip(slabs, bounds[:-1], bounds[1:]):
        s.flow.copy_(fl0[:, :, a:b])
    for it in range(2):
        if smooth:   # what the xGMI halo exchange does
            planes = [s.boundary_planes() for s in slabs]
            for r, s in enumerate(slabs):
                if s.has_lo:
                    s.halo_lo.copy_(planes[r - 1][1])
                if s.has_hi:
                    s.halo_hi.copy_(planes[r + 1][0])
        total = sum(s.local_moments().clone() for s in slabs)      # what the 64-byte all-reduce does
        for s in slabs:
            s.apply(total)
        torch.cuda.synchronize()
        if it == 0:
            check_first_update(torch.cat([s.flow for s in slabs], dim=2), "eight slabs")
            for s in slabs:
                assert abs(s.losses[0, 0].item() - (t64 + reg0)) <= 2e-5 * max(1.0, abs(t64 + reg0))
    for s in slabs:
        assert torch.allclose(s.losses, whole.losses, rtol=1e-5, atol=1e-6)       # every rank records the whole-volume loss
    flow = torch.cat([s.flow for s in slabs], dim=2)
    assert torch.max(torch.abs(flow - whole.flow)).item() <= 1e-5 * max(1.0, whole.flow.abs().max().item())
    if not smooth:
        assert whole.losses[0, 1].item() < whole.losses[0, 0].item()

"""GPU parity of the LDS-tiled F1 kernel across its code paths: interior tiles, boundary tiles (zero
padding through the LDS fix-up), tiles whose pre-image does not fit the LDS box (global-gather fallback),
volumes with W % 4 != 0 (fallback), ragged sizes that are not multiples of the 32x16x8 tile, y-split
columns (small batches) and batches.  Checker: the C oracle in fp64 (tolerance: loss 2e-5 rel,
gradient 2e-4 of max — the fp32 floors of test_gpu_affine.py)."""
import math

import numpy as np
import pytest
import torch

import oracle
import phantoms as ph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torchregister_amd._engine as e
    assert torch.cuda.is_available()
    return e


def rot_theta(ax, ay, az, scale=(1.0, 1.0, 1.0), shift=(0.0, 0.0, 0.0)):
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    m = rz @ ry @ rx @ np.diag(scale)
    return np.concatenate([m, np.asarray(shift, dtype=np.float64)[:, None]], axis=1)


def generic(th):
    """Perturb theta by ~1e-3 irrational-ish amounts: 'nice' decimal entries put whole lines of samples
    EXACTLY on integer coordinates (e.g. theta* with D=40: iz integral wherever 3d+5h = 26 mod 150),
    where the trilinear derivative is one-sided and fp32-vs-fp64 rounding picks different sides."""
    th = np.array(th, dtype=np.float64)
    k = np.arange(th.size, dtype=np.float64).reshape(th.shape)
    return th + 1.3e-3 * np.sin(1.2345 * (k + 1.0)) + 0.7e-3 * np.cos(2.718 * k)


THETAS = {
    "identity": np.eye(3, 4),
    "star": np.asarray(ph.THETA_STAR3, dtype=np.float64),
    "small_rot": rot_theta(0.04, -0.07, 0.09, (1.03, 0.96, 1.01), (0.02, -0.03, 0.05)),
    "shift_out": rot_theta(0.02, 0.01, -0.03, (1.0, 1.0, 1.0), (0.45, -0.4, 0.3)),      # large OOB region
    "rot30": rot_theta(0.1, 0.2, 0.52, (0.9, 1.1, 1.0), (0.05, 0.0, -0.05)),             # box does not fit: fallback
    "rigid_rand": rot_theta(0.77, 0.5, 0.09, (1, 1, 1), (0.03, 0.07, 0.15)),             # typical torch.rand rigid init
    "zoom_out": rot_theta(0.0, 0.0, 0.0, (1.6, 1.5, 1.7), (0.0, 0.0, 0.0)),
    "flip": np.array([[-1.0, 0.02, 0, 0.01], [0.01, 1.0, 0, 0], [0, 0, -0.98, 0.02]]),
}
SHAPES = [(40, 36, 44), (16, 48, 64), (24, 20, 30), (9, 33, 68), (64, 64, 64), (23, 37, 46), (18, 16, 33)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("tname", list(THETAS))
def test_f1_step_vs_oracle(eng, shape, tname):
    tgt = ph.blobs(shape, 77)
    mov = ph.blobs(shape, 78) + 0.1 * ph.vol(shape, 0.013, "sin")
    th64 = THETAS[tname] if tname == "identity" else generic(THETAS[tname])
    th = torch.tensor(th64, dtype=torch.float32)[None]
    kw = dict(w_ncc=1.0, w_mse=0.5)
    s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1)
    s.run(1)
    torch.cuda.synchronize()
    th_used = th[0].double().numpy()          # the fp32-rounded theta, evaluated by the oracle in fp64
    total, _, dth, _ = oracle.c_affine_loss_grad(mov[0, 0].double().numpy(), tgt[0, 0].double().numpy(), th_used, oracle.wts(**kw),
                                                 oracle.base_tables(shape, np.float64))
    loss = s.losses[0, 0].item()
    grad = s.grad[0, :12].cpu().numpy().reshape(3, 4)
    assert abs(loss - total) <= 2e-5 * max(1.0, abs(total)), (loss, total)
    if tname == "identity":
        return  # every sample on a voxel: one-sided derivative, side decided by last-bit rounding
    assert np.max(np.abs(grad - dth)) <= 2e-4 * np.max(np.abs(dth)), (grad, dth)
    # loss-only kernel (MODE 1) agrees
    assert abs(s.eval_loss(th.cuda())[0, 0].item() - loss) <= 1e-6 * max(1.0, abs(loss))


def test_forced_gather_path_matches_tile_path(eng):
    """trx_volumes.flags = TRX_FLAG_GATHER_PATH runs the un-tiled kernel, TRX_FLAG_SINGLE_GEOM the one-geometry tile kernel: same
    results as the default per-pair GeomA / GeomR kernel to fp32 rounding."""
    shape = (40, 48, 64)
    tgt, mov = ph.blobs(shape, 5).cuda(), ph.blobs(shape, 6).cuda()
    th = torch.tensor(generic(THETAS["small_rot"]), dtype=torch.float32)[None]
    outs = []
    from torchregister_amd import _lib
    for flags in (0, _lib.FLAG_GATHER_PATH, _lib.FLAG_SINGLE_GEOM):
        s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=1e-4, init=th, capacity=3, flags=flags)
        s.run(3)
        torch.cuda.synchronize()
        outs.append((s.losses.clone(), s.theta.clone()))
    for o in outs[1:]:
        assert torch.allclose(outs[0][0], o[0], rtol=2e-6, atol=1e-5)
        assert torch.allclose(outs[0][1], o[1], rtol=0, atol=2e-6)


def test_batch_of_mixed_thetas(eng):
    """Pairs of one launch may take different paths (fits / fallback / boundary) independently."""
    shape = (32, 40, 48)
    names = ["star", "rot30", "shift_out", "zoom_out", "identity"]
    tgt = torch.cat([ph.blobs(shape, 100 + i) for i in range(len(names))])
    mov = torch.cat([ph.blobs(shape, 200 + i) for i in range(len(names))])
    th = torch.stack([torch.tensor(THETAS[n] if n == "identity" else generic(THETAS[n]), dtype=torch.float32) for n in names])
    s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=0.0, init=th, capacity=1)
    s.run(1)
    torch.cuda.synchronize()
    for i, n in enumerate(names):
        total, _, dth, _ = oracle.c_affine_loss_grad(mov[i, 0].double().numpy(), tgt[i, 0].double().numpy(), th[i].double().numpy(),
                                                     oracle.wts(w_ncc=1.0), oracle.base_tables(shape, np.float64))
        assert abs(s.losses[i, 0].item() - total) <= 2e-5 * max(1.0, abs(total)), n
        if n != "identity":
            assert np.max(np.abs(s.grad[i, :12].cpu().numpy().reshape(3, 4) - dth)) <= 2e-4 * np.max(np.abs(dth)), n


@pytest.mark.parametrize("shape", [(8, 16, 36), (8, 16, 40), (8, 16, 44), (16, 36, 44), (8, 33, 52), (12, 16, 76)])
def test_partial_x_tiles_identity_mse(eng, shape):
    """Regression: partial x tiles with fewer than 16 active columns (the row constants are broadcast
    from lanes 0..15 with v_readlane, which must not happen under the `active voxel` branch).  At theta =
    identity the warp is the identity, so the fused MSE must equal mean((moving - target)^2)."""
    B = 3
    tgt = torch.cat([ph.blobs(shape, 300 + i) for i in range(B)])
    mov = torch.cat([ph.blobs(shape, 400 + i) for i in range(B)])
    for _ in range(2):   # twice: a stale-LDS bug shows up as run-to-run differences
        s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(w_mse=1.0), lr=0.0, capacity=1)
        s.run(1)
        torch.cuda.synchronize()
        ref = ((mov.double() - tgt.double()) ** 2).mean(dim=(1, 2, 3, 4)).numpy()
        assert np.allclose(s.losses[:, 0].cpu().numpy(), ref, rtol=2e-6, atol=0)


@pytest.mark.parametrize("shape", [(24, 48, 38), (17, 35, 70), (40, 32, 33)])
def test_forward_warp_rows_not_multiple_of_4(eng, shape):
    """Forward warp through the tile kernel (MODE 3) when W % 4 != 0, two channels (the float4 straddling x = W is
    fetched whole and its tail zeroed; the last row of the LAST channel must not be over-read): vs the C oracle in fp64,
    including a theta that puts samples on and beyond the +x face."""
    x = torch.cat([ph.blobs(shape, 90) + 0.1 * ph.vol(shape, 0.013, "sin"), ph.vol(shape, 0.23, "cos")], dim=1)
    for th64 in (generic(THETAS["star"]), generic(rot_theta(0.03, -0.02, 0.05, (1.2, 1.1, 1.15), (0.1, -0.05, 0.02)))):
        th = torch.tensor(th64, dtype=torch.float32)[None]
        w = eng.affine_warp(th.cuda(), x.cuda()).cpu().numpy()
        for c in range(2):
            r64 = oracle.c_affine_warp(x[0, c].double().numpy(), th[0].double().numpy(), oracle.base_tables(shape, np.float64))
            r32 = oracle.c_affine_warp(x[0, c].numpy(), th[0].numpy(), oracle.base_tables(shape, np.float32))
            assert np.max(np.abs(w[0, c] - r32)) <= max(2e-6, 2.0 * np.max(np.abs(r32 - r64)))   # fp32 floor, as in test_gpu_affine.py


POSES = {  # (rot about y, rot about z, rot about x, tx, ty, tz) - ref:utils.py:287-310; translations pass through 0.25 tanh
    "tiny": [0.02, -0.03, 0.01, 0.1, 0.0, -0.1],
    "z04": [0.03, 0.41, -0.02, 0.2, -0.1, 0.0],          # GeomA does not fit: GeomR
    "y03": [0.33, 0.05, 0.04, 0.0, 0.1, 0.1],
    "x05": [0.02, -0.04, 0.52, -0.2, 0.0, 0.1],
    "rand": [0.77, 0.5, 0.09, 0.03, 0.07, 0.15],         # torch.rand-like init (ref:utils.py:316-330 draws every angle from [0,1) rad)
    "max": [0.98, 0.93, 0.99, 0.9, 0.5, 0.2],            # the far corner of that range: GeomR's 28 x 27 x 26 box holds any rotation
}


@pytest.mark.parametrize("shape", [(40, 36, 44), (64, 64, 64), (23, 37, 46), (33, 70, 51)])
@pytest.mark.parametrize("pname", list(POSES))
def test_rigid_step_dual_geometry_vs_oracle(eng, shape, pname):
    """Rigid steps run the dual kernel: per pair GeomA (32 x 16 x 8 tile) where the pre-image fits its box, GeomR (16 x 16 x 8,
    28 x 27 x 26 box) for every other rotation, the global gather for zoom-out / shear beyond that.  Loss and pose gradient vs the C oracle in fp64
    (theta from the oracle's Theta, chain rule through its vjp); batch of two different poses so both geometries run in one launch."""
    tgt = torch.cat([ph.blobs(shape, 77), ph.blobs(shape, 79)])
    mov = torch.cat([ph.blobs(shape, 78), ph.blobs(shape, 80)]) + 0.1 * torch.cat([ph.blobs(shape, 81, nblob=9), ph.blobs(shape, 82, nblob=9)])
    # (smooth phantoms only: a term that oscillates from plane to plane, like vol(shape, 0.013), turns the one-sided derivative at
    # integer coordinates into a 3e-4 pose-gradient difference between fp32 and fp64 coordinate rounding - DESIGN.md section 2)
    poses = np.array([POSES[pname], POSES["tiny"]], dtype=np.float64) + 1.3e-3 * np.sin(1.7 * np.arange(12).reshape(2, 6))
    p32 = torch.tensor(poses, dtype=torch.float32)
    kw = dict(w_ncc=1.0, w_mse=0.5)
    s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="rigid", loss=eng.LossSpec(**kw), lr=0.0, init=p32, capacity=1)
    s.run(1)
    torch.cuda.synchronize()
    tabs = oracle.base_tables(shape, np.float64)
    for b in range(2):
        p64 = p32[b].double().numpy()
        th = oracle.c_theta_fwd(p64)
        total, _, dth, _ = oracle.c_affine_loss_grad(mov[b, 0].double().numpy(), tgt[b, 0].double().numpy(), th, oracle.wts(**kw), tabs)
        dp = oracle.c_theta_vjp(p64, dth)
        # the fp32 run of the oracle sets the floor of the pose gradient (large rotations amplify the coordinate rounding)
        p32n = p32[b].numpy()
        _, _, dth32, _ = oracle.c_affine_loss_grad(mov[b, 0].numpy(), tgt[b, 0].numpy(), oracle.c_theta_fwd(p32n), oracle.wts(**kw),
                                                   oracle.base_tables(shape, np.float32))
        dp32 = oracle.c_theta_vjp(p32n, dth32)
        assert abs(s.losses[b, 0].item() - total) <= 2e-5 * max(1.0, abs(total))
        assert np.max(np.abs(s.grad[b, :6].cpu().numpy() - dp)) <= max(3e-4 * np.max(np.abs(dp)), 2.0 * np.max(np.abs(dp32 - dp)))
    # forward warp and loss-only kernels take the same dual dispatch
    th32 = s.current_theta if hasattr(s, "current_theta") else None
    w = eng.affine_warp(th32, mov.cuda()).cpu().numpy()
    for b in range(2):
        r32 = oracle.c_affine_warp(mov[b, 0].numpy(), th32[b].cpu().numpy(), oracle.base_tables(shape, np.float32))
        r64 = oracle.c_affine_warp(mov[b, 0].double().numpy(), th32[b].cpu().double().numpy(), tabs)
        assert np.max(np.abs(w[b, 0] - r32)) <= max(2e-6, 3.0 * np.max(np.abs(r32 - r64)))


def test_straddling_float4_moves_between_tiles(eng):
    """Regression (found by tests/fuzz_affine.py, seed 5 case 79): W % 4 != 0 and a column of tiles whose box origin shifts by
    4 voxels in x from one tile to the next.  The slot that straddles x = W (fetched whole, tail zeroed after landing) then sits in
    a different float4 of the box while the cached fetch masks - keyed on the slot range only - stayed valid, and the tail of
    the wrong slot was zeroed: 8 voxels of this warp were off by up to 70 %.  The straddler's position is part of the key now."""
    shape = (54, 83, 90)
    th64 = np.array([[0.9194692373275757, -0.213600292801857, -0.11958087980747223, 0.1763327270746231],
                     [0.220373272895813, 1.001311182975769, 0.19802626967430115, 0.39654022455215454],
                     [0.08663784712553024, -0.28822240233421326, 0.7801948189735413, -0.27287742495536804]])
    # three pairs, as in the sweep: the batch size sets how many tiles of a column one block walks (tile_geom's y split)
    x = torch.cat([ph.blobs(shape, 1295 + b) + 0.25 for b in range(3)])
    th = torch.tensor(th64, dtype=torch.float32)[None].expand(3, 3, 4).contiguous()
    w = eng.affine_warp(th.cuda(), x.cuda()).cpu().numpy()
    for b in range(3):
        r64 = oracle.c_affine_warp(x[b, 0].double().numpy(), th[b].double().numpy(), oracle.base_tables(shape, np.float64))
        r32 = oracle.c_affine_warp(x[b, 0].numpy(), th[b].numpy(), oracle.base_tables(shape, np.float32))
        assert np.max(np.abs(w[b, 0] - r32)) <= max(2e-6, 3.0 * np.max(np.abs(r32 - r64)))
    x, th = x[:1], th[:1]
    # the fused step at the same theta (single-geometry kernel: global gather for this rotation) and through the dual kernel
    tgt = ph.blobs(shape, 77)
    for mode_kw in (dict(mode="affine", init=th.reshape(1, 12)),):
        s = eng.AffineSolver(x.cuda(), tgt.cuda(), loss=eng.LossSpec(w_mse=1.0), lr=0.0, capacity=1, **mode_kw)
        s.run(1)
        torch.cuda.synchronize()
        total, _, dth, _ = oracle.c_affine_loss_grad(x[0, 0].double().numpy(), tgt[0, 0].double().numpy(), th[0].double().numpy(), oracle.wts(w_mse=1.0),
                                                     oracle.base_tables(shape, np.float64))
        assert abs(s.losses[0, 0].item() - total) <= 2e-5 * max(1.0, abs(total))
        assert np.max(np.abs(s.grad[0, :12].cpu().numpy().reshape(3, 4) - dth)) <= 2e-4 * np.max(np.abs(dth))


@pytest.mark.parametrize("shape", SHAPES)
def test_identity_gradient_same_on_every_path(eng, shape):
    """At theta = identity every sample sits exactly on a voxel: the trilinear derivative is one-sided there and the side is decided by
    the last bit of the un-normalised coordinate, so the C oracle (fp64) is no arbiter (test_f1_step_vs_oracle checks the loss only).
    All kernels of the library reproduce ATen's coordinate arithmetic bit for bit at the identity (trx_common.h: unnorm<ND>, the
    host-built base tables), so they take the SAME side: the gradient of the tiled kernels (GeomA through the dual kernel, the
    single-geometry kernel) and of the un-tiled row-walking kernel must agree to fp32 summation error - and the small identity fixtures
    ID3 / ID2 (test_gpu_affine.py) pin that common value to the reference's own autograd result."""
    from torchregister_amd import _lib
    tgt = ph.blobs(shape, 77)
    mov = ph.blobs(shape, 78) + 0.1 * ph.vol(shape, 0.013, "sin")
    th = torch.eye(3, 4)[None]
    grads, losses = [], []
    for flags in (0, _lib.FLAG_SINGLE_GEOM, _lib.FLAG_GATHER_PATH):
        s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(w_ncc=1.0, w_mse=0.5), lr=0.0, init=th, capacity=1, flags=flags)
        s.run(1)
        torch.cuda.synchronize()
        grads.append(s.grad[0, :12].cpu().double().numpy())
        losses.append(s.losses[0, 0].item())
    gmax = np.max(np.abs(grads[2]))
    for g, l in zip(grads[:2], losses[:2]):
        assert abs(l - losses[2]) <= 2e-6 * max(1.0, abs(losses[2]))
        assert np.max(np.abs(g - grads[2])) <= 2e-5 * gmax, (g, grads[2])


DEEP_THETAS = {
    "identity": np.eye(3, 4),
    "near": np.eye(3, 4) + 0.02 * np.sin(1.3 * np.arange(12) + 0.4).reshape(3, 4),
    "rot_z": rot_theta(0.0, 0.0, 0.12, (1.01, 0.99, 1.0), (0.03, -0.02, 0.01)),          # about z: the deep box holds what GeomA holds
    "shift_out": rot_theta(0.01, 0.005, -0.02, (1.0, 1.0, 1.0), (0.45, -0.4, 0.3)),      # large out-of-volume region, still the deep tile
    "tilt": rot_theta(0.06, 0.05, 0.02, (1.0, 1.0, 1.0), (0.0, 0.0, 0.0)),               # about x / y beyond the box's one plane of slack: GeomA takes it
}


@pytest.mark.parametrize("loss", ["ncc_mse", "mse"])
@pytest.mark.parametrize("shape", [(40, 36, 44), (16, 48, 64), (33, 20, 68), (64, 64, 64), (23, 37, 46)])
@pytest.mark.parametrize("tname", list(DEEP_THETAS))
def test_deep_tile_vs_oracle(eng, shape, tname, loss):
    """GeomD (32 x 16 x 16 tile, sixteen rows per thread; by default only for big batches) forced on small ragged volumes through
    TRX_FLAG_DEEP_TILE, both step kernels (41 sums with the NCC term, 13 without), against the C oracle in fp64."""
    from torchregister_amd import _lib
    tgt = ph.blobs(shape, 77)
    mov = ph.blobs(shape, 78) + 0.1 * ph.vol(shape, 0.013, "sin")
    th64 = DEEP_THETAS[tname] if tname == "identity" else generic(DEEP_THETAS[tname])
    th = torch.tensor(th64, dtype=torch.float32)[None]
    kw = dict(w_ncc=1.0, w_mse=0.5) if loss == "ncc_mse" else dict(w_mse=1.0, w_ssd=0.01)
    s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1, flags=_lib.FLAG_DEEP_TILE)
    ref = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1)
    s.run(1)
    ref.run(1)
    torch.cuda.synchronize()
    total, _, dth, _ = oracle.c_affine_loss_grad(mov[0, 0].double().numpy(), tgt[0, 0].double().numpy(), th[0].double().numpy(), oracle.wts(**kw),
                                                 oracle.base_tables(shape, np.float64))
    loss_v = s.losses[0, 0].item()
    grad = s.grad[0, :12].cpu().numpy().reshape(3, 4)
    assert abs(loss_v - total) <= 2e-5 * max(1.0, abs(total)), (loss_v, total)
    gref = ref.grad[0, :12].cpu().numpy().reshape(3, 4)
    assert np.max(np.abs(grad - gref)) <= 2e-5 * np.max(np.abs(gref))            # the default geometry choice gives the same numbers
    if tname != "identity":
        assert np.max(np.abs(grad - dth)) <= 2e-4 * np.max(np.abs(dth)), (grad, dth)


ROT_DEEP_THETAS = {
    "rot_z_0.5": rot_theta(0.0, 0.0, 0.5, (1.0, 1.0, 1.0), (0.03, -0.02, 0.01)),
    "rot_z_1.0": rot_theta(0.0, 0.0, 1.0, (1.02, 0.98, 1.0), (0.0, 0.05, 0.0)),
    "rot_x_0.6": rot_theta(0.6, 0.0, 0.0, (1.0, 1.0, 1.0), (0.0, 0.0, 0.02)),
    "rot_y_0.6": rot_theta(0.0, 0.6, 0.0, (1.0, 1.01, 0.97), (0.02, 0.0, 0.0)),
    "general_0.3": rot_theta(0.3, 0.3, 0.3, (1.0, 1.0, 1.0), (0.01, -0.02, 0.015)),        # fits GeomRD's box
    "general_0.45": rot_theta(0.45, 0.45, 0.45, (1.0, 1.0, 1.0), (0.01, -0.02, 0.015)),     # does not: GeomR either way
    "shift_out": rot_theta(0.0, 0.2, 0.5, (1.0, 1.0, 1.0), (0.5, -0.45, 0.3)),             # rotated AND mostly out of the volume
}


@pytest.mark.parametrize("loss", ["ncc_mse", "mse"])
@pytest.mark.parametrize("shape", [(40, 36, 44), (16, 48, 64), (33, 20, 68), (64, 64, 64), (23, 37, 46)])
@pytest.mark.parametrize("tname", list(ROT_DEEP_THETAS))
def test_rotated_deep_tile_vs_oracle(eng, shape, tname, loss):
    """GeomRD (GeomR's 28 x 27 x 26 box under a 16 x 16 x 16 tile, eight rows per thread: the choice of every step whose rotated
    pre-image still fits that box) against the C oracle in fp64 and against GeomR on the same inputs (TRX_FLAG_NO_ROT_DEEP_TILE), both
    step kernels, ragged volumes (partial tiles in every direction)."""
    from torchregister_amd import _lib
    tgt = ph.blobs(shape, 77)
    mov = ph.blobs(shape, 78) + 0.1 * ph.vol(shape, 0.013, "sin")
    th64 = generic(ROT_DEEP_THETAS[tname])
    th = torch.tensor(th64, dtype=torch.float32)[None]
    kw = dict(w_ncc=1.0, w_mse=0.5) if loss == "ncc_mse" else dict(w_mse=1.0, w_ssd=0.01)
    # (TRX_FLAG_DEEP_TILE: these volumes are below the size from which GeomRD is offered by default)
    s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1, flags=_lib.FLAG_DEEP_TILE)
    ref = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1, flags=_lib.FLAG_NO_ROT_DEEP_TILE)
    s.run(1)
    ref.run(1)
    torch.cuda.synchronize()
    total, _, dth, _ = oracle.c_affine_loss_grad(mov[0, 0].double().numpy(), tgt[0, 0].double().numpy(), th[0].double().numpy(), oracle.wts(**kw),
                                                 oracle.base_tables(shape, np.float64))
    loss_v = s.losses[0, 0].item()
    grad = s.grad[0, :12].cpu().numpy().reshape(3, 4)
    assert abs(loss_v - total) <= 2e-5 * max(1.0, abs(total)), (loss_v, total)
    assert abs(loss_v - ref.losses[0, 0].item()) <= 2e-6 * max(1.0, abs(total))
    gref = ref.grad[0, :12].cpu().numpy().reshape(3, 4)
    assert np.max(np.abs(grad - gref)) <= 2e-5 * np.max(np.abs(gref))
    # large rotations put many samples on the zero-padding border, where fp32 coordinates decide which side of the jump a sample sees:
    # the bar is the oracle's own fp32-vs-fp64 gap (x2) where that exceeds the 2e-4 floor, as in tests/fuzz_affine.py
    _, _, dth32, _ = oracle.c_affine_loss_grad(mov[0, 0].numpy(), tgt[0, 0].numpy(), th[0].numpy(), oracle.wts(**kw), oracle.base_tables(shape, np.float32))
    bar = max(2e-4 * np.max(np.abs(dth)), 2.0 * np.max(np.abs(np.asarray(dth32, dtype=np.float64) - dth)))
    assert np.max(np.abs(grad - dth)) <= bar, (grad, dth, bar)


def test_rotated_deep_tile_is_the_default_from_1024_tiles(eng):
    """From 1024 tiles of 16^3 per launch (five pairs of 96^3: 216 tiles each, every volume >= 128 tiles) GeomRD is offered without any
    flag: the default run equals the flagged run bit for bit and differs from the GeomR run (fp32 summation order) while agreeing with
    it to 2e-6.  (Smaller launches run the two-body kernel: their steps are launch-bound, see launch_dual.)"""
    from torchregister_amd import _lib
    shape = (96, 96, 96)
    tgt = ph.blobs(shape, 77).cuda().repeat(5, 1, 1, 1, 1)
    mov = (ph.blobs(shape, 78) + 0.1 * ph.vol(shape, 0.013, "sin")).cuda().repeat(5, 1, 1, 1, 1)
    th = torch.tensor(generic(ROT_DEEP_THETAS["rot_z_0.5"]), dtype=torch.float32)[None].repeat(5, 1, 1)
    runs = {}
    for flags in (0, _lib.FLAG_DEEP_TILE, _lib.FLAG_NO_ROT_DEEP_TILE):
        s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=0.0, init=th, capacity=1, flags=flags)
        s.run(1)
        torch.cuda.synchronize()
        runs[flags] = (s.losses[0, 0].item(), s.grad[0, :12].cpu().numpy())
    assert runs[0][0] == runs[_lib.FLAG_DEEP_TILE][0] and np.array_equal(runs[0][1], runs[_lib.FLAG_DEEP_TILE][1])
    g_r = runs[_lib.FLAG_NO_ROT_DEEP_TILE][1]
    assert not np.array_equal(runs[0][1], g_r)
    assert np.max(np.abs(runs[0][1] - g_r)) <= 2e-5 * np.max(np.abs(g_r)) and abs(runs[0][0] - runs[_lib.FLAG_NO_ROT_DEEP_TILE][0]) <= 2e-6 * abs(runs[0][0])


def test_rotated_deep_tile_mixed_batch(eng):
    """One batch whose pairs choose four different geometries (GeomA, GeomRD, GeomR, and the deep tile through its flag): the batch
    launch gives every pair what a single-pair launch gives it, bit for bit (the finalise kernel repeats each pair's choice)."""
    from torchregister_amd import _lib
    shape = (48, 40, 56)
    names = ["near", "rot_z_0.5", "general_0.45", "rot_x_0.6", "identity"]
    thetas = [DEEP_THETAS.get(n, ROT_DEEP_THETAS.get(n)) for n in names]
    th = torch.tensor(np.stack([t if n == "identity" else generic(t) for n, t in zip(names, thetas)]), dtype=torch.float32)
    B = len(names)
    tgt = torch.cat([ph.blobs(shape, 80 + b) for b in range(B)])
    mov = torch.cat([ph.blobs(shape, 90 + b) + 0.1 * ph.vol(shape, 0.013, "sin") for b in range(B)])
    for flags in (0, _lib.FLAG_DEEP_TILE):
        s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=1e-4, init=th, capacity=3, flags=flags)
        s.run(3)
        for b in range(B):
            one = eng.AffineSolver(mov[b:b + 1].cuda(), tgt[b:b + 1].cuda(), mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=1e-4, init=th[b:b + 1], capacity=3,
                                   flags=flags)
            one.run(3)
            assert torch.equal(one.losses[0, :3], s.losses[b, :3]), (names[b], flags)
            assert torch.equal(one.current_theta[0], s.current_theta[b]), (names[b], flags)

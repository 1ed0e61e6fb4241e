"""The exact-footprint F1 body (csrc/affine_eft.h): rotated poses of the step kernels.

(1) VERDICT r3 #1: the two rotated legs of bench.py at THEIR size - one launch of 8 x 256^3, affine + NCC at theta = R(0.5, 0.4, 0.3)
    diag(1.05, 0.95, 1.02), and rigid + NCC at the reference's initial pose (torch.manual_seed(0); torch.rand(6), ref:utils.py:316-321) -
    two pairs of each against the C oracle in fp64 (loss 2e-5 relative; gradient 2e-4 / 3e-4 of its maximum or twice the oracle's own
    fp32-vs-fp64 gap, the bars of tests/test_gpu_full_size.py and tests/test_gpu_tile_paths.py), and the whole batch against the same launch
    without the body (TRX_FLAG_NO_EFT: GeomR) to the fp32 floors between bodies.
(2) the body forced (TRX_FLAG_EFT) on small and ragged shapes, MSE-only steps and loss-only evaluation, mixed batches, volume faces:
    against the C oracle and against the tile kernels on the same inputs.
"""
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


def rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


def test_bench_rotated_legs_at_full_size_vs_oracle(eng):
    from test_gpu_full_size import blobs_gpu
    import torchregister_amd._lib as lib
    shape, B = (256, 256, 256), 8
    base = [blobs_gpu(shape, 4000 + i) for i in range(4)]
    tgt = torch.cat(base + [b.flip(2) for b in base[:2]] + [b.flip(3) for b in base[2:]])
    gen = np.stack([np.eye(3, 4) + 0.03 * np.cos(0.9 * np.arange(12) + 0.5 * b).reshape(3, 4) for b in range(B)])
    mov = eng.affine_warp(torch.tensor(gen, dtype=torch.float32).cuda(), tgt) + 0.05 * tgt.roll(1, 0)
    kw = dict(w_ncc=1.0)
    t64, t32 = oracle.base_tables(shape, np.float64), oracle.base_tables(shape, np.float32)
    # ---- affine mode at bench.py's value_rot pose (every pair; + a pair-dependent nudge so that the pairs differ)
    th0 = np.concatenate([rot(0.5, 0.4, 0.3) @ np.diag([1.05, 0.95, 1.02]), np.array([[0.01], [-0.02], [0.015]])], axis=1)
    ths = np.stack([th0 + 2e-3 * np.sin(1.1 * np.arange(12) + b).reshape(3, 4) for b in range(B)])
    th = torch.tensor(ths, dtype=torch.float32)
    s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1)
    s.run(1)
    sn = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1, flags=lib.FLAG_NO_EFT)
    sn.run(1)
    torch.cuda.synchronize()
    assert not torch.equal(s.grad, sn.grad), "the exact-footprint body did not run (same bits as the launch without it)"
    for b in range(B):
        assert abs(s.losses[b, 0].item() - sn.losses[b, 0].item()) <= 2e-5 * max(1.0, abs(sn.losses[b, 0].item())), b
        gb = sn.grad[b, :12]
        assert torch.max(torch.abs(s.grad[b, :12] - gb)).item() <= 2e-4 * gb.abs().max().item(), b
    for b in (2, 5):
        m, t = mov[b, 0].cpu().numpy(), tgt[b, 0].cpu().numpy()
        total, _, dth, _ = oracle.c_affine_loss_grad(m.astype(np.float64), t.astype(np.float64), th[b].double().numpy(), oracle.wts(**kw), t64)
        _, _, dth32, _ = oracle.c_affine_loss_grad(m, t, th[b].numpy(), oracle.wts(**kw), t32)
        assert abs(s.losses[b, 0].item() - total) <= 2e-5 * max(1.0, abs(total)), b
        gmax = np.max(np.abs(dth))
        assert np.max(np.abs(s.grad[b, :12].cpu().numpy().reshape(3, 4) - dth)) <= max(2e-4 * gmax, 2.0 * np.max(np.abs(dth32 - dth))), b
    # ---- rigid mode at the reference's initial pose (bench.py's value_rigid_randinit)
    torch.manual_seed(0)
    pose0 = torch.rand(6)
    poses = torch.stack([pose0 + 1e-3 * torch.sin(0.7 * torch.arange(6) + b) for b in range(B)]).float()
    r = eng.AffineSolver(mov, tgt, mode="rigid", loss=eng.LossSpec(**kw), lr=0.0, init=poses, capacity=1)
    r.run(1)
    rn = eng.AffineSolver(mov, tgt, mode="rigid", loss=eng.LossSpec(**kw), lr=0.0, init=poses, capacity=1, flags=lib.FLAG_NO_EFT)
    rn.run(1)
    torch.cuda.synchronize()
    assert not torch.equal(r.grad, rn.grad), "the exact-footprint body did not run (same bits as the launch without it)"
    for b in range(B):
        assert abs(r.losses[b, 0].item() - rn.losses[b, 0].item()) <= 2e-5 * max(1.0, abs(rn.losses[b, 0].item())), b
        gb = rn.grad[b, :6]
        assert torch.max(torch.abs(r.grad[b, :6] - gb)).item() <= 3e-4 * gb.abs().max().item(), b
    for b in (0, 7):
        m, t = mov[b, 0].cpu().numpy(), tgt[b, 0].cpu().numpy()
        p64 = poses[b].double().numpy()
        total, _, dth, _ = oracle.c_affine_loss_grad(m.astype(np.float64), t.astype(np.float64), oracle.c_theta_fwd(p64), oracle.wts(**kw), t64)
        dp = oracle.c_theta_vjp(p64, dth)
        p32n = poses[b].numpy()
        _, _, dth32, _ = oracle.c_affine_loss_grad(m, t, oracle.c_theta_fwd(p32n), oracle.wts(**kw), t32)
        dp32 = oracle.c_theta_vjp(p32n, dth32)
        assert abs(r.losses[b, 0].item() - total) <= 2e-5 * max(1.0, abs(total)), b
        assert np.max(np.abs(r.grad[b, :6].cpu().numpy() - dp)) <= max(3e-4 * np.max(np.abs(dp)), 2.0 * np.max(np.abs(dp32 - dp))), b


POSES = {
    "r543": (rot(0.5, 0.4, 0.3) @ np.diag([1.05, 0.95, 1.02]), [0.01, -0.02, 0.015]),
    "r786": (rot(0.7, 0.8, 0.6), [0.03, 0.01, -0.02]),
    "rneg": (rot(-0.6, 0.9, -0.45) @ np.diag([0.93, 1.06, 0.98]), [-0.2, 0.15, 0.1]),      # a good part of the volume maps outside the source
    "shear": (np.array([[0.9, 0.35, -0.2], [-0.3, 0.85, 0.4], [0.25, -0.35, 0.95]]), [0.05, 0.0, -0.05]),
    "zoomin": (rot(0.3, 0.5, 0.7) * 0.8, [0.0, 0.0, 0.0]),
    "far": (rot(0.4, 0.3, 0.6), [0.9, -0.7, 0.8]),                                          # most tiles miss the volume altogether
}


@pytest.mark.parametrize("shape", [(64, 64, 64), (48, 80, 32), (54, 83, 90), (33, 47, 21), (16, 16, 16), (96, 32, 40)])
@pytest.mark.parametrize("pname", list(POSES))
def test_forced_body_vs_oracle_and_tile_kernels(eng, shape, pname):
    """The body offered to every launch (TRX_FLAG_EFT): ragged shapes (partial tiles in x, y and z, W % 4 != 0: granules that straddle a
    face), poses that put tiles on and beyond the volume's faces, NCC + MSE and MSE-only steps; checked against the C oracle in fp64
    and against the same launch without the body."""
    import torchregister_amd._lib as lib
    A, tr = POSES[pname]
    th0 = np.concatenate([A, np.array(tr)[:, None]], axis=1)
    B = 3
    ths = np.stack([th0 + 3e-3 * np.sin(1.3 * np.arange(12) + b).reshape(3, 4) for b in range(B)])
    th = torch.tensor(ths, dtype=torch.float32)
    tgt = torch.cat([ph.blobs(shape, 300 + b) for b in range(B)])
    mov = torch.cat([ph.blobs(shape, 400 + b) for b in range(B)]) + 0.1 * torch.cat([ph.blobs(shape, 500 + b, nblob=9) for b in range(B)])
    tabs64, tabs32 = oracle.base_tables(shape, np.float64), oracle.base_tables(shape, np.float32)
    for kw in (dict(w_ncc=1.0, w_mse=0.5), dict(w_mse=1.0)):
        s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1, flags=lib.FLAG_EFT | lib.FLAG_DEEP_TILE)
        s.run(1)
        sn = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1, flags=lib.FLAG_NO_EFT | lib.FLAG_DEEP_TILE)
        sn.run(1)
        torch.cuda.synchronize()
        for b in range(B):
            m, t = mov[b, 0].numpy(), tgt[b, 0].numpy()
            total, _, dth, _ = oracle.c_affine_loss_grad(m.astype(np.float64), t.astype(np.float64), th[b].double().numpy(), oracle.wts(**kw), tabs64)
            _, _, dth32, _ = oracle.c_affine_loss_grad(m, t, th[b].numpy(), oracle.wts(**kw), tabs32)
            assert abs(s.losses[b, 0].item() - total) <= 2e-5 * max(1.0, abs(total)), (b, kw)
            gmax = np.max(np.abs(dth))
            bar = max(2e-4 * gmax, 2.0 * np.max(np.abs(dth32 - dth)))
            assert np.max(np.abs(s.grad[b, :12].cpu().numpy().reshape(3, 4) - dth)) <= bar, (b, kw)
            assert np.max(np.abs(sn.grad[b, :12].cpu().numpy().reshape(3, 4) - dth)) <= bar, (b, kw)


def test_mixed_batch_every_body_in_one_launch(eng):
    """One launch whose pairs sit at the identity, a small affine, a rotation about z, a general rotation and beyond what any plan holds
    (zoom-out): every body of the step kernel next to the exact-footprint one; each pair equals its own single-pair launch to the fp32
    floors, and the batch is bit-for-bit reproducible."""
    import torchregister_amd._lib as lib
    shape = (96, 96, 96)
    mats = [np.eye(3), np.eye(3) + 0.02 * np.sin(np.arange(9)).reshape(3, 3), rot(0, 0, 0.5), rot(0.5, 0.4, 0.3), rot(0.7, 0.8, 0.6) * 1.02, np.eye(3) * 1.9,
            rot(0.2, 0.2, 0.2), rot(0.45, 0.75, 0.1)]
    B = len(mats)
    th = torch.tensor(np.stack([np.concatenate([m, 0.01 * np.ones((3, 1))], axis=1) for m in mats]), dtype=torch.float32)
    tgt = torch.cat([ph.blobs_fast(shape, 600 + b, device="cuda") for b in range(B)])
    mov = torch.cat([ph.blobs_fast(shape, 700 + b, device="cuda") for b in range(B)])
    fl = lib.FLAG_EFT | lib.FLAG_DEEP_TILE | lib.FLAG_ZSTREAM
    runs = []
    for _ in range(2):
        s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=0.0, init=th, capacity=1, flags=fl)
        s.run(1)
        torch.cuda.synchronize()
        runs.append((s.losses.clone(), s.grad.clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    for b in range(B):
        s1 = eng.AffineSolver(mov[b:b + 1], tgt[b:b + 1], mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=0.0, init=th[b:b + 1], capacity=1, flags=lib.FLAG_NO_EFT)
        s1.run(1)
        torch.cuda.synchronize()
        assert abs(s1.losses[0, 0].item() - runs[0][0][b, 0].item()) <= 2e-5 * max(1.0, abs(s1.losses[0, 0].item())), b
        gb = s1.grad[0, :12]
        assert torch.max(torch.abs(runs[0][1][b, :12] - gb)).item() <= 2e-4 * gb.abs().max().item(), b


def test_rotated_trajectory_through_the_body(eng):
    """30 Adam iterations of a rigid run from a rotated pose, with and without the body: same loss curve and final pose to the fp32 floors
    (the body is chosen again at every step from the theta of that step)."""
    import torchregister_amd._lib as lib
    shape = (64, 64, 64)
    tgt = torch.cat([ph.blobs(shape, 810), ph.blobs(shape, 811)]).cuda()
    th_true = torch.tensor(np.stack([np.concatenate([rot(0.45, 0.35, 0.25), [[0.02], [0.01], [-0.02]]], axis=1)] * 2), dtype=torch.float32)
    mov = eng.affine_warp(th_true.cuda(), tgt)
    pose0 = torch.tensor([[0.3, 0.5, 0.2, 0.01, 0.0, 0.02], [0.5, 0.3, 0.35, -0.02, 0.01, 0.0]], dtype=torch.float32)
    out = []
    for fl in (lib.FLAG_EFT | lib.FLAG_DEEP_TILE, lib.FLAG_NO_EFT | lib.FLAG_DEEP_TILE):
        s = eng.AffineSolver(mov, tgt, mode="rigid", loss=eng.LossSpec(w_ncc=1.0), optimizer="adam", lr=2e-3, init=pose0, capacity=30, flags=fl)
        s.run(30)
        torch.cuda.synchronize()
        out.append((s.losses.clone(), s.param.clone()))
    lmax = out[1][0].abs().max().item()
    assert torch.max(torch.abs(out[0][0] - out[1][0])).item() <= 1e-4 * lmax
    assert torch.max(torch.abs(out[0][1] - out[1][1])).item() <= 1e-4
    assert (out[0][0][:, -1] < out[0][0][:, 0]).all()


def test_flat_grid_mixed_batch_against_oracle(eng):
    """A launch that runs the FLAT grid (8 x 128^3) whose pairs alternate between poses next to the identity (z-streaming body / GeomD in the
    fused kernel) and rotated ones (exact-footprint kernel in front of it): the two kernels share one partial-row workspace and one
    rows_used array.  Three pairs against the C oracle, every pair against its single-pair launch, rows_used < 0 exactly for the rotated pairs."""
    shape, B = (128, 128, 128), 8
    mats = [np.eye(3), rot(0.5, 0.4, 0.3), np.eye(3) + 0.01 * np.sin(np.arange(9)).reshape(3, 3), rot(0.45, 0.75, 0.1), rot(0, 0, 0.5), rot(0.7, 0.8, 0.6) * 1.03,
            np.eye(3) * 1.6, rot(0.3, 0.3, 0.3)]
    th = torch.tensor(np.stack([np.concatenate([m, 0.01 * np.ones((3, 1))], axis=1) + 1e-3 * np.cos(np.arange(12) + i).reshape(3, 4) for i, m in enumerate(mats)]), dtype=torch.float32)
    tgt = torch.cat([ph.blobs_fast(shape, 900 + b, device="cuda") for b in range(B)])
    mov = torch.cat([ph.blobs_fast(shape, 950 + b, device="cuda") for b in range(B)])
    kw = dict(w_ncc=1.0)
    s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1)
    s.run(1)
    torch.cuda.synchronize()
    rows = s.rows_used().tolist()
    # round 5: the z-streaming kernel runs FIRST and marks its pairs (0 and 2, next to the identity) negative like the exact-footprint kernel marks
    # the rotated ones; the tile kernel behind both keeps pairs 4 (R_z(0.5): GeomRD) and 6 (zoom 1.6: GeomR) and leaves their counts positive
    assert [r < 0 for r in rows] == [True, True, True, True, False, True, False, True], rows
    assert s.bodies() == ["zstream", "eft", "zstream", "eft", "tile-RD", "eft", "tile-R", "eft"], s.bodies()
    # the same launch with the z-streaming body kept inside the tile kernel (the round 3-4 form, TRX_FLAG_ZS_FUSED): same bodies, same rows
    import torchregister_amd._lib as lib
    sf = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1, flags=lib.FLAG_ZS_FUSED)
    sf.run(1)
    torch.cuda.synchronize()
    rf = sf.rows_used().tolist()
    assert [r < 0 for r in rf] == [False, True, False, True, False, True, False, True] and [abs(r) for r in rf] == [abs(r) for r in rows], (rf, rows)
    assert sf.bodies() == ["zstream-fused", "eft", "zstream-fused", "eft", "tile-RD", "eft", "tile-R", "eft"], sf.bodies()
    for b in range(B):
        assert abs(sf.losses[b, 0].item() - s.losses[b, 0].item()) <= 1e-6 * max(1.0, abs(s.losses[b, 0].item())), b
        assert torch.max(torch.abs(sf.grad[b, :12] - s.grad[b, :12])).item() <= 1e-5 * s.grad[b, :12].abs().max().item(), b
    t64, t32 = oracle.base_tables(shape, np.float64), oracle.base_tables(shape, np.float32)
    for b in (0, 1, 2, 5):
        m, t = mov[b, 0].cpu().numpy(), tgt[b, 0].cpu().numpy()
        total, _, dth, _ = oracle.c_affine_loss_grad(m.astype(np.float64), t.astype(np.float64), th[b].double().numpy(), oracle.wts(**kw), t64)
        _, _, dth32, _ = oracle.c_affine_loss_grad(m, t, th[b].numpy(), oracle.wts(**kw), t32)
        assert abs(s.losses[b, 0].item() - total) <= 2e-5 * max(1.0, abs(total)), b
        assert np.max(np.abs(s.grad[b, :12].cpu().numpy().reshape(3, 4) - dth)) <= max(2e-4 * np.max(np.abs(dth)), 2.0 * np.max(np.abs(dth32 - dth))), b
    for b in range(B):
        s1 = eng.AffineSolver(mov[b:b + 1], tgt[b:b + 1], mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th[b:b + 1], capacity=1)
        s1.run(1)
        torch.cuda.synchronize()
        assert abs(s1.losses[0, 0].item() - s.losses[b, 0].item()) <= 2e-5 * max(1.0, abs(s.losses[b, 0].item())), b
        gb = s.grad[b, :12]
        assert torch.max(torch.abs(s1.grad[0, :12] - gb)).item() <= 2e-4 * gb.abs().max().item(), b


def test_drawn_items_equal_assigned_items(eng):
    """Round 5: behind the z-streaming kernel the exact-footprint kernel's blocks DRAW their items (one ticket queue per XCD, stealing at the
    end) instead of taking every gridDim-th one.  Which block computes an item must not matter: a launch of 8 x 192^3 with five rotated
    pairs (720 items, 90 per queue: 26 of them drawn) equals, bit for bit, the launch with TRX_FLAG_ZS_FUSED, whose exact-footprint kernel
    runs in front with the items assigned - and repeats itself bit for bit although the drawing order differs from launch to launch."""
    import torchregister_amd._lib as lib
    shape, B = (192, 192, 192), 8
    mats = [rot(0.5, 0.4, 0.3), np.eye(3), rot(0.45, 0.75, 0.1), rot(0.3, 0.3, 0.3), np.eye(3) * 1.01, rot(0.7, 0.8, 0.6), np.eye(3), rot(0.4, 0.5, 0.6)]
    th = torch.tensor(np.stack([np.concatenate([m, 0.01 * np.ones((3, 1))], axis=1) for m in mats]), dtype=torch.float32)
    tgt = torch.cat([ph.blobs_fast(shape, 1000 + b, device="cuda") for b in range(B)])
    mov = torch.cat([ph.blobs_fast(shape, 1050 + b, device="cuda") for b in range(B)])
    out = []
    for fl in (0, 0, lib.FLAG_ZS_FUSED):
        s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=0.0, init=th, capacity=1, flags=fl)
        s.run(1)
        torch.cuda.synchronize()
        out.append((s.losses.clone(), s.grad.clone(), s.bodies()))
    assert out[0][2] == ["eft", "zstream", "eft", "eft", "zstream", "eft", "zstream", "eft"], out[0][2]
    assert out[2][2] == ["eft", "zstream-fused", "eft", "eft", "zstream-fused", "eft", "zstream-fused", "eft"], out[2][2]
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    rotated = [0, 2, 3, 5, 7]
    assert torch.equal(out[0][0][rotated], out[2][0][rotated]) and torch.equal(out[0][1][rotated], out[2][1][rotated])


def test_work_list_backwards_equals_forwards(eng):
    """Round 5: on odd iterations of a run (TRX_FLAG_WALK_DOWN) the exact-footprint kernel's queues and the tile kernel's work list take the
    pairs in the opposite order (they start on what the Infinity Cache still holds).  Same items, same sums: a step of 8 x 192^3 with rotated
    pairs (exact-footprint kernel), pairs rotated about z (GeomRD in the tile kernel) and pairs next to the identity equals itself bit for
    bit with the flag set - except for the z-streaming pairs, whose WALK along z changes the summation order (their test is in test_gpu_zstream.py)."""
    import torchregister_amd._lib as lib
    shape, B = (192, 192, 192), 8
    mats = [rot(0.5, 0.4, 0.3), rot(0, 0, 0.6), rot(0.45, 0.75, 0.1), np.eye(3), rot(0, 0, 1.0), rot(0.7, 0.8, 0.6), rot(0, 0, 0.5), rot(0.4, 0.5, 0.6)]
    th = torch.tensor(np.stack([np.concatenate([m, 0.01 * np.ones((3, 1))], axis=1) for m in mats]), dtype=torch.float32)
    tgt = torch.cat([ph.blobs_fast(shape, 1100 + b, device="cuda") for b in range(B)])
    mov = torch.cat([ph.blobs_fast(shape, 1150 + b, device="cuda") for b in range(B)])
    out = []
    for fl in (lib.FLAG_NO_PINGPONG, lib.FLAG_NO_PINGPONG | lib.FLAG_WALK_DOWN):
        s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=0.0, init=th, capacity=1, flags=fl)
        s.run(1)
        torch.cuda.synchronize()
        out.append((s.losses.clone(), s.grad.clone(), s.bodies()))
    assert out[0][2] == out[1][2] == ["eft", "tile-RD", "eft", "zstream", "tile-RD", "eft", "tile-RD", "eft"], out[0][2]
    same = [0, 1, 2, 4, 5, 6, 7]
    assert torch.equal(out[0][0][same], out[1][0][same]) and torch.equal(out[0][1][same], out[1][1][same])
    assert abs(out[0][0][3, 0].item() - out[1][0][3, 0].item()) <= 1e-6 * max(1.0, abs(out[0][0][3, 0].item()))

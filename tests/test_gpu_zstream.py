"""GPU parity of the z-streaming F1 body (csrc/affine_zstream.h): the per-pair choice the step kernels make for transforms next
to the identity.  TRX_FLAG_ZSTREAM offers it whatever the batch size (by default only launches that fill the chip get it), so small
volumes exercise every part of it: window origin at volume faces (zero padding in x / y / z), several z segments, the ring wrap,
translations by whole voxels, pairs of one batch choosing different bodies.
Checker: the C oracle in fp64 (tolerance: loss 2e-5 rel, gradient 2e-4 of max - the fp32 floors of test_gpu_affine.py) and the
tile kernels on the same inputs (TRX_FLAG_NO_ZSTREAM)."""
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


def near_identity(seed, eps, shift=(0.0, 0.0, 0.0)):
    """identity + eps * (irrational-ish pattern in [-1, 1]) - never puts whole lines of samples on integer coordinates"""
    k = np.arange(12, dtype=np.float64).reshape(3, 4)
    th = np.eye(3, 4) + eps * np.sin(1.2345 * (k + 1.0) + 0.77 * seed)
    th[:, 3] += np.asarray(shift)
    return th


# (D, H, W): W % 64 == 0, H % 32 == 0 - the shapes the body tiles
SHAPES = [(16, 32, 64), (40, 64, 64), (24, 32, 128), (33, 96, 64), (70, 64, 128)]
CASES = [("eps3e-3", 3e-3, (0, 0, 0)), ("eps1e-2", 1e-2, (0, 0, 0)), ("shift", 4e-3, (0.11, -0.07, 0.2)), ("eps2e-2", 2e-2, (0.01, 0.02, -0.03))]


def ran_zstream(solver, eng):
    """rows_used[b] (the dual kernel's note to the finalise kernel) == the z-streaming body's block count?"""
    return solver.rows_used().tolist()


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("kw", [dict(w_ncc=1.0, w_mse=0.5), dict(w_mse=1.0, w_ssd=0.3)], ids=["ncc", "mse"])
def test_zstream_step_vs_oracle(eng, shape, case, kw):
    from torchregister_amd import _lib
    _, eps, shift = case
    tgt = ph.blobs(shape, 31)
    mov = ph.blobs(shape, 32) + 0.1 * ph.vol(shape, 0.013, "sin")
    th = torch.tensor(near_identity(sum(shape), eps, shift), dtype=torch.float32)[None]
    out = {}
    for name, flags in (("zs", _lib.FLAG_ZSTREAM), ("tile", _lib.FLAG_NO_ZSTREAM)):
        s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1, flags=flags)
        s.run(1)
        torch.cuda.synchronize()
        out[name] = (s.losses[0, 0].item(), s.grad[0, :12].cpu().numpy().reshape(3, 4), s.rows_used().tolist())
    total, _, dth, _ = oracle.c_affine_loss_grad(mov[0, 0].double().numpy(), tgt[0, 0].double().numpy(), th[0].double().numpy(), oracle.wts(**kw),
                                                 oracle.base_tables(shape, np.float64))
    for name in ("zs", "tile"):
        loss, grad, _ = out[name]
        assert abs(loss - total) <= 2e-5 * max(1.0, abs(total)), (name, loss, total)
        assert np.max(np.abs(grad - dth)) <= 2e-4 * np.max(np.abs(dth)), (name, grad, dth)
    assert out["zs"][2] != out["tile"][2] or eps >= 2e-2, "the z-streaming body did not run (same row count as the tile kernels)"


def test_zstream_identity_is_exact(eng):
    """theta = identity: every sample sits on a voxel, the warp is the identity: fused MSE == mean((moving - target)^2), and the
    loss is bitwise the tile kernels' (same per-voxel arithmetic; only the order of the block sums differs -> 1e-6)."""
    from torchregister_amd import _lib
    shape = (24, 64, 128)
    tgt, mov = ph.blobs(shape, 5), ph.blobs(shape, 6)
    th = torch.eye(3, 4)[None]
    vals = []
    for flags in (_lib.FLAG_ZSTREAM, _lib.FLAG_NO_ZSTREAM):
        s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(w_mse=1.0), lr=0.0, init=th, capacity=1, flags=flags)
        s.run(1)
        torch.cuda.synchronize()
        vals.append(s.losses[0, 0].item())
    ref = ((mov.double() - tgt.double()) ** 2).mean().item()
    assert abs(vals[0] - ref) <= 2e-6 * ref and abs(vals[0] - vals[1]) <= 2e-6 * ref


def test_zstream_mixed_batch_and_trajectory(eng):
    """One batch whose pairs choose different bodies (z-streaming / GeomA / GeomR) follows the single-geometry launches; ten Adam
    iterations from the identity stay on the tile kernels' trajectory."""
    from torchregister_amd import _lib
    shape = (32, 64, 64)
    n = 4
    tgt = torch.cat([ph.blobs(shape, 300 + i) for i in range(n)])
    mov = torch.cat([ph.blobs(shape, 400 + i) for i in range(n)])
    ths = [near_identity(1, 4e-3), near_identity(2, 0.08), np.eye(3, 4), near_identity(3, 0.3)]
    th = torch.stack([torch.tensor(t, dtype=torch.float32) for t in ths])
    res = []
    for flags in (_lib.FLAG_ZSTREAM, _lib.FLAG_NO_ZSTREAM):
        s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(w_ncc=1.0), optimizer="adam", lr=1e-3, init=th, capacity=10, flags=flags)
        s.run(10)
        torch.cuda.synchronize()
        res.append((s.losses.clone(), s.theta.clone(), s.rows_used().tolist()))
    assert len(set(res[0][2])) > 1, res[0][2]            # pairs of the launch ran different bodies
    assert torch.allclose(res[0][0], res[1][0], rtol=3e-5, atol=3e-5)
    assert torch.allclose(res[0][1], res[1][1], rtol=0, atol=3e-5)


def test_zstream_default_choice_at_headline_share(eng):
    """8 x 128 x 128 x 128 ... the default (no flag) offers the body to launches that fill the chip: 8 pairs of 256^3 is the
    headline; here a smaller launch that still qualifies (B = 16 x 64 x 128 x 128: 16 x 8 columns x 4 segments) against the oracle."""
    shape = (64, 128, 128)
    B = 16
    tgt = torch.cat([ph.blobs(shape, 500 + i) for i in range(2)]).repeat(B // 2, 1, 1, 1, 1)
    mov = torch.cat([ph.blobs(shape, 600 + i) for i in range(2)]).repeat(B // 2, 1, 1, 1, 1)
    th = torch.tensor(near_identity(7, 5e-3), dtype=torch.float32)[None].repeat(B, 1, 1)
    s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=0.0, init=th, capacity=1)
    s.run(1)
    torch.cuda.synchronize()
    for i in (0, 1):
        total, _, dth, _ = oracle.c_affine_loss_grad(mov[i, 0].double().numpy(), tgt[i, 0].double().numpy(), th[i].double().numpy(), oracle.wts(w_ncc=1.0),
                                                     oracle.base_tables(shape, np.float64))
        assert abs(s.losses[i, 0].item() - total) <= 2e-5 * max(1.0, abs(total))
        assert np.max(np.abs(s.grad[i, :12].cpu().numpy().reshape(3, 4) - dth)) <= 2e-4 * np.max(np.abs(dth))
    assert torch.equal(s.losses[0], s.losses[2]) and torch.equal(s.grad[1], s.grad[3])   # slot independence, bit for bit
    assert set(s.bodies()) == {"zstream"}, s.bodies()   # (round 5: the z-streaming kernel, in front of the tile kernel, took every pair)


@pytest.mark.parametrize("B", [64, 65])
def test_largest_flat_grid_batch_and_the_first_classic_one(eng, B):
    """64 pairs is the largest batch the flat persistent grid takes (per-pair choices live in one wave's lanes), 65 the first that goes
    back to the (blocks, pairs) grid: both against single-pair launches of a few of their pairs (fp32 floors between bodies) and with
    per-pair bodies mixed in the batch (every fourth pair far from the identity)."""
    shape = (64, 64, 64)
    base_t = [ph.blobs(shape, 900 + i) for i in range(4)]
    base_m = [ph.blobs(shape, 950 + i) for i in range(4)]
    tgt = torch.cat([base_t[i % 4] for i in range(B)])
    mov = torch.cat([base_m[(i + i // 4) % 4] for i in range(B)])
    ths = [near_identity(i, 0.25 if i % 4 == 3 else 4e-3 + 1e-4 * i) for i in range(B)]
    th = torch.stack([torch.tensor(t, dtype=torch.float32) for t in ths])
    s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(w_ncc=1.0, w_mse=0.2), lr=0.0, init=th, capacity=1)
    s.run(1)
    torch.cuda.synchronize()
    rows = s.rows_used().tolist()
    assert len(set(rows)) > 1, rows                       # streaming pairs and tile pairs in one launch
    assert torch.isfinite(s.losses[:, 0]).all() and torch.isfinite(s.grad).all()
    for b in (0, 3, B // 2, B - 2, B - 1):
        s1 = eng.AffineSolver(mov[b:b + 1].cuda(), tgt[b:b + 1].cuda(), mode="affine", loss=eng.LossSpec(w_ncc=1.0, w_mse=0.2), lr=0.0, init=th[b:b + 1], capacity=1)
        s1.run(1)
        torch.cuda.synchronize()
        assert abs(s1.losses[0, 0].item() - s.losses[b, 0].item()) <= 2e-5 * max(1.0, abs(s.losses[b, 0].item())), b
        gb = s.grad[b, :12]
        assert torch.max(torch.abs(s1.grad[0, :12] - gb)).item() <= 2e-4 * gb.abs().max().item(), b


def test_zstream_offer_boundary_never_trips(eng):
    """VERDICT r4 #7 / ADVICE r3: the body's own per-anchor window test (exact, from the block's corners) reports a violation as NaN rows - it
    "cannot happen while zs_nsub holds" (the offer rule is the per-anchor test's worst case over the alignment of the window origin, with 0.3
    voxels to spare).  This sweep walks TO the edge of the offer rule: for random directions in theta space (all twelve entries, or only the
    shears / only the diagonal / only the translations scaled up) it bisects the largest step from the identity that the rule still accepts
    (AffineSolver.bodies() says which body ran), and checks there - where the window is as full as the rule ever lets it get, for every
    block of the pair at once - that the body's result is finite and equal to the tile kernels' to the fp32 floor.  60 directions x 3 shapes."""
    from torchregister_amd import _lib
    rng = np.random.default_rng(20261003)
    shapes = [(40, 32, 64), (72, 64, 128), (130, 96, 64)]     # one z segment without / with re-anchoring (>= 64 and >= 128 planes per segment)
    tripped, edges = 0, []
    for shape in shapes:
        tgt = ph.blobs(shape, 77).cuda()
        mov = (ph.blobs(shape, 78) + 0.1 * ph.vol(shape, 0.013, "sin")).cuda()

        def run(th, flags):
            s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0, w_mse=0.3), lr=0.0, init=torch.tensor(th, dtype=torch.float32)[None], capacity=1, flags=flags)
            s.run(1)
            torch.cuda.synchronize()
            return s.losses[0, 0].item(), s.grad[0, :12].cpu().numpy(), s.bodies()[0]
        for k in range(20):
            d = rng.standard_normal((3, 4))
            kind = k % 4
            if kind == 1: d[:, 3] = 0; d[np.arange(3), np.arange(3)] = 0          # shears / rotations only
            if kind == 2: d = np.diag(rng.standard_normal(3)) @ np.eye(3, 4)     # zooms only
            if kind == 3: d[:, :3] *= 0.05                                          # mostly translation
            d /= np.max(np.abs(d))
            lo, hi = 0.0, 0.5
            for _ in range(14):
                mid = 0.5 * (lo + hi)
                body = run(np.eye(3, 4) + mid * d, _lib.FLAG_ZSTREAM)[2]
                if body.startswith("zstream"): lo = mid
                else: hi = mid
            th = np.eye(3, 4) + lo * d
            loss, grad, body = run(th, _lib.FLAG_ZSTREAM)
            assert body.startswith("zstream"), (shape, k, lo)
            loss_t, grad_t, body_t = run(th, _lib.FLAG_NO_ZSTREAM)
            assert not body_t.startswith("zstream")
            if not (np.isfinite(loss) and np.all(np.isfinite(grad))):
                tripped += 1
                continue
            edges.append(lo)
            assert abs(loss - loss_t) <= 2e-5 * max(1.0, abs(loss_t)), (shape, k, lo, loss, loss_t)
            assert np.max(np.abs(grad - grad_t)) <= 4e-4 * np.max(np.abs(grad_t)), (shape, k, lo)
    assert tripped == 0, f"{tripped} poses at the edge of the offer rule tripped the body's own window test"
    assert min(edges) > 1e-3 and np.median(edges) < 0.2, (min(edges), np.median(edges), max(edges))   # the bisection found the rule's edge (directions that are mostly translation have none: the window follows)


def _rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    return np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])


def test_zstream_flat_tile_in_the_convergence_basin(eng):
    """Round 5: the z-streaming kernel's FLAT tile (64 x 16 voxels per plane under a ring of 8 planes of 80 x 30: csrc/affine_zstream.h ZSF) takes the
    pairs of a chip-filling launch that the 64 x 32 tile's window does not hold - rotations up to ~0.15 rad about z, zooms to ~1.1, a few
    planes of tilt: where an affine run converges to.  One launch of 16 pairs of 64 x 128 x 128 at sixteen such poses (and two that neither tile
    takes): every pair against the same launch with TRX_FLAG_NO_ZS_FLAT (the tile kernels) to the fp32 floor, four of them against the C oracle."""
    from torchregister_amd import _lib
    shape, B = (64, 128, 128), 16
    rng = np.random.default_rng(7)
    mats = []
    for i in range(B):
        A = _rot(rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02), rng.choice([-1.0, 1.0]) * rng.uniform(0.07, 0.14)) @ np.diag(1.0 + rng.uniform(-0.04, 0.08, 3))
        if i == 5: A = _rot(0.0, 0.0, 0.5)              # beyond both tiles: GeomRD / exact-footprint
        if i == 11: A = np.eye(3) + 0.002               # inside the 64 x 32 tile's window
        mats.append(np.concatenate([A, rng.uniform(-0.05, 0.05, (3, 1))], axis=1))
    th = torch.tensor(np.stack(mats), dtype=torch.float32)
    tgt = torch.cat([ph.blobs(shape, 1200 + b) for b in range(B)]).cuda()
    mov = torch.cat([ph.blobs(shape, 1300 + b) + 0.1 * ph.vol(shape, 0.017, "sin") for b in range(B)]).cuda()
    kw = dict(w_ncc=1.0, w_mse=0.3)
    s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1)
    s.run(1)
    sn = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1, flags=_lib.FLAG_NO_ZS_FLAT)
    sn.run(1)
    torch.cuda.synchronize()
    bodies, bodies_n = s.bodies(), sn.bodies()
    assert bodies.count("zstream-flat") >= 8 and bodies[11] == "zstream" and not bodies[5].startswith("zstream"), bodies
    assert "zstream-flat" not in bodies_n, bodies_n
    for b in range(B):
        assert abs(s.losses[b, 0].item() - sn.losses[b, 0].item()) <= 2e-5 * max(1.0, abs(sn.losses[b, 0].item())), (b, bodies[b])
        gb = sn.grad[b, :12]
        assert torch.max(torch.abs(s.grad[b, :12] - gb)).item() <= 3e-4 * gb.abs().max().item(), (b, bodies[b])
    t64, t32 = oracle.base_tables(shape, np.float64), oracle.base_tables(shape, np.float32)
    flat = [b for b in range(B) if bodies[b] == "zstream-flat"]
    for b in flat[:4]:
        m, t = mov[b, 0].cpu().numpy(), tgt[b, 0].cpu().numpy()
        total, _, dth, _ = oracle.c_affine_loss_grad(m.astype(np.float64), t.astype(np.float64), th[b].double().numpy(), oracle.wts(**kw), t64)
        _, _, dth32, _ = oracle.c_affine_loss_grad(m, t, th[b].numpy(), oracle.wts(**kw), t32)
        assert abs(s.losses[b, 0].item() - total) <= 2e-5 * max(1.0, abs(total)), b
        assert np.max(np.abs(s.grad[b, :12].cpu().numpy().reshape(3, 4) - dth)) <= max(2e-4 * np.max(np.abs(dth)), 2.0 * np.max(np.abs(dth32 - dth))), b


@pytest.mark.parametrize("shape", [(16, 32, 64), (40, 64, 64), (70, 64, 128), (150, 32, 64)])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_zstream_walk_down_equals_walk_up(eng, shape, case):
    """Round 5: TRX_FLAG_WALK_DOWN - the z-streaming columns walked from the last plane to the first (what every second iteration of
    trx_affine_run does, so that a pass starts on the planes the previous one left in the Infinity Cache).  Same voxels, same per-voxel
    arithmetic, another order of the sums along z: loss and gradient equal the upward walk's to the fp32 floor and meet the oracle's bars.
    (150 planes: a column that re-anchors its window; the whole-voxel shifts of the 'shift' case put samples on lattice planes.)"""
    from torchregister_amd import _lib
    _, eps, shift = case
    tgt = ph.blobs(shape, 41)
    mov = ph.blobs(shape, 42) + 0.1 * ph.vol(shape, 0.013, "sin")
    th = torch.tensor(near_identity(sum(shape), eps, shift), dtype=torch.float32)[None]
    kw = dict(w_ncc=1.0, w_mse=0.5)
    out = {}
    for name, flags in (("up", _lib.FLAG_ZSTREAM), ("down", _lib.FLAG_ZSTREAM | _lib.FLAG_WALK_DOWN)):
        s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1, flags=flags)
        s.run(1)
        torch.cuda.synchronize()
        out[name] = (s.losses[0, 0].item(), s.grad[0, :12].cpu().numpy().reshape(3, 4), s.bodies()[0])
    assert out["up"][2] == out["down"][2]
    from fuzz_affine import kink_variants
    m64, t64, tu, tabs = mov[0, 0].double().numpy(), tgt[0, 0].double().numpy(), th[0].double().numpy(), oracle.base_tables(shape, np.float64)
    total, _, dth, _ = oracle.c_affine_loss_grad(m64, t64, tu, oracle.wts(**kw), tabs)
    gmax = np.max(np.abs(dth))
    # (next to the identity bands of voxels sample within fp32 rounding of a lattice plane: the oracle's own sensitivity to a one-ulp nudge of
    #  the translations widens the bar as in tests/fuzz_zstream.py - it is the same for both directions, which share every coordinate)
    ksens = max(np.max(np.abs(oracle.c_affine_loss_grad(m64, t64, t, oracle.wts(**kw), tabs)[2] - dth)) for t in kink_variants(tu)) / gmax
    gbar = max(2e-4, 2.0 * ksens)
    for name in ("up", "down"):
        loss, grad, _ = out[name]
        assert abs(loss - total) <= 2e-5 * max(1.0, abs(total)), (name, loss, total)
        assert np.max(np.abs(grad - dth)) <= gbar * gmax, (name, np.max(np.abs(grad - dth)) / gmax, gbar)
    if out["up"][2].startswith("zstream"):   # the same voxels with the same coordinates in another order of summation
        assert abs(out["up"][0] - out["down"][0]) <= 2e-6 * max(1.0, abs(total))
        assert np.max(np.abs(out["up"][1] - out["down"][1])) <= 2e-5 * gmax, np.max(np.abs(out["up"][1] - out["down"][1])) / gmax


def test_run_alternates_the_walk_and_follows_the_one_way_trajectory(eng):
    """trx_affine_run toggles TRX_FLAG_WALK_DOWN on odd iterations (TRX_FLAG_NO_PINGPONG: never).  A 12-iteration Adam run of a chip-filling
    launch (16 x 64 x 128 x 128: z-streaming kernel, both tiles) with and without the alternation: same loss curve and theta to the fp32 floor."""
    from torchregister_amd import _lib
    shape, B = (64, 128, 128), 16
    tgt = torch.cat([ph.blobs(shape, 500 + i) for i in range(2)]).repeat(B // 2, 1, 1, 1, 1).cuda()
    mov = torch.cat([ph.blobs(shape, 600 + i) for i in range(2)]).repeat(B // 2, 1, 1, 1, 1).cuda()
    th = torch.stack([torch.tensor(near_identity(7 + i, 5e-3 if i % 3 else 0.09), dtype=torch.float32) for i in range(B)])
    res = []
    for flags in (0, _lib.FLAG_NO_PINGPONG):
        s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), optimizer="adam", lr=1e-3, init=th, capacity=12, flags=flags)
        s.run(12)
        torch.cuda.synchronize()
        res.append((s.losses.clone(), s.theta.clone(), s.bodies()))
    assert any(b.startswith("zstream") for b in res[0][2]), res[0][2]
    assert torch.allclose(res[0][0], res[1][0], rtol=3e-5, atol=3e-5)
    assert torch.allclose(res[0][1], res[1][1], rtol=0, atol=3e-5)

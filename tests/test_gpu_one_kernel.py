"""Round 6: the ONE-KERNEL form of a step (TRX_FLAG_ONE_KERNEL, include/trx.h).  Launches that fill the chip run the z-streaming kernel alone -
no exact-footprint kernel, no tile kernel behind it - and that kernel runs the pairs outside its two windows itself on GeomR's body.  The flag
is a hint about speed: every test here checks that results do not depend on it beyond the fp32 floors between kernel bodies.
Checkers: the C oracle in fp64 (loss 2e-5 rel, gradient 2e-4 of its maximum - the floors of test_gpu_affine.py) and the three-kernel form of
the same step (one_kernel=False)."""
import ctypes

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


def near_identity(seed, eps):
    k = np.arange(12, dtype=np.float64).reshape(3, 4)
    return np.eye(3, 4) + eps * np.sin(1.2345 * (k + 1.0) + 0.77 * seed)


def rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


def oracle_check(mov, tgt, th, i, loss, grad, kw, floor=2e-4):
    """loss 2e-5 rel; gradient `floor` of its maximum, widened as the randomised sweeps do (tests/fuzz_zstream.py): twice the oracle's own
    fp32-vs-fp64 gap and twice its sensitivity to a one-ulp nudge of the translations (next to the identity whole bands of voxels sample
    within fp32 rounding of a lattice plane, where the trilinear derivative is one-sided)."""
    from fuzz_affine import kink_variants
    m64, t64, tu = mov[i, 0].double().cpu().numpy(), tgt[i, 0].double().cpu().numpy(), th[i].double().numpy()
    shape = tuple(mov.shape[2:])
    t64s, t32s = oracle.base_tables(shape, np.float64), oracle.base_tables(shape, np.float32)
    total, _, dth, _ = oracle.c_affine_loss_grad(m64, t64, tu, oracle.wts(**kw), t64s)
    _, _, dth32, _ = oracle.c_affine_loss_grad(mov[i, 0].cpu().numpy(), tgt[i, 0].cpu().numpy(), th[i].numpy(), oracle.wts(**kw), t32s)
    gmax = np.max(np.abs(dth))
    ksens = max(np.max(np.abs(oracle.c_affine_loss_grad(m64, t64, t, oracle.wts(**kw), t64s)[2] - dth)) for t in kink_variants(tu)) / gmax
    bar = max(floor, 2.0 * np.max(np.abs(dth32 - dth)) / gmax, 2.0 * ksens)
    assert abs(loss - total) <= 2e-5 * max(1.0, abs(total)), (i, loss, total)
    e = np.max(np.abs(grad.cpu().numpy().reshape(3, 4) - dth)) / gmax
    assert e <= bar, (i, e, bar, ksens)


SHAPE, B = (64, 128, 128), 16   # 16 pairs x 8 columns x 4 z segments: the smallest launch of tests/test_gpu_zstream.py that fills the chip


def batch(seed, rough=True):
    """rough: + 0.1 sin(0.013 i) over the FLAT voxel index - smooth along x, a ~4-row ripple along y and z: the near-identity tests keep it (as
    tests/test_gpu_zstream.py does), the rotated poses use smooth volumes (under a rotation the ripple turns into an fp32-vs-fp64 gap of 1e-3 of the
    gradient in EVERY body and in the oracle's own fp32 run - tests/fuzz_affine.py, "smooth phantoms only")."""
    tgt = torch.cat([ph.blobs_fast(SHAPE, seed + i, device="cuda") for i in range(B)])
    mov = torch.cat([ph.blobs_fast(SHAPE, seed + 50 + i, device="cuda") + (0.1 * ph.vol(SHAPE, 0.013 + 0.001 * i, "sin").cuda() if rough else 0.0) for i in range(B)])
    return mov, tgt


def test_near_the_identity_the_flag_sets_itself_and_changes_nothing(eng):
    """Every pair inside the streaming windows: "auto" sets the flag from the initial thetas (trx_affine_near_identity), the step is the z-streaming
    kernel alone, and losses / gradients equal the three-kernel form's (same body, same sums) and the oracle's."""
    mov, tgt = batch(100)
    th = torch.stack([torch.tensor(near_identity(i, 4e-3 + 5e-4 * i), dtype=torch.float32) for i in range(B)])
    out = {}
    for name, ok in (("one", "auto"), ("three", False)):
        s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=0.0, init=th, capacity=1, one_kernel=ok)
        assert s.one_kernel == (name == "one")
        s.run(1)
        torch.cuda.synchronize()
        out[name] = (s.losses[:, 0].clone(), s.grad.clone(), s.bodies())
    assert set(out["one"][2]) <= {"zstream", "zstream-flat"} and out["one"][2] == out["three"][2], (out["one"][2], out["three"][2])
    assert torch.allclose(out["one"][0], out["three"][0], rtol=2e-6, atol=2e-6)
    assert torch.max(torch.abs(out["one"][1] - out["three"][1])).item() <= 2e-5 * out["three"][1].abs().max().item()
    for i in (0, 9):
        oracle_check(mov, tgt, th, i, out["one"][0][i].item(), out["one"][1][i, :12], dict(w_ncc=1.0))


@pytest.mark.parametrize("kw", [dict(w_ncc=1.0), dict(w_mse=1.0, w_ssd=0.2)], ids=["ncc", "mse"])
def test_stray_pairs_run_inside_the_kernel(eng, kw):
    """The flag FORCED on a batch in which every third pair is far outside the windows (general rotations, a zoom, a flip): those pairs run GeomR's body
    inside the streaming kernel ("tile-R"), the others stream; all of them against the three-kernel form (fp32 floors between bodies), and a streaming
    pair, a rotated pair and the flipped pair against the oracle."""
    mov, tgt = batch(300, rough=False)
    mats = [rot(0.5, 0.4, 0.3), rot(0, 0, 0.6) * 1.05, np.diag([1.3, 0.8, 1.1]), rot(0.7, 0.8, 0.6), np.diag([-1.0, 1.0, 1.0]), rot(0.2, 0.0, 0.0)]
    ths = []
    for i in range(B):
        if i % 3 == 2:
            m = mats[(i // 3) % len(mats)]
            ths.append(np.concatenate([m, [[0.011], [-0.017], [0.013]]], axis=1) + 1e-3 * np.sin(np.arange(12.0).reshape(3, 4) + i))
        else:
            ths.append(near_identity(i, 5e-3))
    th = torch.tensor(np.stack(ths), dtype=torch.float32)
    out = {}
    for name, ok in (("one", True), ("three", False)):
        s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1, one_kernel=ok)
        s.run(1)
        torch.cuda.synchronize()
        out[name] = (s.losses[:, 0].clone(), s.grad.clone(), s.bodies())
    # (the mirror image diag(-1, 1, 1) is INSIDE the streaming window - the pre-image of a tile is as large as at the identity - and streams in both forms)
    stream = ("zstream", "zstream-flat")
    assert out["one"][2] == [b if b in stream else "tile-R" for b in out["three"][2]], (out["one"][2], out["three"][2])
    assert [b in stream for b in out["one"][2]] == [i % 3 != 2 or i == 14 for i in range(B)], out["one"][2]
    assert all(b != "tile-R" for b in out["three"][2])   # (the three-kernel form runs the strays on the exact-footprint / tile kernels)
    for i in range(B):
        l1, l3 = out["one"][0][i].item(), out["three"][0][i].item()
        assert abs(l1 - l3) <= 2e-5 * max(1.0, abs(l3)), (i, l1, l3)
        g3 = out["three"][1][i, :12]
        assert torch.max(torch.abs(out["one"][1][i, :12] - g3)).item() <= 3e-4 * g3.abs().max().item(), i
    for i in (0, 2, 14):
        oracle_check(mov, tgt, th, i, out["one"][0][i].item(), out["one"][1][i, :12], kw, floor=3e-4)


def test_a_run_follows_the_same_trajectory_and_the_policy_follows_the_run(eng):
    """(a) 12 Adam iterations next to the identity with and without the flag: one trajectory (loss curve, theta) to the fp32 floors.
    (b) the "auto" policy: a solver that starts at rotated poses leaves the flag off; one that starts at the identity sets it, and drops it for the call
    AFTER a call that ended with a pair outside the windows (a large step drives the pairs out) - read from pinned memory, never waited for."""
    mov, tgt = batch(500)
    th = torch.stack([torch.tensor(near_identity(i, 3e-3), dtype=torch.float32) for i in range(B)])
    res = []
    for ok in ("auto", False):
        s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), optimizer="adam", lr=5e-4, init=th, capacity=12, one_kernel=ok)
        s.run(12)
        torch.cuda.synchronize()
        res.append((s.losses.clone(), s.theta.clone()))
    assert torch.allclose(res[0][0], res[1][0], rtol=3e-5, atol=3e-5)
    assert torch.allclose(res[0][1], res[1][1], rtol=0, atol=3e-5)
    assert (res[0][0][:, -1] < res[0][0][:, 0]).all()
    # (b)
    th_rot = torch.tensor(np.concatenate([rot(0.5, 0.4, 0.3), np.zeros((3, 1))], axis=1), dtype=torch.float32)[None].repeat(B, 1, 1)
    assert not eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=0.0, init=th_rot, capacity=1).one_kernel
    s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), optimizer="adam", lr=0.08, capacity=40)   # (theta = identity)
    assert s.one_kernel
    s.run(6)                      # Adam moves every entry of theta by ~lr per step: 0.5 after six - far outside the windows
    torch.cuda.synchronize()      # (the test waits so that the copy of the notes HAS landed; run() itself never does)
    assert "tile-R" in s.bodies(), s.bodies()
    s.run(1)
    assert not s.one_kernel
    torch.cuda.synchronize()
    assert torch.isfinite(s.losses[:, :7]).all()   # (the three-kernel form ran this step: its tile kernel reports GeomR as "tile-R" too)
    # ... and back: a solver that starts rotated, is handed the identity, and finds the flag again after one call
    s2 = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=0.0, init=th_rot, capacity=8)
    s2.theta.copy_(eng.pad_theta(torch.eye(3, 4, device="cuda")[None].repeat(B, 1, 1), 3)); s2.param.copy_(s2.theta)
    s2.run(1)
    torch.cuda.synchronize()
    s2.run(1)
    assert s2.one_kernel


def test_near_identity_helper_is_the_kernels_test(eng):
    """trx_affine_near_identity (host thetas) answers what the kernel's window test answers on the device: bisect along a direction of theta space to
    the edge of the windows and compare with the body the one-kernel step reports on both sides."""
    from torchregister_amd import _lib
    lib = _lib.load()
    mov, tgt = batch(700)
    d = np.sin(1.7 * np.arange(12.0).reshape(3, 4) + 0.3)
    probe = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=0.0, capacity=1, one_kernel=True)

    def helper(eps):
        th = torch.tensor(np.eye(3, 4) + eps * d, dtype=torch.float32)[None].repeat(B, 1, 1)
        padded = eng.pad_theta(th, 3)   # (host memory: kept alive across the call)
        return bool(lib.trx_affine_near_identity(ctypes.byref(probe.vol), ctypes.c_void_p(padded.data_ptr()))), th

    lo, hi = 0.0, 0.5
    assert helper(lo)[0] and not helper(hi)[0]
    for _ in range(12):
        mid = 0.5 * (lo + hi)
        lo, hi = (mid, hi) if helper(mid)[0] else (lo, mid)
    for eps, inside in ((lo, True), (hi, False)):
        s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), lr=0.0, init=helper(eps)[1], capacity=1, one_kernel=True)
        s.run(1)
        torch.cuda.synchronize()
        assert all((b != "tile-R") == inside for b in s.bodies()), (eps, inside, s.bodies())
        assert torch.isfinite(s.losses[:, 0]).all()

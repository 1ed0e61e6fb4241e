"""Multi-iteration parity at the BASELINE.json sizes: full-length trajectories, not single evaluations.

  cfg2  3-D 128^3, one pair, affine + NCC, 200 iterations, SGD and Adam - through the tile geometries and (TRX_FLAG_ZSTREAM) through
        the z-streaming body: loss curve, final theta, best theta / index;
  cfg3  3-D 256^3, one pair, direct flow field + NCC + smoothness regulariser, 10 iterations of Adam (and SGD at 128^3): loss curve and
        final flow;
  cfg4  the per-GPU share of config 4 - ONE launch of 8 pairs of 256^3, affine + NCC, Adam, 20 iterations (the headline workload: the
        z-streaming body with its flat grid): three of the eight pairs against the arbiter, loss curve and theta; and the same launch in
        RIGID mode from the reference's random initial pose (the exact-footprint kernel at every step): one pair against the arbiter.

Arbiter: oracle/compose.py (the reference's loop re-composed from the ATen CPU ops at the reference's call sites,
ref:warpings.py:67-93 / :208-233) run in fp32 AND fp64 on the host; bar = max(stated floor, 2 x the arbiter's own fp32-vs-fp64 gap)
(conftest.bar); the arbiter's runs execute side by side in worker processes (tests/traj_workers.py).  Floors: loss curve 2e-5 of its maximum, theta 2e-6, flow 1e-5 voxels - the fp32 floors of tests/test_r2_fixtures.py.
Every run starts OFF the voxel lattice (a generic theta / a smooth non-zero flow): on the lattice the trilinear derivative is one-sided
and the arbiter's own two precisions disagree by 1e-3 ... 2e-2."""
import concurrent.futures
import multiprocessing

import numpy as np
import pytest
import torch

import traj_workers as tw
from conftest import bar

pytestmark = pytest.mark.gpu

S128, S256 = (128, 128, 128), (256, 256, 256)
CFG2 = {"sgd": 2e-6, "adam": 5e-4}
CFG3 = [("adam", 0.01, S256), ("sgd", 1.0, S128)]      # (the SGD variant of the flow loop at 128^3: the arbiter needs 9 s per 256^3 iteration)
CFG4_LR, CFG4_PAIRS = 1e-3, (0, 5, 7)
CFG4R_PAIR = 3
JOBS = {}
for _opt, _lr in CFG2.items():
    for _dt in ("float32", "float64"):
        JOBS[("cfg2", _opt, _dt)] = ("affine", S128, 1000, _dt, _opt, _lr, 200, 0)
for _opt, _lr, _shape in CFG3:
    for _dt in ("float32", "float64"):
        JOBS[("cfg3", _opt, _dt)] = ("flow", _shape, 1000, _dt, _opt, _lr, 10, 1.0)
for _dt in ("float32", "float64"):
    for _p in CFG4_PAIRS:
        JOBS[("cfg4", _p, _dt)] = ("affine", S256, 1000 + _p, _dt, "adam", CFG4_LR, 20, _p)
    JOBS[("cfg4r", "adam", _dt)] = ("rigid", S256, 1000 + CFG4R_PAIR, _dt, "adam", CFG4_LR, 20, CFG4R_PAIR)


_POOL = {}


def start_arbiter():
    """Submit every arbiter run of this module to worker processes (spawned: they must not inherit an initialised GPU).  conftest.py calls
    this when the collection ends and moves this module's tests behind all others: the arbiter's ~2 minutes of host time then pass under
    the rest of the GPU suite instead of in front of this module (round 6: the suite had grown to 800-1000 s of a 1200 s limit)."""
    if not _POOL:
        ctx = multiprocessing.get_context("spawn")
        ex = concurrent.futures.ProcessPoolExecutor(max_workers=len(JOBS), mp_context=ctx)
        _POOL["ex"] = ex
        _POOL["futs"] = {k: ex.submit(tw.run, j) for k, j in JOBS.items()}
    return _POOL


@pytest.fixture(scope="module")
def refs():
    """every arbiter run of this module (started at collection time, joined here)"""
    pool = start_arbiter()
    try:
        return {k: f.result() for k, f in pool["futs"].items()}
    finally:
        pool["ex"].shutdown(wait=False, cancel_futures=True)


def batch_of_pairs(B, compared):
    """The 8 x 256^3 batch: the pairs that go to the arbiter exactly as its workers build them (tw.pair, on the host), the others from the
    same family on the GPU (phantoms.blobs_fast; nobody compares them with anything but themselves)."""
    import phantoms as ph
    import torchregister_amd._engine as e
    movs, tgts = [], []
    for i in range(B):
        if i in compared:
            m, t = tw.pair(S256, 1000 + i)
            m, t = m.cuda(), t.cuda()
        else:
            t = ph.blobs_fast(S256, 1000 + i, device="cuda")
            m = e.affine_warp(torch.tensor(ph.THETA_STAR3, device="cuda")[None], t)
        movs.append(m); tgts.append(t)
    return torch.cat(movs), torch.cat(tgts)


@pytest.fixture(scope="module")
def eng():
    import torchregister_amd._engine as e
    assert torch.cuda.is_available()
    return e


def check_affine(got_losses, got_thetas, ref32, ref64, tag):
    l32, l64 = ref32["losses"], ref64["losses"]
    e = np.max(np.abs(got_losses - l64))
    b = bar(l32, l64, 2e-5 * np.max(np.abs(l64)))
    assert e <= b, (tag, "loss curve", e, b)
    t32, t64 = ref32["thetas"][-1], ref64["thetas"][-1]
    e = np.max(np.abs(got_thetas - t64))
    b = bar(t32, t64, 2e-6)
    assert e <= b, (tag, "final theta", e, b)


@pytest.mark.parametrize("optimizer", list(CFG2))
@pytest.mark.parametrize("zs", [False, True], ids=["tiles", "zstream"])
def test_cfg2_affine_ncc_128_200_iterations(eng, refs, optimizer, zs):
    from torchregister_amd import _lib
    iters, lr = 200, CFG2[optimizer]
    mov, tgt = tw.pair(S128, 1000)
    th0 = torch.from_numpy(tw.theta0_np(seed=0))
    s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(w_ncc=1.0), optimizer=optimizer, lr=lr, init=th0[None], capacity=iters,
                         flags=_lib.FLAG_ZSTREAM if zs else _lib.FLAG_NO_ZSTREAM)
    s.run(iters)
    torch.cuda.synchronize()
    r32, r64 = refs[("cfg2", optimizer, "float32")], refs[("cfg2", optimizer, "float64")]
    check_affine(s.losses[0].cpu().numpy().astype(np.float64), s.theta[0, :12].cpu().numpy().reshape(3, 4), r32, r64, (optimizer, zs))
    # best = first strict minimum of the recorded curve, theta of that forward (ref:warpings.py:85-93)
    bi = int(s.best_idx[0].item())
    assert bi == r64["best_idx"] or abs(r64["losses"][bi] - r64["losses"][r64["best_idx"]]) <= bar(r32["losses"], r64["losses"], 1e-7)
    assert np.max(np.abs(s.best_theta[0, :12].cpu().numpy().reshape(3, 4) - r64["thetas"][bi])) <= bar(r32["thetas"][bi], r64["thetas"][bi], 2e-6)
    assert r64["losses"][-1] < 0.9 * r64["losses"][0] or optimizer == "sgd"   # the run goes somewhere (Adam: a tenth of the way to theta*)


@pytest.mark.parametrize("optimizer,lr,shape", CFG3, ids=[c[0] for c in CFG3])
def test_cfg3_flow_ncc_smooth_10_iterations(eng, refs, optimizer, lr, shape):
    iters, sw = 10, 1.0
    mov, tgt = tw.pair(shape, 1000)
    s = eng.FlowSolver(mov.cuda(), tgt.cuda(), loss=eng.LossSpec(w_ncc=1.0), optimizer=optimizer, lr=lr, init=tw.smooth_flow0(shape), capacity=iters,
                       smooth_weight=sw)
    s.run(iters)
    torch.cuda.synchronize()
    r32, r64 = refs[("cfg3", optimizer, "float32")], refs[("cfg3", optimizer, "float64")]
    e, b = np.max(np.abs(s.losses[0].cpu().numpy() - r64["losses"])), bar(r32["losses"], r64["losses"], 2e-5 * np.max(np.abs(r64["losses"])))
    assert e <= b, ("loss curve", e, b)
    e, b = np.max(np.abs(s.flow.cpu().numpy() - r64["flow"])), bar(r32["flow"], r64["flow"], 1e-5)
    assert e <= b, ("flow", e, b)


@pytest.mark.parametrize("walk", ["pingpong", "one-way"])
def test_cfg4_share_of_one_gpu_8x256_20_iterations(eng, refs, walk):
    """walk: trx_affine_run alternates the z-streaming kernel's walk direction from iteration to iteration (the product) / every iteration walks upward
    (TRX_FLAG_NO_PINGPONG, the order of rounds 3-4): both against the same arbiter, so that the one-way order stays a tested path."""
    from torchregister_amd import _lib
    iters, B = 20, 8
    mov, tgt = batch_of_pairs(B, CFG4_PAIRS)
    th0 = torch.stack([torch.from_numpy(tw.theta0_np(seed=i)) for i in range(B)])
    s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0),
                         optimizer="adam", lr=CFG4_LR, init=th0, capacity=iters, flags=0 if walk == "pingpong" else _lib.FLAG_NO_PINGPONG)
    s.run(iters)
    torch.cuda.synchronize()
    rows = s.rows_used().tolist()
    assert len(set(rows)) == 1 and abs(rows[0]) == 64 and set(s.bodies()) == {"zstream"}, (rows, s.bodies())   # the z-streaming kernel (64 blocks per 256^3 pair) served all eight pairs
    for i in CFG4_PAIRS:   # (round 4: three of the eight pairs instead of one)
        check_affine(s.losses[i].cpu().numpy().astype(np.float64), s.theta[i, :12].cpu().numpy().reshape(3, 4), refs[("cfg4", i, "float32")],
                     refs[("cfg4", i, "float64")], ("pair", i))


def test_cfg4_rigid_from_the_reference_initial_pose_8x256_20_iterations(eng, refs):
    """Round 4: the rotated path at the headline's size as a TRAJECTORY - ONE launch of 8 x 256^3, rigid mode + NCC, Adam, 20 iterations from
    the reference's initial pose (torch.manual_seed(0); torch.rand(6), ref:utils.py:316-321: ~0.5 / 0.77 / 0.09 rad), which the
    exact-footprint kernel runs at every step; pair 3 against the arbiter (loss curve and theta after the last step)."""
    iters, B = 20, 8
    mov, tgt = batch_of_pairs(B, (CFG4R_PAIR,))
    pose0 = torch.stack([tw.rigid_pose0(i) for i in range(B)])
    s = eng.AffineSolver(mov, tgt, mode="rigid", loss=eng.LossSpec(w_ncc=1.0),
                         optimizer="adam", lr=CFG4_LR, init=pose0, capacity=iters)
    s.run(iters)
    torch.cuda.synchronize()
    rows = s.rows_used().tolist()
    assert all(r < 0 for r in rows) and set(s.bodies()) == {"eft"}, (rows, s.bodies())            # every pair of the last step ran the exact-footprint kernel
    i = CFG4R_PAIR
    check_affine(s.losses[i].cpu().numpy().astype(np.float64), s.theta[i, :12].cpu().numpy().reshape(3, 4), refs[("cfg4r", "adam", "float32")],
                 refs[("cfg4r", "adam", "float64")], ("rigid pair", i))

"""Round 6: the CARRY form of trx_affine_run (csrc/affine.hip, CarryKArgs) - launch-bound 3-D steps (one pair up to ~128^3: the two-body GeomA / GeomR
kernel on the classic grid) as ONE launch per iteration: the finalise of iteration k rides in the prologue of iteration k + 1's kernel (every block of a
pair reduces that pair's partial rows of the previous launch and computes the same theta; the pair's first block writes the state), one finalise kernel
behind the last launch flushes the run.  TRX_FLAG_NO_CARRY keeps the two-launch form of rounds 1-5: every test runs both and compares - the folded finalise
sums the rows in another fixed order (fp64), so the two agree to fp32 rounding of the 12-float update, not bit for bit.
The trajectories against the fp32 + fp64 arbiter at BASELINE cfg2's size (128^3, 200 iterations, SGD and Adam) run through this path in
tests/test_gpu_baseline_trajectories.py; the golden trajectories of the reference in tests/test_gpu_affine.py."""
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
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


def both(eng, mov, tgt, iters, **kw):
    """the same run in the carry form and in the two-launch form: (losses, theta, best theta, best idx, grad, step) of each"""
    from torchregister_amd import _lib
    out = []
    for fl in (0, _lib.FLAG_NO_CARRY):
        s = eng.AffineSolver(mov, tgt, capacity=sum(iters), flags=fl, **kw)
        for n in iters:
            s.run(n)
        torch.cuda.synchronize()
        out.append(dict(losses=s.losses.clone(), theta=s.theta.clone(), best=s.best_theta.clone(), best_idx=s.best_idx.clone(), best_loss=s.best_loss.clone(),
                        grad=s.grad.clone(), step=s.step.clone(), m=s.adam_m.clone(), v=s.adam_v.clone(), bodies=s.bodies(), rows=s.rows_used().tolist()))
    return out


def agree(a, b, iters_total, ltol=2e-6, ttol=2e-6):
    assert torch.equal(a["step"], b["step"]) and int(a["step"][0]) == iters_total
    la, lb = a["losses"][:, :iters_total], b["losses"][:, :iters_total]
    assert torch.isfinite(la).all()
    assert torch.max(torch.abs(la - lb) / torch.clamp(lb.abs(), min=1.0)).item() <= ltol, (la - lb).abs().max()
    assert torch.max(torch.abs(a["theta"] - b["theta"])).item() <= ttol
    assert torch.max(torch.abs(a["grad"] - b["grad"])).item() <= 1e-4 * max(b["grad"].abs().max().item(), 1e-30)
    assert a["bodies"] == b["bodies"] and a["rows"] == b["rows"]          # the notes of the LAST iteration sit in the primary buffers in both forms
    # best = first strict minimum: same index unless two recorded losses are within rounding of each other
    for p in range(la.shape[0]):
        ia, ib = int(a["best_idx"][p]), int(b["best_idx"][p])
        assert ia == ib or abs(lb[p, ia].item() - lb[p, ib].item()) <= ltol * max(1.0, abs(lb[p, ib].item())), (p, ia, ib)
        if ia == ib:
            assert torch.max(torch.abs(a["best"][p] - b["best"][p])).item() <= ttol


@pytest.mark.parametrize("shape", [(64, 64, 64), (40, 52, 36), (128, 128, 128)], ids=["64", "ragged", "128"])
@pytest.mark.parametrize("optimizer,lr", [("sgd", 2e-6), ("adam", 5e-4)])
def test_carry_run_equals_the_two_launch_run(eng, shape, optimizer, lr):
    """One pair, affine + NCC: 9 iterations in one call (odd count: the first launch writes the primary buffers) and 2 + 5 + 4 over three calls (the state
    crosses call boundaries through the caller's arrays; an even count starts on the second buffers)."""
    tgt = ph.blobs_fast(shape, 71, device="cuda")
    mov = eng.affine_warp(torch.tensor(ph.THETA_STAR3, device="cuda")[None], tgt) + 0.05 * ph.blobs_fast(shape, 72, device="cuda")
    th0 = torch.tensor(np.eye(3, 4) + 2.5e-3 * np.sin(1.2345 * (np.arange(12.0).reshape(3, 4) + 1.0)), dtype=torch.float32)[None]
    kw = dict(mode="affine", loss=eng.LossSpec(w_ncc=1.0), optimizer=optimizer, lr=lr, init=th0)
    for iters in ((9,), (2, 5, 4)):
        c, n = both(eng, mov, tgt, iters, **kw)
        agree(c, n, sum(iters))
        assert c["losses"][0, sum(iters) - 1] < c["losses"][0, 0] or optimizer == "sgd"


def test_carry_batch_of_pairs_with_their_own_bodies_rigid_and_mse(eng):
    """Three pairs of 48 x 56 x 64 in one launch - near the identity (GeomA), rotated (GeomR), and zoomed out of the volume - in RIGID mode from three poses
    (the pose chain rule rides in the prologue) and in affine mode with an MSE + SSD loss (the 13-sum rows of the MSE-only step kernel)."""
    shape, B = (48, 56, 64), 3
    tgt = torch.cat([ph.blobs_fast(shape, 200 + i, device="cuda") for i in range(B)])
    mov = torch.cat([ph.blobs_fast(shape, 210 + i, device="cuda") for i in range(B)])
    poses = torch.tensor([[0.02, -0.01, 0.03, 0.01, 0.0, -0.02], [0.5, 0.77, 0.09, 0.3, -0.2, 0.1], [0.9, 0.1, 0.6, -0.4, 0.5, 0.2]], dtype=torch.float32)
    c, n = both(eng, mov, tgt, (6,), mode="rigid", loss=eng.LossSpec(w_ncc=1.0), optimizer="adam", lr=2e-3, init=poses)
    agree(c, n, 6, ltol=5e-6, ttol=5e-6)
    assert len(set(c["bodies"])) >= 2, c["bodies"]
    ths = np.stack([np.eye(3, 4) + 4e-3 * np.sin(np.arange(12.0).reshape(3, 4)), np.concatenate([rot(0.4, 0.3, 0.5), [[0.02], [0.01], [-0.03]]], axis=1),
                    np.concatenate([1.6 * np.eye(3), [[0.1], [-0.2], [0.15]]], axis=1)])
    c, n = both(eng, mov, tgt, (5,), mode="affine", loss=eng.LossSpec(w_mse=1.0, w_ssd=0.1), optimizer="sgd", lr=1e-3, init=torch.tensor(ths, dtype=torch.float32))
    agree(c, n, 5, ltol=5e-6, ttol=5e-6)


def test_carry_first_iteration_against_the_oracle_and_two_iterations_by_hand(eng):
    """run(2) in the carry form: the recorded loss / gradient of iteration 0 against the C oracle (fp64), and theta after two SGD steps against the two
    oracle evaluations chained by hand (theta1 = theta0 - lr g0; theta2 = theta1 - lr g1) - the prologue's update IS the reference's optimizer.step()."""
    shape = (56, 48, 64)
    tgt = ph.blobs_fast(shape, 301, device="cuda")
    mov = ph.blobs_fast(shape, 302, device="cuda")
    th0 = np.asarray(ph.THETA_ROT3, dtype=np.float64)
    lr = 1e-5
    s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(w_ncc=1.0), optimizer="sgd", lr=lr, init=torch.tensor(th0, dtype=torch.float32)[None], capacity=2)
    s.run(2)
    torch.cuda.synchronize()
    m64, t64, tabs = mov[0, 0].double().cpu().numpy(), tgt[0, 0].double().cpu().numpy(), oracle.base_tables(shape, np.float64)
    l0, _, g0, _ = oracle.c_affine_loss_grad(m64, t64, np.asarray(th0, dtype=np.float32).astype(np.float64), oracle.wts(w_ncc=1.0), tabs)
    th1 = np.asarray(th0, dtype=np.float32).astype(np.float64) - lr * g0
    l1, _, g1, _ = oracle.c_affine_loss_grad(m64, t64, th1, oracle.wts(w_ncc=1.0), tabs)
    th2 = th1 - lr * g1
    assert abs(s.losses[0, 0].item() - l0) <= 2e-5 * max(1.0, abs(l0)) and abs(s.losses[0, 1].item() - l1) <= 2e-5 * max(1.0, abs(l1))
    assert np.max(np.abs(s.grad[0, :12].cpu().numpy().reshape(3, 4) - g1)) <= 3e-4 * np.max(np.abs(g1))
    assert np.max(np.abs(s.theta[0, :12].cpu().numpy().reshape(3, 4) - th2)) <= 2e-6 + 3e-4 * lr * np.max(np.abs(g1))

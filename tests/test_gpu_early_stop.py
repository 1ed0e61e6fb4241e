"""Early stop of the flow loop (ref:warpings.py:231-233: `if losses_train[-1] <= self.stop_crit: break`, AFTER optimizer.step()).

The direct-flow path tests the criterion on the device, per pair, with no host sync (csrc/flow.hip: flow_coef_kernel sets a per-pair
flag, later iterations are no-ops).  What must hold, exactly as in the reference's loop:
  * the number of recorded losses is k + 1 where k is the first iteration with loss <= stop_crit,
  * the update of iteration k HAS been applied (final flow = flow after k + 1 updates),
  * `.flow` is the flow of the LAST FORWARD (before update k).
Checked bit for bit against un-stopped solvers run for exactly k + 1 and k iterations, for every kernel variant (SGD / Adam, with and
without the smoothness term whose loss arrives one coefficient kernel late, 2-D / 3-D, one call / several calls / single-iteration
calls), and through flow_register for both flow models.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

import phantoms as ph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torchregister_amd._engine as e
    assert torch.cuda.is_available()
    return e


def _pair(shape, seed=0):
    from oracle import compose
    tgt = ph.blobs(shape, 1100 + seed)
    th = torch.tensor(ph.THETA_STAR3 if len(shape) == 3 else ph.THETA_STAR2)[None]
    mov = compose.affine_warp(th, tgt)
    return mov.cuda(), tgt.cuda()


CASES = [("sgd", 2.0, 0.0), ("adam", 0.02, 0.0), ("sgd", 1.0, 3.0), ("adam", 0.02, 2.0)]


@pytest.mark.parametrize("calls", ["one", "split", "single"])
@pytest.mark.parametrize("shape", [(20, 24, 28), (48, 56)])
@pytest.mark.parametrize("optimizer,lr,smooth", CASES)
def test_direct_flow_stops_exactly(eng, optimizer, lr, smooth, shape, calls):
    mov, tgt = _pair(shape)
    N, k = 14, 6
    kw = dict(loss=eng.LossSpec(w_mse=1.0, w_ncc=0.01), optimizer=optimizer, lr=lr, smooth_weight=smooth)
    free = eng.FlowSolver(mov, tgt, capacity=N, **kw)
    free.run(N)
    L = free.losses[0].cpu().numpy().astype(np.float64)
    assert np.all(np.diff(L[: k + 2]) < 0), "the phantom run must descend so that a threshold between two losses is well defined"
    crit = 0.5 * (L[k] + L[k - 1])          # first loss <= crit is L[k]

    s = eng.FlowSolver(mov, tgt, capacity=N, stop_crit=crit, **kw)
    if calls == "one":
        s.run(N)
    elif calls == "split":                  # the hit falls inside the second call; a third call must stay a no-op
        s.run(4)
        s.run(5)
        s.run(N - 9)
    else:                                   # single-iteration calls: no fusion, no lag, the double buffer is settled by a copy
        for _ in range(N):
            s.run(1)
    torch.cuda.synchronize()
    assert int(s.step[0]) == k + 1 and int(s.stopped[0]) != 0
    got = s.losses[0].cpu().numpy()
    assert np.array_equal(got[: k + 1], free.losses[0, : k + 1].cpu().numpy())
    assert np.all(np.isnan(got[k + 1:]))    # nothing is recorded after the stop

    after = eng.FlowSolver(mov, tgt, capacity=N, **kw)      # exactly k + 1 iterations: the final flow
    after.run(k + 1)
    before = eng.FlowSolver(mov, tgt, capacity=N, **kw)     # exactly k iterations: the flow of the last forward
    before.run(k)
    torch.cuda.synchronize()
    assert torch.equal(s.flow, after.flow)
    assert torch.equal(s.flow_last, before.flow)
    if optimizer == "adam":
        assert torch.equal(s.adam_m, after.adam_m) and torch.equal(s.adam_v, after.adam_v)


def test_stop_on_first_iteration_and_never(eng):
    mov, tgt = _pair((16, 20, 24))
    kw = dict(loss=eng.LossSpec(w_mse=1.0), lr=1.0)
    s = eng.FlowSolver(mov, tgt, capacity=5, stop_crit=1e9, **kw)       # the very first loss is below the bar
    s.run(5)
    one = eng.FlowSolver(mov, tgt, capacity=5, **kw)
    one.run(1)
    torch.cuda.synchronize()
    assert int(s.step[0]) == 1
    assert torch.equal(s.flow, one.flow) and torch.count_nonzero(s.flow_last).item() == 0
    n = eng.FlowSolver(mov, tgt, capacity=5, stop_crit=-1.0, **kw)      # never: flow_last = flow before the last update of the run
    n.run(5)
    four = eng.FlowSolver(mov, tgt, capacity=5, **kw)
    four.run(4)
    torch.cuda.synchronize()
    assert int(n.step[0]) == 5 and int(n.stopped[0]) == 0
    assert torch.equal(n.flow_last, four.flow)


def test_pairs_of_a_batch_stop_independently(eng):
    """A batch is B independent registrations: each pair stops at its own iteration (or never) and keeps its own last forward."""
    shape = (20, 24, 28)
    pairs = [_pair(shape, s) for s in range(3)]
    sc = torch.tensor([1.0, 1.7, 6.0], device="cuda").view(-1, 1, 1, 1, 1)      # MSE grows with the square of the intensity scale
    mov, tgt = torch.cat([p[0] for p in pairs]) * sc, torch.cat([p[1] for p in pairs]) * sc
    N = 12
    kw = dict(loss=eng.LossSpec(w_mse=1.0), optimizer="adam", lr=0.02, smooth_weight=1.5)
    free = eng.FlowSolver(mov, tgt, capacity=N, **kw)
    free.run(N)
    L = free.losses.cpu().numpy().astype(np.float64)

    def stops(crit):
        out = []
        for b in range(3):
            hit = np.nonzero(L[b] <= crit)[0]
            out.append(int(hit[0]) + 1 if len(hit) else N)
        return out

    # a threshold between two recorded losses at which the pairs really do differ: one stops mid-run, another later or never
    vals = np.unique(L)
    cands = [0.5 * (a + c) for a, c in zip(vals[:-1], vals[1:])]
    good = [c for c in cands if len(set(stops(c))) > 1 and any(2 <= w <= N - 2 for w in stops(c))]
    assert good, L
    crit = good[len(good) // 2]
    want = stops(crit)
    s = eng.FlowSolver(mov, tgt, capacity=N, stop_crit=crit, **kw)
    s.run(5)
    s.run(N - 5)
    torch.cuda.synchronize()
    assert s.step.cpu().tolist() == want
    for b in range(3):
        assert np.array_equal(s.losses[b, :want[b]].cpu().numpy(), free.losses[b, :want[b]].cpu().numpy())
        assert np.all(np.isnan(s.losses[b, want[b]:].cpu().numpy()))
        one = eng.FlowSolver(mov[b:b + 1], tgt[b:b + 1], capacity=N, **kw)
        one.run(want[b] - 1)
        torch.cuda.synchronize()
        assert torch.equal(s.flow_last[b], one.flow[0])
        one.run(1)
        torch.cuda.synchronize()
        assert torch.equal(s.flow[b], one.flow[0])


@pytest.mark.parametrize("shape", [(40, 44), (16, 20, 24)])
def test_flow_register_direct_stops_like_the_reference_loop(shape):
    """flow_model='direct' through the public class: count of recorded losses, final flow and `.flow` (the flow of the last forward)
    against the reference's own loop structure driven by torch (same HIP warp, torch SGD, host-side break)."""
    import torchregister_amd as tr
    mov, tgt = _pair(shape)
    N = 12
    probe = tr.flow_register(shape, criterions=[nn.MSELoss()], weights=[1.0], lr=1.0, max_epochs=N, stop_crit=-1.0, flow_model="direct")
    probe.optimize(mov, tgt, debug=False)
    L = probe.losses[0].cpu().numpy().astype(np.float64)
    assert probe.losses.shape[1] == N and int(probe.iterations[0]) == N
    k = 5
    crit = 0.5 * (L[k] + L[k - 1])
    fr = tr.flow_register(shape, criterions=[nn.MSELoss()], weights=[1.0], lr=1.0, max_epochs=N, stop_crit=crit, flow_model="direct")
    fr.optimize(mov, tgt, debug=False)
    assert fr.losses.shape[1] == k + 1 and int(fr.iterations[0]) == k + 1
    # the reference's loop, literally: forward, loss, backward, step, append, test
    fl = torch.zeros(1, len(shape), *shape, device="cuda", requires_grad=True)
    opt = torch.optim.SGD([fl], 1.0)
    warp = tr.SpatialTransformer(shape)
    losses = []
    for _ in range(N):
        opt.zero_grad()
        last = fl.detach().clone()
        err = nn.MSELoss()(tgt, warp(mov, fl))
        err.backward()
        opt.step()
        losses.append(err.item())
        if losses[-1] <= crit:
            break
    assert len(losses) == k + 1
    assert np.max(np.abs(np.asarray(losses) - fr.losses[0].cpu().numpy())) <= 2e-6 * max(1.0, abs(losses[0]))
    assert torch.max(torch.abs(fr.flow - last)).item() <= 1e-5            # flow of the last forward
    assert torch.max(torch.abs(fr.final_flow - fl.detach())).item() <= 1e-5
    assert not torch.equal(fr.flow, fr.final_flow)
    # deform() uses the flow of the last forward, like the reference (ref:warpings.py:238-242)
    assert torch.equal(fr.deform(mov), warp(mov, fr.flow))


def test_flow_register_unet_stops_like_the_reference_loop():
    """flow_model='unet' (the reference's model): the loop is the reference's own (host-side test after every step).  Two same-seed GPU
    runs of the U-Net are not bit-reproducible (MIOpen's convolution backward), so the checks are self-consistent ones: a run stops at
    the first of ITS OWN recorded losses that is <= stop_crit, never earlier, never later; `.flow` is the flow of the last forward (the
    weights have moved since)."""
    import torchregister_amd as tr
    shape = (160, 160)
    mov, tgt = _pair(shape)

    def run(max_epochs, crit):
        torch.manual_seed(7)
        fr = tr.flow_register(shape, n=2, criterions=[nn.MSELoss()], weights=[1.0], lr=1e-2, max_epochs=max_epochs, stop_crit=crit).cuda()
        fr.optimize(mov, tgt, debug=False)
        return fr

    probe = run(6, -1.0)
    L = probe.losses[0].cpu().numpy().astype(np.float64)
    assert len(L) == 6                                           # never
    assert run(6, 1e9).losses.shape[1] == 1                      # at once: one forward, one step, stop
    crit = float(np.sort(L)[2])                                  # somewhere inside the curve's range
    fr = run(6, crit)
    mine = fr.losses[0].cpu().numpy()
    assert 1 <= len(mine) <= 6
    assert np.all(mine[:-1] > crit) and (mine[-1] <= crit or len(mine) == 6)
    with torch.no_grad():
        assert not torch.allclose(fr.flow, fr.model.features(mov), atol=1e-7)   # the weights moved after the last forward


def test_capacity_is_enforced(eng):
    from torchregister_amd import _lib
    mov, tgt = _pair((16, 20, 24))
    s = eng.FlowSolver(mov, tgt, capacity=4)
    s.run(3)
    with pytest.raises(_lib.TrxError, match="capacity"):
        s.run(2)
    a = eng.AffineSolver(mov, tgt, capacity=4)
    a.run(4)
    with pytest.raises(_lib.TrxError, match="capacity"):
        a.run(1)


def test_smoothness_term_reaches_every_flow_path():
    """Register(mode='flow', smooth_weight=...) optimises the same objective whichever path evaluates it: the fused
    direct path and the torch-driven generic path (a criterion list with no fused form) record the same first loss."""
    import torchregister_amd as tr
    from torchregister_amd.warpings import smooth_regulariser
    shape = (16, 20, 24)
    mov, tgt = _pair(shape)
    lam = 4.0
    init = (0.3 * ph.flow_field(shape, 1.0, 0.05)).cuda()
    s = tr._engine.FlowSolver(mov, tgt, loss=tr._engine.LossSpec(w_mse=1.0), lr=0.0, init=init, capacity=1, smooth_weight=lam)
    s.run(1)
    mse = nn.MSELoss()(tgt, tr.SpatialTransformer(shape)(mov, init))
    want = mse + smooth_regulariser(init, lam)
    assert abs(s.losses[0, 0].item() - want.item()) <= 2e-6 * max(1.0, abs(want.item()))

    class L2(nn.Module):                                 # no fused form -> _optimize_generic
        def forward(self, a, b):
            return ((a - b) ** 2).mean()

    g = tr.flow_register(shape, criterions=[L2()], weights=[1.0], lr=0.5, max_epochs=3, stop_crit=-1.0, flow_model="direct", smooth_weight=lam)
    g.optimize(mov, tgt, debug=False)
    f = tr.flow_register(shape, criterions=[nn.MSELoss()], weights=[1.0], lr=0.5, max_epochs=3, stop_crit=-1.0, flow_model="direct", smooth_weight=lam)
    f.optimize(mov, tgt, debug=False)
    assert np.max(np.abs(g.losses[0].cpu().numpy() - f.losses[0].cpu().numpy())) <= 1e-5 * abs(f.losses[0, 0].item())
    assert torch.max(torch.abs(g.final_flow - f.final_flow)).item() <= 1e-5

"""Peer-mapped transport of the Z-slab partition (include/trx.h: trx_peer_*; torchregister_amd.SlabPeers): the boundary planes and the
sum of the 8 moments travel as direct writes into the peers' mailboxes instead of torch.distributed calls.

  * one process, three slabs on the one GPU of the test box, driven in lock step (run_slabs_lockstep) - the "self-peer" arrangement:
    the mailboxes are plain tensors of the same device; must reproduce the un-partitioned FlowSolver like the torch.distributed path
    (tests/test_gpu_flow.py::test_slab_partition_equals_whole_volume), and every slab must record the SAME loss curve bit for bit
    (the N slots are added in rank order on every rank);
  * two processes on the one GPU, mailboxes mapped through HIP IPC (SlabPeers.exchange over a gloo group), SlabFlowSolver.run();
  * a wait nobody answers times out, sets the status word and SlabPeers.check() raises - it does not hang the device."""
import os
import tempfile

import pytest
import torch

import phantoms as ph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torchregister_amd._engine as e
    assert torch.cuda.is_available()
    return e


@pytest.mark.parametrize("optimizer,lr,smooth", [("sgd", 2.0, 0.0), ("adam", 0.05, 0.0), ("sgd", 1.0, 4.0), ("adam", 0.05, 2.0)])
def test_peer_lockstep_three_slabs_equal_whole_volume(eng, optimizer, lr, smooth):
    shape = (36, 28, 40)
    tgt, mov = ph.blobs(shape, 1021).cuda(), ph.blobs(shape, 1022).cuda()
    iters = 6
    kw = dict(loss=eng.LossSpec(w_ncc=1.0, w_mse=0.3), optimizer=optimizer, lr=lr, capacity=iters, smooth_weight=smooth)
    whole = eng.FlowSolver(mov, tgt, **kw)
    whole.run(iters)
    bounds = [0, 10, 25, 36]
    world = len(bounds) - 1
    boxes = [eng.SlabPeers.allocate(mov.device, shape[1], shape[2], world) for _ in range(world)]
    slabs = [eng.SlabFlowSolver(mov, tgt[:, :, a:b].contiguous(), a, peers=eng.SlabPeers(r, boxes, shape[1], shape[2]), **kw)
             for r, (a, b) in enumerate(zip(bounds[:-1], bounds[1:]))]
    eng.run_slabs_lockstep(slabs, 4)
    eng.run_slabs_lockstep(slabs, iters - 4)          # a second call continues (flags keep counting)
    torch.cuda.synchronize()
    for s in slabs:
        s.peers.check()
        assert torch.allclose(s.losses, whole.losses, rtol=1e-5, atol=1e-6)
        assert torch.equal(s.losses, slabs[0].losses)                                  # the same fp64 additions on every rank
    flow = torch.cat([s.flow for s in slabs], dim=2)
    tol = 2e-3 if (optimizer == "adam" and smooth) else 1e-5                           # (see test_slab_partition_equals_whole_volume)
    assert torch.max(torch.abs(flow - whole.flow)).item() <= tol * max(1.0, whole.flow.abs().max().item())


def test_peer_path_equals_hand_exchange_bitwise(eng):
    """The transport moves bytes, it must not change them: peer-mapped lock step == the same slabs with planes copied and sums added by
    hand in rank order (what test_slab_partition_equals_whole_volume does), bit for bit."""
    shape = (30, 24, 32)
    tgt, mov = ph.blobs(shape, 77).cuda(), ph.blobs(shape, 78).cuda()
    iters, bounds = 5, [0, 12, 30]
    kw = dict(loss=eng.LossSpec(w_ncc=1.0), optimizer="adam", lr=0.03, capacity=iters, smooth_weight=1.5)
    hand = [eng.SlabFlowSolver(mov, tgt[:, :, a:b].contiguous(), a, **kw) for a, b in zip(bounds[:-1], bounds[1:])]
    for _ in range(iters):
        planes = [s.boundary_planes() for s in hand]
        hand[0].halo_hi.copy_(planes[1][0]); hand[1].halo_lo.copy_(planes[0][1])
        ms = [s.local_moments_without_halo().clone() for s in hand]
        for s, m in zip(hand, ms):
            edge = torch.zeros_like(m)
            s.add_boundary_smooth(edge)
            m += edge
        total = ms[0] + ms[1]
        for s in hand:
            s.apply(total.clone())
    boxes = [eng.SlabPeers.allocate(mov.device, shape[1], shape[2], 2) for _ in range(2)]
    peer = [eng.SlabFlowSolver(mov, tgt[:, :, a:b].contiguous(), a, peers=eng.SlabPeers(r, boxes, shape[1], shape[2]), **kw)
            for r, (a, b) in enumerate(zip(bounds[:-1], bounds[1:]))]
    eng.run_slabs_lockstep(peer, iters)
    torch.cuda.synchronize()
    for h, p in zip(hand, peer):
        p.peers.check()
        assert torch.equal(h.losses, p.losses) and torch.equal(h.flow, p.flow)


def test_peer_wait_times_out_instead_of_hanging(eng):
    boxes = [eng.SlabPeers.allocate(torch.device("cuda", 0), 8, 8, 2) for _ in range(2)]
    pr = eng.SlabPeers(0, boxes, 8, 8)
    pr.TIMEOUT_US = 2000
    pr.wait_halo("hi")                      # nobody raises the flag
    out = torch.zeros(1, 8, dtype=torch.float64, device="cuda")
    pr.publish(out)                         # rank 1 never publishes
    pr.gather(out)
    torch.cuda.synchronize()
    # the first time-out is STICKY: the gather behind it returns at once - it does not spin through its own time-out - and
    # hands NaN sums to the update, so the loss curve shows the failure even if nobody calls check()
    assert int(pr.status.item()) == 1
    assert torch.isnan(out).all()
    with pytest.raises(Exception, match="timed out"):
        pr.check()
    # a gather that times out on its own (no earlier failure) reports 2, also with NaN sums; later waits return immediately
    pr2 = eng.SlabPeers(0, [eng.SlabPeers.allocate(torch.device("cuda", 0), 8, 8, 2) for _ in range(2)], 8, 8)
    pr2.TIMEOUT_US = 2000
    out2 = torch.zeros(1, 8, dtype=torch.float64, device="cuda")
    pr2.publish(out2)
    pr2.gather(out2)
    pr2.TIMEOUT_US = 60_000_000             # (would hang the test for a minute if the failure were not sticky)
    pr2.wait_halo("lo")
    torch.cuda.synchronize()
    assert int(pr2.status.item()) == 2 and torch.isnan(out2).all()


def _ipc_worker(rank, world, port, tmp, shape, bounds, iters, kw, finegrained=True):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
    import torch.distributed as dist
    import phantoms as ph2
    import torchregister_amd._engine as e
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    tgt, mov = ph2.blobs(shape, 1021).cuda(), ph2.blobs(shape, 1022).cuda()
    box = e.SlabPeers.allocate(mov.device, shape[1], shape[2], world, finegrained=finegrained)
    assert (getattr(box, "_trx_mem", None) is not None) == finegrained
    boxes = e.SlabPeers.exchange(box)
    a, b = bounds[rank], bounds[rank + 1]
    s = e.SlabFlowSolver(mov, tgt[:, :, a:b].contiguous(), a, peers=e.SlabPeers(rank, boxes, shape[1], shape[2]), **kw)
    s.run(iters)
    torch.cuda.synchronize()
    s.peers.check()
    torch.save({"flow": s.flow.cpu(), "losses": s.losses.cpu()}, os.path.join(tmp, f"rank{rank}.pt"))
    dist.barrier()                          # nobody unmaps a mailbox a peer may still be writing
    dist.destroy_process_group()


@pytest.mark.parametrize("optimizer,lr,smooth,finegrained", [("sgd", 1.0, 4.0, True), ("adam", 0.05, 0.0, True), ("sgd", 1.0, 4.0, False)])
def test_two_processes_ipc_mailboxes_equal_whole_volume(eng, optimizer, lr, smooth, finegrained):
    """finegrained: mailboxes from trx_peer_alloc (hipExtMallocWithFlags, fine-grained) shared by their raw HIP IPC handles - the default;
    False: torch caching-allocator tensors shared through torch's storage IPC (round 3's path, kept for comparison)."""
    import torch.multiprocessing as mp
    shape, bounds, iters = (36, 28, 40), [0, 14, 36], 6
    kw = dict(loss=eng.LossSpec(w_ncc=1.0, w_mse=0.3), optimizer=optimizer, lr=lr, capacity=iters, smooth_weight=smooth)
    tgt, mov = ph.blobs(shape, 1021).cuda(), ph.blobs(shape, 1022).cuda()
    whole = eng.FlowSolver(mov, tgt, **kw)
    whole.run(iters)
    torch.cuda.synchronize()
    with tempfile.TemporaryDirectory() as tmp:
        port = 29300 + (os.getpid() % 250)
        mp.spawn(_ipc_worker, args=(2, port, tmp, shape, bounds, iters, kw, finegrained), nprocs=2, join=True)
        parts = [torch.load(os.path.join(tmp, f"rank{r}.pt")) for r in range(2)]
    flow = torch.cat([p["flow"] for p in parts], dim=2)
    for p in parts:
        assert torch.allclose(p["losses"], whole.losses.cpu(), rtol=1e-5, atol=1e-6)
    assert torch.equal(parts[0]["losses"], parts[1]["losses"])
    assert torch.max(torch.abs(flow - whole.flow.cpu())).item() <= 1e-5 * max(1.0, whole.flow.abs().max().item())


def test_peer_early_stop_and_single_rank(eng):
    """Early stop over the peer transport: the whole-volume loss is bit-identical on every rank (rank-order sum), so every slab stops at the
    same iteration without exchanging anything else - as the un-partitioned solver does; and a world of one rank (its own mailbox) equals
    the plain solver bit for bit."""
    shape = (30, 24, 32)
    tgt, mov = ph.blobs(shape, 401).cuda(), ph.blobs(shape, 402).cuda()
    iters = 12
    kw = dict(loss=eng.LossSpec(w_ncc=1.0), optimizer="sgd", lr=2.0, capacity=iters)
    probe = eng.FlowSolver(mov, tgt, **kw)
    probe.run(iters)
    torch.cuda.synchronize()
    crit = 0.5 * (probe.losses[0, 4] + probe.losses[0, 5]).item()          # met by iteration 5's loss
    whole = eng.FlowSolver(mov, tgt, stop_crit=crit, **kw)
    whole.run(iters)
    bounds = [0, 13, 30]
    boxes = [eng.SlabPeers.allocate(mov.device, shape[1], shape[2], 2) for _ in range(2)]
    slabs = [eng.SlabFlowSolver(mov, tgt[:, :, a:b].contiguous(), a, stop_crit=crit, peers=eng.SlabPeers(r, boxes, shape[1], shape[2]), **kw)
             for r, (a, b) in enumerate(zip(bounds[:-1], bounds[1:]))]
    eng.run_slabs_lockstep(slabs, iters)
    torch.cuda.synchronize()
    n = int(whole.step[0].item())
    for s in slabs:
        s.peers.check()
        assert int(s.step_t.item()) == int(slabs[0].step_t.item())
    k = int(slabs[0].step_t.item())
    for s in slabs:
        assert torch.equal(s.losses[0, :k], slabs[0].losses[0, :k]) and torch.isnan(s.losses[0, k:]).all()   # nothing recorded after the stop
    assert 0 < k < iters
    assert k == n
    assert torch.allclose(slabs[0].losses[0, :k], whole.losses[0, :k], rtol=1e-5, atol=1e-6)
    # one rank
    one_box = [eng.SlabPeers.allocate(mov.device, shape[1], shape[2], 1)]
    one = eng.SlabFlowSolver(mov, tgt, 0, peers=eng.SlabPeers(0, one_box, shape[1], shape[2]), **kw)
    one.run(iters)
    plain = eng.SlabFlowSolver(mov, tgt, 0, **kw)
    plain.run(iters)
    torch.cuda.synchronize()
    one.peers.check()
    assert torch.equal(one.losses, plain.losses) and torch.equal(one.flow, plain.flow)


def test_apply_outside_run_keeps_flow_last_current(eng):
    """a caller that drives the slab building blocks itself (local_moments -> sum -> apply) ends with flow_last = the flow its
    LAST update started from, without knowing about TRX_FLAG_SAVE_LAST; apply(last=False) opts out."""
    import phantoms as ph
    shape = (12, 16, 20)
    mov, tgt = ph.blobs(shape, 5).cuda(), ph.blobs(shape, 6).cuda()
    s = eng.SlabFlowSolver(mov, tgt, 0, loss=eng.LossSpec(w_ncc=1.0), optimizer="sgd", lr=0.5, capacity=8, stop_crit=-1.0)   # (stop_crit: allocates flow_last)
    assert s.flow_last is not None
    for it in range(3):
        before = s.flow.clone()
        s.apply(s.local_moments())
        torch.cuda.synchronize()
        assert torch.equal(s.flow_last, before), it
    keep = s.flow_last.clone()
    s.apply(s.local_moments(), last=False)
    torch.cuda.synchronize()
    assert torch.equal(s.flow_last, keep)


def test_finegrained_mailbox_is_wrapped_in_place_and_freed(eng):
    """SlabPeers.allocate: fine-grained memory from trx_peer_alloc, zeroed, seen by torch at the same address (no copy), exportable as a
    64-byte HIP IPC handle; the allocation is released when the last tensor on it goes away."""
    box = eng.SlabPeers.allocate(torch.device("cuda", 0), 8, 8, 2)
    mem = box._trx_mem
    assert box.dtype == torch.uint8 and box.is_cuda and box.data_ptr() == mem.ptr and box.numel() == eng.SlabPeers.layout(8, 8, 2)[4]
    assert int(box.sum().item()) == 0
    box[:16] = 7
    torch.cuda.synchronize()
    assert int(box[:32].sum().item()) == 7 * 16
    h = mem.export()
    assert isinstance(h, bytes) and len(h) == 64 and any(h)
    view = box[4:12].view(torch.float32)
    del box
    assert view.is_cuda and mem.ptr != 0          # a view keeps the memory
    del view
    import gc
    gc.collect()
    assert mem.ptr != 0                           # ... and so does our own reference to the holder
    mem.close()
    assert mem.ptr == 0

"""world_size-2 gloo test (CPU) of the multi-GPU plumbing: shard assignment, MAX timing reduce and the
result gather.  The data path itself has no collective (pairs are independent)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_pairs, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from torchregister_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.pair_range(rank, world, total_pairs)
    # fake per-shard results: theta encodes the global pair id, losses the rank
    theta = torch.stack([torch.full((3, 4), float(i)) for i in range(lo, hi)]) if hi > lo else torch.zeros(0, 3, 4)
    losses = torch.full((hi - lo, 5), float(rank))
    t = sharding.max_over_ranks(1.0 + rank)
    th_all, l_all = sharding.gather_results(theta, losses)
    # slab mode (config 5): the only per-iteration exchange is the sum of 8 fp64 moments
    z0, z1 = sharding.slab_range(rank, world, 37)
    mom = torch.full((1, 8), float(z1 - z0), dtype=torch.float64)
    dist.all_reduce(mom, op=dist.ReduceOp.SUM)
    assert mom.tolist() == [[37.0] * 8], mom
    q.put((rank, lo, hi, t, th_all[:, 0, 0].tolist(), l_all[:, 0].tolist(), sharding.weak_pair_ids(rank, 8)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total_pairs", [64, 7])
def test_two_rank_sharding_gloo(total_pairs):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, t0, th0, l0, w0), (r1, lo1, hi1, t1, th1, l1, w1) = res
    assert (lo0, hi1) == (0, total_pairs) and hi0 == lo1                     # shards tile the batch
    assert abs((hi0 - lo0) - (hi1 - lo1)) <= 1
    assert t0 == t1 == 2.0                                                   # MAX over ranks
    assert th0 == th1 == [float(i) for i in range(total_pairs)]               # gathered in pair order on every rank
    assert l0 == [0.0] * (hi0 - lo0) + [1.0] * (hi1 - lo1)
    assert w0 == list(range(0, 8)) and w1 == list(range(8, 16))              # weak-scaling ids (bench.py)


def test_single_process_identity():
    sys.path.insert(0, ROOT)
    from torchregister_amd import sharding
    assert sharding.pair_range(0, 1, 5) == (0, 5)
    assert sharding.max_over_ranks(3.5) == 3.5
    a, b = sharding.gather_results(torch.ones(2, 3, 4), torch.zeros(2, 7))
    assert a.shape == (2, 3, 4) and b.shape == (2, 7)


def _subgroup_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from torchregister_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    groups = [dist.new_group([0, 1]), dist.new_group([2, 3])]     # two independent slab partitions in one job; the second starts at global rank 2
    g = groups[rank // 2]
    below, above = sharding.neighbour_global_ranks(g)
    # the halo exchange of SlabFlowSolver.exchange_halos with these peers: every rank sends its global rank as the "boundary plane"
    mine = torch.full((4,), float(rank))
    got_lo, got_hi = torch.full((4,), -1.0), torch.full((4,), -1.0)
    ops = []
    if below is not None:
        ops += [dist.P2POp(dist.isend, mine, below, g), dist.P2POp(dist.irecv, got_lo, below, g)]
    if above is not None:
        ops += [dist.P2POp(dist.isend, mine, above, g), dist.P2POp(dist.irecv, got_hi, above, g)]
    for r in dist.batch_isend_irecv(ops):
        r.wait()
    q.put((rank, below, above, got_lo[0].item(), got_hi[0].item()))
    dist.barrier()
    dist.destroy_process_group()


def test_slab_neighbours_inside_a_subgroup_gloo():
    """P2POp peers are GLOBAL ranks.  Four ranks, two sub-groups {0,1} and {2,3}: inside the second one the lower slab is
    global rank 2 and its upper neighbour global rank 3 (group ranks 0 and 1) - the planes must travel between exactly those."""
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, None, 1, -1.0, 1.0), (1, 0, None, 0.0, -1.0), (2, None, 3, -1.0, 3.0), (3, 2, None, 2.0, -1.0)]


def _fallback_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torchregister_amd._engine import SlabPeers
    out = {}
    out["all_ok"] = SlabPeers.agree(True)
    out["one_bad"] = SlabPeers.agree(rank != 1)
    # the mapping fails on rank 1 only (what a missing peer access or an invisible device looks like): EVERY rank must fall back
    cpu_box = lambda d, H, W, n: torch.zeros(SlabPeers.layout(H, W, n)[4], dtype=torch.uint8)   # (the real allocate() is fine-grained GPU memory)
    export = lambda box: ("cpu", rank, box.numel())

    def opener(box, handles, r):
        assert [h[1] for h in handles] == list(range(world))   # the handles really travelled through all_gather_object
        if rank == 1:
            raise RuntimeError("no peer access from device 1 to device 0")
        return [box] * world   # rank 0 maps fine - and must still fall back because rank 1 could not
    peers, why = SlabPeers.try_exchange(torch.device("cpu"), 8, 8, rank, _alloc=cpu_box, _export=export, _open=opener)
    out["peers_none"], out["why"] = peers is None, why
    # the ALLOCATION fails on rank 0 only, before any collective of the exchange: rank 0 must not run ahead into the final
    # agree() while rank 1 sits in all_gather_object - both fall back, nobody hangs, and the next collective still lines up
    def bad_alloc(d, H, W, n):
        if rank == 0:
            raise MemoryError("hipExtMallocWithFlags: out of memory")
        return cpu_box(d, H, W, n)
    peers2, why2 = SlabPeers.try_exchange(torch.device("cpu"), 8, 8, rank, _alloc=bad_alloc, _export=export, _open=lambda box, h, r: [box] * world)
    out["alloc_none"], out["alloc_why"] = peers2 is None, why2
    # the export fails on rank 1 only
    def bad_export(box):
        if rank == 1:
            raise RuntimeError("hipIpcGetMemHandle: invalid argument")
        return export(box)
    peers3, why3 = SlabPeers.try_exchange(torch.device("cpu"), 8, 8, rank, _alloc=cpu_box, _export=bad_export, _open=lambda box, h, r: [box] * world)
    out["export_none"], out["export_why"] = peers3 is None, why3
    t = torch.tensor([rank + 1.0])
    dist.all_reduce(t)   # the group is still in step after three failed set-ups
    out["sum"] = float(t.item())
    q.put((rank, out))
    dist.destroy_process_group()


def test_peer_transport_fallback_is_decided_by_all_ranks_together():
    """when the mailbox mapping fails on ANY rank (IPC import, peer access, device visibility) every rank of the group
    must fall back to torch.distributed - a MIN all-reduce of the local outcome - and say why (tools/run_slab.py prints the transport
    that actually ran)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + os.getpid() % 200
    ps = [ctx.Process(target=_fallback_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    for r in (0, 1):
        assert res[r]["all_ok"] is True and res[r]["one_bad"] is False
        assert res[r]["peers_none"] is True and res[r]["why"]
        assert res[r]["alloc_none"] is True and res[r]["export_none"] is True
        assert res[r]["sum"] == 3.0
    assert "peer access" in res[1]["why"]
    assert "out of memory" in res[0]["alloc_why"] and "another rank" in res[1]["alloc_why"]
    assert "hipIpcGetMemHandle" in res[1]["export_why"] and "another rank" in res[0]["export_why"]

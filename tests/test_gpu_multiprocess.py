"""Z-slab mode (BASELINE config 5) with REAL process-group traffic: two processes share the one GPU of the test box and
talk through a gloo group (all-reduce of the 8 fp64 sums, batched isend/irecv of the boundary flow planes) - the same
SlabFlowSolver.run() code that runs over RCCL / xGMI on a multi-GPU node; the slabs together must reproduce the
un-partitioned FlowSolver.  (RCCL itself refuses two ranks on one device, and the test box has one.)"""
import os
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, tmp, shape, bounds, iters, kw):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
    import torch.distributed as dist
    import phantoms as ph
    import torchregister_amd._engine as eng
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    tgt = ph.blobs(shape, 1021).cuda()
    mov = ph.blobs(shape, 1022).cuda()
    a, b = bounds[rank], bounds[rank + 1]
    s = eng.SlabFlowSolver(mov, tgt[:, :, a:b].contiguous(), a, **kw)
    s.run(iters)
    torch.cuda.synchronize()
    torch.save({"flow": s.flow.cpu(), "losses": s.losses.cpu()}, os.path.join(tmp, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("optimizer,lr,smooth", [("sgd", 1.0, 4.0), ("adam", 0.05, 0.0)])
def test_two_process_slabs_equal_whole_volume(optimizer, lr, smooth):
    import torch.multiprocessing as mp
    import phantoms as ph
    import torchregister_amd._engine as eng
    shape, bounds, iters = (36, 28, 40), [0, 14, 36], 6
    kw = dict(loss=eng.LossSpec(w_ncc=1.0, w_mse=0.3), optimizer=optimizer, lr=lr, capacity=iters, smooth_weight=smooth)
    tgt, mov = ph.blobs(shape, 1021).cuda(), ph.blobs(shape, 1022).cuda()
    whole = eng.FlowSolver(mov, tgt, **kw)
    whole.run(iters)
    torch.cuda.synchronize()
    with tempfile.TemporaryDirectory() as tmp:
        port = 29600 + (os.getpid() % 300)
        mp.spawn(_worker, args=(2, port, tmp, shape, bounds, iters, kw), nprocs=2, join=True)
        parts = [torch.load(os.path.join(tmp, f"rank{r}.pt")) for r in range(2)]
    flow = torch.cat([p["flow"] for p in parts], dim=2)
    for p in parts:
        assert torch.allclose(p["losses"], whole.losses.cpu(), rtol=1e-5, atol=1e-6)     # every rank records the whole-volume loss
    assert torch.max(torch.abs(flow - whole.flow.cpu())).item() <= 1e-5 * max(1.0, whole.flow.abs().max().item())


def _pair_worker(rank, world, port, tmp, shape, total, iters):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
    import torch.distributed as dist
    import phantoms as ph
    import torchregister_amd as tr
    from torchregister_amd import sharding
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    lo, hi = sharding.pair_range(rank, world, total)          # 5 pairs over 2 ranks: 3 + 2 (remainder on the first rank)
    tgt = torch.cat([ph.blobs(shape, 60 + p) for p in range(lo, hi)]).cuda()
    mov = torch.cat([ph.blobs(shape, 80 + p) for p in range(lo, hi)]).cuda()
    s = tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=1e-3, capacity=iters)
    s.run(iters)
    theta, losses = sharding.gather_results(s.current_theta, s.losses[:, :iters])
    t = sharding.max_over_ranks(0.5 + rank, torch.device("cuda", 0))
    # the user-facing driver of the same path: the WHOLE batch is described on every rank (here by generator callables), each
    # rank solves its shard, everyone gets everything back in pair order
    gen = lambda seed0: (lambda a, b: torch.cat([ph.blobs(shape, seed0 + p) for p in range(a, b)]))  # noqa: E731
    r = tr.register_sharded(gen(80), gen(60), mode="affine", loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=1e-3, iters=iters, pairs=total)
    assert r["shard"] == (lo, hi) and r["theta"].shape == (total, 3, 4) and r["losses"].shape == (total, iters)
    if rank == 0:
        torch.save({"theta": theta.cpu(), "losses": losses.cpu(), "tmax": t, "sharded_final": r["final_theta"].cpu(), "sharded_losses": r["losses"].cpu(),
                    "sharded_best_idx": r["best_idx"].cpu(), "sharded_best": r["theta"].cpu()}, os.path.join(tmp, "gathered.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_pair_sharding_two_processes_equal_one_batch():
    """The data-parallel path of bench.py / config 4: independent pairs sharded over ranks with NO data-path collective; the
    gathered theta / loss curves of 2 ranks (3 + 2 pairs) are bitwise those of one process running all 5 pairs."""
    import torch.multiprocessing as mp
    import phantoms as ph
    import torchregister_amd as tr
    shape, total, iters = (24, 32, 40), 5, 8
    tgt = torch.cat([ph.blobs(shape, 60 + p) for p in range(total)]).cuda()
    mov = torch.cat([ph.blobs(shape, 80 + p) for p in range(total)]).cuda()
    s = tr.AffineSolver(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=1e-3, capacity=iters)
    s.run(iters)
    torch.cuda.synchronize()
    with tempfile.TemporaryDirectory() as tmp:
        port = 29900 + (os.getpid() % 90)
        mp.spawn(_pair_worker, args=(2, port, tmp, shape, total, iters), nprocs=2, join=True)
        got = torch.load(os.path.join(tmp, "gathered.pt"))
    assert torch.equal(got["theta"], s.current_theta.cpu()) and torch.equal(got["losses"], s.losses[:, :iters].cpu())
    assert got["tmax"] == 1.5
    # register_sharded (torchregister_amd.sharding): the same numbers through the user-facing driver
    assert torch.equal(got["sharded_final"], s.current_theta.cpu()) and torch.equal(got["sharded_losses"], s.losses[:, :iters].cpu())
    assert torch.equal(got["sharded_best"], s.best.cpu()) and torch.equal(got["sharded_best_idx"], s.best_idx.cpu())
    one = tr.register_sharded(mov, tgt, mode="affine", loss=tr.LossSpec(w_ncc=1.0), optimizer="adam", lr=1e-3, iters=iters)   # no process group: one GPU, all pairs
    assert torch.equal(one["final_theta"], s.current_theta) and one["shard"] == (0, total)


def test_bench_two_ranks_control_flow():
    """bench.py's N > 1 control flow (rendezvous from the torchrun environment, barrier + max-over-ranks timing, whole-job
    aggregate, ONE JSON line from rank 0) with two ranks on the one GPU over gloo; the driver runs the same code over RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + os.getpid() % 200), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--size", "64", "--backend", "gloo", "--single-device", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["warmup"] == 2 and j["scaling"] == "weak" and j["higher_is_better"] is True
    assert abs(j["value"] - 2 * 8 * 6 / (j["ms_per_step"] * 6e-3)) < 1e-6 * j["value"]      # whole-job aggregate over both ranks
    assert "roofline" not in j and "cpu_baseline" not in j                                   # N = 1 only


def test_bench_strong_scaling_two_ranks():
    """`bench.py --scaling strong --pairs-total P` - the second half of SURVEY 8d's table (64 / G pairs per GPU) - with two
    ranks on the one GPU over gloo: 5 pairs split 3 + 2 (sharding.pair_range), value = P x K / max-over-ranks(elapsed), "scaling": "strong"."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29900 + os.getpid() % 90), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--size", "64", "--backend", "gloo", "--single-device", "--no-cpu-baseline", "--scaling", "strong", "--pairs-total", "5"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["pairs_total"] == 5 and j["config"]["pairs_per_gpu"] == 3   # (rank 0's shard)
    assert abs(j["value"] - 5 * 6 / (j["ms_per_step"] * 6e-3)) < 1e-6 * j["value"]


def test_bench_under_torchrun_nccl_single_rank():
    """The RCCL path on hardware: bench.py under torch.distributed.run with ONE rank and the `nccl` backend (the launcher starts before
    any GPU call) - process-group init on the device, barriers and the MAX all-reduce of the timing run through RCCL exactly as they
    do on an 8-GPU node (a 1-GPU test box cannot hold a second RCCL rank)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(29400 + os.getpid() % 150), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
           "--size", "64", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 6 and j["value"] > 0 and "roofline" in j

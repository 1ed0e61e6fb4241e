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

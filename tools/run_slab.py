#!/usr/bin/env python3
"""BASELINE config 5: one S^3 volume, direct flow + NCC (+ smoothness), Z-slabs over the ranks of one node.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/run_slab.py --size 512 --iters 100
    python tools/run_slab.py --size 256 --iters 20            # single process: one full-depth slab

Every rank builds the (closed-form) phantom, keeps the whole moving volume and its own slab of target / flow /
Adam state, and runs SlabFlowSolver: per iteration one 64-byte all-reduce (+ two one-plane P2P halo exchanges with
the smoothness term) over torch.distributed, or (--transport peer) the same two exchanges as direct writes into the peers'
IPC-mapped mailboxes.  Rank 0 prints one JSON line."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import blobs_gpu, THETA_STAR   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--optimizer", default="adam")
    ap.add_argument("--smooth", type=float, default=1.0)
    ap.add_argument("--transport", choices=("dist", "peer"), default="dist",
                    help="dist: torch.distributed (RCCL) P2P + all_reduce; peer: direct writes into IPC-mapped mailboxes (SlabPeers)")
    a = ap.parse_args()
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    import torchregister_amd as tr
    from torchregister_amd.sharding import slab_range, max_over_ranks
    shape = (a.size,) * 3
    tgt = blobs_gpu(shape, 1000, dev)
    mov = tr.get_affine_warp(torch.tensor(THETA_STAR, device=dev)[None], tgt)
    z0, z1 = slab_range(rank, world, a.size)
    peers, transport = None, ("dist" if world > 1 else "none")
    if a.transport == "peer" and world > 1:
        # mailboxes if EVERY rank can map them (IPC export / import, peer access, device visibility), else torch.distributed - decided
        # together, and the line below says which transport actually ran
        peers, why = tr.SlabPeers.try_exchange(dev, a.size, a.size, rank)
        transport = "peer" if peers is not None else f"dist (peer transport unavailable: {why})"
    s = tr.SlabFlowSolver(mov, tgt[:, :, z0:z1].contiguous(), z0, loss=tr.LossSpec(w_ncc=1.0), optimizer=a.optimizer,
                          lr=0.01 if a.optimizer == "adam" else 1.0, capacity=a.iters + a.warmup, smooth_weight=a.smooth, peers=peers)
    del tgt
    s.run(a.warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    s.run(a.iters)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = max_over_ranks(time.perf_counter() - t0, dev)
    if peers is not None:
        peers.check()
    if rank == 0:
        ls = s.losses[0, : a.iters + a.warmup]
        print(json.dumps({"config": f"{a.size}^3 direct flow + NCC + {a.optimizer} + smooth {a.smooth}, {world} Z-slab(s)", "iters": a.iters,
                          "ms_per_iter": 1e3 * el / a.iters, "iters_per_s": a.iters / el, "loss_first": ls[0].item(), "loss_last": ls[-1].item(),
                          "n_gpus": world, "transport": transport, "transport_requested": a.transport}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

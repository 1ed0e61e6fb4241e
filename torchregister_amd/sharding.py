"""Multi-GPU sharding of independent (moving, target) pairs: one process per GPU, NO data-path
collective (pairs never interact; the reference is strictly batch-1, ref:torchregister.py:52-55).
Only scalars / KB-sized results cross ranks: the timing MAX and the gathered theta / loss curves.
Works on any torch.distributed backend ("nccl" = RCCL on ROCm; "gloo" in the CPU tests)."""
import torch


def pair_range(rank, world, total_pairs):
    """Contiguous shard [lo, hi) of `total_pairs` for `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(int(total_pairs), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def weak_pair_ids(rank, pairs_per_rank):
    """Weak scaling (bench.py): rank r owns global pair ids [r*p, (r+1)*p)."""
    return list(range(rank * pairs_per_rank, (rank + 1) * pairs_per_rank))


def max_over_ranks(value, device="cpu"):
    """MAX all-reduce of a python float (elapsed time); identity when not distributed."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_results(theta, losses, device=None):
    """Gather per-rank results (theta [b,nd,nd+1], losses [b,T]) on every rank, ordered by rank.
    Shards may differ in size (strong scaling with a remainder)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return theta, losses
    world = dist.get_world_size()
    device = device or theta.device
    n = torch.tensor([theta.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)

    def pad(t):
        out = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=device)
        out[: t.shape[0]] = t.to(device)
        return out

    outs = []
    for t in (theta, losses):
        buf = [torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=device) for _ in range(world)]
        dist.all_gather(buf, pad(t))
        outs.append(torch.cat([b[:c] for b, c in zip(buf, counts)]))
    return outs[0], outs[1]


def slab_range(rank, world, depth):
    """Z-slab [z0, z1) of a D-deep volume owned by `rank` (BASELINE config 5: 512 planes / 8 ranks = 64 each)."""
    return pair_range(rank, world, depth)

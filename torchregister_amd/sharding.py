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


def neighbour_global_ranks(group=None):
    """(lower, upper) Z neighbours of this rank inside `group` as GLOBAL ranks (None at the ends): torch.distributed.P2POp addresses
    peers by global rank, the position of a slab along Z is the rank inside the group."""
    import torch.distributed as dist
    r, n = dist.get_rank(group), dist.get_world_size(group)
    to_global = (lambda k: k) if group is None else (lambda k: dist.get_global_rank(group, k))
    return (to_global(r - 1) if r > 0 else None), (to_global(r + 1) if r + 1 < n else None)


def register_sharded(moving, target, mode="affine", loss=None, optimizer="sgd", lr=1e-5, iters=1000, init=None, device=None,
                     pairs=None, betas=(0.9, 0.999), eps=1e-8):
    """BASELINE config 4 for callers: N independent (moving, target) pairs spread over the ranks of the default process group (one
    process per GPU, launched with torchrun), 64 pairs -> 8 per GPU on an 8-GPU node.  The reference is batch-1
    (ref:torchregister.py:52-55): every pair is its own registration, so there is NO data-path collective - each rank uploads and
    solves only its shard (AffineSolver: all pairs of the shard in one launch per iteration) and the KB-sized results are
    all-gathered at the end.

    moving, target: [N,1,*spatial] tensors (CPU, pinned or GPU; identical on every rank - only the rank's slice is moved to its GPU),
                    or callables `f(lo, hi) -> tensor [hi-lo,1,*spatial]` that produce the shard (then give pairs=N).
    init:           optional [N, ...] initial theta (affine) / pose (rigid), sliced the same way.
    Returns a dict on EVERY rank, in global pair order: theta (best, Q8) [N,nd,nd+1], final_theta, losses [N,iters], best_idx [N],
    shard (lo, hi) of this rank.  Without an initialised process group one GPU solves all N pairs."""
    import torch.distributed as dist
    from ._engine import AffineSolver, LossSpec
    distributed = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if distributed else (0, 1)
    n = int(pairs) if pairs is not None else int(moving.shape[0])
    lo, hi = pair_range(rank, world, n)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())

    def shard(t):
        if callable(t):
            return t(lo, hi).to(device=device, dtype=torch.float32)
        return t[lo:hi].to(device=device, dtype=torch.float32, non_blocking=True)

    nd = None
    out = {}
    if hi > lo:
        mov, tgt = shard(moving), shard(target)
        nd = mov.dim() - 2
        solver = AffineSolver(mov, tgt, mode=mode, loss=loss or LossSpec(w_mse=1.0), optimizer=optimizer, lr=lr,
                              init=None if init is None else init[lo:hi], capacity=max(1, iters), betas=betas, eps=eps)
        solver.run(iters)
        res = (solver.best, solver.current_theta, solver.losses[:, :iters], solver.best_idx.to(torch.float32)[:, None])
    else:   # more ranks than pairs: an empty shard still takes part in the gather
        first = moving(0, 1) if callable(moving) else moving[:1]
        nd = first.dim() - 2
        res = (torch.zeros(0, nd, nd + 1, device=device), torch.zeros(0, nd, nd + 1, device=device), torch.zeros(0, iters, device=device),
               torch.zeros(0, 1, device=device))
    best, final = gather_results(res[0], res[1], device)
    losses, bidx = gather_results(res[2], res[3], device)
    out.update(theta=best, final_theta=final, losses=losses, best_idx=bidx[:, 0].to(torch.int32), shard=(lo, hi))
    return out

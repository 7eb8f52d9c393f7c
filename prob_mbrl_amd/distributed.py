"""Multi-GPU decomposition of the rollout: particle rows are independent except
inside a moment-matching group, so rank r owns a contiguous block of whole groups
(row = particle*S + sample, group = particle).  Weights, reward constants and the
discount vector are replicated; every rank produces a partial flat policy
gradient already scaled by 1/B_global; ONE sum all-reduce (RCCL over xGMI,
`backend="nccl"`) per optimiser iteration; then the identical clip + Adam runs
on every rank (deterministic replicas, no parameter broadcast)."""
import torch


def shard_bounds(B, mm_groups, world, rank):
    """[lo, hi) rows of `rank`: whole groups, as even as possible."""
    G = mm_groups if mm_groups else B
    if B % G:
        raise ValueError('B must be divisible by mm_groups')
    M = B // G
    base, extra = divmod(G, world)
    g_lo = rank * base + min(rank, extra)
    g_hi = g_lo + base + (1 if rank < extra else 0)
    return g_lo * M, g_hi * M


def local_groups(B, mm_groups, world, rank):
    lo, hi = shard_bounds(B, mm_groups, world, rank)
    if not mm_groups:
        return None
    return (hi - lo) // (B // mm_groups)


def allreduce_sum_(t, group=None):
    """In-place sum all-reduce of the flat gradient (no-op without a process group)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def max_over_ranks(x, device, group=None):
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())

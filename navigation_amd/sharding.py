"""Fleet sharding across ranks: robots are independent, so the instance range is split
statically and the only collective is an all-reduce of throughput counters.  Pure host logic —
works with any torch.distributed backend (nccl == RCCL on the GPUs, gloo in the CPU tests)."""


def shard_range(n_total, rank, world):
    """Contiguous block of instances owned by `rank`: first, count (remainder goes to the low ranks)."""
    base, rem = divmod(n_total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def owner_of(instance, n_total, world):
    base, rem = divmod(n_total, world)
    cut = rem * (base + 1)
    if instance < cut:
        return instance // (base + 1)
    return rem + (instance - cut) // base if base else world - 1


def reduce_counters(dist, elapsed_s, counters, device="cpu"):
    """Whole-job view: MAX of the per-rank elapsed time, SUM of the per-rank counters."""
    import torch
    el = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    cnt = torch.tensor([float(c) for c in counters], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    return float(el.item()), [float(x) for x in cnt.tolist()]

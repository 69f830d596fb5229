"""Slice sharding for multi-GPU inference: slices are independent units, split contiguously over ranks,
no data-path collective (the reference gets the same effect from PTL's DistributedSampler under `strategy: ddp`)."""


def shard_range(n_items: int, rank: int, world_size: int):
    """Contiguous [start, stop) of `n_items` for `rank`; the first n_items % world_size ranks get one extra item."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} / world_size {world_size}")
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_metric_sums(values, device=None):
    """Sum a list of per-rank scalar metrics over all ranks (the reference's DistributedMetricSum, models/base.py:35-53).
    Works with or without an initialised process group."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
